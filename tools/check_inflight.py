#!/usr/bin/env python
"""Lint for the inline-asm weight stream of the fast sweep kernels.

pmbrl_fast.h issues some global loads through inline asm and waits for them with explicit
s_waitcnt (see the comment above pm_ldw).  The compiler does not know those registers are
still being written, so a register-allocator copy / spill / reuse of a destination register
between the load and the wait that covers it would silently read stale data.  This script
walks the control-flow graph of the generated ISA (hipcc -S), tracking on every path which
inline-asm loads are still outstanding (an inline-asm `s_waitcnt vmcnt(N)` retires all but the
N youngest vector-memory operations -- stores and the compiler's own loads take their places in that queue too), and
reports any instruction that touches a register that is in flight.

usage: check_inflight.py file.s [function-name-substring]
"""
import re
import sys

REG = re.compile(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b')
LABEL = re.compile(r'^(\.LBB\d+_\d+):')
FUNC = re.compile(r'^(_Z\w+):')
VMEM = re.compile(r'^(global|buffer|flat|scratch)_(load|store|atomic)')
LOAD = re.compile(r'^(global|buffer)_load')


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def parse_functions(path):
    funcs, cur, name, in_asm = {}, None, None, False
    for ln, raw in enumerate(open(path), 1):
        m = FUNC.match(raw)
        if m:
            name, cur, in_asm = m.group(1), [], False
            funcs[name] = cur
            continue
        if cur is None:
            continue
        st = raw.strip()
        if st.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if st.startswith(';;#ASMEND'):
            in_asm = False
            continue
        m = LABEL.match(st)
        if m:
            cur.append(('label', m.group(1), ln, raw))
            continue
        code = st.split(';')[0].strip()
        if code.startswith('.Lfunc_end'):     # (not the first s_endpgm: blocks may follow it)
            cur = None
            continue
        if not code or code.startswith('.') or code.endswith(':'):
            continue
        cur.append(('asm' if in_asm else 'ins', code, ln, raw))
    return funcs


def check_function(name, items):
    # split into basic blocks
    blocks, label_of, cur = [], {}, []
    pending_labels = []
    for it in items:
        if it[0] == 'label':
            if cur:
                blocks.append(cur)
                cur = []
            pending_labels.append(it[1])
            continue
        if not cur:
            for l in pending_labels:
                label_of[l] = len(blocks)
            pending_labels = []
        cur.append(it)
        code = it[1]
        if code.startswith('s_branch') or code.startswith('s_cbranch') or code.startswith('s_endpgm'):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    dst_regs = {}
    bad = set()
    nload = 0
    seen = set()
    work = [(0, ())]
    while work:
        bi, state = work.pop()
        if bi >= len(blocks) or (bi, state) in seen:
            continue
        seen.add((bi, state))
        st = list(state)
        succ = [bi + 1]
        for kind, code, ln, raw in blocks[bi]:
            used = None
            if VMEM.match(code) and not (kind == 'asm' and LOAD.match(code)):
                # any other vector-memory operation (the compiler's own loads, every store): it takes a place in the
                # in-order return queue that vmcnt counts, and has no destination to watch
                used = regs_of(code)
                for at in st:
                    if not at:
                        continue
                    hit = used & dst_regs[at]
                    if hit:
                        bad.add((ln, at, raw.strip(), tuple(sorted(hit))))
                if st:            # (anonymous, and of interest only while a watched load is older than it)
                    st.append(0)
                    if len(st) > 64:      # vmcnt counts to 63: forgetting an anonymous entry only errs on the safe side
                        st.remove(0)      # (fewer entries behind a load = a wait retires less of what is watched)
                continue
            if kind == 'asm' and LOAD.match(code):
                dst = code.split()[1].rstrip(',')
                dst_regs[ln] = regs_of(dst)
                used = regs_of(code.split(',', 1)[1]) | dst_regs[ln]
                for at in st:
                    if not at:
                        continue
                    hit = used & dst_regs[at]
                    if hit:
                        bad.add((ln, at, raw.strip(), tuple(sorted(hit))))
                # the same load site re-issued (loop): only its youngest instance matters for the
                # register check; this also bounds the abstract state
                if ln in st:
                    st.remove(ln)
                st.append(ln)
                continue
            if code.startswith('s_waitcnt'):      # (hand-written or the compiler's: the hardware does not care)
                mm = re.search(r'vmcnt\((\d+)\)', code)
                if mm:
                    keep = int(mm.group(1))
                    st = st[len(st) - keep:] if keep else []
                    while st and st[0] == 0:
                        st.pop(0)
                continue
            if code.startswith('s_endpgm'):
                succ = []
                break
            if code.startswith('s_branch'):
                succ = [label_of[code.split()[1]]]
                break
            if code.startswith('s_cbranch'):
                succ = [label_of[code.split()[1]], bi + 1]
                break
            used = regs_of(code)
            for at in st:
                if not at:
                    continue
                hit = used & dst_regs[at]
                if hit:
                    bad.add((ln, at, raw.strip(), tuple(sorted(hit))))
        for s in succ:
            work.append((s, tuple(st)))
    nload = sum(1 for v in dst_regs.values() if v)
    for ln, at, raw, hit in sorted(bad):
        print('%s:%d: touches v%s, in flight since line %d\n    %s' % (name, ln, list(hit), at, raw))
    return len(bad), nload, len(seen)


if __name__ == '__main__':
    only = sys.argv[2] if len(sys.argv) > 2 else 'fast'
    tot_bad = tot_load = 0
    for name, items in parse_functions(sys.argv[1]).items():
        if only not in name:
            continue
        b, n, nstates = check_function(name, items)
        print('%s: %d inline-asm loads, %d (block, state) pairs, %d violations' % (name, n, nstates, b))
        tot_bad += b
        tot_load += n
    print('check_inflight: %d inline-asm loads scanned, %d violations' % (tot_load, tot_bad))
    sys.exit(1 if tot_bad else 0)
