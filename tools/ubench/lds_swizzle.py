#!/usr/bin/env python
"""LDS bank model of the wide layers' piece planes (pmbrl_wide.h: pw_sw): conflicts of the B-operand reads (ds_read_b128)
and of the epilogue's writes (ds_write_b64) with and without the chunk swizzle, lane groups and bank functions as
/opt/skills/guides/MI355X_MICROARCH.md lists them.  Prints the worst multiplicity of a bank inside one lane group."""
LDB = 528
GROUPS_R = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
            list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def worst(swizzle):
    sig = lambda c: ((c >> 2) & 3) if swizzle else 0
    wr = ww = 0
    for kb in range(16):                    # reads: lane (g, c) -> row c, chunk 4 kb + g
        for grp in GROUPS_R:
            cnt = {}
            for l in grp:
                g, c = l >> 4, l & 15
                a = (c * LDB + ((kb * 4 + g) ^ sig(c)) * 8) * 2
                for d in range(4):
                    b = (a // 4 + d) % 64
                    cnt[b] = cnt.get(b, 0) + 1
            wr = max(wr, max(cnt.values()))
    for wid in range(8):                    # writes: 4 x 16 contiguous lanes; lane (g, c) -> row c, column 64 wid + 16 k + 4 g
        for k in range(4):
            for grp in range(4):
                cnt = {}
                for l in range(grp * 16, grp * 16 + 16):
                    g, c = l >> 4, l & 15
                    col = wid * 64 + k * 16 + 4 * g
                    a = (c * LDB + ((col >> 3) ^ sig(c)) * 8 + (col & 7)) * 2
                    for d in range(2):
                        b = (a // 4 + d) % 32
                        cnt[b] = cnt.get(b, 0) + 1
                ww = max(ww, max(cnt.values()))
    return wr, ww


if __name__ == '__main__':
    for s in (False, True):
        print('swizzle %-5s: reads %d-way, writes %d-way' % (s, *worst(s)))
