"""The LDS layout of the wide layers' piece planes (csrc/pmbrl_wide.h: pw_sw), checked against the bank model of
/opt/skills/guides/MI355X_MICROARCH.md (tools/ubench/lds_swizzle.py): the chunk swizzle leaves the B-operand reads
(ds_read_b128) conflict-free and halves the conflicts of the epilogue's ds_write_b64; and the swizzle is a permutation of
a row's 16-byte chunks that never leaves a K32 block (what lets a K = 32 first layer use the same planes)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'ubench'))
import lds_swizzle as M  # noqa: E402


def pw_sw(row, col):
    return (((col >> 3) ^ ((row >> 2) & 3)) << 3) | (col & 7)


def test_reads_stay_conflict_free_and_writes_halve():
    assert M.worst(False) == (1, 4)
    assert M.worst(True) == (1, 2)


def test_swizzle_is_a_permutation_inside_every_k32_block():
    for row in range(64):
        for kb in range(16):
            cols = [pw_sw(row, c) for c in range(kb * 32, kb * 32 + 32)]
            assert sorted(cols) == list(range(kb * 32, kb * 32 + 32))
            # 8-element chunks stay contiguous (a B operand is one 16-byte read)
            for c in range(kb * 32, kb * 32 + 32, 8):
                assert [pw_sw(row, c + e) for e in range(8)] == list(range(pw_sw(row, c), pw_sw(row, c) + 8))
