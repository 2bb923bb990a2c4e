#!/usr/bin/env python3
"""CPU check of the LDS images of corrla_rs_amd/csrc/mixed_kernels.hpp (no GPU needed): replays, with the same formulas,
what the loader waves' LDS-DMA writes (physical 16-byte slot <- logical slot, rule 21 of cdna_hip_programming.md: linear
destination, swizzled SOURCE) and what the MFMA waves read back, and verifies
  * every fragment read returns the reduction indices kmap(g, j) of the right row / column, for both kernels and all
    planes / column tiles / row tiles,
  * every ds_read_b128 (four 16-lane groups, MI355X_MICROARCH.md LDS table) and ds_read_b32 (two 32-lane groups) of the
    kernels is bank-conflict free,
  * split_planes_kernel's position -> reduction index map equals the fragment order the big operand is read in.
Run: python tools/lds_layout_check.py"""
import itertools
import sys

KT, BIG, OUTER = 32, 32768, 256
B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
               list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def kmap(p):
    return 16 + 4 * (p >> 3) + (p & 3) if (p & 4) else 4 * (p >> 3) + (p & 3)


def big_swz(row):
    return (row >> 1) & 7


def plane_swz(row):
    return (-(row >> 2)) & 3


def tn_swz(kr):
    return ((kr >> 2) & 1) << 2


def fill_stage(nt, np_, tn):
    """LDS 16-byte slot index -> what it holds after the DMA of one tile."""
    lds = {}
    plane = nt * 16 * KT * 2
    for c in range(BIG // 1024):
        for lane in range(64):
            dst = (c * 1024 + 16 * lane) // 16
            if not tn:
                row = 8 * c + (lane >> 3)
                ls = (lane & 7) ^ big_swz(row)
                lds[dst] = ("big", row, tuple(4 * ls + e for e in range(4)))       # outer row, reduction indices
            else:
                kr = c
                ls = lane ^ tn_swz(kr)
                lds[dst] = ("big", kr, tuple(4 * ls + e for e in range(4)))        # reduction row, outer columns
    for p in range(np_):
        for ct in range(nt):
            for lane in range(64):
                row = 16 * ct + (lane >> 2)
                ls = (lane & 3) ^ plane_swz(row)
                dst = (BIG + p * plane + ct * 1024 + 16 * lane) // 16
                lds[dst] = ("plane", p, row, tuple(kmap(8 * ls + j) for j in range(8)))  # column, reduction indices held
    return lds, plane


def conflict_free_b128(addrs):
    for grp in B128_GROUPS:
        slots = {}
        for lane in grp:
            s = (addrs[lane] // 16) % 16
            if s in slots and slots[s] != addrs[lane]:
                return False
            slots[s] = addrs[lane]
    return True


def conflict_free_b32(addrs):
    for grp in (range(0, 32), range(32, 64)):
        banks = {}
        for lane in grp:
            b = (addrs[lane] // 4) % 32
            if b in banks and banks[b] != addrs[lane]:
                return False
            banks[b] = addrs[lane]
    return True


def check(nt, np_, tn):
    lds, plane = fill_stage(nt, np_, tn)
    want_k = [[kmap(8 * g + j) for j in range(8)] for g in range(4)]
    for wave, mw in itertools.product(range(8), range(2)):
        if not tn:
            for half in range(2):
                addrs = {}
                for lane in range(64):
                    fr, fg = lane & 15, lane >> 4
                    a_base = (32 * wave + fr) * 128 + ((fg ^ big_swz(fr)) << 4)
                    addr = (a_base + mw * 2048) ^ (64 * half)
                    addrs[lane] = addr
                    kind, row, ks = lds[addr // 16]
                    assert kind == "big" and row == 32 * wave + 16 * mw + fr, (wave, mw, lane, row)
                    assert list(ks) == want_k[fg][4 * half:4 * half + 4], (lane, ks)
                assert conflict_free_b128(addrs), ("nn big", wave, mw, half)
        else:
            for j in range(8):
                addrs = {}
                for lane in range(64):
                    fr, fg = lane & 15, lane >> 4
                    a_base = (((32 * wave + fr) >> 2) << 4) + ((fr & 3) << 2)
                    off = (a_base + mw * 64) ^ ((fg & 1) << 6)
                    kr = 4 * fg + (j if j < 4 else 12 + j)
                    assert kr == want_k[fg][j]
                    addr = kr * 1024 + off
                    addrs[lane] = addr
                    kind, row, cols = lds[addr // 16]
                    assert kind == "big" and row == kr, (lane, j, row, kr)
                    assert cols[(addr % 16) // 4] == 32 * wave + 16 * mw + fr, (lane, j, cols, addr)
                assert conflict_free_b32(addrs), ("tn big", wave, mw, j)
    for t, p in itertools.product(range(nt), range(np_)):
        addrs = {}
        for lane in range(64):
            fr, fg = lane & 15, lane >> 4
            b_base = BIG + fr * 64 + ((fg ^ plane_swz(fr)) << 4)
            addr = b_base + t * 1024 + p * plane
            addrs[lane] = addr
            kind, pp, col, ks = lds[addr // 16]
            assert kind == "plane" and pp == p and col == 16 * t + fr, (t, p, lane, pp, col)
            assert list(ks) == want_k[fg], (lane, ks)
        assert conflict_free_b128(addrs), ("plane", t, p)


if __name__ == "__main__":
    assert sorted(kmap(p) for p in range(32)) == list(range(32))
    n = 0
    for nt, np_, tn in itertools.product(range(1, 10), (2, 3), (False, True)):
        check(nt, np_, tn)
        n += 1
    print(f"{n} kernel variants: fragment maps and bank-conflict freedom OK")
