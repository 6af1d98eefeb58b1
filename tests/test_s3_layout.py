"""The LDS image layout of the split-arithmetic minibatch kernel (elegantrl_amd/csrc/split_bf16.h: swz<>; ppo_step_s3_impl.h: phi; s3_image.h:
s3_swz) restated in Python and checked against the three access patterns it was solved for, with the LDS lane groups and bank
moduli of /opt/skills/guides/MI355X_MICROARCH.md (reads: 64 dword banks; writes: 32; ds_read_b128 in four 16-lane groups
{0-3,12-15,20-27}, {4-11,16-19,28-31}, +32; transposing reads in two 32-lane passes; ds_write_b128 in groups of 8 consecutive lanes).
No GPU: this pins the formulas the kernel and the optimiser's image writer share (a change to one without the other, or a swizzle
that reintroduces conflicts, fails here before it costs a 3.7x slowdown on the device)."""
import re
from pathlib import Path

import pytest

CSRC = Path(__file__).resolve().parents[1] / "elegantrl_amd" / "csrc"


def swz(CP, r):
    r0, r1, r2, r3 = r & 1, (r >> 1) & 1, (r >> 2) & 1, (r >> 3) & 1
    if CP == 16:
        return ((r1 ^ r3) << 3) | (r0 << 2) | ((r1 ^ r2) << 1) | r2
    if CP == 8:
        return (r1 << 2) | (r2 << 1) | (r0 ^ r3)
    return (r2 << 1) | (r1 ^ r3)


def phi(i):
    return (i & 0x13) | ((i & 4) << 1) | ((i & 8) >> 1)


def addr(CP, row, part, chunk, byte=0):
    """byte address of (row, part, 16-byte chunk) inside an image with CP chunks per part"""
    return row * 48 * CP + part * 16 * CP + 16 * (chunk ^ swz(CP, row)) + byte


def test_formulas_match_the_sources():
    """the C sources carry exactly these expressions (both copies: the kernel's template and the optimiser's runtime form)"""
    for f in ("split_bf16.h", "s3_image.h"):
        src = (CSRC / f).read_text()
        flat = re.sub(r"\s+", " ", src)
        assert "((r1 ^ r3) << 3) | (r0 << 2) | ((r1 ^ r2) << 1) | r2" in flat, f
        assert "(r1 << 2) | (r2 << 1) | (r0 ^ r3)" in flat, f
        assert "(r2 << 1) | (r1 ^ r3)" in flat, f
    assert "(i & 0x13) | ((i & 4) << 1) | ((i & 8) >> 1)" in (CSRC / "ppo_step_s3_impl.h").read_text()


def test_phi_is_an_involution_within_16():
    assert sorted(phi(i) for i in range(32)) == list(range(32))
    assert all(phi(phi(i)) == i and (phi(i) >> 4) == (i >> 4) for i in range(32))


B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]


@pytest.mark.parametrize("CP", [4, 8, 16])
def test_forward_operand_reads_are_conflict_free(CP):
    """ds_read_b128 of the forward A operand: lane (i = lane & 31, hi = lane >> 5) reads chunk 2 ks + hi of row 32 To + phi(i)"""
    for To in range(4):
        for ks in range(CP // 2):
            for part in range(3):
                for group in B128_GROUPS:
                    banks = set()
                    for lane in group:
                        a = addr(CP, 32 * To + phi(lane & 31), part, 2 * ks + (lane >> 5))
                        b = {(a // 4 + d) % 64 for d in range(4)}
                        assert not (b & banks), (CP, To, ks, part, lane)
                        banks |= b


@pytest.mark.parametrize("CP", [4, 8, 16])
def test_transposing_reads_are_conflict_free(CP):
    """ds_read_b64_tr_b16 (backward's W2^T, both weight-gradient operands): lane (q = lane >> 4: kb = q >> 1, half = q & 1;
    t = lane & 15: rr = t >> 2, u = t & 3) reads 8 bytes at row 16 ks + 8 kb + 4 ridx + rr, columns 32 tile + 16 half + 4 u' .. + 3
    (u' = u or sigma(u): the same set of half-chunks); the two 32-lane passes must each touch 64 distinct banks"""
    for tile in range(CP // 4):
        for ks in range(8):
            for ridx in range(2):
                for part in range(3):
                    for first in (0, 32):
                        banks = set()
                        for lane in range(first, first + 32):
                            q, t = lane >> 4, lane & 15
                            kb, half, rr, u = q >> 1, q & 1, t >> 2, t & 3
                            row = 16 * ks + 8 * kb + 4 * ridx + rr
                            a = addr(CP, row, part, 4 * tile + 2 * half + (u >> 1), 8 * (u & 1))
                            b = {(a // 4 + d) % 64 for d in range(2)}
                            assert not (b & banks), (CP, tile, ks, ridx, part, lane)
                            banks |= b


@pytest.mark.parametrize("CP", [4, 8, 16])
def test_staging_stores_are_conflict_free(CP):
    """ds_write_b128 of a sample-major image: lane (sample s = 32 wave + (lane & 31), hi) stores chunk 2 ks + hi of row s; groups
    of 8 consecutive lanes, 32 dword banks"""
    for wave in range(4):
        for ks in range(CP // 2):
            for part in range(3):
                for g0 in range(0, 64, 8):
                    banks = set()
                    for lane in range(g0, g0 + 8):
                        a = addr(CP, 32 * wave + (lane & 31), part, 2 * ks + (lane >> 5))
                        b = {(a // 4 + d) % 32 for d in range(4)}
                        assert not (b & banks), (CP, wave, ks, part, lane)
                        banks |= b


@pytest.mark.parametrize("CP", [4, 8, 16])
def test_image_is_a_bijection(CP):
    """every (row, part, column) of a 128-row image owns its own two bytes: the optimiser's element writer (s3_image_put: column c
    -> chunk c >> 3, byte 2 (c & 7)) and the kernel's 8-column operands address the same cells"""
    K = 8 * CP
    seen = set()
    for row in range(128):
        for part in range(3):
            for c in range(K):
                a = addr(CP, row, part, c >> 3, 2 * (c & 7))
                assert a not in seen and a < 128 * 6 * K
                seen.add(a)
    assert len(seen) == 128 * 3 * K
