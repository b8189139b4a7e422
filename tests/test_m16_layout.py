"""The register layout of the v_mfma_f32_16x16x32 kernels (nerf_amd/csrc/m16.h, nrf_common.h: m16_slot), restated in numpy
and checked for the invariants the kernels rely on -- no GPU, no library call:

* a block's register 8 nb + 4 mh + r of lane (c, g) is feature 16 mh + 4 g + r of point 16 nb + c, and the eight registers of one
  nb are, in order, the lane's eight k-values of the next layer's B fragment (slot m16_slot(g, jj));
* the packers reach the same weight through (register t, lane half h) of the 32x32 form: acc_row(t, h) == slot;
* to_m16 (v_permlane32_swap then v_permlane16_swap per register pair) takes a block made in the 32x32 form -- register t of
  lane (j, h) = slot acc_row(t, h) of point j -- to that layout, and the mask-word conversions are its bit-level inverse / copy;
* saved rows: both forms address every (row, point) of a block's [32 rows][32 points] exactly once.

GPU parity of the kernels built on this is tests/test_gpu_bf16x3.py, test_gpu_bf16.py and the training tests."""
import numpy as np


def acc_row(t, h):                      # nrf_common.h
    return (t & 3) + 8 * (t >> 2) + 4 * h


def m16_slot(g, jj):                    # nrf_common.h
    return 16 * (jj >> 2) + 4 * g + (jj & 3)


def permlane32_swap(x, y):              # upper 32 lanes of x <-> lower 32 lanes of y
    x, y = x.copy(), y.copy()
    x[32:], y[:32] = y[:32].copy(), x[32:].copy()
    return x, y


def permlane16_swap(x, y):              # odd rows (of 16 lanes) of x <-> even rows of y
    x, y = x.copy(), y.copy()
    for r in (0, 2):
        a, b = slice(16 * (r + 1), 16 * (r + 2)), slice(16 * r, 16 * (r + 1))
        x[a], y[b] = y[b].copy(), x[a].copy()
    return x, y


def to_m16(e):                          # e: (64 lanes, 16 registers) -> the same shape (m16.h)
    r = np.empty_like(e)
    for mh in range(2):
        for rr in range(4):
            p, q = permlane32_swap(e[:, 8 * mh + rr], e[:, 8 * mh + 4 + rr])
            d0, d1 = permlane16_swap(p, q)
            r[:, 4 * mh + rr], r[:, 8 + 4 * mh + rr] = d0, d1
    return r


def mask_to_old(n):                     # n: (64,) uint32 words of 16 sign bits, 16x16x32 order -> 32x32 order
    o = np.zeros_like(n)
    for mh in range(2):
        p, q = permlane16_swap((n >> (4 * mh)) & 0xF, (n >> (8 + 4 * mh)) & 0xF)
        a, b = permlane32_swap(p, q)
        o |= (a << (8 * mh)) | (b << (8 * mh + 4))
    return o


def mask_from_old(o):
    n = np.zeros_like(o)
    for mh in range(2):
        p, q = permlane32_swap((o >> (8 * mh)) & 0xF, (o >> (8 * mh + 4)) & 0xF)
        d0, d1 = permlane16_swap(p, q)
        n |= (d0 << (4 * mh)) | (d1 << (8 + 4 * mh))
    return n


def old_block(fn):
    """(64, 16) block in the 32x32 form: register t of lane (j, h) = fn(point j, slot acc_row(t, h))."""
    e = np.empty((64, 16), dtype=np.int64)
    for lane in range(64):
        j, h = lane & 31, lane >> 5
        for t in range(16):
            e[lane, t] = fn(j, acc_row(t, h))
    return e


def test_slot_numbering_and_packer_inverse():
    seen = set()
    for g in range(4):
        for jj in range(8):
            phi = m16_slot(g, jj)
            seen.add(phi)
            t, h = 4 * (phi >> 3) + (phi & 3), (phi >> 2) & 1       # xpack_fetch / bpack_layer_kernel / kilo packer
            assert 0 <= t < 16 and acc_row(t, h) == phi
            # register 8 nb + 4 mh + r with 4 mh + r == jj is this slot: feature 16 mh + 4 g + r
            mh, r = jj >> 2, jj & 3
            assert phi == 16 * mh + 4 * g + r
    assert seen == set(range(32))


def test_to_m16_is_the_documented_permutation():
    e = old_block(lambda point, slot: 100 * point + slot)
    n = to_m16(e)
    for lane in range(64):
        c, g = lane & 15, lane >> 4
        for nb in range(2):
            for mh in range(2):
                for r in range(4):
                    assert n[lane, 8 * nb + 4 * mh + r] == 100 * (16 * nb + c) + 16 * mh + 4 * g + r
    # ... i.e. register 4 (2 mh + (g >> 1)) + r of lane 32 (g & 1) + 16 nb + c (the comment in m16.h)
    for lane in range(64):
        c, g = lane & 15, lane >> 4
        for nb in range(2):
            for mh in range(2):
                for r in range(4):
                    assert n[lane, 8 * nb + 4 * mh + r] == e[32 * (g & 1) + 16 * nb + c, 4 * (2 * mh + (g >> 1)) + r]


def test_b_fragment_order():
    """The eight registers of one point half, in order, are the k-values m16_slot(g, 0..7): what the unchanged operand split
    (register pair q -> dword q % 4 of fragment q / 4) hands the MFMA matches what the packer put into the A fragment."""
    n = to_m16(old_block(lambda point, slot: slot))
    for lane in range(64):
        g = lane >> 4
        for nb in range(2):
            assert [int(v) for v in n[lane, 8 * nb:8 * nb + 8]] == [m16_slot(g, jj) for jj in range(8)]


def test_mask_words_round_trip_and_match_the_register_move():
    rng = np.random.default_rng(0)
    bits_old = rng.integers(0, 2, size=(64, 16))
    bits_new = to_m16(bits_old)
    word = lambda b: (b.astype(np.uint32) << np.arange(16, dtype=np.uint32)).sum(-1).astype(np.uint32)
    assert np.array_equal(mask_from_old(word(bits_old)), word(bits_new))
    assert np.array_equal(mask_to_old(word(bits_new)), word(bits_old))


def test_saved_rows_cover_the_block_once():
    def off(m16, t):                    # m16.h: saved_row_off
        return (16 * ((t >> 2) & 1) + (t & 3)) * 32 + 16 * (t >> 3) if m16 else ((t & 3) + 8 * (t >> 2)) * 32
    for m16 in (False, True):
        hit = np.zeros(32 * 32, dtype=int)
        for lane in range(64):
            base = 4 * (lane >> 4) * 32 + (lane & 15) if m16 else 4 * (lane >> 5) * 32 + (lane & 31)
            for t in range(16):
                hit[base + off(m16, t)] += 1
        assert (hit == 1).all()
    # and the same (row, point) gets the same value in both forms
    e = old_block(lambda point, slot: 100 * point + slot)
    n = to_m16(e)
    mem_old, mem_new = np.zeros(1024, dtype=np.int64), np.zeros(1024, dtype=np.int64)
    for lane in range(64):
        for t in range(16):
            mem_old[4 * (lane >> 5) * 32 + (lane & 31) + off(False, t)] = e[lane, t]
            mem_new[4 * (lane >> 4) * 32 + (lane & 15) + off(True, t)] = n[lane, t]
    assert np.array_equal(mem_old, mem_new)
