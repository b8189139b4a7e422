"""Numerics of the kernels' positional encoding, restated in numpy (no GPU, no library).

nerf_amd/csrc/mlp_chain.h evaluates sin / cos with its own Cody-Waite reduction + minimax polynomials
(`trig_reduce` / `trig_finish`) and, in the KiloNeRF and bf16 kernels, takes the odd octaves from the even ones by
one double-angle step (`encode_pe_doubling`).  This file restates both forms operation by operation in fp32 and
pins the error bounds the header quotes, over the argument range the scene produces (|x| <= far = 6, octaves
2^0..2^9 as in src/models/encoding/freq.py:14-29), plus the slot -> reference-column map of the packed weights.
"""
import numpy as np

f32 = np.float32


def _bits(u):
    return np.array([u], dtype=np.uint32).view(np.float32)[0]


def _fma(a, b, c):
    # exact product in fp64 (24 x 24 bits), one fp64 add, one rounding to fp32: an fp32 fma up to rare double rounding
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def kernel_sincos(a):
    a = a.astype(f32)
    c = lambda v: np.full_like(a, f32(v))
    k = np.rint((a * f32(0.63661977236758134308)).astype(f32)).astype(f32)
    r = _fma(k, c(_bits(0xBFC90FDA)), a)
    r = _fma(k, c(_bits(0xB3A22168)), r)
    r = _fma(k, c(_bits(0xA7C234C4)), r)
    z = (r * r).astype(f32)
    sp = _fma(_fma(c(-1.9515295891e-4), z, c(8.3321608736e-3)), z, c(-1.6666654611e-1))
    cp = _fma(_fma(c(2.443315711809948e-5), z, c(-1.388731625493765e-3)), z, c(4.166664568298827e-2))
    s = _fma((sp * z).astype(f32), r, r)
    co = _fma((cp * z).astype(f32), z, _fma(c(-0.5), z, c(1.0)))
    ki = k.astype(np.int64)

    def finish(h):
        q = ki + h
        v = np.where(q & 1, co, s)
        return np.where(q & 2, -v, v).astype(f32)
    return finish(0), finish(1)


def double_angle(s, c):
    d = (s + s).astype(f32)
    return (d * c).astype(f32), _fma(-d, s, np.ones_like(s))


def _args(octave, n=400_000, seed=0):
    x = np.random.default_rng(seed + octave).uniform(-6, 6, n).astype(f32)
    x[:4] = [0.0, 6.0, -6.0, 1e-30]
    return (x * f32(2 ** octave)).astype(f32)


def test_direct_polynomial_error():
    for o in range(10):
        a = _args(o)
        s, c = kernel_sincos(a)
        assert np.abs(s - np.sin(a.astype(np.float64))).max() < 1.0e-7, o
        assert np.abs(c - np.cos(a.astype(np.float64))).max() < 1.0e-7, o


def test_double_angle_step_error():
    for o in range(0, 10, 2):                      # even octaves are evaluated, the next odd one follows
        a = _args(o, seed=100)
        s2, c2 = double_angle(*kernel_sincos(a))
        a2 = a.astype(np.float64) * 2
        es, ec = np.abs(s2 - np.sin(a2)), np.abs(c2 - np.cos(a2))
        assert es.max() < 2.0e-7 and ec.max() < 3.2e-7, (o, es.max(), ec.max())
        assert np.sqrt((es ** 2).mean()) < 5e-8 and np.sqrt((ec ** 2).mean()) < 8e-8


def _pe_col(n, h, L):                              # nrf_common.h: pe_col
    if n < 3 * L:
        return 3 + 6 * (n // 3) + 3 * h + (n % 3)
    if n == 3 * L:
        return h
    if n == 3 * L + 1:
        return -1 if h else 2
    return -1


def test_doubling_slots_cover_the_reference_columns():
    """encode_pe_doubling writes slot 6g + c (octave 2g) and 6g + 3 + c (octave 2g + 1) for exact argument
    e = 3g + c: together with the raw-input slots every reference column 0..3+6L-1 appears exactly once."""
    for L in (10, 4):
        slots = set()
        for e in range(3 * (L // 2)):
            g, c = divmod(e, 3)
            slots |= {6 * g + c, 6 * g + 3 + c}
        assert slots == set(range(3 * L))
        cols = sorted(_pe_col(n, h, L) for n in list(slots) + [3 * L, 3 * L + 1] for h in (0, 1))
        assert cols == [-1] + list(range(3 + 6 * L))
        for n in slots:                            # slot n is (coordinate n % 3, octave n // 3): sin | cos
            assert _pe_col(n, 0, L) == 3 + 6 * (n // 3) + (n % 3)
            assert _pe_col(n, 1, L) == _pe_col(n, 0, L) + 3
