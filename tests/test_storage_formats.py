"""CPU: the 16-bit DDF storage formats of the oracle against numpy's IEEE half and against their definition."""
import ctypes as C

import numpy as np

from oracle import harness as H

lib = C.CDLL(H.PORT_LIB)
lib.oracle_half_to_float.restype = C.c_float
lib.oracle_half_to_float.argtypes = [C.c_uint16]
lib.oracle_float_to_half.restype = C.c_uint16
lib.oracle_float_to_half.argtypes = [C.c_float]
lib.oracle_fp16c_to_float.restype = C.c_float
lib.oracle_fp16c_to_float.argtypes = [C.c_uint16]
lib.oracle_float_to_fp16c.restype = C.c_uint16
lib.oracle_float_to_fp16c.argtypes = [C.c_float]


def test_half_decode_matches_numpy_for_all_65536_codes():
    codes = np.arange(65536, dtype=np.uint16)
    want = codes.view(np.float16).astype(np.float32)
    got = np.array([lib.oracle_half_to_float(int(c)) for c in codes], dtype=np.float32)
    finite = np.isfinite(want)
    assert np.array_equal(got[finite].view(np.uint32), want[finite].view(np.uint32))
    assert np.isnan(got[np.isnan(want)]).all() and np.array_equal(got[np.isinf(want)], want[np.isinf(want)])


def test_half_encode_is_round_to_nearest_even():
    rng = np.random.RandomState(7)
    x = np.concatenate([rng.uniform(-70000, 70000, 20000), rng.uniform(-1, 1, 20000) * 10.0 ** rng.uniform(-9, 0, 20000),
                        np.float32([0.0, -0.0, 65504.0, 65519.9, 65520.0, 2.0 ** -24, 2.0 ** -25, 2.0 ** -25 * 1.0000001, 2.0 ** -14])]).astype(np.float32)
    codes = np.arange(65536, dtype=np.uint16)  # exact halves and midpoints between neighbours
    h = codes.view(np.float16).astype(np.float32)
    h = h[np.isfinite(h)]
    mid = ((h[:-1].astype(np.float64) + h[1:].astype(np.float64)) / 2).astype(np.float32)
    x = np.concatenate([x, h, mid])
    with np.errstate(over="ignore"):
        want = x.astype(np.float16).view(np.uint16)
    got = np.array([lib.oracle_float_to_half(float(v)) for v in x], dtype=np.uint16)
    assert np.array_equal(got, want)


def test_fp16c_format_definition():
    # 1 sign, 4 exponent (bias 15), 11 mantissa bits; subnormals m*2^-25 (FluidX3D-master/src/kernel.cpp:858-863)
    for h in range(65536):
        s, e, m = h >> 15, (h >> 11) & 15, h & 0x7FF
        want = (m * 2.0 ** -25) if e == 0 else (1 + m / 2048.0) * 2.0 ** (e - 15)
        want = -want if s else want
        assert lib.oracle_fp16c_to_float(h) == np.float32(want)
    assert lib.oracle_fp16c_to_float(0x7FFF) == np.float32(1.99951171875)


def test_fp16c_round_trip_and_rounding():
    for h in list(range(0, 0x8000, 7)) + [0x7FFF, 0x0001, 0x07FF, 0x0800]:
        assert lib.oracle_float_to_fp16c(lib.oracle_fp16c_to_float(h)) == h  # representable values survive
    # between two normal neighbours the upper one wins from the exact midpoint on (add half, truncate)
    lo, hi = lib.oracle_fp16c_to_float(0x7800), lib.oracle_fp16c_to_float(0x7801)
    assert lib.oracle_float_to_fp16c(float(np.float32(lo + (hi - lo) * 0.49))) == 0x7800
    assert lib.oracle_float_to_fp16c(float(np.float32(lo + (hi - lo) * 0.5))) == 0x7801
    assert lib.oracle_float_to_fp16c(0.0) == 0 and lib.oracle_float_to_fp16c(1e-12) == 0
    assert lib.oracle_float_to_fp16c(-1.0) == 0x8000 | (15 << 11)
