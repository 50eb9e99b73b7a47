"""GPU: the DDF storage conversions the lattice kernels use (through lbmhip_convert_ddfs) against the reference's
definitions, bit for bit.  FP16C: decode of all 65536 codes against the oracle; encode of every sign/exponent with a dense
mantissa sample and of EVERY float in the subnormal range of the format against the formula of
FluidX3D-master/src/kernel.cpp:864-869 restated in numpy (itself pinned to the oracle's C restatement on a sample), both
for one value at a time and for the packed two-at-a-time form.  LBM_FULL_CODEC=1 sweeps all 2^32 floats."""
import ctypes as C
import os

import numpy as np
import pytest

import cfd_amd
from oracle import harness as H

pytestmark = pytest.mark.gpu

oracle = C.CDLL(H.PORT_LIB)
oracle.oracle_fp16c_to_float.restype = C.c_float
oracle.oracle_fp16c_to_float.argtypes = [C.c_uint16]
oracle.oracle_float_to_fp16c.restype = C.c_uint16
oracle.oracle_float_to_fp16c.argtypes = [C.c_float]
oracle.oracle_half_to_float.restype = C.c_float
oracle.oracle_half_to_float.argtypes = [C.c_uint16]
oracle.oracle_float_to_half.restype = C.c_uint16
oracle.oracle_float_to_half.argtypes = [C.c_float]


def convert(storage, to_storage, pairs, a, out_dtype):
    a = np.ascontiguousarray(a)
    out = np.empty(a.size, dtype=out_dtype)
    cfd_amd.lbm.check(cfd_amd.lib().lbmhip_convert_ddfs(0, cfd_amd.STORAGE[storage], int(to_storage), int(pairs), a.ctypes.data_as(C.c_void_p),
                                                         out.ctypes.data_as(C.c_void_p), C.c_uint64(a.size)))
    return out


def fp16c_encode_reference(bits):
    """kernel.cpp:864-869 on an array of FP32 bit patterns"""
    b = (bits.astype(np.uint64) + 0x800).astype(np.uint32)
    e = ((b & 0x7F800000) >> 23).astype(np.int64)
    m = (b & 0x007FFFFF).astype(np.int64)
    normal = (((e - 112) << 11) & 0x7800) | (m >> 12)
    shift = np.clip(124 - e, 0, 31)
    sub = (((0x007FF800 + m) >> shift) + 1) >> 1
    r = ((b.astype(np.int64) & 0x80000000) >> 16) | np.where(e > 112, normal, np.where(e > 100, sub, 0))
    return r.astype(np.uint16)


def test_fp16c_decode_all_codes(gpu):
    codes = np.arange(65536, dtype=np.uint16)
    want = np.array([oracle.oracle_fp16c_to_float(int(c)) for c in codes], dtype=np.float32)
    for pairs in (0, 1, 2):
        got = convert("FP16C", 0, pairs, codes, np.float32)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "pairs=%d" % pairs


def test_fp16s_both_directions(gpu):
    codes = np.arange(65536, dtype=np.uint16)
    finite = np.isfinite(codes.view(np.float16))
    want = (codes.view(np.float16).astype(np.float32) * np.float32(3.0517578E-5))
    rng = np.random.RandomState(3)
    x = np.concatenate([rng.uniform(-2, 2, 1 << 16), rng.uniform(-1, 1, 1 << 16) * 10.0 ** rng.uniform(-12, 0, 1 << 16), want[finite]]).astype(np.float32)
    with np.errstate(over="ignore"):
        want_codes = (x * np.float32(32768.0)).astype(np.float16).view(np.uint16)
    for pairs in (0, 1, 2):
        got = convert("FP16S", 0, pairs, codes, np.float32)
        assert np.array_equal(got[finite].view(np.uint32), want[finite].view(np.uint32))
        got = convert("FP16S", 1, pairs, x, np.uint16)
        bad = np.flatnonzero(got != want_codes)
        assert bad.size == 0, "pairs=%d: %d mismatches, first x=%r (bits 0x%08x): got 0x%04x, want 0x%04x" % (
            pairs, bad.size, float(x[bad[0]]), x.view(np.uint32)[bad[0]], got[bad[0]], want_codes[bad[0]])


def check_encode(bits):
    want = fp16c_encode_reference(bits)
    finite = (bits & 0x7F800000) != 0x7F800000  # the payload of a NaN is not preserved; DDFs are finite
    for pairs in (0, 1, 2):
        got = convert("FP16C", 1, pairs, bits.view(np.float32), np.uint16)
        bad = np.flatnonzero((got != want) & finite)
        assert bad.size == 0, "pairs=%d: %d mismatches, first at bits 0x%08x: got 0x%04x, want 0x%04x" % (pairs, bad.size, bits[bad[0]], got[bad[0]], want[bad[0]])


def test_fp16c_encode_reference_formula_is_the_oracles(gpu):
    rng = np.random.RandomState(5)
    bits = np.concatenate([rng.randint(0, 1 << 32, 20000, dtype=np.uint64).astype(np.uint32),
                           (rng.randint(99 << 23, 116 << 23, 20000, dtype=np.uint64).astype(np.uint32))])
    bits = bits[(bits & 0x7F800000) != 0x7F800000]
    want = np.array([oracle.oracle_float_to_fp16c(float(v)) for v in bits.view(np.float32)], dtype=np.uint16)
    assert np.array_equal(fp16c_encode_reference(bits), want)


def test_fp16c_encode_every_exponent(gpu):
    top = np.arange(2048, dtype=np.uint32) << 12
    low = np.array([0, 1, 0x7FE, 0x7FF, 0x800, 0x801, 0xFFE, 0xFFF], dtype=np.uint32)
    mant = (top[:, None] | low[None, :]).ravel()
    se = np.arange(512, dtype=np.uint32) << 23
    check_encode((se[:, None] | mant[None, :]).ravel())


def test_fp16c_encode_whole_subnormal_range(gpu):
    # every float from 2^-28 up to 2^-12, both signs: below, inside and just above the range where the format is subnormal
    mant = np.arange(1 << 23, dtype=np.uint32)
    for e in range(99, 116):
        for sign in (0, 0x80000000):
            check_encode(mant | np.uint32(e << 23) | np.uint32(sign))


@pytest.mark.skipif(os.environ.get("LBM_FULL_CODEC") != "1", reason="all 2^32 floats: set LBM_FULL_CODEC=1 (about five minutes)")
def test_fp16c_encode_all_floats(gpu):
    chunk = 1 << 26
    for start in range(0, 1 << 32, chunk):
        check_encode(np.arange(start, start + chunk, dtype=np.uint64).astype(np.uint32))
