"""GPU: the device numerics (fcz_math.h) against the host libm, bit for bit.
 * acos_deg(c) == (float)(acos((double)c)*180.0/M_PI) on strided sweeps over every float in [-1,1]
 * sinf/cosf restatement == libm sinf/cosf on sweeps over |x| < 8"""
import ctypes
import os
import struct

import numpy as np
import pytest

import _harness as H

pytestmark = pytest.mark.gpu


def _b(f):
    return struct.unpack("<I", struct.pack("<f", f))[0]


def _host_acos(start, stride, count):
    lib = H.load_oracle()
    lib.fcz_oracle_acos_deg_sweep.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int]
    out = np.zeros(count, np.float32)
    lib.fcz_oracle_acos_deg_sweep(start, stride, count, out.ctypes.data, min(32, os.cpu_count() or 1))
    return out


def _host_sincos(is_cos, start, stride, count):
    lib = H.load_oracle()
    lib.fcz_oracle_sincos_sweep.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int]
    out = np.zeros(count, np.float32)
    lib.fcz_oracle_sincos_sweep(is_cos, start, stride, count, out.ctypes.data, min(32, os.cpu_count() or 1))
    return out


@pytest.mark.parametrize("sign", [0, 0x80000000])
def test_acos_deg_strided_sweep(codec, sign):
    one = _b(1.0)
    stride = 61                      # odd stride: hits every mantissa residue class over the sweep
    count = one // stride + 1
    dev = codec.selftest_math(0, sign, stride, count)
    host = _host_acos(sign, stride, count)
    a, b = dev.view(np.uint32), host.view(np.uint32)
    bad = np.nonzero(a != b)[0]
    assert len(bad) == 0, (len(bad), [(hex(sign + int(i) * stride), float(dev[i]), float(host[i])) for i in bad[:8]])


def test_acos_deg_dense_near_one(codec):
    """the steep end: every float in [0.999, 1] and [-1, -0.999] plus out-of-range / NaN inputs"""
    for sign in (0, 0x80000000):
        lo, hi = _b(0.999), _b(1.0)
        dev = codec.selftest_math(0, sign + lo, 1, hi - lo + 8)
        host = _host_acos(sign + lo, 1, hi - lo + 8)
        a, b = dev.view(np.uint32), host.view(np.uint32)
        nan = np.isnan(dev) & np.isnan(host)
        assert ((a == b) | nan).all()


@pytest.mark.parametrize("is_cos", [0, 1])
def test_sincos_sweep(codec, is_cos):
    for sign in (0, 0x80000000):
        stride = 13
        count = _b(8.0) // stride
        dev = codec.selftest_math(1 + is_cos, sign, stride, count)
        host = _host_sincos(is_cos, sign, stride, count)
        a, b = dev.view(np.uint32), host.view(np.uint32)
        bad = np.nonzero(a != b)[0]
        assert len(bad) == 0, (len(bad), [(hex(sign + int(i) * stride), float(dev[i]), float(host[i])) for i in bad[:8]])


@pytest.mark.parametrize("is_cos", [0, 1])
def test_sincos_pair_sweep(codec, is_cos):
    """the branch-free sincosf_pair the kernels call (one reduction for every |x|, both polynomials, quadrant select)
    against libm: every 7th float below 8 rad, and every float below 2^-10 (the |x| < 2^-12 early-out and its neighbourhood)"""
    for sign in (0, 0x80000000):
        for start, stride, count in ((0, 7, _b(8.0) // 7), (_b(2.0 ** -14), 1, _b(2.0 ** -10) - _b(2.0 ** -14)), (0, 1, 1 << 20)):
            dev = codec.selftest_math(9 + is_cos, sign + start, stride, count)
            host = _host_sincos(is_cos, sign + start, stride, count)
            a, b = dev.view(np.uint32), host.view(np.uint32)
            bad = np.nonzero(a != b)[0]
            assert len(bad) == 0, (len(bad), [(hex(sign + start + int(i) * stride), float(dev[i]), float(host[i])) for i in bad[:8]])


@pytest.mark.parametrize("is_cos", [0, 1])
def test_sincos_of_any_float(codec, is_cos):
    """sincosf_pair_any (the decoder's form for chains whose quantiser parameters can reach 120 radians: records made by hand or
    damaged) against libm over the whole float range, both signs: every 997th float from 0 to infinity (glibc's three
    reductions: none, fast, large), every float around 120.0 where they switch, the infinities and NaNs (NaN on both sides)"""
    for sign in (0, 0x80000000):
        for start, stride, count in ((0, 997, 0x7f800000 // 997 + 1), (_b(119.99), 1, _b(120.01) - _b(119.99)), (_b(3e38), 1, 0x7f800010 - _b(3e38)), (0x7fc00000, 1, 4)):
            dev = codec.selftest_math(12 + is_cos, sign + start, stride, count)
            host = _host_sincos(is_cos, sign + start, stride, count)
            a, b = dev.view(np.uint32), host.view(np.uint32)
            bad = np.nonzero((a != b) & ~(np.isnan(dev) & np.isnan(host)))[0]
            assert len(bad) == 0, (len(bad), [(hex(sign + start + int(i) * stride), float(dev[i]), float(host[i])) for i in bad[:8]])


def _host_math(mode, start, stride, count):
    lib = H.load_oracle()
    lib.fcz_oracle_math_sweep.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int]
    out = np.zeros(count, np.float32)
    lib.fcz_oracle_math_sweep(mode, start, stride, count, out.ctypes.data, min(32, os.cpu_count() or 1))
    return out


def _same(dev, host):
    a, b = dev.view(np.uint32), host.view(np.uint32)
    return (a == b) | (np.isnan(dev) & np.isnan(host))


def test_deg2rad_sweep(codec):
    """every 7th float with |x| < 400 degrees, both signs"""
    for sign in (0, 0x80000000):
        stride = 7
        count = _b(400.0) // stride
        dev = codec.selftest_math(3, sign, stride, count)
        host = _host_math(3, sign, stride, count)
        bad = np.nonzero(~_same(dev, host))[0]
        assert len(bad) == 0, (len(bad), [hex(sign + int(i) * stride) for i in bad[:8]])


@pytest.mark.parametrize("mode", [4, 5, 6, 7, 8])
def test_hashed_geometry_functions(codec, mode):
    """norm / getCosineTheta / place_atom on 2^26 hashed inputs: device fast paths == host reference order"""
    count = 1 << 26
    dev = codec.selftest_math(mode, 12345, 1, count)
    host = _host_math(mode, 12345, 1, count)
    bad = np.nonzero(~_same(dev, host))[0]
    assert len(bad) == 0, (mode, len(bad), [(int(i), float(dev[i]), float(host[i])) for i in bad[:8]])


def test_acos_deg_f32_error_bound(codec):
    """the float acos behind the side-chain torsion byte stays within 1e-4 degrees of the exact acos_deg on every 61st float
    of (-1, 1) and on every float of the steep ends [0.999, 1): the kernel's guard band is 5e-4 degrees"""
    worst = 0.0
    for sign in (0, 0x80000000):
        one = _b(1.0)
        for start, stride, count in ((0, 61, one // 61), (_b(0.999), 1, one - _b(0.999))):
            approx = codec.selftest_math(11, sign + start, stride, count).astype(np.float64)
            exact = codec.selftest_math(0, sign + start, stride, count).astype(np.float64)
            worst = max(worst, float(np.abs(approx - exact).max()))
    assert worst < 1e-4, worst
