"""CPU: pin the restated glibc sinf/cosf (oracle + the same algorithm in fcz_math.h) against the host
libm. Every angle the codec converts to radians lies in |x| <= 2*pi (+ rounding), far below 17.27 rad,
the first float where glibc's FMA and non-FMA ifunc variants disagree (measured exhaustively over all
1.12e9 floats |x| < 120)."""
import ctypes
import os
import struct

import _harness as H


def _bits(f):
    return struct.unpack("<I", struct.pack("<f", f))[0]


def test_restated_sincos_matches_libm_all_floats_below_8():
    lib = H.load_oracle()
    lib.fcz_oracle_trig_mismatches.restype = ctypes.c_long
    lib.fcz_oracle_trig_mismatches.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int]
    nt = max(1, min(8, os.cpu_count() or 1))
    # all floats with 2^-14 <= |x| < 8 (both signs); smaller |x| return x / 1 exactly
    lo, hi = _bits(2.0 ** -14), _bits(8.0)
    assert lib.fcz_oracle_trig_mismatches(lo, hi, nt) == 0
    assert lib.fcz_oracle_trig_mismatches(0, 4096, 1) == 0
