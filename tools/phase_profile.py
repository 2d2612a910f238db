#!/usr/bin/env python3
"""debug: per-phase cycle split of k_compress_tiled (needs foldcomp_amd/libfcz_hip_prof.so built with -DFCZ_PROFILE_PHASES)"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from foldcomp_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "foldcomp_amd", "libfcz_hip_prof.so")
from foldcomp_amd import synthetic
from foldcomp_amd.codec import Codec
import bench
C = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
d = bench.generate_resident(C, 350, 25, 32768, "cuda:0", 0)
codec = Codec(0); lib = codec.lib
cb = bench.c_batch(d)
off = torch.zeros(C + 1, dtype=torch.int64, device="cuda:0")
torch.cuda.synchronize()
lib.fcz_compress_sizes_dev(codec.ctx, ctypes.byref(cb), off.data_ptr()); codec.synchronize()
blob = torch.zeros(int(off[-1]), dtype=torch.uint8, device="cuda:0"); st = torch.zeros(C, dtype=torch.int32, device="cuda:0")
torch.cuda.synchronize()
lib.fcz_debug_phase_cycles.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
out = (ctypes.c_ulonglong * 16)()
for it in range(2):
    lib.fcz_compress_batch_dev(codec.ctx, ctypes.byref(cb), off.data_ptr(), blob.data_ptr(), st.data_ptr())
    lib.fcz_debug_phase_cycles(codec.ctx, out, 1)
v = np.array(list(out), np.float64)[:9]
names = ["validate", "tile meta", "staging", "slot table", "anchors+scan", "items", "tile fence", "minmax+pack", "title+header"]
tot = v.sum()
for n, x in zip(names, v):
    print(f"{n:14s} {x / C:12.0f} ticks/chain  {100 * x / tot:5.1f}%")
print("total ticks/chain", tot / C)
