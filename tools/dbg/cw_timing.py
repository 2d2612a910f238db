"""phase timing of k_compress_angles_w: run with FCZ_HIP_LIB pointing at a library built with -DFCZ_CW_TIMING
(wavefront-cycles between the kernel's phase boundaries, summed over all wavefronts; tools/dbg, not part of the product)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from foldcomp_amd.codec import Codec

C = int(os.environ.get("CHAINS", 262144))
codec = Codec(0)
d = bench.generate_resident(C, 350, 25, 32768, "cuda:0", seed_base=1)
w = bench.Workload(codec, d, "cuda:0")
lib = codec.lib
lib.fcz_debug_cw_timing.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 8)()
w.compress(); codec.synchronize()
lib.fcz_debug_cw_timing(buf)
reps = 3
for _ in range(reps):
    w.compress()
codec.synchronize()
lib.fcz_debug_cw_timing(buf)
tiles = (w.R + 62) // 63 * reps
names = ["top: meta wait, issue next meta, punt test, wave_sync", "stage: atom/code loads -> LDS", "table row + item lists", "backbone items",
         "side-chain items", "stores issued"]
tot = sum(buf[i] for i in range(6))
print(f"{C} chains, {tiles} wave-tiles; wavefront-cycles per tile (s_memtime units), share")
for i, n in enumerate(names):
    print(f"  {n:58s} {buf[i] / tiles:10.0f}  {100.0 * buf[i] / tot:5.1f} %")
print(f"  {'total per tile':58s} {tot / tiles:10.0f}")
