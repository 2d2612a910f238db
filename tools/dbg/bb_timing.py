"""phase timing of k_backbone<0>: run with FCZ_HIP_LIB pointing at a library built with -DFCZ_BB_TIMING
(wavefront-cycles in the kernel's parts, summed over all wavefronts; tools/dbg, not part of the product)"""
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
lib.fcz_debug_bb_timing.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 8)()
w.compress(); w.decompress(); codec.synchronize()
lib.fcz_debug_bb_timing(buf)
reps = 3
for _ in range(reps):
    w.decompress()
codec.synchronize()
lib.fcz_debug_bb_timing(buf)
res = w.R * reps / 64       # residue-steps per wavefront lane group
names = ["prologue (header, first anchor, first words)", "segment top (next anchor, ring head)", "forward pass", "reverse pass + blend (without flushes)", "window flushes"]
tot = sum(buf[i] for i in range(5))
print(f"{C} chains; wavefront-cycles per residue (64 chains in lock step), share")
for i, n in enumerate(names):
    print(f"  {n:50s} {buf[i] / res:10.0f}  {100.0 * buf[i] / tot:5.1f} %")
print(f"  {'total per residue':50s} {tot / res:10.0f}")
