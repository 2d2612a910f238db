"""phase timing of k_sidechain: run with FCZ_HIP_LIB pointing at a library built with -DFCZ_SC_TIMING
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
lib.fcz_debug_sc_timing.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 12)()
w.compress(); w.decompress(); codec.synchronize()
lib.fcz_debug_sc_timing(buf)
reps = 3
for _ in range(reps):
    w.decompress()
codec.synchronize()
lib.fcz_debug_sc_timing(buf)
tiles = (w.R + 255) // 256 * reps * 4      # wavefront-tiles
names = ["tile top: next tile's loads issued, this tile's data waited for", "thread = residue: N, CA, C staged, O placed", "wave scan + barrier",
         "list build", "barrier after list build", "depth rounds: items", "depth rounds: barriers", "write-back stores issued", "barrier after write-back"]
tot = sum(buf[i] for i in range(9))
print(f"{C} chains, {tiles} wave-tiles; wavefront-cycles per tile, share")
for i, n in enumerate(names):
    print(f"  {n:66s} {buf[i] / tiles:10.0f}  {100.0 * buf[i] / tot:5.1f} %")
print(f"  {'total per tile':66s} {tot / tiles:10.0f}")
