#!/usr/bin/env python3
"""Where the wavefronts of k_ingest_parse spend their cycles: build/lib_igt.so = the library built with -DFCZ_IG_TIMING.
usage (GPU box): FCZ_HIP_LIB=$PWD/build/lib_igt.so python tools/dbg/ig_timing.py [files]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from foldcomp_amd.codec import Codec
from foldcomp_amd import synthetic
sys.path.insert(0, os.path.join(ROOT, "oracle"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
c = Codec(0)
b = synthetic.to_chain_batch(synthetic.generate(64, [350] * 64, seed=3))
blob, off, st = c.compress_batch(b)
texts, _ = c.decompress_pdb(blob, off)
texts = [texts[i % 64] for i in range(n)]
names = [f"s{i:06d}.pdb" for i in range(n)]
out = (ctypes.c_ulonglong * 8)()
c.ingest_pdb(texts, names)
c.lib.fcz_debug_ig_timing(out)
c.ingest_pdb(texts, names)
c.lib.fcz_debug_ig_timing(out)
v = list(out); tot = sum(v) or 1
for name, x in zip(("setup + tables", "stage chunk (global -> LDS)", "line-end scan", "line table", "lines (parse, keep, store)", "title + tail"), v):
    print(f"{name:32s} {x / tot * 100:5.1f} %   {x / n / 1e3:8.1f} kcycles per file")
