#!/usr/bin/env python3
"""Where the wavefronts of k_ingest_parse_cif spend their cycles: build/lib_cift.so = the library built with
-DFCZ_IG_TIMING -DFCZ_CIF_TIMING (hipcc ... -DFCZ_IG_TIMING -DFCZ_CIF_TIMING -o build/lib_cift.so foldcomp_amd/csrc/fcz_abi.hip).
usage (GPU box): FCZ_HIP_LIB=$PWD/build/lib_cift.so python tools/dbg/cif_timing.py [files]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench
from foldcomp_amd.codec import Codec
from foldcomp_amd import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
c = Codec(0)
b = synthetic.to_chain_batch(synthetic.generate(64, [350] * 64, seed=3))
blob, off, st = c.compress_batch(b)
texts, _ = c.decompress_pdb(blob, off)
cifs = [bench.cif_from_pdb_text(t, f"S{i}") for i, t in enumerate(texts)]
texts = [cifs[i % 64] for i in range(n)]
names = [f"s{i:06d}.cif" for i in range(n)]
out = (ctypes.c_ulonglong * 8)()
c.ingest_pdb(texts, names)
c.lib.fcz_debug_ig_timing(out)
r = c.ingest_pdb(texts, names)
assert (r[3] == 0).all(), "files were handed back"
c.lib.fcz_debug_ig_timing(out)
v = list(out); tot = sum(v) or 1
for name, x in zip(("setup + tables", "stage chunk (global -> LDS)", "line-end scan + table", "blanks by dword", "token bounds", "token starts, quotes, class",
                    "grammar walk", "rows"), v):
    print(f"{name:32s} {x / tot * 100:5.1f} %   {x / n / 1e3:8.1f} kcycles per file")
