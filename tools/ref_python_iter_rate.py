#!/usr/bin/env python3
"""What a program written against the reference's Python module waits for per entry: `for name, pdb in foldcomp.open(db)` is one
Foldcomp::read + Foldcomp::decompress + writeAtomCoordinatesToPDB per entry (foldcomp/foldcomp.cxx:44-90, :197-250). The module
itself is a CPython extension that this image cannot pip-build; its per-entry C++ work is exactly oracle/_ref's fcz_ref_decompress_pdb
(oracle/ref_shim.cpp), so the loop below -- one ctypes call per entry, single thread like the module -- is the reference's rate minus
its Python object construction (i.e. an upper bound of it). Run in the BUILD container (needs /root/reference via oracle/_ref):
  python tools/ref_python_iter_rate.py [--entries 2000] [--residues 350]  -> one JSON line (committed as profiles/r6_python_iter_reference.json)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--entries", type=int, default=2000)
    ap.add_argument("--residues", type=int, default=350)
    a = ap.parse_args()
    import numpy as np
    import _harness as H
    from foldcomp_amd import synthetic
    b = synthetic.to_chain_batch(synthetic.generate(a.entries, [a.residues] * a.entries, seed=5))
    blob, off, st = H.oracle_compress(b, n_threads=os.cpu_count() or 1)
    assert (st == 0).all()
    recs = [blob[int(off[i]):int(off[i + 1])].tobytes() for i in range(a.entries)]
    H.ref_decompress_pdb(recs[0])
    t0 = time.perf_counter()
    nbytes = 0
    for r in recs:
        nbytes += len(H.ref_decompress_pdb(r))
    dt = time.perf_counter() - t0
    print(json.dumps({"what": "per-entry loop over the reference's own C++ (oracle/_ref: Foldcomp::read + decompress + writeAtomCoordinatesToPDB), one thread, "
                              "one call per entry -- the work behind the reference module's FoldcompDatabase.__getitem__",
                      "entries": a.entries, "residues_per_entry": a.residues, "seconds": round(dt, 3), "entries_per_s": round(a.entries / dt, 1),
                      "residues_per_s": round(a.entries * a.residues / dt), "text_MB_per_s": round(nbytes / dt / 1e6, 1),
                      "host": "build container (8 CPUs), " + os.uname().machine}))


if __name__ == "__main__":
    main()
