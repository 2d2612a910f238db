#!/usr/bin/env python3
"""k_ingest_parse_cif's time under library variants: FCZ_HIP_LIB=<lib> python tools/dbg/cif_ab.py [files] -> ms of the kernel over
`files` AFDB-shaped mmCIF files of 350 residues (the e2e leg's shape), checked against the PDB-text ingest of the same chains."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from foldcomp_amd.codec import Codec
from foldcomp_amd import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
c = Codec(0)
b = synthetic.to_chain_batch(synthetic.generate(64, [350] * 64, seed=3))
blob, off, st = c.compress_batch(b)
texts, _ = c.decompress_pdb(blob, off)
cifs = [bench.cif_from_pdb_text(t, f"S{i}") for i, t in enumerate(texts)]
tc = [cifs[i % 64] for i in range(n)]; tp = [texts[i % 64] for i in range(n)]
names = [f"s{i:06d}.cif" for i in range(n)]
ref = c.ingest_pdb(tp[:64], [f"s{i}.pdb" for i in range(64)])
c.ingest_pdb(tc, names)
c.enable_timing(True); c.reset_timing()
for _ in range(3):
    r = c.ingest_pdb(tc, names)
assert (r[3] == 0).all(), "files were handed back"
ms, k = c.kernel_time("ingest_parse_cif")
ms2, k2 = c.kernel_time("ingest_rows_cif")             # (0 in builds before the rows had a kernel of their own)
na = int(ref[0].atom_off[-1])
same = all(np.array_equal(np.asarray(getattr(r[0], f))[:na], np.asarray(getattr(ref[0], f))[:na]) for f in ("x", "y", "z", "atom_code")) \
    and np.array_equal(np.asarray(r[0].bfac_ca)[: ref[0].n_residues], np.asarray(ref[0].bfac_ca)) and np.array_equal(np.asarray(r[0].first_atom_index)[:64], np.asarray(ref[0].first_atom_index))
print(os.environ.get("FCZ_HIP_LIB", "default"), "k_ingest_parse_cif ms per call = %.3f" % (ms / max(k, 1)), "k_ingest_rows_cif = %.3f" % (ms2 / max(k2, 1)), "files", n, "bytes", sum(len(t) for t in tc), "same_as_pdb_text", same)
