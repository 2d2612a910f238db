#!/usr/bin/env python3
"""Mint the ingest / database-container goldens from the REAL reference (oracle/_ref/libfoldcomp_ref.so: StructureReader with
the vendored gemmi, removeAlternativePosition / identifyChains / identifyDiscontinousResInd, make_writer / make_reader) and
from the data files the reference's own tests hold. Runs in the build container only.

Output: tests/golden/reference_ingest.npz (data only)
  file:<name>            the reference's test data files, byte for byte (test.pdb, test_af.pdb, multichain.pdb, test.cif.gz,
                         example_db + .index / .lookup / .dbtype)
  ingest:<name>/...      what the reference's reader makes of the file: atom table after removeAlternativePosition (names,
                         residue names, chain ids, serials, residue numbers, float32 coordinates and B-factors), the structure
                         title, the fragment ranges the compress lambda hands to the codec (src/main.cpp:457-474)
  cif:test/...           test.cif.gz through the reference codec (FCZ bytes, decompressed coordinates in both atom orders):
                         the case behind the second RMSD pin of the reference's build.sh
  dbw:...                the reference writer's output (data / .index / .lookup / .dbtype) for the example_db entries appended
                         in a scrambled key order
"""
import os, sys, tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _harness as H
from foldcomp_amd.structure import Chain, build_batch

REF_TEST = "/root/reference/test"
out = {}


def put_bytes(key, b):
    out[key] = np.frombuffer(b, np.uint8)


def names(strs, w):
    a = np.zeros((len(strs), w), np.uint8)
    for i, s in enumerate(strs):
        e = s.encode()[:w]; a[i, :len(e)] = np.frombuffer(e, np.uint8)
    return a


for fn in ("test.pdb", "test_af.pdb", "multichain.pdb", "test.cif.gz"):
    data = open(os.path.join(REF_TEST, fn), "rb").read()
    put_bytes(f"file:{fn}", data)
    t, title, frag, nch = H.ref_load_structure(data, fn)
    k = f"ingest:{fn}"
    out[f"{k}/atom"] = names(t.atom, 4); out[f"{k}/residue"] = names(t.residue, 3)
    out[f"{k}/chain"] = np.asarray([ord(c) for c in t.chain], np.uint8)
    out[f"{k}/atom_index"] = t.atom_index; out[f"{k}/res_index"] = t.res_index
    out[f"{k}/xyz"] = t.xyz; out[f"{k}/bfac"] = t.bfac
    put_bytes(f"{k}/title", title.encode("latin-1"))
    out[f"{k}/frag"] = np.asarray(frag, np.int32).reshape(-1, 4)
    out[f"{k}/n_chains"] = np.asarray([nch], np.int32)
    print(fn, len(t), "atoms,", len(frag), "fragments, title", repr(title))

# ---- test.cif.gz through the reference codec (title rule of src/main.cpp:465: the structure's own title is kept) ----
data = open(os.path.join(REF_TEST, "test.cif.gz"), "rb").read()
t, title, frag, nch = H.ref_load_structure(data, "test.cif.gz")
assert len(frag) == 1
ft = t.take(slice(frag[0][0], frag[0][1]))
b = build_batch([Chain(title, ft)])
fcz = H.mask_pad(H.ref_compress(ft, title, 25))
for kk in ("res_off", "atom_off", "x", "y", "z", "atom_code", "res_code", "bfac_ca", "first_res_index", "first_atom_index", "chain_id", "titles", "title_off"):
    out[f"cif:test/in/{kk}"] = getattr(b, kk)
out["cif:test/in/anchor_threshold"] = np.asarray([25], np.int32)
put_bytes("cif:test/fcz", fcz)
for alt in (0, 1):
    d = H.ref_decompress(fcz, bool(alt))
    out[f"cif:test/xyz{alt}"] = np.stack([d["x"], d["y"], d["z"]], 1)
# the reference's own pins (build.sh:35,37): all-atom RMSD, atoms paired in file order, float accumulation
for fn, alt, pin in (("test.pdb", 0, 0.0826751), ("test.cif.gz", 1, 0.130284)):
    tt, ti, fr, _ = H.ref_load_structure(open(os.path.join(REF_TEST, fn), "rb").read(), fn)
    f = H.ref_compress(tt.take(slice(fr[0][0], fr[0][1])), ti, 25)
    d = H.ref_decompress(f, bool(alt))
    got = np.stack([d["x"], d["y"], d["z"]], 1)
    rmsd = float(np.sqrt(((got.astype(np.float64) - tt.xyz[fr[0][0]:fr[0][1]].astype(np.float64)) ** 2).sum(1).mean()))
    print(fn, "reference round-trip RMSD", rmsd, "pin", pin)
    assert abs(rmsd - pin) < 1e-3

# ---- database container ----
for suffix in ("", ".index", ".lookup", ".dbtype"):
    put_bytes(f"file:example_db{suffix}", open(os.path.join(REF_TEST, "example_db" + suffix), "rb").read())
rows = H.ref_db_read(os.path.join(REF_TEST, "example_db"))
out["dbr:keys"] = np.asarray([r[0] for r in rows], np.int64)
out["dbr:offsets"] = np.asarray([r[1] for r in rows], np.int64)
out["dbr:lengths"] = np.asarray([r[2] for r in rows], np.int64)
put_bytes("dbr:names", "\n".join(r[3] for r in rows).encode())
order = np.random.default_rng(11).permutation(len(rows))
with tempfile.TemporaryDirectory() as tmp:
    H.ref_db_write(tmp + "/w", [rows[i][4] for i in order], [rows[i][0] for i in order], [rows[i][3] for i in order])
    out["dbw:order"] = order.astype(np.int64)
    for suffix in ("", ".index", ".lookup", ".dbtype"):
        put_bytes(f"dbw:file{suffix}", open(tmp + "/w" + suffix, "rb").read())
path = os.path.join(ROOT, "tests", "golden", "reference_ingest.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes")
