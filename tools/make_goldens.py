#!/usr/bin/env python3
"""Mint golden vectors from the REAL reference (oracle/_ref/libfoldcomp_ref.so, built from
/root/reference/src by oracle/build_ref.sh) and from the reference's own test fixtures.

Runs in the build container only. Output: tests/golden/reference_vectors.npz (data only: inputs as SoA
arrays and the reference's outputs; no reference source text).

Cases
  pdb:<name>      reference fixtures test/test.pdb, test/test_af.pdb, test/multichain.pdb fragments,
                  compressed with the reference, then decompressed with the reference (both atom orders)
  db:<i>          the 24 FCZ entries of test/example_db decompressed with the reference
  syn:<i>         seeded synthetic chains covering edge cases (n < 25, n == 25, n == 26, long chain,
                  no OXT, PRO-rich, GLY-only, missing side-chain atoms, constant B-factor, UNK residue,
                  shuffled atom order, anchor threshold 10 / 200)
  fixture:test_af.fcz   the committed reference output test/test_af.fcz (header floats are
                  platform-variant, see SURVEY.md §4: only its packed words/side-chain/temp bytes are pinned)
"""
import os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _harness as H
from foldcomp_amd import synthetic
from foldcomp_amd._aa_tables import ATOM_NAMES, RES3
from foldcomp_amd.structure import (AtomTable, Chain, build_batch, identify_chains, identify_discontinuous, parse_pdb,
                                    remove_alternative_position)

REF_TEST = "/root/reference/test"
out = {}
index = []


def chain_table(b, c):
    """AtomTable of chain c of a ChainBatch (to feed the reference shim)."""
    r0, r1 = b.res_off[c], b.res_off[c + 1]; a0, a1 = b.atom_off[r0], b.atom_off[r1]
    xyz = np.stack([b.x, b.y, b.z], 1)
    atom = [ATOM_NAMES[k] if k < 37 else "H" for k in b.atom_code[a0:a1]]
    residx = np.zeros(a1 - a0, np.int32); res = []; bf = np.zeros(a1 - a0, np.float32)
    for r in range(r0, r1):
        s, e = b.atom_off[r] - a0, b.atom_off[r + 1] - a0
        residx[s:e] = b.first_res_index[c] + (r - r0)
        res += [RES3[b.res_code[r]]] * (e - s)
        bf[s:e] = b.bfac_ca[r]
    return AtomTable(atom, res, [chr(b.chain_id[c])] * (a1 - a0), np.arange(a1 - a0, dtype=np.int32) + b.first_atom_index[c],
                     residx, xyz[a0:a1].copy(), bf)


def add_batch_case(name, b, thr=25):
    """one-chain batch -> reference FCZ + reference decompressed coordinates"""
    assert b.n_chains == 1
    t = chain_table(b, 0)
    title = bytes(b.titles).decode("latin-1")
    fcz = H.mask_pad(H.ref_compress(t, title, thr))
    for k in ("res_off", "atom_off", "x", "y", "z", "atom_code", "res_code", "bfac_ca", "first_res_index", "first_atom_index",
              "chain_id", "titles", "title_off"):
        out[f"{name}/in/{k}"] = getattr(b, k)
    out[f"{name}/in/anchor_threshold"] = np.asarray([thr], np.int32)
    out[f"{name}/fcz"] = np.frombuffer(fcz, np.uint8)
    has_unk = bool((b.res_code > 19).any())
    for alt in ((0,) if has_unk else (0, 1)):
        d = H.ref_decompress(fcz, bool(alt))
        out[f"{name}/xyz{alt}"] = np.stack([d["x"], d["y"], d["z"]], 1)
    out[f"{name}/bfac"] = d["bfac"]
    a = H.ref_angles(t)
    for k, v in a.items():
        out[f"{name}/angle/{k}"] = v
    add_text_goldens(name, fcz)
    index.append(name)


def add_text_goldens(name, fcz):
    """reference PDB text of the decompressed record and the `extract` strings"""
    out[f"{name}/pdb0"] = np.frombuffer(H.ref_decompress_pdb(fcz, False).encode("latin-1"), np.uint8)
    for dgt in (1, 2, 3, 4):
        out[f"{name}/plddt{dgt}"] = np.frombuffer(H.ref_extract(fcz, 0, dgt).encode("latin-1"), np.uint8)
    out[f"{name}/fasta"] = np.frombuffer(H.ref_extract(fcz, 1, 0).encode("latin-1"), np.uint8)


# ---- reference fixtures ------------------------------------------------------------------------
def pdb_chains(path, stem):
    t = remove_alternative_position(parse_pdb(open(path).read()))
    frags = []
    chains = identify_chains(t)
    for cs in chains:
        parts = identify_discontinuous(t, cs)
        for j, sl in enumerate(parts):
            name = stem + (t.chain[sl.start] if len(chains) > 1 else "") + (f"_{j}" if len(parts) > 1 else "")
            frags.append((name, t.take(sl)))
    return frags

for fn, stem in (("test.pdb", "test"), ("test_af.pdb", "test_af"), ("multichain.pdb", "multichain")):
    for name, tab in pdb_chains(os.path.join(REF_TEST, fn), stem):
        add_batch_case("pdb:" + name, build_batch([Chain(name, tab)]))

# ---- example_db entries ------------------------------------------------------------------------
idx = [l.split("\t") for l in open(os.path.join(REF_TEST, "example_db.index")).read().splitlines()]
data = open(os.path.join(REF_TEST, "example_db"), "rb").read()
lookup = {l.split("\t")[0]: l.split("\t")[1] for l in open(os.path.join(REF_TEST, "example_db.lookup")).read().splitlines()}
for key, offs, ln in idx:
    e = data[int(offs):int(offs) + int(ln)]
    name = f"db:{int(key):02d}"
    out[f"{name}/fcz"] = np.frombuffer(e, np.uint8)
    out[f"{name}/name"] = np.frombuffer(lookup[key].encode(), np.uint8)
    n_res = int.from_bytes(e[4:6], "little"); n_anchor = e[12]; tl = int.from_bytes(e[24:28], "little")
    w0 = 76 + 4 * n_anchor + tl + 36 * n_anchor + 13
    has_unk = any((e[w0 + 8 * k] >> 3) > 19 for k in range(n_res))
    # `-a` on an UNK residue reads past the end of an empty altAtoms vector in the reference
    # (_reorderAtoms, src/foldcomp.cpp:1563-1577): undefined behaviour, so no golden for it
    for alt in ((0,) if has_unk else (0, 1)):
        d = H.ref_decompress(e, bool(alt))
        out[f"{name}/xyz{alt}"] = np.stack([d["x"], d["y"], d["z"]], 1)
    out[f"{name}/bfac"] = d["bfac"]
    add_text_goldens(name, e)
    index.append(name)

# ---- the committed reference output test_af.fcz -------------------------------------------------
out["fixture:test_af.plddt"] = np.frombuffer(open(os.path.join(REF_TEST, "test_af.plddt"), "rb").read(), np.uint8)
out["fixture:test_af.plddt.tsv"] = np.frombuffer(open(os.path.join(REF_TEST, "test_af.plddt.tsv"), "rb").read(), np.uint8)
out["fixture:test_af.fcz"] = np.frombuffer(open(os.path.join(REF_TEST, "test_af.fcz"), "rb").read(), np.uint8)

# ---- synthetic edge cases ----------------------------------------------------------------------
def syn(n, seed, **kw):
    return synthetic.to_chain_batch(synthetic.generate(1, n, seed=seed, anchor_threshold=kw.get("thr", 25)))

cases = []
for i, n in enumerate([2, 3, 7, 24, 25, 26, 49, 50, 51, 64, 65, 127, 128, 129, 350, 351, 700, 1400]):
    cases.append((f"syn:len{n}", syn(n, 1000 + i), 25))
cases.append(("syn:thr10", syn(123, 2001), 10))
cases.append(("syn:thr200", syn(450, 2002), 200))
b = syn(90, 2003); b.res_code[:] = 14
cases.append(("syn:pro_codes", b, 25))            # all residues flagged PRO (atoms of other types -> missing atoms too)
b = syn(77, 2004)                                  # no OXT: drop last atom
b.x, b.y, b.z, b.atom_code = b.x[:-1].copy(), b.y[:-1].copy(), b.z[:-1].copy(), b.atom_code[:-1].copy(); b.atom_off[-1] -= 1
cases.append(("syn:no_oxt", b, 25))
b = syn(60, 2005); b.bfac_ca[:] = 87.5
cases.append(("syn:const_bfac", b, 25))
b = syn(80, 2006); b.res_code[10] = 23; b.res_code[40] = 23   # UNK residues (0 torsions, backbone only on decode)
cases.append(("syn:unk", b, 25))
b = syn(100, 2007)                                 # missing side-chain atoms: rename some atoms to 'other'
rng = np.random.default_rng(5); m = (rng.random(b.n_atoms) < 0.15) & (b.atom_code > 3); b.atom_code[m] = 255
cases.append(("syn:missing_atoms", b, 25))
b = syn(70, 2008)                                  # shuffled atom order inside residues (N,CA,C kept in order)
for r in range(b.n_residues):
    s, e = b.atom_off[r], b.atom_off[r + 1]
    if r == b.n_residues - 1: e -= 1               # keep OXT last
    perm = np.arange(s + 3, e); rng.shuffle(perm)
    for arr in (b.x, b.y, b.z, b.atom_code):
        arr[s + 3:e] = arr[perm]
cases.append(("syn:shuffled", b, 25))
b = syn(40, 2009); b.res_code[:] = 7              # GLY-only codes
cases.append(("syn:gly_codes", b, 25))
for name, b, thr in cases:
    b.anchor_threshold = thr
    add_batch_case(name, b, thr)

out["index"] = np.frombuffer("\n".join(index).encode(), np.uint8)
path = os.path.join(ROOT, "tests", "golden", "reference_vectors.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes;", len(index), "cases")
