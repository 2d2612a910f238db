#!/usr/bin/env python3
"""Structure ingest on the device against the host reader (gemmi's rules, pinned to the live reference) on TEXT of the input
variants of the differential fuzz: every variant's chains decoded to PDB text (columns overflow for huge coordinates, B-factors
and numbers) and re-rendered as AFDB-shaped and archive-shaped mmCIF. A file is either read into the same batch or handed back.
usage (GPU box): python tools/dbg/ingest_variants_fuzz.py [chains per variant] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import bench
from _cases import input_variants
from foldcomp_amd.codec import Codec
import test_gpu_ingest as T

def run(N, seed, codec=None):
    rng = np.random.default_rng(seed)
    own = codec is None
    if own: codec = Codec(0)
    bad = 0; n = 0; taken = 0; handed = 0
    if True:
        for name, b in input_variants(rng, N):
            n += 1
            blob, off, st = codec.compress_batch(b, strict=False)
            keep = [i for i in range(b.n_chains) if st[i] == 0]
            if not keep:
                continue
            entries = [blob[off[i]:off[i + 1]].tobytes() for i in keep]
            eoff = np.zeros(len(entries) + 1, np.uint64); eoff[1:] = np.cumsum([len(e) for e in entries])
            texts, status = codec.decompress_pdb(np.frombuffer(b"".join(entries), np.uint8).copy(), eoff)
            texts = [t for t, s in zip(texts, status) if s == 0]
            files = [(t, f"v{i}.pdb") for i, t in enumerate(texts)]
            for i, t in enumerate(texts):
                try:
                    files.append((bench.cif_from_pdb_text(t, f"V{i}"), f"v{i}.cif"))
                    files.append((bench.cif_archive_from_pdb_text(t, f"W{i}", bench.ARCHIVE_STYLES[(i * 7) % len(bench.ARCHIVE_STYLES)]), f"w{i}.cif"))
                except Exception as e:                       # (a text whose columns overflowed is not something the converters split)
                    pass
            ftexts = [f[0] for f in files]; fnames = [f[1] for f in files]
            bdev, cfile, cmeta, fstat, refused = codec.ingest_pdb(ftexts, fnames)
            ok = [i for i in range(len(files)) if fstat[i] in (0, 4)]
            taken += len(ok); handed += len(files) - len(ok)
            remap = {f: k for k, f in enumerate(ok)}
            try:
                exp, exp_names, exp_file, exp_ref, failed = T._host_expect([ftexts[i] for i in ok], [fnames[i] for i in ok], reader=T._read_any)
                assert not failed, ("the device took files the host reader fails", [fnames[ok[i]] for i in failed])
                if exp is None:
                    assert bdev.n_chains == 0
                else:
                    T._same_batch(bdev, exp)
                    cn = codec.chain_names(bdev.n_chains)
                    assert [T._name_of(fnames[f], int(m), c) for f, m, c in zip(cfile, cmeta, cn)] == exp_names
                    assert [remap[int(f)] for f in cfile] == exp_file
                assert sorted((remap[int(f)], T._name_of(fnames[int(f)], int(m))) for f, m in refused) == sorted(exp_ref)
            except AssertionError as e:
                print(f"[{name}] {str(e)[:300]}"); bad += 1
    # files as depositions look (_cases.composite_pdb: several chains, gaps, alternative locations, insertion codes, waters, CRLF)
    from _cases import composite_pdb, _variant_base
    pool = _variant_base(rng, 40, 4, 160)
    for k in ("x", "y", "z"):
        setattr(pool, k, (np.round((getattr(pool, k).astype(np.float64) + rng.normal(0, 0.05, pool.n_atoms)) * 1000.0) / 1000.0).astype(np.float32))
    ftexts = [composite_pdb(rng, pool, f"C{i}") for i in range(10 * N)]; fnames = [f"c{i}.pdb" for i in range(10 * N)]
    for skip in (False, True):
        bdev, cfile, cmeta, fstat, refused = codec.ingest_pdb(ftexts, fnames, 25, skip)
        ok = [i for i in range(len(ftexts)) if fstat[i] in (0, 4)]
        taken += len(ok); handed += len(ftexts) - len(ok)
        remap = {f: k for k, f in enumerate(ok)}
        try:
            exp, exp_names, exp_file, exp_ref, failed = T._host_expect([ftexts[i] for i in ok], [fnames[i] for i in ok], 25, skip, reader=T._read_any)
            assert not failed, ("the device took files the host reader fails", failed)
            if exp is None:
                assert bdev.n_chains == 0
            else:
                T._same_batch(bdev, exp)
                cn = codec.chain_names(bdev.n_chains)
                assert [T._name_of(fnames[f], int(m), c) for f, m, c in zip(cfile, cmeta, cn)] == exp_names
                assert [remap[int(f)] for f in cfile] == exp_file
            assert sorted((remap[int(f)], T._name_of(fnames[int(f)], int(m))) for f, m in refused) == sorted(exp_ref)
        except AssertionError as e:
            print(f"[composite files, skip_discontinuous={skip}] {str(e)[:300]}"); bad += 1
    if own: codec.close()
    print(f"{n} variants, {taken} files read on the device, {handed} handed back, {bad} differences")
    return n, bad



if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 12, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
