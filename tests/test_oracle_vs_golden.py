"""CPU: the C restatement (oracle/) against vectors minted from the real reference (tests/golden/,
made by tools/make_goldens.py) -- FCZ bytes bit-exact (4 uninitialised header pad bytes masked),
pre-quantisation angles and decompressed float32 coordinates bit-exact."""
import numpy as np
import pytest

import _harness as H
from _cases import compress_cases, db_cases, entries_blob, golden_batch


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_golden_index(golden):
    z, index = golden
    assert len(index) >= 50
    assert sum(n.startswith("db:") for n in index) == 24


def test_oracle_compress_matches_reference_bytes(golden):
    z, index = golden
    for name in compress_cases(index):
        b = golden_batch(z, name)
        blob, off, st = H.oracle_compress(b)
        assert st[0] == 0, name
        assert blob.tobytes() == z[f"{name}/fcz"].tobytes(), name


def test_oracle_decompress_matches_reference_bits(golden):
    z, index = golden
    names = compress_cases(index) + db_cases(index)
    for alt in (0, 1):
        use = [n for n in names if f"{n}/xyz{alt}" in z]
        blob, off = entries_blob([z[f"{n}/fcz"].tobytes() for n in use])
        d = H.oracle_decompress(blob, off, alt_order=bool(alt))
        for i, n in enumerate(use):
            a0, a1 = d["atom_off"][i], d["atom_off"][i + 1]
            exp = z[f"{n}/xyz{alt}"]
            assert a1 - a0 == len(exp), n
            got = np.stack([d["x"][a0:a1], d["y"][a0:a1], d["z"][a0:a1]], 1)
            assert np.array_equal(_bits(got), _bits(exp)), (n, alt)


def test_oracle_decompress_restated_trig_is_identical(golden):
    """the restated glibc sinf/cosf gives the same coordinates as the host libm"""
    z, index = golden
    use = [n for n in compress_cases(index) + db_cases(index)]
    blob, off = entries_blob([z[f"{n}/fcz"].tobytes() for n in use])
    a = H.oracle_decompress(blob, off)
    b = H.oracle_decompress(blob, off, restated_trig=True)
    for k in "xyz":
        assert np.array_equal(_bits(a[k]), _bits(b[k]))


def test_committed_reference_fcz_payload(golden):
    """test/test_af.fcz was produced on another platform: its header floats differ by 1 ulp
     but packed words, side-chain bytes and B-factor bytes must be identical."""
    z, _ = golden
    ref = z["fixture:test_af.fcz"].tobytes()
    b = golden_batch(z, "pdb:test_af")
    blob, off, st = H.oracle_compress(b)
    mine = blob.tobytes()
    n, na, tl_ref = 28, ref[12], int.from_bytes(ref[24:28], "little")
    tl = int.from_bytes(mine[24:28], "little")
    w_ref = 76 + 4 * na + tl_ref + 36 * na + 13
    w = 76 + 4 * na + tl + 36 * na + 13
    assert ref[w_ref:w_ref + 8 * n] == mine[w:w + 8 * n]
    nsc = int.from_bytes(ref[16:20], "little")
    assert ref[w_ref + 8 * n: w_ref + 8 * n + nsc] == mine[w + 8 * n: w + 8 * n + nsc]
    assert ref[-n:] == mine[-n:]


def test_oracle_check_and_errors(golden):
    z, index = golden
    lib = H.load_oracle()
    e = z["pdb:test_af/fcz"].tobytes()
    assert lib.fcz_oracle_check(e, len(e)) == 0
    bad = b"XXXX" + e[4:]
    assert lib.fcz_oracle_check(bad, len(bad)) == -4


@pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref not built (no /root/reference)")
def test_oracle_vs_live_reference_on_fresh_synthetic():
    """fresh seeds, not in the goldens: oracle == real reference, compress and decompress"""
    from foldcomp_amd import synthetic
    from tools_chain_table import chain_table
    b = synthetic.to_chain_batch(synthetic.generate(6, [33, 150, 350, 351, 12, 77], seed=424242))
    blob, off, st = H.oracle_compress(b)
    assert (st == 0).all()
    d = H.oracle_decompress(blob, off)
    for c in range(b.n_chains):
        t = chain_table(b, c)
        title = bytes(b.titles[b.title_off[c]:b.title_off[c + 1]]).decode()
        ref = H.mask_pad(H.ref_compress(t, title, b.anchor_threshold))
        assert ref == blob[off[c]:off[c + 1]].tobytes()
        r = H.ref_decompress(ref)
        a0, a1 = d["atom_off"][c], d["atom_off"][c + 1]
        for k in "xyz":
            assert np.array_equal(_bits(r[k]), _bits(d[k][a0:a1]))


@pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref not built (no /root/reference)")
def test_oracle_vs_live_reference_on_degenerate_coordinates():
    """zero-length bonds, straight lines, coinciding atoms, overflow, every atom at one point: NaN angles and what the quantisers
    make of them. The oracle's records and its decode of them == the real reference's (NaN for NaN where coordinates are NaN)"""
    from _cases import degenerate_batch, degenerate_cases
    from tools_chain_table import chain_table
    for name, mutate in degenerate_cases():
        b = degenerate_batch(mutate)
        blob, off, st = H.oracle_compress(b)
        assert (st == 0).all(), name
        d = H.oracle_decompress(blob, off)
        for c in range(b.n_chains):
            title = bytes(b.titles[b.title_off[c]:b.title_off[c + 1]]).decode()
            ref = H.mask_pad(H.ref_compress(chain_table(b, c), title, b.anchor_threshold))
            assert ref == blob[off[c]:off[c + 1]].tobytes(), (name, c)
            r = H.ref_decompress(ref)
            a0, a1 = d["atom_off"][c], d["atom_off"][c + 1]
            for k in "xyz":
                got, exp = d[k][a0:a1], np.asarray(r[k], np.float32)
                assert np.all((_bits(got) == _bits(exp)) | (np.isnan(got) & np.isnan(exp))), (name, c, k)


@pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref not built (no /root/reference)")
def test_oracle_decode_of_mutated_records_equals_live_reference(golden):
    """random angle words, side-chain torsion bytes and B-factor bytes (_cases.payload_mutations: conformations no structure has):
    coordinates and B-factors of the oracle's decode == Foldcomp::decompress of the live reference, both atom orders; the PDB text
    restatement == the reference's text; `extract` (pLDDT with 1..4 digits, from random quantiser parameters too) == the reference's.
    The reference runs in a child process (on a few of these records it crashes; they have no reference answer). The device is held
    to the oracle on the same records in tests/test_gpu_fcz_fuzz.py"""
    from _cases import golden_records, payload_mutations
    from foldcomp_amd import fczfile
    from host_text import extract_plddt, pdb_from_result
    ref = H.RefWorker()
    recs = payload_mutations(golden_records(golden), per_record=4)
    blob, off = entries_blob(recs)
    same = crashed = 0
    for alt in (False, True):
        o = H.oracle_decompress(blob, off, alt_order=alt, n_threads=4)
        for i, e in enumerate(recs):
            assert o["info"][i].status == 0
            got_r = ref.call("ref_decompress", e, alt)
            if got_r[0] == "crash":
                crashed += 1; continue
            assert got_r[0] == "ok", (i, alt)
            r = got_r[1]
            a0, a1 = o["atom_off"][i], o["atom_off"][i + 1]
            assert a1 - a0 == len(r["x"]), i
            for k in "xyz":
                got, exp = o[k][a0:a1], np.asarray(r[k], np.float32)
                assert np.all((_bits(got) == _bits(exp)) | (np.isnan(got) & np.isnan(exp))), (i, alt, k)
            if i % 3 == 0:
                t = ref.call("ref_decompress_pdb", e, alt)
                assert t[0] == "ok" and pdb_from_result(fczfile.parse(e), o, i, alt) == t[1], (i, alt)
            same += 1
    for i, e in enumerate(payload_mutations(golden_records(golden), per_record=3, seed=5, temp_params=True)):
        for digits in (1, 2, 3, 4):
            t = ref.call("ref_extract", e, 0, digits)
            if t[0] == "crash":
                crashed += 1; continue
            assert t[0] == "ok" and extract_plddt(fczfile.parse(e), digits) == t[1], (i, digits)
    ref.close()
    assert same > 150 and crashed < same // 4, (same, crashed)


@pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref not built (no /root/reference)")
def test_oracle_decode_of_records_with_any_header_floats_equals_live_reference(golden):
    """the twelve angle parameters, anchor coordinates and the OXT atom of golden records set to anything (tools/dbg/param_fuzz.py:
    huge, tiny, negative, NaN, infinite -- angles of thousands of radians go through glibc's large-argument sine / cosine): the
    restatement's decode == Foldcomp::decompress of the live reference, both atom orders. The device is held to the restatement on
    the same kind of records in tests/test_gpu_parity_fuzz.py"""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("param_fuzz", os.path.join(root, "tools", "dbg", "param_fuzz.py"))
    pf = importlib.util.module_from_spec(spec); spec.loader.exec_module(pf)
    from _cases import golden_records
    recs = pf.mutated(golden_records(golden), np.random.default_rng(20261001), per_record=3)
    blob, off = entries_blob(recs)
    ref = H.RefWorker()
    same = crashed = 0
    for alt in (False, True):
        o = H.oracle_decompress(blob, off, alt_order=alt, n_threads=4)
        for i, e in enumerate(recs):
            if o["info"][i].status != 0:
                continue
            got_r = ref.call("ref_decompress", e, alt)
            if got_r[0] != "ok":
                crashed += 1; continue
            r = got_r[1]
            a0, a1 = o["atom_off"][i], o["atom_off"][i + 1]
            assert a1 - a0 == len(r["x"]), i
            for k in "xyz":
                got, exp = o[k][a0:a1], np.asarray(r[k], np.float32)
                assert np.all((_bits(got) == _bits(exp)) | (np.isnan(got) & np.isnan(exp))), (i, alt, k, np.frombuffer(e, np.float32, 12, 28))
            same += 1
    ref.close()
    assert same > 200 and crashed < same // 4, (same, crashed)
