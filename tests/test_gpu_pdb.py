"""GPU: PDB text produced on the device (k_pdb_sizes / k_pdb_format) against the reference's own text (goldens) and, for
column overflows the goldens do not contain, against the host restatement in oracle/host_text.py (itself pinned to the
goldens in test_host_formats.py)."""
import numpy as np
import pytest

from foldcomp_amd import fczfile

pytestmark = pytest.mark.gpu


def _blob(entries):
    off = np.zeros(len(entries) + 1, np.uint64)
    off[1:] = np.cumsum([len(e) for e in entries])
    return np.frombuffer(b"".join(entries), np.uint8), off


def test_pdb_text_equals_reference_for_every_golden(codec, golden):
    z, index = golden
    names = [n for n in index if f"{n}/pdb0" in z.files and f"{n}/fcz" in z.files]
    assert len(names) >= 20
    entries = [z[f"{n}/fcz"].tobytes() for n in names]
    blob, off = _blob(entries)
    texts, status = codec.decompress_pdb(blob, off)
    assert (status == 0).all()
    for n, t in zip(names, texts):
        exp = z[f"{n}/pdb0"].tobytes()
        assert t == exp, (n, len(t), len(exp), next((i for i in range(min(len(t), len(exp))) if t[i] != exp[i]), None))


def _host_text(codec, entries, alt_order=False):
    from host_text import pdb_from_result as _pdb_from_result
    blob, off = _blob(entries)
    d = codec.decompress_batch(blob, off, alt_order=alt_order)
    return [_pdb_from_result(fczfile.parse(e), d, i, alt_order).encode("latin-1") for i, e in enumerate(entries)]


def test_pdb_text_column_overflows_and_long_titles(codec):
    from foldcomp_amd import synthetic
    from foldcomp_amd.structure import ChainBatch  # noqa: F401
    lens = [40, 5200, 33, 64, 65, 2]
    d = synthetic.generate(len(lens), lens, seed=99)
    b = synthetic.to_chain_batch(d)
    # chain 0: residue numbers run past 9999; chain 1: atom serials run past 99999; chain 2: coordinates and B-factors
    # wider than their columns; chain 3: very long title (continuation numbers > 99); chain 4: no title
    b.first_res_index[0] = 9985
    b.first_atom_index[1] = 65000
    a0, a1 = int(b.atom_off[b.res_off[2]]), int(b.atom_off[b.res_off[3]])
    b.x[a0:a1] += np.float32(20000.0); b.y[a0:a1] -= np.float32(3000.0); b.z[a0:a1] += np.float32(123456.0)
    b.bfac_ca[b.res_off[2]:b.res_off[3]] = np.linspace(900.0, 1800.0, int(b.res_off[3] - b.res_off[2])).astype(np.float32)
    titles = [b"t0", b"chain with many atoms", b"far away", bytes((65 + i % 26) for i in range(7300)), b"", b"x" * 70]
    b.titles = np.frombuffer(b"".join(titles), np.uint8).copy()
    b.title_off = np.concatenate([[0], np.cumsum([len(t) for t in titles])]).astype(np.uint32)
    blob, off, st = codec.compress_batch(b)
    assert (st == 0).all()
    entries = [blob[off[i]:off[i + 1]].tobytes() for i in range(len(lens))]
    for alt in (False, True):
        texts, status = codec.decompress_pdb(blob, off, alt_order=alt)
        assert (status == 0).all()
        exp = _host_text(codec, entries, alt_order=alt)
        for i, (t, e) in enumerate(zip(texts, exp)):
            assert t == e, (alt, i, len(t), len(e), next((k for k in range(min(len(t), len(e))) if t[k] != e[k]), None))
    # the overflow cases really are in the data
    t = texts[0].decode("latin-1")
    assert any(len(line) > 80 for line in t.split("\n"))


def test_pdb_text_skips_bad_entries(codec, golden):
    z, index = golden
    good = z["pdb:test_af/fcz"].tobytes()
    blob, off = _blob([good, b"NOPE" + good[4:], good[:100], good])
    texts, status = codec.decompress_pdb(blob, off)
    assert status[0] == 0 and status[3] == 0 and status[1] != 0 and status[2] != 0
    assert texts[1] == b"" and texts[2] == b"" and texts[0] == texts[3] == z["pdb:test_af/pdb0"].tobytes()


def test_pdb_text_with_database_terminators(codec, golden):
    """FCZ_PDB_NUL_TERMINATED: every text that decodes is followed by the NUL of a database record (src/main.cpp:656-664), the
    entries that do not decode take no byte; twice in a row on one ctx (the second call's buffer held the first call's text)"""
    z, index = golden
    names = [n for n in index if f"{n}/pdb0" in z.files and f"{n}/fcz" in z.files]
    entries = [z[f"{n}/fcz"].tobytes() for n in names]
    bad = b"NOPE" + entries[0][4:]
    for order in (entries[:7] + [bad] + entries[7:], [bad] + entries[::-1] + [bad]):
        blob, off = _blob(order)
        plain, st0 = codec.decompress_pdb(blob, off)
        term, st1 = codec.decompress_pdb(blob, off, nul_terminated=True)
        assert np.array_equal(st0, st1)
        for e, a, b, s in zip(order, plain, term, st0):
            assert b == (a + b"\0" if s == 0 else b""), (len(a), len(b), s)
        assert sum(len(t) for t in term) == sum(len(t) for t in plain) + int((st0 == 0).sum())


def test_pdb_text_of_degenerate_records(codec):
    """NaN coordinates in the text ("(.00(": what the reference prints for them, test_host_formats.py pins the restatement to the
    live reference on the same records): device text == restatement, both atom orders"""
    import _harness as H
    from _cases import degenerate_batch, degenerate_cases
    seen_nan = False
    for name, mutate in degenerate_cases():
        b = degenerate_batch(mutate)
        blob, off, st = codec.compress_batch(b)
        assert (st == 0).all()
        entries = [blob[off[i]:off[i + 1]].tobytes() for i in range(b.n_chains)]
        for alt in (False, True):
            texts, status = codec.decompress_pdb(blob, off, alt_order=alt)
            assert (status == 0).all()
            o = H.oracle_decompress(blob, off, alt_order=alt)
            from host_text import pdb_from_result
            for i, (t, e) in enumerate(zip(texts, entries)):
                exp = pdb_from_result(fczfile.parse(e), o, i, alt).encode("latin-1")
                assert t == exp, (name, alt, i, next((k for k in range(min(len(t), len(exp))) if t[k] != exp[k]), None))
                seen_nan = seen_nan or b"(.00(" in t
    assert seen_nan


def test_extract_equals_reference_for_every_golden(codec, golden):
    """k_extract (pLDDT digits 1..4, sequence) against the reference's `foldcomp extract` strings"""
    z, index = golden
    names = [n for n in index if f"{n}/plddt2" in z.files and f"{n}/fcz" in z.files]
    assert len(names) >= 20
    entries = [z[f"{n}/fcz"].tobytes() for n in names]
    blob, off = _blob(entries)

    for digits in (1, 2, 3, 4):
        got = codec.extract(blob, off, mode=0, digits=digits)
        for n, g in zip(names, got):
            assert g == z[f"{n}/plddt{digits}"].tobytes(), (n, digits)
    got = codec.extract(blob, off, mode=1)
    for n, g in zip(names, got):
        assert g == z[f"{n}/fasta"].tobytes(), n
    # unreadable entries give empty strings
    bad = codec.extract(*_blob([entries[0], b"FCMPxx", entries[0][:90]]), mode=0, digits=2)
    assert bad[1] == b"" and bad[2] == b"" and bad[0] == z[f"{names[0]}/plddt2"].tobytes()
