"""GPU: seeded mutations of FCZ records through the device decoder.

  payload mutations   random angle words (residue codes kept), side-chain torsion bytes and B-factor bytes: every record still
                      decodes, and decodes to the oracle's bits -- conformations no real structure has (the rare branches of the
                      exact numerics: re-measured bond angles near 0 / 180 degrees, clashing atoms, zero-length normals)
  header mutations    counts, anchor indices, lengths, truncation, random bytes anywhere: the call never fails as a whole; an entry
                      is either refused with a status or -- when nothing the decoder depends on was hit -- decodes like the oracle
"""
import numpy as np
import pytest

import _harness as H
from _cases import compress_cases, db_cases, entries_blob
from foldcomp_amd import fczfile

pytestmark = pytest.mark.gpu


def _same_bits(a, b):
    """float32 arrays: equal bits, or NaN on both sides (the payload of a NaN is not part of the contract)"""
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def _records(golden):
    z, index = golden
    recs = []
    for n in compress_cases(index) + db_cases(index):
        e = z[f"{n}/fcz"].tobytes()
        recs.append(e[:fczfile.record_size(e)])
    return recs


def _compare(codec, entries, alt):
    blob, off = entries_blob(entries)
    d = codec.decompress_batch(blob, off, alt_order=alt)
    o = H.oracle_decompress(blob, off, alt_order=alt)
    assert [d["info"][i].status for i in range(len(entries))] == [o["info"][i].status for i in range(len(entries))]
    assert np.array_equal(d["res_off"], o["res_off"]) and np.array_equal(d["atom_off"], o["atom_off"])
    for k in ("x", "y", "z", "bfac_res"):
        assert _same_bits(d[k], o[k]), (k, int((np.ascontiguousarray(d[k]).view(np.uint32) != np.ascontiguousarray(o[k]).view(np.uint32)).sum()))
    assert np.array_equal(d["res_code"], o["res_code"])
    return d


def test_payload_mutations_decode_like_the_oracle(codec, golden):
    from _cases import golden_records, payload_mutations
    entries = payload_mutations(golden_records(golden))        # the records test_oracle_vs_golden.py puts to the live reference
    assert len(entries) >= 250
    for alt in (False, True):
        d = _compare(codec, entries, alt)
        assert all(d["info"][i].status == 0 for i in range(len(entries)))


def test_extract_of_mutated_records(codec, golden):
    """pLDDT strings from random B-factor bytes and random quantiser parameters (negative, tiny, huge): the device's `extract` ==
    the restatement that test_oracle_vs_golden.py pins to the live reference on the same records"""
    from _cases import golden_records, payload_mutations
    from host_text import extract_plddt
    entries = payload_mutations(golden_records(golden), per_record=3, seed=5, temp_params=True)
    blob, off = entries_blob(entries)
    for digits in (1, 2, 3, 4):
        got = codec.extract(blob, off, mode=0, digits=digits)
        for i, e in enumerate(entries):
            assert got[i].decode("latin-1") == extract_plddt(fczfile.parse(e), digits), (i, digits)


def test_header_mutations_are_refused_or_decode_like_the_oracle(codec, golden):
    rng = np.random.default_rng(7)
    recs = _records(golden)
    entries = []
    for i in range(1500):
        e = recs[i % len(recs)]
        r = fczfile.parse(e)
        b = bytearray(e)
        kind = int(rng.integers(0, 9))
        if kind == 0:                                   # a count of the header
            o = [4, 6, 8, 10, 12, 16, 24][int(rng.integers(0, 7))]
            b[o] = int(rng.integers(0, 256)); b[o + 1] = int(rng.integers(0, 256)) if o != 12 else b[o + 1]
        elif kind == 1:                                 # an anchor index
            o = 76 + 4 * int(rng.integers(0, r.n_anchors))
            b[o:o + 4] = int(rng.integers(-5, r.n_residues + 5)).to_bytes(4, "little", signed=True)
        elif kind == 2:                                 # cut short
            b = b[:int(rng.integers(0, len(b)))]
        elif kind == 3:                                 # bytes appended
            b += bytes(rng.integers(0, 256, int(rng.integers(1, 64)), dtype=np.uint8))
        elif kind == 4:                                 # the magic
            b[int(rng.integers(0, 4))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 5:                                 # a residue code
            b[r.o_words + 8 * int(rng.integers(0, r.n_residues))] ^= 0xf8 & int(rng.integers(8, 256))
        elif kind == 6:                                 # first / last residue letters, the OXT flag
            o = [20, 21, r.o_words - 13][int(rng.integers(0, 3))]
            b[o] = int(rng.integers(0, 256))
        elif kind == 7:                                 # an anchor atom: a random float
            o = 76 + 4 * r.n_anchors + len(r.title) + 4 * int(rng.integers(0, 9 * r.n_anchors))
            b[o:o + 4] = np.float32(rng.normal(0, 1) * 10.0 ** int(rng.integers(-3, 6))).tobytes()
        else:                                           # a few random bytes anywhere past the quantiser parameters
            for _ in range(int(rng.integers(1, 4))):
                o = int(rng.integers(76, len(b)))
                b[o] = int(rng.integers(0, 256))
        entries.append(bytes(b))
    blob, off = entries_blob(entries)
    d = codec.decompress_batch(blob, off)
    st = np.asarray([d["info"][i].status for i in range(len(entries))])
    assert (st != 0).sum() > 300 and (st == 0).sum() > 300
    # what the device takes, the oracle takes too and decodes to the same bits
    ok = [entries[i] for i in range(len(entries)) if st[i] == 0]
    for alt in (False, True):
        d2 = _compare(codec, ok, alt)
        assert all(d2["info"][i].status == 0 for i in range(len(ok)))
    # PDB text and extract on the same entries: refused entries give nothing, the others their text
    texts, st_t = codec.decompress_pdb(blob, off)
    assert [int(s != 0) for s in st_t] == [int(s != 0) for s in st]
    assert all((len(t) == 0) == (s != 0) for t, s in zip(texts, st))
