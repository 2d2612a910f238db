"""GPU: the Python surface that mirrors the reference's `foldcomp` module (compress / decompress /
get_data / open), end to end through libfcz_hip.so, against reference-minted goldens."""
import numpy as np
import pytest

import foldcomp_amd as foldcomp
from _cases import db_cases, golden_batch
import host_text as pdbio   # oracle/host_text.py: host restatement of the reference writer
from foldcomp_amd._aa_tables import RES3
from foldcomp_amd.database import DatabaseWriter

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _input_pdb_text(z, name):
    """render the golden SoA input of a one-chain case as PDB text (what a user would pass in)"""
    b = golden_batch(z, name)
    n_at = b.n_atoms
    res_of_atom = np.repeat(np.arange(b.n_residues), np.diff(b.atom_off.astype(np.int64)))
    bf = b.bfac_ca[res_of_atom]
    return pdbio.format_pdb("", b.atom_code, b.res_code[res_of_atom], int(b.first_res_index[0]) + res_of_atom, chr(b.chain_id[0]),
                            int(b.first_atom_index[0]), b.x, b.y, b.z, bf), bytes(b.titles).decode()


@pytest.fixture(scope="module", autouse=True)
def _codec_for_api(codec):
    from foldcomp_amd import api
    api.set_codec(codec)
    yield
    api.set_codec(None)


@pytest.mark.parametrize("name", ["pdb:test_af", "pdb:test", "pdb:multichainA", "syn:len350", "syn:len26"])
def test_compress_matches_reference_bytes(golden, name):
    z, _ = golden
    text, title = _input_pdb_text(z, name)
    thr = int(z[f"{name}/in/anchor_threshold"][0])
    fcz = foldcomp.compress(title, text, anchor_residue_threshold=thr)
    assert isinstance(fcz, bytes)
    assert fcz == z[f"{name}/fcz"].tobytes()


def test_compress_errors(golden):
    with pytest.raises(foldcomp.error, match="No ATOM lines found"):
        foldcomp.compress("x", "HEADER nothing\n")
    z, _ = golden
    text, _ = _input_pdb_text(z, "pdb:test_af")
    two = text + text.replace(" A ", " B ")
    with pytest.raises(foldcomp.error, match="Multiple chains"):
        foldcomp.compress("x", two)
    with pytest.raises(TypeError):
        foldcomp.compress("x", text, anchor_residue_threshold="25")
    assert len(foldcomp.split_pdb_by_chain(two)) == 2


@pytest.mark.parametrize("name", ["pdb:test_af", "pdb:test", "pdb:multichainB_1", "syn:len350", "db:00", "db:13"])
def test_decompress_matches_reference_text(golden, name):
    z, _ = golden
    title, pdb = foldcomp.decompress(z[f"{name}/fcz"].tobytes())
    exp = z[f"{name}/pdb0"].tobytes().decode("latin-1")
    assert pdb == exp
    if exp.startswith("TITLE"):
        assert exp.startswith("TITLE     " + title[:70])


def test_decompress_error():
    with pytest.raises(foldcomp.error):
        foldcomp.decompress(b"not an fcz record at all, definitely not")


def test_get_data_fcz(golden):
    z, _ = golden
    for name in ("pdb:test_af", "db:05"):
        d = foldcomp.get_data(z[f"{name}/fcz"].tobytes())
        exp = z[f"{name}/xyz0"]
        got = np.asarray(d["coordinates"], np.float32)
        assert np.array_equal(_bits(got), _bits(exp))
        assert set(d) == {"phi", "psi", "omega", "torsion_angles", "bond_angles", "residues", "b_factors", "coordinates"}
        assert d["residues"] == z[f"{name}/fasta"].tobytes().decode()


def test_get_data_pdb_angles_bit_exact(golden):
    z, _ = golden
    # syn:len700 / syn:len1400: chains beyond the register path of the pack kernel (angles finished in place, two passes)
    # ... and every length class of the rows kernels (four chains to a wavefront: 1 / 2 / 4 / 8 rounds of 16 residues; the finished
    # angles and the first residue's N-CA-C angle are left in the scratch by a 16-lane group there)
    for name in ("pdb:test_af", "pdb:test", "syn:len129", "syn:len351", "syn:len700", "syn:len1400",
                 "syn:len2", "syn:len3", "syn:len7", "syn:len24", "syn:len26", "syn:len49", "syn:len64", "syn:len65", "syn:len127", "syn:len128", "syn:thr10"):
        text, _ = _input_pdb_text(z, name)
        d = foldcomp.get_data(text)
        for k in ("phi", "psi", "omega"):
            assert np.array_equal(_bits(np.asarray(d[k], np.float32)), _bits(z[f"{name}/angle/{k}"])), (name, k)
        n = len(d["residues"])
        assert len(d["bond_angles"]) == 3 * n - 2 and len(d["torsion_angles"]) == 3 * n - 3
        ba = np.asarray(d["bond_angles"], np.float32)
        assert np.array_equal(_bits(ba[1::3]), _bits(z[f"{name}/angle/ca_c_n"]))
        assert np.array_equal(_bits(ba[3::3]), _bits(z[f"{name}/angle/n_ca_c"]))
    with pytest.raises(ValueError):
        foldcomp.get_data(b"")


def test_open_database(tmp_path, golden):
    z, index = golden
    names = db_cases(index)
    w = DatabaseWriter(str(tmp_path / "db"))
    for i, n in enumerate(names):
        # MMseqs2-made databases carry a trailing NUL per entry; the module strips one byte (foldcomp.cxx:66)
        e = z[f"{n}/fcz"].tobytes()
        w.append(e if e.endswith(b"\0") else e + b"\0", i, bytes(z[f"{n}/name"]).decode())
    w.close()
    with foldcomp.open(str(tmp_path / "db")) as db:
        assert len(db) == 24
        title, pdb = db[3]
        assert pdb == z[f"{names[3]}/pdb0"].tobytes().decode("latin-1")
        allp = list(db.decompress_all(batch=16))
        assert len(allp) == 24 and allp[3][1] == pdb
        with pytest.raises(IndexError):
            db[24]
    want = [bytes(z[f"{names[5]}/name"]).decode(), "missing_id", bytes(z[f"{names[1]}/name"]).decode()]
    with foldcomp.open(tmp_path / "db", ids=want, decompress=False) as db:
        assert len(db) == 2
        assert db[0].rstrip(b"\0") == z[f"{names[5]}/fcz"].tobytes().rstrip(b"\0")
    with pytest.raises(KeyError):
        foldcomp.open(str(tmp_path / "db"), ids=want, err_on_missing=True)
    with pytest.raises(TypeError):
        foldcomp.open(str(tmp_path / "db"), ids="d1asha_")


def test_open_database_written_without_terminators(tmp_path, golden):
    """databases made by `compress -d` (here and in the reference, src/main.cpp:516) carry no trailing NUL per entry: db[i] must
    not drop the record's last byte (the reference's module does, foldcomp.cxx:66, and reads one byte short)"""
    z, index = golden
    names = db_cases(index)[:6]
    w = DatabaseWriter(str(tmp_path / "db"))
    from foldcomp_amd import fczfile
    recs = []
    for i, n in enumerate(names):
        e = z[f"{n}/fcz"].tobytes()
        recs.append(e[:fczfile.record_size(e)])          # the record alone: the example_db entries end in a NUL
        assert len(recs[-1]) in (len(e), len(e) - 1)
        w.append(recs[-1], i, bytes(z[f"{n}/name"]).decode())
    w.close()
    with foldcomp.open(str(tmp_path / "db")) as db:
        for i, n in enumerate(names):
            title, pdb = db[i]
            assert pdb == z[f"{n}/pdb0"].tobytes().decode("latin-1"), n
    with foldcomp.open(str(tmp_path / "db"), decompress=False) as db:
        assert db[2] == recs[2]


def test_iterating_a_database_runs_in_gpu_batches_behind_the_per_entry_surface(tmp_path, codec):
    """`for name, pdb in foldcomp.open(db)` -- the loop a program written against the reference's module runs (foldcomp.cxx:44-90,
    :197-220: one read + decompress + PDB text per entry). Here the sequential walk is served from a read-ahead window (one GPU call
    per 1 024 entries): the same (name, text) entry by entry as the per-entry call, an entry that does not decode raises at ITS index,
    random access still works -- and the walk is faster than the reference's per-entry C++ loop measured in the build container
    (profiles/r6_python_iter_reference.json: 254 entries/s at 350 residues, one thread)."""
    import json, os, time
    from foldcomp_amd import synthetic
    from foldcomp_amd.api import FoldcompDatabase
    n, n_res = 20000, 350
    w = DatabaseWriter(str(tmp_path / "db"))
    recs = []
    for s in range(0, n, 5000):
        b = synthetic.to_chain_batch(synthetic.generate(5000, [n_res] * 5000, seed=100 + s, first_chain_id=s))
        blob, off, st = codec.compress_batch(b)
        assert (st == 0).all()
        raw = blob.tobytes()
        for i in range(5000):
            r = raw[int(off[i]):int(off[i + 1])]
            if s + i == 7777:
                r = r[:40] + bytes(len(r) - 40)          # an entry whose payload is gone: decodes to nothing
            recs.append(r)
            w.append(r, s + i, f"e{s + i:06d}")
    w.close()
    db = foldcomp.open(str(tmp_path / "db"))
    assert len(db) == n
    t0 = time.perf_counter()
    got, bad = 0, []
    it = iter(range(n))
    for i in it:
        try:
            name, pdb = db[i]
        except foldcomp.error:
            bad.append(i); continue
        got += 1
        if i % 997 == 0:
            assert (name, pdb) == foldcomp.decompress(recs[i]), i          # the per-entry call: same title, same text
    dt = time.perf_counter() - t0
    assert bad in ([7777], []) and got + len(bad) == n
    if bad == []:                                                           # (zeroed angles still decode to a structure: then the text must say so)
        assert db[7777] == foldcomp.decompress(recs[7777])
    # random access next to the walk, and the plain iterator protocol
    assert db[12345] == foldcomp.decompress(recs[12345]) and db[3] == foldcomp.decompress(recs[3])
    first = next(iter(db))
    assert first == foldcomp.decompress(recs[0])
    db.close()
    ref = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r6_python_iter_reference.json")))
    rate = n / dt
    print(f"database walk: {rate:.0f} entries/s ({rate * n_res / 1e6:.1f} M residues/s) in windows of {FoldcompDatabase.READAHEAD}; reference per-entry loop {ref['entries_per_s']} entries/s")
    assert rate > 3 * ref["entries_per_s"], (rate, ref["entries_per_s"])
    # the same walk with a window of one entry (a GPU call per entry, round 5's shape) for the record
    old = FoldcompDatabase.READAHEAD
    FoldcompDatabase.READAHEAD = 1
    try:
        db1 = foldcomp.open(str(tmp_path / "db"))
        t0 = time.perf_counter()
        for i in range(500):
            db1[i]
        per_entry = 500 / (time.perf_counter() - t0)
        db1.close()
    finally:
        FoldcompDatabase.READAHEAD = old
    print(f"  a GPU call per entry: {per_entry:.0f} entries/s -> the window is {rate / per_entry:.1f}x")
    with open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r6_python_iter.json"), "w") as fh:
        json.dump({"entries": n, "residues_per_entry": n_res, "readahead": old, "entries_per_s": round(rate, 1), "residues_per_s": round(rate * n_res),
                   "per_entry_gpu_call_entries_per_s": round(per_entry, 1), "reference_per_entry_loop_entries_per_s": ref["entries_per_s"],
                   "speedup_vs_reference_loop": round(rate / ref["entries_per_s"], 1)}, fh)
