"""CPU: the two hosts' ingest (PDB / mmCIF / .gz parsing, alternative positions, chain and gap fragmenting, titles) and the
database container (reader and writer) against the REAL reference: goldens minted by tools/make_ingest_goldens.py from
StructureReader (gemmi) + src/main.cpp:457-474 and from make_writer / make_reader, over the reference's own test data files
(tests/golden/reference_ingest.npz); where oracle/_ref exists (the build container) the same comparisons also run live."""
import os
import subprocess

import numpy as np
import pytest

import _harness as H
from foldcomp_amd.__main__ import load_structure
from foldcomp_amd.database import DatabaseReader, DatabaseWriter
from foldcomp_amd.structure import (Chain, build_batch, identify_chains, identify_discontinuous, remove_alternative_position)
from test_host_cpp import _dump, _run, _same

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ("test.pdb", "test_af.pdb", "multichain.pdb", "test.cif.gz")


@pytest.fixture(scope="module")
def ing():
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_ingest.npz"))


def _ref_table(z, fn):
    k = f"ingest:{fn}"
    strs = lambda a: [bytes(r).rstrip(b"\0").decode() for r in a]
    return dict(atom=strs(z[f"{k}/atom"]), residue=strs(z[f"{k}/residue"]), chain=[chr(c) for c in z[f"{k}/chain"]],
                atom_index=z[f"{k}/atom_index"], res_index=z[f"{k}/res_index"], xyz=z[f"{k}/xyz"], bfac=z[f"{k}/bfac"],
                title=bytes(z[f"{k}/title"]).decode("latin-1"), frag=[tuple(int(v) for v in f) for f in z[f"{k}/frag"]],
                n_chains=int(z[f"{k}/n_chains"][0]))


def _python_ingest(fn, data):
    t, title = load_structure(fn, data)
    t = remove_alternative_position(t)
    chains = identify_chains(t)
    frag = []
    for i, cs in enumerate(chains):
        for j, sl in enumerate(identify_discontinuous(t, cs)):
            frag.append((sl.start, sl.stop, i, j))
    return t, title, frag, len(chains)


@pytest.mark.parametrize("fn", FILES)
def test_python_ingest_equals_reference_reader(ing, fn):
    """parser output == what gemmi hands the reference, field by field and bit for bit; fragments == src/main.cpp:467-480"""
    ref = _ref_table(ing, fn)
    t, title, frag, nch = _python_ingest(fn, ing[f"file:{fn}"].tobytes())
    assert len(t) == len(ref["atom"])
    assert t.atom == ref["atom"] and t.residue == ref["residue"] and t.chain == ref["chain"]
    assert np.array_equal(t.atom_index, ref["atom_index"]) and np.array_equal(t.res_index, ref["res_index"])
    assert np.array_equal(t.xyz.view(np.uint32), ref["xyz"].view(np.uint32))
    assert np.array_equal(t.bfac.view(np.uint32), ref["bfac"].view(np.uint32))
    assert title == ref["title"]
    assert frag == ref["frag"] and nch == ref["n_chains"]


@pytest.mark.parametrize("fn", FILES)
def test_cpp_ingest_equals_reference_reader(ing, fn, tmp_path):
    """the C++ host's batch (dump-batch) == the batch built from the REFERENCE's atom table and fragments"""
    ref = _ref_table(ing, fn)
    p = tmp_path / fn
    p.write_bytes(ing[f"file:{fn}"].tobytes())
    from foldcomp_amd.structure import AtomTable
    t = AtomTable(ref["atom"], ref["residue"], ref["chain"], ref["atom_index"], ref["res_index"], ref["xyz"], ref["bfac"])
    stem = fn[:-3] if fn.endswith(".gz") else fn
    stem = stem.rsplit(".", 1)[0]
    # output names / titles: src/main.cpp:444-508 (title = output stem when the structure's title is the file name)
    base_stem = fn.rsplit(".", 1)[0]                                   # getFileParts splits at the LAST dot: test.cif.gz -> test.cif
    title = base_stem if ref["title"] == fn else ref["title"]
    names, chains = [], []
    for (a, b, ci, fj) in ref["frag"]:
        n_in_chain = sum(1 for f in ref["frag"] if f[2] == ci)
        nm = base_stem + (ref["chain"][a] if ref["n_chains"] > 1 else "") + (f"_{fj}" if n_in_chain > 1 else "") + ".fcz"
        names.append(nm); chains.append(Chain(title, t.take(slice(a, b))))
    _same(_dump(p), names, build_batch(chains, 25))


@pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref is built only where /root/reference exists")
@pytest.mark.parametrize("fn", FILES)
def test_goldens_are_what_the_live_reference_says(ing, fn):
    t, title, frag, nch = H.ref_load_structure(ing[f"file:{fn}"].tobytes(), fn)
    ref = _ref_table(ing, fn)
    assert t.atom == ref["atom"] and title == ref["title"] and frag == ref["frag"] and nch == ref["n_chains"]
    assert np.array_equal(t.xyz.view(np.uint32), ref["xyz"].view(np.uint32))


# ---- database container -------------------------------------------------------------------------------------------------
def _write_example_db(ing, d):
    for suffix in ("", ".index", ".lookup", ".dbtype"):
        (d / ("example_db" + suffix)).write_bytes(ing[f"file:example_db{suffix}"].tobytes())
    return str(d / "example_db")


def test_python_reader_on_the_reference_made_database(ing, golden, tmp_path):
    """DatabaseReader on test/example_db (an MMseqs2-made database) == what the reference's reader reports, == the goldens"""
    z, index = golden
    path = _write_example_db(ing, tmp_path)
    r = DatabaseReader(path)
    names = bytes(ing["dbr:names"]).decode().split("\n")
    assert len(r) == 24
    assert np.array_equal(r.keys, ing["dbr:keys"]) and np.array_equal(r.offsets, ing["dbr:offsets"]) and np.array_equal(r.lengths, ing["dbr:lengths"])
    for i in range(24):
        assert r.name(i) == names[i]
        assert r.data(i) == z[f"db:{i:02d}/fcz"].tobytes()
        assert r.id_of_name(names[i]) == i
    assert r.id_of_name("no_such_entry") == -1
    r.close()


def test_cpp_reader_on_the_reference_made_database(ing, golden, tmp_path):
    z, index = golden
    path = _write_example_db(ing, tmp_path)
    outd = tmp_path / "unpacked"
    r = _run("db-unpack", path, str(outd))
    assert r.returncode == 0, r.stderr
    names = bytes(ing["dbr:names"]).decode().split("\n")
    assert sorted(os.listdir(outd)) == sorted(names)
    for i, n in enumerate(names):
        assert (outd / n).read_bytes() == z[f"db:{i:02d}/fcz"].tobytes()


def test_python_writer_equals_the_reference_writer(ing, golden, tmp_path):
    """entries appended in a scrambled key order: data file, .index, .lookup and .dbtype == free_writer's bytes"""
    z, index = golden
    order = ing["dbw:order"]
    names = bytes(ing["dbr:names"]).decode().split("\n")
    w = DatabaseWriter(str(tmp_path / "w"))
    for i in order:
        w.append(z[f"db:{int(i):02d}/fcz"].tobytes(), int(ing["dbr:keys"][i]), names[i])
    w.close()
    for suffix in ("", ".index", ".lookup", ".dbtype"):
        assert (tmp_path / ("w" + suffix)).read_bytes() == ing[f"dbw:file{suffix}"].tobytes(), suffix


def test_cpp_writer_equals_the_reference_writer(ing, golden, tmp_path):
    """db-pack of the unpacked example_db: same files as the reference writer makes of the same entries in the same order"""
    z, index = golden
    names = bytes(ing["dbr:names"]).decode().split("\n")
    src = tmp_path / "files"
    src.mkdir()
    for i, n in enumerate(names):
        (src / n).write_bytes(z[f"db:{i:02d}/fcz"].tobytes())
    r = _run("db-pack", str(src), str(tmp_path / "packed"))
    assert r.returncode == 0, r.stderr
    # db-pack walks the directory in sorted order and numbers the entries 0.. : the example_db's own order
    for suffix in ("", ".index", ".lookup", ".dbtype"):
        assert (tmp_path / ("packed" + suffix)).read_bytes() == ing[f"file:example_db{suffix}"].tobytes(), suffix


@pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref is built only where /root/reference exists")
def test_reference_reader_reads_our_writers(ing, golden, tmp_path):
    """the other direction, live: the reference's reader on a database written by the Python writer"""
    z, index = golden
    names = bytes(ing["dbr:names"]).decode().split("\n")
    w = DatabaseWriter(str(tmp_path / "w"))
    for i in (5, 2, 9):
        w.append(z[f"db:{i:02d}/fcz"].tobytes(), i, names[i])
    w.close()
    rows = H.ref_db_read(str(tmp_path / "w"))
    assert [(r[0], r[3]) for r in rows] == [(2, names[2]), (5, names[5]), (9, names[9])]
    assert rows[1][4] == z["db:05/fcz"].tobytes()
    assert H.ref_db_lookup(str(tmp_path / "w"), names[9]) == 2


def _mutation_bases(ing):
    """test_af.pdb, and the head of multichain.pdb: some header records, the first 300 coordinate records (two chains, HETATM)"""
    mc = ing["file:multichain.pdb"].tobytes().decode("latin-1").splitlines()
    coord = [l for l in mc if l.startswith(("ATOM", "HETATM", "TER", "ANISOU"))]
    cut = [l for l in coord if l[21:22] == "A"][:200] + [l for l in coord if l[21:22] == "B"][:100]
    return [ing["file:test_af.pdb"].tobytes().decode("latin-1").splitlines(), mc[:6] + cut + ["END"]]


_RefWorker = H.RefWorker


def test_database_text_files_are_read_as_the_reference_reads_them(tmp_path):
    """.index / .lookup files a person has touched: no line end after the last line (the reference counts entries by line ends: the
    last one is not an entry), blanks for tabs, CR LF, a '+' before a key, a name with a blank in it (the name ends there), keys twice,
    out of order. The Python reader == the live reference's reader (the C++ reader holds the same rules; its databases are compared
    with the Python host's in test_host_cpp.py). Lines the reference reads undefined memory for (blank, fewer than three columns) are
    refused by both hosts, not compared"""
    if not H.have_ref():
        pytest.skip("oracle/_ref is not built (it only exists where /root/reference does)")
    rng = np.random.default_rng(3)
    ref = H.RefWorker(timeout=5)
    same = crashed = 0
    for it in range(120):
        n = int(rng.integers(1, 8))
        datas = [bytes(rng.integers(65, 91, int(rng.integers(1, 30)), dtype=np.uint8)) + b"\0" for _ in range(n)]
        offs = np.concatenate([[0], np.cumsum([len(d) for d in datas])])
        keys = [int(k) for k in rng.permutation(n * 2)[:n]]
        idx = [f"{keys[i]}\t{offs[i]}\t{len(datas[i])}" for i in range(n)]
        lk = [f"{keys[i]}\tname{keys[i]}\t0" for i in range(n)]
        for _ in range(int(rng.integers(0, 3))):
            kind = int(rng.integers(0, 7)); j = int(rng.integers(0, n))
            if kind == 0:
                idx[j] = idx[j].replace("\t", "  ")
            elif kind == 1:
                idx[j] = idx[j] + "\r"
            elif kind == 2:
                idx[j] = " " + idx[j]
            elif kind == 3:
                lk.insert(j, lk[j].replace("name", "other"))
            elif kind == 4:
                idx.insert(j, idx[j])
            elif kind == 5:
                lk[j] = lk[j].replace("name", "na me")
            else:
                idx[j] = "+" + idx[j]
        p = str(tmp_path / f"db{it}")
        open(p, "wb").write(b"".join(datas))
        open(p + ".index", "w").write("\n".join(idx) + ("\n" if rng.random() < 0.7 else ""))
        open(p + ".lookup", "w").write("\n".join(lk) + "\n")
        open(p + ".dbtype", "wb").write((12).to_bytes(4, "little"))
        r = ref.call("ref_db_read", p)
        if r[0] != "ok":
            crashed += 1; continue
        rd = DatabaseReader(p)
        mine = [(int(rd.keys[i]), int(rd.offsets[i]), int(rd.lengths[i]), rd.name(i), rd.data(i)) for i in range(len(rd))]
        rd.close()
        assert mine == [tuple(e) for e in r[1]], (it, idx, lk)
        same += 1
    ref.close()
    assert same > 80, (same, crashed)


def test_python_pdb_reader_equals_live_reference_on_mutated_files(ing):
    """the command line's PDB reader (foldcomp_amd.structure.parse_pdb_gemmi = gemmi's read_pdb + StructureReader::updateStructure
    restated) against the LIVE reference (oracle/_ref: gemmi 0.5.1 itself) on 1 500 seeded mutations of two files: same atoms in
    the same order -- residues regrouped, END / MODEL / ANISOU rules, lenient number fields, default B-factor 20 --, same title,
    same fragments, and the same verdict on the files the reader fails."""
    if not H.have_ref():
        pytest.skip("oracle/_ref is not built (it only exists where /root/reference does)")
    from _cases import mutated_pdb, reference_would_spin
    from foldcomp_amd.structure import StructureError, parse_pdb_gemmi
    bases = _mutation_bases(ing)
    rng = np.random.default_rng(20260927)
    same = failed = crashed = 0
    ref = _RefWorker()
    for i in range(1500):
        data = mutated_pdb(bases[i % 2], rng)
        name = f"fz{i}.pdb"
        try:
            t, title = parse_pdb_gemmi(data)
            t = remove_alternative_position(t)
        except StructureError:
            t = None
        if t is not None and reference_would_spin(t):
            continue
        r = ref.load(data, name)
        if r[0] == "crash":
            crashed += 1; continue
        rt, rtitle, rfrag, rnch = r[1:] if r[0] == "ok" else (None, None, None, None)
        assert (t is None) == (rt is None), (i, "only one of the two readers fails the file")
        if t is None:
            failed += 1; continue
        assert len(t) == len(rt), i
        assert t.atom == rt.atom and t.residue == rt.residue and [c[:1] or " " for c in t.chain] == rt.chain, i
        assert np.array_equal(t.atom_index, rt.atom_index) and np.array_equal(t.res_index, rt.res_index), i
        assert np.array_equal(t.xyz.view(np.uint32), rt.xyz.view(np.uint32)) and np.array_equal(t.bfac.view(np.uint32), rt.bfac.view(np.uint32)), i
        assert (title if title else name) == rtitle, i
        if len(t):
            chains = identify_chains(t)
            frag = [(sl.start, sl.stop, ci, j) for ci, cs in enumerate(chains) for j, sl in enumerate(identify_discontinuous(t, cs))]
            assert frag == rfrag and len(chains) == rnch, i
        same += 1
    ref.close()
    assert same > 900 and failed > 100 and crashed < 100, (same, failed, crashed)


def test_oracle_compress_equals_live_reference_on_mutated_files(ing):
    """what the codec is handed after the reader: the fragments of 500 mutated files (atoms out of order, doubled, missing, foreign
    records, cut lines ...). Every fragment the host accepts (build_batch) is compressed by the oracle and by the LIVE reference
    (Foldcomp::compress on the same atoms): same bytes. What the reference compresses and the host refuses are chains whose residues
    the reference counts differently (it works on the flat list of all N / CA / C atoms: a residue with a missing, a second or an
    out-of-order backbone atom shifts everything behind it) and one-residue fragments; the device's ingest is held to the host's
    verdicts and batches on such files in tests/test_gpu_ingest.py."""
    if not H.have_ref():
        pytest.skip("oracle/_ref is not built (it only exists where /root/reference does)")
    from _cases import mutated_pdb
    from foldcomp_amd.structure import StructureError, parse_pdb_gemmi
    bases = _mutation_bases(ing)
    rng = np.random.default_rng(99)
    ref = _RefWorker()
    same = refused = ref_failed = crashed = 0
    for i in range(400):
        data = mutated_pdb(bases[i % 2], rng)
        try:
            t, title = parse_pdb_gemmi(data)
        except StructureError:
            continue
        t = remove_alternative_position(t)
        if len(t) == 0:
            continue
        title = title or f"fz{i}"
        for cs in identify_chains(t):
            for sl in identify_discontinuous(t, cs):
                ft = t.take(sl)
                try:
                    b = build_batch([Chain(title, ft)], 25)
                except StructureError:
                    refused += 1; continue
                blob, off, st = H.oracle_compress(b)
                r = ref.compress(ft, title)
                if r[0] == "crash":
                    crashed += 1; continue
                if st[0] != 0:
                    assert st[0] == -7 or r[0] == "fail", (i, st[0], "the codec refuses a chain the reference compresses")   # -7: fewer than two residues
                    ref_failed += 1; continue
                if r[0] == "fail":
                    raise AssertionError((i, sl, "the reference refuses a chain the codec compresses"))
                assert r[0] == "ok", (i, sl)
                assert r[1] == blob.tobytes(), (i, sl, next(k for k in range(min(len(r[1]), len(blob))) if r[1][k] != blob[k]))
                same += 1
    ref.close()
    assert same > 300 and crashed < 60, (same, refused, ref_failed, crashed)


def _short_cif(ing, rows=260):
    import gzip
    lines = gzip.decompress(ing["file:test.cif.gz"].tobytes()).decode("latin-1").split("\n")
    at = [i for i, l in enumerate(lines) if l.startswith("ATOM")]
    return "\n".join(lines[:at[rows]] + lines[at[-1] + 1:])


def test_python_cif_reader_equals_live_reference_on_mutated_files(ing):
    """the mmCIF reader of the command line (foldcomp_amd.structure.parse_cif_gemmi: gemmi's CIF grammar, table look-ups and
    make_structure restated; parse_structure_gemmi: the format read off the content as StructureReader::loadFromBuffer does) against
    the LIVE reference on 900 seeded mutations of the reference's own mmCIF test file (_cases.mutated_cif): same verdict on the files
    the reader fails; same atoms in the same order (residues regrouped, models, chains of any name), numbers (uncertainties in
    brackets, nulls, defaults), title. Where the reference does not survive its own exception there is no answer to compare"""
    if not H.have_ref():
        pytest.skip("oracle/_ref is not built (it only exists where /root/reference does)")
    from _cases import mutated_cif
    from foldcomp_amd.structure import StructureError, parse_structure_gemmi
    text = _short_cif(ing)
    rng = np.random.default_rng(20260927)
    ref = H.RefWorker(timeout=5)
    same = failed = crashed = 0
    for i in range(900):
        data = mutated_cif(text, rng)
        try:
            t, title = parse_structure_gemmi(data)
            t = remove_alternative_position(t)
        except StructureError:
            t = None
        r = ref.load(data, "x.cif")
        if r[0] == "crash":
            crashed += 1; continue
        assert (t is None) == (r[0] != "ok"), (i, "only one of the two readers fails the file")
        if t is None:
            failed += 1; continue
        rt, rtitle = r[1], r[2]
        assert len(t) == len(rt), i
        # (the shim hands names on as 4 / 3 characters and the chain as one)
        assert [a[:4] for a in t.atom] == rt.atom and [x[:3] for x in t.residue] == rt.residue and [c[:1] or " " for c in t.chain] == rt.chain, i
        assert np.array_equal(t.atom_index, rt.atom_index) and np.array_equal(t.res_index, rt.res_index), i
        assert np.all((t.xyz.view(np.uint32) == rt.xyz.view(np.uint32)) | (np.isnan(t.xyz) & np.isnan(rt.xyz))), i
        assert np.array_equal(t.bfac.view(np.uint32), rt.bfac.view(np.uint32)) and (title if title else "x.cif") == rtitle, i
        same += 1
    ref.close()
    assert same > 350 and failed > 250 and crashed < 60, (same, failed, crashed)


def test_cpp_cif_reader_equals_python_reader_on_mutated_files(ing, tmp_path):
    """the C++ host's mmCIF reader (parse_cif_gemmi / parse_structure_gemmi in host/foldcomp_hip.cpp, through dump-batch on a
    directory) == the Python one on 300 mutated files -- named .cif and .pdb alike: the content decides"""
    from _cases import mutated_cif
    from foldcomp_amd.structure import StructureError, parse_structure_gemmi
    text = _short_cif(ing, rows=120)
    rng = np.random.default_rng(11)
    d = tmp_path / "fz"
    d.mkdir()
    names, chains = [], []
    for i in range(300):
        data = mutated_cif(text, rng)
        stem = f"f{i:03d}"
        ext = ".cif" if i % 3 else ".pdb"
        (d / (stem + ext)).write_bytes(data)
        try:
            t, title = parse_structure_gemmi(data)
        except StructureError:
            continue
        if len(t) == 0:
            continue
        title = stem if (not title or title == stem + ext) else title
        t = remove_alternative_position(t)
        cs_all = identify_chains(t)
        for cs in cs_all:
            frags = identify_discontinuous(t, cs)
            for j, sl in enumerate(frags):
                ch = Chain(title, t.take(sl))
                try:
                    build_batch([ch], 25)
                except Exception:
                    continue
                names.append(stem + (t.chain[cs.start] if len(cs_all) > 1 else "") + (f"_{j}" if len(frags) > 1 else "") + ".fcz")
                chains.append(ch)
    assert len(chains) > 100
    _same(_dump(d), names, build_batch(chains, 25))


def test_host_library_readers_equal_python_readers_on_mutated_files(ing):
    """host/libfcz_host.so (fcz_host_read_structure: what `python -m foldcomp_amd` reads files with) == the Python restatement
    (structure.parse_structure_gemmi) on 600 mutated PDB files and 600 mutated mmCIF files, plain and gzipped: the same atoms, names
    of any length, chain names, numbers bit for bit, title -- or both fail the file"""
    import gzip
    from _cases import mutated_cif, mutated_pdb
    from foldcomp_amd import _hostlib
    from foldcomp_amd.structure import StructureError, parse_structure_gemmi
    assert _hostlib.load() is not None, "host/libfcz_host.so is not built (make -C host)"
    bases = _mutation_bases(ing)
    cif = _short_cif(ing, rows=150)
    rng = np.random.default_rng(5)
    same = failed = 0
    for i in range(1200):
        data = mutated_pdb(bases[i % 2], rng) if i % 2 == 0 else mutated_cif(cif, rng)
        gz = i % 5 == 0
        try:
            p, ptitle = parse_structure_gemmi(data)
        except StructureError:
            p = None
        try:
            t, title = _hostlib.read_structure(gzip.compress(data, 1) if gz else data, gz=gz)
        except StructureError:
            t = None
        assert (p is None) == (t is None), (i, "only one of the two fails the file")
        if p is None:
            failed += 1; continue
        assert t.atom == p.atom and t.residue == p.residue and t.chain == p.chain, i
        assert np.array_equal(t.atom_index, p.atom_index) and np.array_equal(t.res_index, p.res_index), i
        assert np.all((t.xyz.view(np.uint32) == p.xyz.view(np.uint32)) | (np.isnan(t.xyz) & np.isnan(p.xyz))), i
        assert np.array_equal(t.bfac.view(np.uint32), p.bfac.view(np.uint32)) and title == ptitle, i
        same += 1
    assert same > 500 and failed > 200, (same, failed)


def test_cpp_pdb_reader_equals_python_reader_on_mutated_files(ing, tmp_path):
    """the C++ host's reader (parse_pdb_gemmi in host/foldcomp_hip.cpp, through dump-batch on a directory) == the Python one on
    300 mutated files: fragments, names, every array of the batch; files the reader fails and fragments the codec refuses drop out
    of both"""
    from _cases import mutated_pdb
    from foldcomp_amd.structure import StructureError, parse_pdb_gemmi
    bases = _mutation_bases(ing)
    rng = np.random.default_rng(7)
    d = tmp_path / "fz"
    d.mkdir()
    names, chains = [], []
    for i in range(300):
        data = mutated_pdb(bases[i % 2], rng)
        stem = f"f{i:03d}"
        (d / (stem + ".pdb")).write_bytes(data)
        try:
            t, title = parse_pdb_gemmi(data)
        except StructureError:
            continue
        if len(t) == 0:
            continue
        title = stem if (not title or title == stem + ".pdb") else title
        t = remove_alternative_position(t)
        cs_all = identify_chains(t)
        for cs in cs_all:
            frags = identify_discontinuous(t, cs)
            for j, sl in enumerate(frags):
                ch = Chain(title, t.take(sl))
                try:
                    build_batch([ch], 25)
                except Exception:
                    continue
                names.append(stem + (t.chain[cs.start] if len(cs_all) > 1 else "") + (f"_{j}" if len(frags) > 1 else "") + ".fcz")
                chains.append(ch)
    _same(_dump(d), names, build_batch(chains, 25))


def _cpp_names_in_order(db, threads):
    """the entries of a database as the C++ host's reader walks them: `check` names every entry, valid or not, in reader order"""
    from test_host_cpp import BIN
    env = dict(os.environ, OMP_NUM_THREADS=str(threads))
    r = subprocess.run([BIN, "check", db], capture_output=True, text=True, timeout=120, env=env)
    names = [ln[len("[Error] "):].rsplit(": ", 1)[0] for ln in r.stderr.splitlines() if ln.startswith("[Error] ") and ln.endswith(": not a valid FCZ entry")]
    return r, names


def test_cpp_reader_reads_database_text_like_the_python_reader(tmp_path):
    """host/foldcomp_hip.cpp's DbReader (words read in place, both text files cut into one piece per host thread above 64 KB) against
    the Python reader -- which the test above holds to the live reference -- on the same hand-touched .index / .lookup files, and on
    a 60 000-entry database whose lines are out of key order with keys that come twice: one thread and eight threads walk the same
    entries under the same names, and `db-unpack` finds the same bytes under every name"""
    _run("version")
    rng = np.random.default_rng(5)
    compared = refused = 0
    for it in range(80):
        n = int(rng.integers(1, 8))
        datas = [bytes(rng.integers(65, 91, int(rng.integers(1, 30)), dtype=np.uint8)) + b"\0" for _ in range(n)]
        offs = np.concatenate([[0], np.cumsum([len(d) for d in datas])])
        keys = [int(k) for k in rng.permutation(n * 2)[:n]]
        idx = [f"{keys[i]}\t{offs[i]}\t{len(datas[i])}" for i in range(n)]
        lk = [f"{keys[i]}\tname{keys[i]}\t0" for i in range(n)]
        for _ in range(int(rng.integers(0, 3))):
            kind = int(rng.integers(0, 8)); j = int(rng.integers(0, n))
            if kind == 0:
                idx[j] = idx[j].replace("\t", "  ")
            elif kind == 1:
                idx[j] = idx[j] + "\r"
            elif kind == 2:
                idx[j] = " " + idx[j]
            elif kind == 3:
                lk.insert(j, lk[j].replace("name", "other"))
            elif kind == 4:
                idx.insert(j, idx[j])
            elif kind == 5:
                lk[j] = lk[j].replace("name", "na me")
            elif kind == 6:
                idx[j] = "+" + idx[j]
            else:
                lk[j] = lk[j].replace("\t", " ", 1)
        p = str(tmp_path / f"db{it}")
        open(p, "wb").write(b"".join(datas))
        open(p + ".index", "w").write("\n".join(idx) + ("\n" if rng.random() < 0.7 else ""))
        open(p + ".lookup", "w").write("\n".join(lk) + ("\n" if rng.random() < 0.7 else ""))
        open(p + ".dbtype", "wb").write((12).to_bytes(4, "little"))
        try:
            rd = DatabaseReader(p)
            want = [(rd.name(i), rd.data(i)) for i in range(len(rd))]
            rd.close()
        except Exception:
            want = None
        r, names = _cpp_names_in_order(p, 1)
        if want is None:
            refused += 1
            assert r.returncode != 0 or not names, (it, idx)
            continue
        assert r.returncode == 0 and names == [w[0] for w in want], (it, idx, lk, r.stderr)
        out = tmp_path / f"un{it}"
        assert _run("db-unpack", p, str(out)).returncode == 0
        last = {}
        for nm, d in want:
            last[os.path.basename(nm)] = d
        assert {f: open(out / f, "rb").read() for f in os.listdir(out)} == last, it
        compared += 1
    assert compared > 50, (compared, refused)

    # the pieces: 60 000 lines (1.3 MB of index, 1.6 MB of lookup), written out of key order, 300 keys twice in the lookup
    n = 60000
    order = rng.permutation(n)
    recs = [b"%08d" % i + b"\0" for i in range(n)]
    p = str(tmp_path / "big")
    open(p, "wb").write(b"".join(recs))
    open(p + ".index", "w").write("".join(f"{int(k)}\t{int(k) * 9}\t9\n" for k in order))
    lk = [f"{int(k)}\tAF-{int(k):08d}-F1-model_v4\t0\n" for k in rng.permutation(n)]
    for k in rng.integers(0, n, 300):
        lk.insert(int(rng.integers(0, len(lk))), f"{int(k)}\ttwice-{int(k)}\t0\n")
    open(p + ".lookup", "w").write("".join(lk))
    open(p + ".dbtype", "wb").write((12).to_bytes(4, "little"))
    rd = DatabaseReader(p)
    want = [rd.name(i) for i in range(len(rd))]
    assert [rd.data(i) for i in range(0, n, 997)] == [recs[i] for i in range(0, n, 997)]
    rd.close()
    r1, names1 = _cpp_names_in_order(p, 1)
    r8, names8 = _cpp_names_in_order(p, 8)
    assert r1.returncode == 0 and r8.returncode == 0
    assert names1 == want and names8 == want and r1.stderr == r8.stderr


def test_readers_equal_live_reference_on_rendered_variants_and_composite_files():
    """the command line's readers (the Python restatement, and the C++ library's where it is built) against the LIVE reference on the corpus of the differential fuzz: the input variants written as PDB
    text ("-0.000", columns that overflow or run together, B-factors of every kind, UNK, long chains ...), the same chains as mmCIF
    (AFDB's shape and the archive's), and composite files as depositions look (several chains, gaps, alternative locations,
    insertion codes, waters, CRLF): same verdict, same atoms in the same order bit for bit, same title, same fragments"""
    if not H.have_ref():
        pytest.skip("oracle/_ref is not built (it only exists where /root/reference does)")
    import sys
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import bench
    import host_text
    from _cases import composite_pdb, input_variants, reference_would_spin, _variant_base
    from foldcomp_amd import _hostlib
    from foldcomp_amd.structure import StructureError, parse_structure_gemmi
    rng = np.random.default_rng(20261001)
    files = []
    for vi, (name, b) in enumerate(input_variants(rng, 3)):
        if name.startswith("side chains: extra"):
            continue
        res_of_atom = np.repeat(np.arange(b.n_residues), np.diff(b.atom_off.astype(np.int64)))
        for c in range(min(b.n_chains, 2)):
            r0, r1 = int(b.res_off[c]), int(b.res_off[c + 1])
            if r1 - r0 > 500:
                continue
            sl = slice(int(b.atom_off[r0]), int(b.atom_off[r1]))
            text = host_text.format_pdb(f"V{vi}", b.atom_code[sl], b.res_code[res_of_atom[sl]], int(b.first_res_index[c]) + res_of_atom[sl] - r0,
                                        chr(b.chain_id[c]) if 32 < b.chain_id[c] < 127 else "A", int(b.first_atom_index[c]), b.x[sl], b.y[sl], b.z[sl],
                                        b.bfac_ca[res_of_atom[sl]]).encode("latin-1")
            files.append((f"v{vi}_{c}.pdb", text))
            if c == 0:
                try:
                    files.append((f"v{vi}.cif", bench.cif_from_pdb_text(text, f"V{vi}") if vi % 2 == 0 else
                                  bench.cif_archive_from_pdb_text(text, f"V{vi}", bench.ARCHIVE_STYLES[(7 * vi) % len(bench.ARCHIVE_STYLES)])))
                except Exception:
                    pass
    pool = _variant_base(rng, 40, 4, 150)
    files += [(f"w{i}.pdb", composite_pdb(rng, pool, f"COMPOSITE {i}")) for i in range(80)]
    ref = H.RefWorker(timeout=5)
    same = failed = crashed = spun = 0
    for name, data in files:
        try:
            t, title = parse_structure_gemmi(data)
        except StructureError:
            t = None
        if _hostlib.load() is not None:                       # the C++ readers (what the hosts read files with) == the Python restatement
            try:
                ct, ctitle = _hostlib.read_structure(data, gz=False)
            except StructureError:
                ct = None
            assert (ct is None) == (t is None), (name, "only one of the C++ and the Python reader fails the file")
            if t is not None:
                assert ct.atom == t.atom and ct.residue == t.residue and ct.chain == t.chain and ctitle == title, name
                assert np.array_equal(ct.atom_index, t.atom_index) and np.array_equal(ct.res_index, t.res_index), name
                assert np.all((ct.xyz.view(np.uint32) == t.xyz.view(np.uint32)) | (np.isnan(ct.xyz) & np.isnan(t.xyz))) and np.array_equal(ct.bfac.view(np.uint32), t.bfac.view(np.uint32)), name
        if t is not None:
            t = remove_alternative_position(t)
        if t is not None and reference_would_spin(t):
            spun += 1; continue
        r = ref.load(data, name)
        if r[0] == "crash":
            crashed += 1; continue
        assert (t is None) == (r[0] != "ok"), (name, "only one of the two readers fails the file")
        if t is None:
            failed += 1; continue
        rt, rtitle = r[1], r[2]
        assert len(t) == len(rt), name
        assert [a[:4] for a in t.atom] == rt.atom and [x[:3] for x in t.residue] == rt.residue and [c[:1] or " " for c in t.chain] == rt.chain, name
        assert np.array_equal(t.atom_index, rt.atom_index) and np.array_equal(t.res_index, rt.res_index), name
        assert np.all((t.xyz.view(np.uint32) == rt.xyz.view(np.uint32)) | (np.isnan(t.xyz) & np.isnan(rt.xyz))), name
        assert np.array_equal(t.bfac.view(np.uint32), rt.bfac.view(np.uint32)) and (title if title else name) == rtitle, name
        if len(r) > 4 and len(t):
            chains = identify_chains(t)
            frag = [(sl.start, sl.stop, ci, j) for ci, cs in enumerate(chains) for j, sl in enumerate(identify_discontinuous(t, cs))]
            assert frag == r[3] and len(chains) == r[4], name
        same += 1
    ref.close()
    print({"files": len(files), "same": same, "failed by both": failed, "reference crashed": crashed, "not put to it (identifyChains spins)": spun})
    assert same > 300 and crashed < 40, (same, failed, crashed, spun)
