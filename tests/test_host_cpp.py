"""The C++ host (host/foldcomp-hip) over the C-ABI.
CPU: `dump-batch` -- the SoA batch it hands to fcz_compress_batch -- equals what the Python host builds from the same PDB
text (parser, alternative positions, chain and gap splitting, residue splitting, codes, CA B-factors, titles, names).
GPU: compress / decompress / extract / check of files and directories against the reference-minted goldens."""
import os
import subprocess

import numpy as np
import pytest

from _cases import golden_batch
import host_text as pdbio   # oracle/host_text.py
from foldcomp_amd.structure import Chain, build_batch, identify_chains, identify_discontinuous, parse_pdb, remove_alternative_position

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "host", "foldcomp-hip")


def _pdb_text(z, name, chain=None, first_res=None):
    b = golden_batch(z, name)
    res_of_atom = np.repeat(np.arange(b.n_residues), np.diff(b.atom_off.astype(np.int64)))
    fr = int(b.first_res_index[0]) if first_res is None else first_res
    return pdbio.format_pdb("", b.atom_code, b.res_code[res_of_atom], fr + res_of_atom,
                            chain or chr(b.chain_id[0]), int(b.first_atom_index[0]), b.x, b.y, b.z, b.bfac_ca[res_of_atom])


def _run(*args):
    if not os.path.exists(BIN):   # normally built by __graft_entry__.build(); g++ and the library are enough to build it here
        subprocess.run(["make", "-C", os.path.join(ROOT, "host")], capture_output=True, timeout=300)
    assert os.path.exists(BIN), "host/foldcomp-hip missing: run __graft_entry__.build()"
    return subprocess.run([BIN, *args], capture_output=True, text=True, timeout=120)


def _dump(path, brk=25):
    r = _run("dump-batch", "-b", str(brk), str(path))
    assert r.returncode == 0, r.stderr
    d = {"fragment": [], "rejected": []}
    for line in r.stdout.splitlines():
        k, _, v = line.partition(" ")
        if k in ("fragment", "rejected"):
            d[k].append(v)
        else:
            d[k] = v
    return d


def _python_batch(text, stem, brk=25):
    from foldcomp_amd.structure import parse_pdb_gemmi
    t = remove_alternative_position(parse_pdb_gemmi(text.encode("latin-1") if isinstance(text, str) else text)[0])
    chains = identify_chains(t)
    names, cs_list = [], []
    for cs in chains:
        frags = identify_discontinuous(t, cs)
        for j, sl in enumerate(frags):
            nm = stem + (t.chain[cs.start] if len(chains) > 1 else "") + (f"_{j}" if len(frags) > 1 else "") + ".fcz"
            names.append(nm); cs_list.append(Chain(stem, t.take(sl)))
    return names, build_batch(cs_list, brk)


def _same(d, names, b):
    assert d["fragment"] == names
    ints = lambda k: np.asarray(d[k].split(), np.int64) if d[k] else np.zeros(0, np.int64)
    hexf = lambda k: np.asarray([int(v, 16) for v in d[k].split()], np.uint32)
    assert np.array_equal(ints("res_off"), b.res_off) and np.array_equal(ints("atom_off"), b.atom_off)
    assert np.array_equal(ints("title_off"), b.title_off)
    assert np.array_equal(ints("res_code"), b.res_code) and np.array_equal(ints("atom_code"), b.atom_code)
    for k, a in (("x", b.x), ("y", b.y), ("z", b.z), ("bfac_ca", b.bfac_ca)):
        assert np.array_equal(hexf(k), a.view(np.uint32)), k
    assert np.array_equal(ints("first_res"), b.first_res_index) and np.array_equal(ints("first_atom"), b.first_atom_index)
    assert d["chain_id"] == bytes(b.chain_id).decode() and d["titles"] == bytes(b.titles).decode()


@pytest.mark.parametrize("name", ["pdb:test_af", "pdb:test", "syn:len350", "syn:len26"])
def test_cpp_host_batch_equals_python_host(tmp_path, golden, name):
    z, _ = golden
    text = _pdb_text(z, name)
    p = tmp_path / "in.pdb"
    p.write_text(text)
    names, b = _python_batch(text, "in")
    _same(_dump(p), names, b)


def test_cpp_host_multichain_gaps_altloc_hetatm(tmp_path, golden):
    z, _ = golden
    a = _pdb_text(z, "pdb:multichainA")
    b0, b1 = _pdb_text(z, "pdb:multichainB_0"), _pdb_text(z, "pdb:multichainB_1")
    # duplicate every 7th ATOM line (an alternative position), add HETATM-free noise lines and a HEADER without id
    la = a.splitlines()
    dup = []
    for i, l in enumerate(la):
        dup.append(l)
        if l.startswith("ATOM") and i % 7 == 3:
            dup.append(l[:30] + "   9.999   9.999   9.999" + l[54:])
    text = "REMARK test\n" + "\n".join(dup) + "\n" + b0 + b1 + "END\n"
    p = tmp_path / "multi.pdb"
    p.write_text(text)
    names, bb = _python_batch(text, "multi")
    assert len(names) == 3 and names[0].startswith("multi") and names[1].endswith("_0.fcz")
    _same(_dump(p, brk=10), names, _python_batch(text, "multi", 10)[1])
    # a residue the codec cannot take is reported, not compressed
    bad = a.replace(" ALA ", " MSE ", 3)
    q = tmp_path / "bad.pdb"
    q.write_text(bad)
    d = _dump(q)
    if " MSE " in bad:
        assert d["rejected"] and "MSE" in d["rejected"][0] and d["n_chains"].startswith("0")


@pytest.mark.gpu
def test_cpp_cli_files_and_directories(tmp_path, golden):
    z, _ = golden
    d = tmp_path / "in"
    d.mkdir()
    names = ["pdb:test_af", "pdb:test", "syn:len350", "syn:len26"]
    for n in names:
        (d / (n.split(":")[1] + ".pdb")).write_text(_pdb_text(z, n))
    r = _run("compress", str(d), str(tmp_path / "fcz"))
    assert r.returncode == 0, r.stderr

    def no_title(f):   # everything except lenTitle and the title bytes (CLI title = file stem)
        na, tl = f[12], int.from_bytes(f[24:28], "little")
        return f[:24] + f[28:76 + 4 * na] + f[76 + 4 * na + tl:]
    for n in names:
        stem = n.split(":")[1]
        got = (tmp_path / "fcz" / f"{stem}.fcz").read_bytes()
        assert no_title(got) == no_title(z[f"{n}/fcz"].tobytes()), n
    r = _run("decompress", str(tmp_path / "fcz"), str(tmp_path / "pdb"))
    assert r.returncode == 0, r.stderr
    for n in names:
        stem = n.split(":")[1]
        back = (tmp_path / "pdb" / f"{stem}.pdb").read_text()
        ref = z[f"{n}/pdb0"].tobytes().decode("latin-1")
        assert [l for l in back.splitlines() if l.startswith(("ATOM", "TER"))] == [l for l in ref.splitlines() if l.startswith(("ATOM", "TER"))], n
    r = _run("extract", "--plddt", "-p", "2", str(tmp_path / "fcz"), str(tmp_path / "out.tsv"))
    assert r.returncode == 0, r.stderr
    # (an entry goes by its name as the run met it -- a directory's file with its path, src/main.cpp:780-781; held against the
    #  reference's own command line in tests/test_cli_vs_reference.py)
    rows = {os.path.basename(l.split("\t")[0]): l.split("\t") for l in (tmp_path / "out.tsv").read_text().splitlines()}
    assert rows["test_af.fcz"][0] == str(tmp_path / "fcz" / "test_af.fcz")
    assert rows["test_af.fcz"][2] == z["pdb:test_af/plddt2"].tobytes().decode()
    r = _run("extract", "--fasta", str(tmp_path / "fcz" / "test.fcz"), str(tmp_path / "seq.fasta"))
    assert (tmp_path / "seq.fasta").read_text().splitlines()[1] == z["pdb:test/fasta"].tobytes().decode()
    r = _run("check", str(tmp_path / "fcz"))
    assert r.stdout.count("is valid") == 4
    # single file, alternative atom order
    r = _run("decompress", "-a", str(tmp_path / "fcz" / "test_af.fcz"), str(tmp_path / "alt.pdb"))
    assert r.returncode == 0 and (tmp_path / "alt.pdb").read_text().count("ATOM") == z["pdb:test_af/pdb0"].tobytes().count(b"ATOM")


def test_cpp_host_database_container(tmp_path, golden):
    """db-pack / db-unpack (no GPU): the C++ reader and writer interoperate with the Python ones (same .index / .lookup /
    .dbtype text, NUL-terminated entries stripped on read)"""
    from foldcomp_amd.database import DatabaseReader, DatabaseWriter
    z, index = golden
    names = [n for n in index if n.startswith("db:")][:6]
    src = tmp_path / "files"
    src.mkdir()
    for n in names:
        (src / (n.replace(":", "_") + ".fcz")).write_bytes(z[f"{n}/fcz"].tobytes())
    r = _run("db-pack", str(src), str(tmp_path / "cdb"))
    assert r.returncode == 0, r.stderr
    rd = DatabaseReader(str(tmp_path / "cdb"))
    assert len(rd) == len(names)
    got = {rd.name(i): rd.data(i) for i in range(len(rd))}
    rd.close()
    for n in names:
        assert got[n.replace(":", "_")] == z[f"{n}/fcz"].tobytes()
    assert (tmp_path / "cdb.dbtype").read_bytes() == (12).to_bytes(4, "little")
    # Python writer (entries with the MMseqs NUL terminator, keys out of order) -> C++ reader
    w = DatabaseWriter(str(tmp_path / "pdb"))
    for k, n in reversed(list(enumerate(names))):
        w.append(z[f"{n}/fcz"].tobytes() + b"\0", k, n.replace(":", "_") + ".fcz")
    w.close()
    r = _run("db-unpack", str(tmp_path / "pdb"), str(tmp_path / "out"))
    assert r.returncode == 0, r.stderr
    for n in names:   # stored bytes, terminator included (a record may itself end in zero bytes: nothing is stripped)
        assert (tmp_path / "out" / (n.replace(":", "_") + ".fcz")).read_bytes() == z[f"{n}/fcz"].tobytes() + b"\0"
    # check runs on the host only: a database input is walked entry by entry; trailing terminators do not matter
    for db in ("cdb", "pdb"):
        r = _run("check", str(tmp_path / db))
        assert r.stdout.count("is valid") == len(names), (db, r.stderr)


def _cif_text(z, name, entry_id="1ABC"):
    from foldcomp_amd._aa_tables import ATOM_NAMES, RES3
    b = golden_batch(z, name)
    res_of_atom = np.repeat(np.arange(b.n_residues), np.diff(b.atom_off.astype(np.int64)))
    # (the columns gemmi's reader requires: id, type_symbol, label_alt_id, label_asym_id, Cartn_*, occupancy, B_iso_or_equiv,
    # auth_seq_id, and a comp_id / atom_id of either kind -- lib/gemmi/mmcif.hpp:565-598; without one of them there are no atoms)
    cols = ["group_PDB", "id", "type_symbol", "label_atom_id", "label_alt_id", "label_comp_id", "label_asym_id", "auth_asym_id", "auth_seq_id",
            "Cartn_x", "Cartn_y", "Cartn_z", "occupancy", "B_iso_or_equiv"]
    out = ["data_" + entry_id, "_entry.id " + entry_id, "#", "loop_"] + ["_atom_site." + c for c in cols]
    for i in range(b.n_atoms):
        an = ATOM_NAMES[b.atom_code[i]]
        r = int(res_of_atom[i])
        out.append("ATOM %d %s %s . %s %s %s %d %.3f %.3f %.3f 1.00 %.2f" % (
            int(b.first_atom_index[0]) + i, an[0], ('"%s"' % an) if "'" in an else an, RES3[b.res_code[r]], chr(b.chain_id[0]), chr(b.chain_id[0]),
            int(b.first_res_index[0]) + r, b.x[i], b.y[i], b.z[i], b.bfac_ca[r]))
    out.append("#")
    return "\n".join(out) + "\n"


def test_cpp_host_reads_mmcif_and_gzip_like_the_python_host(tmp_path, golden):
    import gzip
    from foldcomp_amd.__main__ import load_structure
    z, _ = golden
    cif = _cif_text(z, "pdb:test_af")
    pdb = _pdb_text(z, "pdb:test")
    (tmp_path / "a.cif").write_text(cif)
    (tmp_path / "b.cif.gz").write_bytes(gzip.compress(cif.encode()))
    (tmp_path / "c.pdb.gz").write_bytes(gzip.compress(pdb.encode()))
    # stems as getFileParts makes them (reference src/utility.cpp:118-126: split at the LAST dot; pinned against the real reference
    # in test_ingest_vs_reference.py)
    for fname, stem in (("a.cif", "a"), ("b.cif.gz", "b.cif"), ("c.pdb.gz", "c.pdb")):
        path = tmp_path / fname
        t, title = load_structure(str(path), path.read_bytes())
        if title == fname:
            title = stem
        t = remove_alternative_position(t)
        chains = identify_chains(t)
        assert len(chains) == 1
        frags = identify_discontinuous(t, chains[0])
        b = build_batch([Chain(title, t.take(sl)) for sl in frags], 25)
        names = [stem + (f"_{j}" if len(frags) > 1 else "") + ".fcz" for j in range(len(frags))]
        _same(_dump(path), names, b)
    assert _dump(tmp_path / "a.cif")["titles"] == "1ABC"


def test_cpp_host_directory_is_parsed_in_file_order(tmp_path, golden):
    """a directory goes through the multi-threaded parse: fragments and arrays must come out in sorted file order"""
    z, _ = golden
    d = tmp_path / "many"
    d.mkdir()
    cases = ["pdb:test_af", "pdb:test", "syn:len350", "syn:len26", "pdb:multichainA"]
    texts = {}
    for rep in range(6):
        for n in cases:
            stem = f"{n.split(':')[1]}_{rep:02d}"
            texts[stem] = _pdb_text(z, n)
            (d / (stem + ".pdb")).write_text(texts[stem])
    names, chains = [], []
    for stem in sorted(texts):                      # list_files sorts by path
        nm, b1 = _python_batch(texts[stem], stem)
        names += nm
    # one Python batch over everything, in the same order
    from foldcomp_amd.structure import parse_pdb_gemmi
    all_chains = []
    for stem in sorted(texts):
        t = remove_alternative_position(parse_pdb_gemmi(texts[stem].encode("latin-1"))[0])
        for cs in identify_chains(t):
            for sl in identify_discontinuous(t, cs):
                all_chains.append(Chain(stem, t.take(sl)))
    _same(_dump(d), names, build_batch(all_chains, 25))


def test_cpp_host_rmsd_matches_python_host(tmp_path, golden, capsys):
    from foldcomp_amd.__main__ import main as py_main
    z, _ = golden
    a = tmp_path / "a.pdb"; b = tmp_path / "b.pdb"
    a.write_text(_pdb_text(z, "pdb:test_af"))
    b.write_text(z["pdb:test_af/pdb0"].tobytes().decode("latin-1"))
    r = _run("rmsd", str(a), str(b))
    assert r.returncode == 0, r.stderr
    py_main(["rmsd", str(a), str(b)])
    py = capsys.readouterr().out.strip().split("\t")
    cc = r.stdout.strip().split("\t")
    assert cc[:4] == py[:4]                                  # files, residues, atoms
    assert abs(float(cc[4]) - float(py[4])) < 1e-3 and abs(float(cc[5]) - float(py[5])) < 1e-3
    assert _run("rmsd", str(a), str(a)).stdout.strip().split("\t")[4:] == ["0", "0"]


@pytest.mark.gpu
def test_cpp_compress_pipeline_many_jobs(tmp_path, golden):
    """`compress -d` as a pipeline: 700 files -> several jobs over 3 worker threads (own ctx and stream each, page-locked
    buffers, pwrite at sequenced offsets). The database must not depend on which worker finished first: keys in file order,
    every record == the record a single-file run makes of the same file, data file = records back to back."""
    from foldcomp_amd.database import DatabaseReader
    z, _ = golden
    d = tmp_path / "in"
    d.mkdir()
    srcs = ["pdb:test_af", "syn:len26", "syn:len129", "pdb:test", "syn:len350"]
    texts = {n: _pdb_text(z, n) for n in srcs}
    order = []
    for i in range(700):
        n = srcs[(i * 7) % len(srcs)]
        (d / f"f{i:04d}.pdb").write_text(texts[n]); order.append(n)
    (d / "f0350.pdb").write_text("HEADER    nothing to see\n")          # a file without atoms: reported, skipped, no hole in the keys
    r = _run("compress", "-d", "-y", "-t", "8", "--gpus", "1", "--workers-per-gpu", "3", "--json-stats", str(d), str(tmp_path / "db"))
    assert r.returncode == 0, r.stderr
    import json
    st = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert st["records"] == 699 and st["workers"] == 3 and st["files"] == 700
    single = {}
    for n in srcs:
        p = tmp_path / f"one_{n.split(':')[1]}.pdb"
        p.write_text(texts[n])
        rr = _run("compress", "-y", str(p), str(tmp_path / f"one_{n.split(':')[1]}.fcz"))
        assert rr.returncode == 0, rr.stderr
        single[n] = (tmp_path / f"one_{n.split(':')[1]}.fcz").read_bytes()

    def no_title(f):
        na, tl = f[12], int.from_bytes(f[24:28], "little")
        return f[:24] + f[28:76 + 4 * na] + f[76 + 4 * na + tl:]
    rd = DatabaseReader(str(tmp_path / "db"))
    assert len(rd) == 699 and list(rd.keys) == list(range(699))
    names = [f"f{i:04d}" for i in range(700) if i != 350]
    kept = [o for i, o in enumerate(order) if i != 350]
    for i in range(699):
        assert rd.name(i) == names[i]
        assert no_title(rd.data(i)) == no_title(single[kept[i]]), i
    assert int(rd.offsets[-1] + rd.lengths[-1]) == os.path.getsize(tmp_path / "db")
    assert all(int(rd.offsets[i] + rd.lengths[i]) == int(rd.offsets[i + 1]) for i in range(698))
    rd.close()
    # more GPUs than the box has: refused, not silently run on fewer
    import torch
    if torch.cuda.device_count() < 8:
        r = _run("compress", "-d", "-y", "--gpus", "8", str(d), str(tmp_path / "db8"))
        assert r.returncode != 0 and "device" in r.stderr


@pytest.mark.gpu
def test_cpp_decompress_pipeline_database_round_trip(tmp_path, golden):
    """database -> database through the decompress pipeline (3 workers): PDB text entries carry the MMseqs terminator like the
    reference's `decompress --db` (src/main.cpp:659), keys follow the input order, every text == the reference's text"""
    from foldcomp_amd.database import DatabaseReader, DatabaseWriter
    z, index = golden
    names = [n for n in index if n.startswith("db:")]
    w = DatabaseWriter(str(tmp_path / "fczdb"))
    for rep in range(30):                                   # 720 entries: several jobs when the job size is small
        for i, n in enumerate(names):
            w.append(z[f"{n}/fcz"].tobytes(), rep * len(names) + i, f"{bytes(z[f'{n}/name']).decode()}_{rep}")
    w.close()
    r = _run("decompress", "-d", "-y", "--gpus", "1", "--workers-per-gpu", "3", "--json-stats", str(tmp_path / "fczdb"), str(tmp_path / "pdbdb"))
    assert r.returncode == 0, r.stderr
    rd = DatabaseReader(str(tmp_path / "pdbdb"))
    assert len(rd) == 720 and list(rd.keys) == list(range(720))
    for k in (0, 1, 23, 24, 100, 719):
        n = names[k % len(names)]
        e = rd.data(k)
        assert e.endswith(b"\0") and e[:-1].decode("latin-1") == z[f"{n}/pdb0"].tobytes().decode("latin-1"), k
        assert rd.name(k) == f"{bytes(z[f'{n}/name']).decode()}_{k // len(names)}"
    rd.close()
    # a directory of FCZ files -> a directory of PDB files through the same workers
    r = _run("db-unpack", str(tmp_path / "fczdb"), str(tmp_path / "fczdir"))
    assert r.returncode == 0
    r = _run("decompress", "-y", "--gpus", "1", str(tmp_path / "fczdir"), str(tmp_path / "pdbdir"))
    assert r.returncode == 0, r.stderr
    assert len(os.listdir(tmp_path / "pdbdir")) == 720
    n0 = names[5]
    assert (tmp_path / "pdbdir" / f"{bytes(z[f'{n0}/name']).decode()}_7.pdb").read_bytes().decode("latin-1") == z[f"{n0}/pdb0"].tobytes().decode("latin-1")


def _golden_db(tmp_path, z, index, n_names=8):
    """a database of golden FCZ records (keys out of order, one corrupted record) + what it holds"""
    from foldcomp_amd.database import DatabaseWriter
    names = [n for n in index if n.startswith("db:")][:n_names]
    w = DatabaseWriter(str(tmp_path / "gdb"))
    recs = {}
    for k, n in reversed(list(enumerate(names))):
        nm = bytes(z[f"{n}/name"]).decode()
        recs[nm] = (10 + k, n)
        w.append(z[f"{n}/fcz"].tobytes(), 10 + k, nm)
    w.close()
    return names, recs


def test_cpp_cli_id_list_and_file_input_on_the_host(tmp_path, golden):
    """--id-list / --id-mode (src/input_processor.h:287-299) and -f (src/main.cpp:304-325) through `check`, which needs no GPU:
    only the listed entries are visited, by key (mode 0) or by name (mode 1, the default); a missing id is a warning"""
    z, index = golden
    names, recs = _golden_db(tmp_path, z, index)
    by_name = list(recs)[:3]
    (tmp_path / "ids_name.txt").write_text("\n".join(by_name + ["no_such_entry"]) + "\n")
    r = _run("check", "--id-list", str(tmp_path / "ids_name.txt"), str(tmp_path / "gdb"))
    assert r.returncode == 0 and r.stdout.count("is valid") == 3 and "no_such_entry not found" in r.stderr
    for nm in by_name:
        assert f"{nm} is valid" in r.stdout
    keys = [recs[nm][0] for nm in by_name[:2]]
    (tmp_path / "ids_key.txt").write_text("\n".join(str(k) for k in keys) + "\n999\n")
    r = _run("check", "-l", str(tmp_path / "ids_key.txt"), "-m", "0", str(tmp_path / "gdb"))
    assert r.returncode == 0 and r.stdout.count("is valid") == 2 and "999 not found" in r.stderr
    assert _run("check", "-l", str(tmp_path / "ids_key.txt"), "--id-mode", "2", str(tmp_path / "gdb")).returncode != 0
    # -f: a list of inputs; container inputs first, then the single files
    d = tmp_path / "dir"
    d.mkdir()
    for n in names[:2]:
        (d / (n.replace(":", "_") + ".fcz")).write_bytes(z[f"{n}/fcz"].tobytes())
    one = tmp_path / "one.fcz"
    one.write_bytes(z[f"{names[2]}/fcz"].tobytes())
    (tmp_path / "inputs.txt").write_text(f"{one}\n{d}\n{tmp_path / 'gdb'}\n")
    r = _run("check", "-f", str(tmp_path / "inputs.txt"))
    assert r.returncode == 0 and r.stdout.count("is valid") == 2 + len(names) + 1
    assert r.stdout.rstrip().splitlines()[-1].startswith(f"[Info] {one}")


@pytest.mark.gpu
def test_cpp_cli_flags_equal_the_python_cli(tmp_path, golden):
    """--check, --id-list, --no-merge, -f and the reference's default output names: the pipelined C++ host and the Python
    command line produce the same files from the same database (reference src/main.cpp:171-195, 356-369, 629-636)"""
    from foldcomp_amd.__main__ import main as py_main
    z, index = golden
    names, recs = _golden_db(tmp_path, z, index)
    # a database with one entry that fails checkValidity (backbone angle bytes zeroed = "empty backbone angles")
    from foldcomp_amd.database import DatabaseReader, DatabaseWriter
    rd = DatabaseReader(str(tmp_path / "gdb"))
    w = DatabaseWriter(str(tmp_path / "bad"))
    for i in range(len(rd)):
        e = bytearray(rd.data(i))
        if i == 1:
            na, tl, n = e[12], int.from_bytes(e[24:28], "little"), int.from_bytes(e[4:6], "little")
            o_words = 76 + 4 * na + tl + 36 * na + 13
            for k in range(n):
                e[o_words + 8 * k] &= 0xf8
                e[o_words + 8 * k + 1:o_words + 8 * k + 5] = b"\0\0\0\0"
        w.append(bytes(e), int(rd.keys[i]), rd.name(i))
    w.close(); rd.close()

    def tree(p):
        return {f: (p / f).read_bytes() for f in sorted(os.listdir(p))}
    # decompress --check: the invalid entry is skipped by both hosts, the others decompress to the same text
    r = _run("decompress", "--check", "-y", str(tmp_path / "bad"), str(tmp_path / "c_chk"))
    assert r.returncode == 0 and "[Error] All backbone angles are empty: " in r.stderr, r.stderr    # (printValidityError's line, with the record's title)
    py_main(["decompress", "--check", "-y", str(tmp_path / "bad"), str(tmp_path / "p_chk")])
    assert len(os.listdir(tmp_path / "c_chk")) == len(names) - 1 and tree(tmp_path / "c_chk") == tree(tmp_path / "p_chk")
    # --id-list by name into a database; the texts are the reference's
    by_name = list(recs)[2:5]
    (tmp_path / "ids.txt").write_text("\n".join(by_name) + "\n")
    r = _run("decompress", "-d", "-y", "-l", str(tmp_path / "ids.txt"), str(tmp_path / "gdb"), str(tmp_path / "c_sel"))
    assert r.returncode == 0, r.stderr
    sel = DatabaseReader(str(tmp_path / "c_sel"))
    assert [sel.name(i) for i in range(len(sel))] == by_name
    for i, nm in enumerate(by_name):
        assert sel.data(i)[:-1].decode("latin-1") == z[f"{recs[nm][1]}/pdb0"].tobytes().decode("latin-1")
    sel.close()
    # extract --no-merge: one file per entry, named like the reference names them; merged default name = <input>_<suffix>
    r = _run("extract", "--plddt", "-p", "3", "--no-merge", str(tmp_path / "gdb"), str(tmp_path / "c_nm"))
    assert r.returncode == 0, r.stderr
    py_main(["extract", "--plddt", "-p", "3", "--no-merge", str(tmp_path / "gdb"), str(tmp_path / "p_nm")])
    assert len(os.listdir(tmp_path / "c_nm")) == len(names) and tree(tmp_path / "c_nm") == tree(tmp_path / "p_nm")
    r = _run("extract", "--fasta", str(tmp_path / "gdb"))
    assert r.returncode == 0 and os.path.isfile(str(tmp_path / "gdb") + "_fasta"), r.stderr
    assert open(str(tmp_path / "gdb") + "_fasta").read().count(">") == len(names)
    # -f with a database and a single file, default output directory <list>_pdb
    one = tmp_path / "one.fcz"
    one.write_bytes(z[f"{names[0]}/fcz"].tobytes())
    (tmp_path / "inputs.txt").write_text(f"{one}\n{tmp_path / 'gdb'}\n")
    r = _run("decompress", "-y", "-f", str(tmp_path / "inputs.txt"))
    assert r.returncode == 0, r.stderr
    outd = str(tmp_path / "inputs.txt") + "_pdb"
    assert len(os.listdir(outd)) == len(names) + 1 and "one.pdb" in os.listdir(outd)


@pytest.mark.gpu
def test_cpp_compress_device_ingest_equals_host_parse(tmp_path, golden):
    """`compress -d` with the structure ingest on the device (the default for directories) and with --host-parse: the same
    database, byte for byte -- also for what the device hands back to the host parser (a number outside the fixed layout), what
    it never reads (mmCIF, gzip), multi-chain files with gaps, alternative positions, and files the codec refuses"""
    import gzip, json
    from foldcomp_amd.database import DatabaseReader
    z, _ = golden
    d = tmp_path / "in"
    d.mkdir()
    srcs = ["pdb:test_af", "syn:len26", "syn:len129", "pdb:test", "syn:len350", "pdb:multichainA"]
    texts = {n: _pdb_text(z, n) for n in srcs}
    for i in range(300):
        (d / f"f{i:04d}.pdb").write_text(texts[srcs[(i * 5) % len(srcs)]])
    t_af = texts["pdb:test_af"]
    lines = t_af.splitlines()
    k = next(i for i, l in enumerate(lines) if l.startswith("ATOM"))
    (d / "f0007_sci.pdb").write_text("\n".join(lines[:k] + [lines[k][:30] + " 1.0e+01" + lines[k][38:]] + lines[k + 1:]) + "\n")
    (d / "f0011.cif").write_text(_cif_text(z, "pdb:test_af"))
    (d / "f0013.pdb.gz").write_bytes(gzip.compress(texts["pdb:test"].encode()))
    (d / "f0017_multi.pdb").write_text(texts["pdb:multichainA"] + _pdb_text(z, "pdb:multichainB_0") + _pdb_text(z, "pdb:multichainB_1"))
    ala = next(l[22:26] for l in lines if l.startswith("ATOM") and l[17:20] == "ALA")
    (d / "f0019_mse.pdb").write_text("\n".join(l[:17] + "MSE" + l[20:] if l.startswith("ATOM") and l[22:26] == ala else l for l in lines) + "\n")
    (d / "f0023_empty.pdb").write_text("HEADER    nothing\n")
    (d / "f0029_alt.pdb").write_text("\n".join(l for ln in lines for l in ([ln, ln[:30] + "   1.000   2.000   3.000" + ln[54:]] if ln.startswith("ATOM") and ln[12:16].strip() == "CB" else [ln])) + "\n")
    # gzipped PDB text: inflated and parsed on the device; one of them holds a record the device hands back (zlib + the host reader then)
    (d / "f0331.ent.gz").write_bytes(gzip.compress(texts["syn:len129"].encode()))
    (d / "f0337_sci.pdb.gz").write_bytes(gzip.compress(("\n".join(lines[:k] + [lines[k][:30] + " 1.0e+01" + lines[k][38:]] + lines[k + 1:]) + "\n").encode()))
    (d / "f0341_bad.pdb.gz").write_bytes(b"not a gzip stream")
    # two chains, the first with a BLANK chain id: a blank names nothing ("f0043_blank.fcz", not "f0043_blank .fcz")
    blank = "".join((l[:21] + " " + l[22:] if l.startswith(("ATOM", "TER")) and len(l) > 22 else l) + "\n" for l in texts["pdb:multichainA"].splitlines() if not l.startswith("END"))
    (d / "f0043_blank.pdb").write_text(blank + _pdb_text(z, "pdb:multichainB_0"))
    # mmCIF goes through the device too (k_ingest_parse_cif): the reference's own AFDB file, gzipped as it ships; a synthetic one is
    # f0011.cif above. A file with a quoted atom name (the primes of archive files force quotes) is read there as well
    ing = np.load(os.path.join(ROOT, "tests", "golden", "reference_ingest.npz"))
    (d / "f0047_af.cif.gz").write_bytes(ing["file:test.cif.gz"].tobytes())
    (d / "f0051_quoted.cif").write_text(_cif_text(z, "syn:len26").replace(" CA ", ' "CA" ', 1))
    outs = {}
    for tag, extra in (("dev", []), ("host", ["--host-parse"])):
        r = _run("compress", "-d", "-y", "-t", "8", "--gpus", "1", "--json-stats", *extra, str(d), str(tmp_path / f"db_{tag}"))
        assert r.returncode == 0, r.stderr
        st = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        outs[tag] = (st, r.stderr)
    assert outs["dev"][0].get("ingest") == "device" and outs["dev"][0]["host_parsed_files"] == 2      # the scientific-notation files (the quoted atom name is read on the device since round 6)
    # gzip members are inflated on the device (k_inflate); the one that is no gzip stream goes back to zlib, which says so
    assert outs["dev"][0]["device_inflated_files"] == 4 and outs["dev"][0]["host_inflated_after_device_refusal"] == 1
    assert outs["dev"][0]["records"] == outs["host"][0]["records"] == 300 + 1 + 1 + 1 + 3 + 1 + 2 + 2 + 2
    assert all("f0341_bad" in outs[tag][1] for tag in ("dev", "host"))
    for ext in ("", ".index", ".lookup", ".dbtype"):
        assert (tmp_path / f"db_dev{ext}").read_bytes() == (tmp_path / f"db_host{ext}").read_bytes(), ext
    for tag in ("dev", "host"):
        assert "f0019_mse" in outs[tag][1] and "No atoms found" in outs[tag][1]
    rd = DatabaseReader(str(tmp_path / "db_dev"))
    names = [rd.name(i) for i in range(len(rd))]
    assert names == sorted(names) and names.count("f0017_multi") == 3 and "f0011" in names and "f0013.pdb" in names and "f0331.ent" in names and "f0337_sci.pdb" in names
    rd.close()
    # directory output: the same files either way
    for tag, extra in (("dev", []), ("host", ["--host-parse"])):
        r = _run("compress", "-y", "-t", "8", *extra, str(d), str(tmp_path / f"dir_{tag}"))
        assert r.returncode == 0, r.stderr
    a, b = sorted(os.listdir(tmp_path / "dir_dev")), sorted(os.listdir(tmp_path / "dir_host"))
    assert a == b and "f0017_multiB_1.fcz" in a and "f0043_blank.fcz" in a and "f0043_blankB.fcz" in a
    for f in a:
        assert (tmp_path / "dir_dev" / f).read_bytes() == (tmp_path / "dir_host" / f).read_bytes(), f


@pytest.mark.gpu
def test_cpp_compress_database_of_gzipped_entries(tmp_path, golden):
    """`compress -d <database> <out>` where the entries are file images under names that end in .gz (StructureReader::loadFromBuffer:
    the NAME says gzipped, the CONTENT says PDB or mmCIF, src/structure_reader.cpp:73-97): the members go to the device as they are
    (their text's first bytes, inflated on the host in microseconds, decide the format), next to plain entries, an entry with the
    MMseqs NUL behind its member (the device refuses bytes behind the trailer: zlib takes it, like gunzip ignores them), a member that
    is no gzip stream and a gzipped entry that is neither PDB nor mmCIF -- the same database as with zlib on the reader threads"""
    import gzip, json
    from foldcomp_amd.database import DatabaseReader, DatabaseWriter
    z, _ = golden
    pdb = {n: _pdb_text(z, n).encode() for n in ("pdb:test_af", "syn:len129", "pdb:test")}
    cif = _cif_text(z, "syn:len26").encode()
    w = DatabaseWriter(str(tmp_path / "src"))
    k = 0
    for i in range(150):
        n = ("pdb:test_af", "syn:len129", "pdb:test")[i % 3]
        w.append(gzip.compress(pdb[n], 6), k, f"g{i:04d}.pdb.gz"); k += 1
    w.append(gzip.compress(cif, 9), k, "c0001.cif.gz"); k += 1
    w.append(pdb["syn:len129"], k, "p0001.pdb"); k += 1
    w.append(gzip.compress(pdb["pdb:test_af"], 6) + b"\0", k, "nul01.pdb.gz"); k += 1          # an MMseqs-made entry: member + NUL
    w.append(b"\x1f\x8b\x08 not a member at all", k, "bad01.pdb.gz"); k += 1
    w.append(gzip.compress(b"{\"mmjson\": 1}" * 50), k, "json1.pdb.gz"); k += 1
    w.close()
    outs = {}
    for tag, extra in (("dev", []), ("zlib", ["--host-inflate"])):
        r = _run("compress", "-d", "-y", "-t", "8", "--gpus", "1", "--json-stats", *extra, str(tmp_path / "src"), str(tmp_path / f"db_{tag}"))
        assert r.returncode == 0, r.stderr
        outs[tag] = (json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]), r.stderr)
    assert outs["dev"][0]["records"] == outs["zlib"][0]["records"] == 150 + 1 + 1 + 1
    # 150 + 1 members inflated on the device; the NUL-terminated one went there too and came back to zlib (which ignores what follows
    # a member); the broken one and the mmJSON one never left the host (their first bytes do not inflate to PDB / mmCIF text)
    assert outs["dev"][0]["device_inflated_files"] == 151 and outs["dev"][0]["host_inflated_after_device_refusal"] == 1
    assert outs["zlib"][0]["device_inflated_files"] == 0
    for tag in ("dev", "zlib"):
        assert "bad01" in outs[tag][1] and "json1" in outs[tag][1]
    for ext in ("", ".index", ".lookup", ".dbtype"):
        assert (tmp_path / f"db_dev{ext}").read_bytes() == (tmp_path / f"db_zlib{ext}").read_bytes(), ext
    rd = DatabaseReader(str(tmp_path / "db_dev"))
    assert rd.id_of_name("nul01.pdb") >= 0 and rd.id_of_name("c0001.cif") >= 0
    rd.close()
