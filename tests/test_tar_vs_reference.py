"""tar archives in and out -- the container AFDB ships its bulk downloads in (a plain tar of .pdb.gz / .cif.gz members) -- through
both hosts (host/foldcomp-hip = host/foldcomp, python -m foldcomp_amd), against the reference's OWN command line
(oracle/_ref/foldcomp_ref: src/main.cpp + lib/microtar compiled from where they lie by oracle/build_ref.sh; test infrastructure only).

CPU: which members a tar yields and under which names (ustar / GNU / pax archives, long names, directories, links, nested paths,
     99- and 100-character names, plain and gzipped, truncated and damaged archives) == what the reference's run over the same
     archive writes; the archive writer's bytes == the reference's archive (records' four uninitialised header bytes masked).
GPU: compress / decompress / extract / check with a tar on either side == the reference's outputs member by member; the gzip members
     of a plain tar are inflated on the device."""
import gzip
import io
import json
import os
import subprocess
import tarfile

import numpy as np
import pytest

from foldcomp_amd.__main__ import file_parts, iter_tar, tar_header

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "host", "foldcomp-hip")
REF = os.path.join(ROOT, "oracle", "_ref", "foldcomp_ref")

needs_ref = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/foldcomp_ref not built (oracle/build_ref.sh needs the reference tree)")


@pytest.fixture(scope="module")
def files():
    z = np.load(os.path.join(ROOT, "tests", "golden", "reference_ingest.npz"))
    return {k[5:]: z[k].tobytes() for k in z.keys() if k.startswith("file:") and not k.startswith("file:example_db")}


def _host(*args, timeout=600):
    assert os.path.exists(BIN), "host/foldcomp-hip missing: run __graft_entry__.build()"
    return subprocess.run([BIN, *args], capture_output=True, text=True, timeout=timeout)


def _ref(*args, cwd=None):
    return subprocess.run([REF, *args], capture_output=True, text=True, timeout=600, cwd=cwd)


def _python(*args):
    import sys
    return subprocess.run([sys.executable, "-m", "foldcomp_amd", *args], capture_output=True, text=True, timeout=600, cwd=ROOT)


def _members(raw: bytes):
    """(name, header, data) of every member of an archive the reference's writer made, and what follows the last one"""
    out, pos = [], 0
    while pos + 512 <= len(raw) and raw[pos + 148] != 0:
        size = int(raw[pos + 124:pos + 136].split(b"\0")[0], 8)
        out.append((raw[pos:pos + 100].split(b"\0")[0].decode(), raw[pos:pos + 512], raw[pos + 512:pos + 512 + size]))
        pos += 512 + (size + 511) // 512 * 512
    return out, raw[pos:]


def _mask(fcz: bytes) -> bytes:
    """an FCZ record with the four header bytes the reference leaves uninitialised (file offsets 14, 15, 22, 23) zeroed"""
    b = bytearray(fcz)
    for i in (14, 15, 22, 23):
        if i < len(b):
            b[i] = 0
    return bytes(b)


def _add(tf, name, data, **kw):
    ti = tarfile.TarInfo(name); ti.size = len(data)
    for k, v in kw.items():
        setattr(ti, k, v)
    tf.addfile(ti, io.BytesIO(data))


def _archive(path, files, fmt=tarfile.GNU_FORMAT, gz=False, long_names=True):
    """an archive of the shapes tar writers produce: files at the top and in directories, a directory member, a symbolic link, a
    name of exactly 99 and of exactly 100 characters, one beyond 100 (GNU: an 'L' record; pax: an 'x' record; ustar: the prefix
    field), a member that is no structure"""
    pdb, af = files["test.pdb"], files["test_af.pdb"]
    n99 = "a" * 95 + ".pdb"
    n100 = "b" * 96 + ".pdb"
    nlong = "deep/" * 18 + "c" * 40 + ".pdb"
    with tarfile.open(path, "w:gz" if gz else "w", format=fmt) as tf:
        _add(tf, "test.pdb", pdb)
        d = tarfile.TarInfo("sub"); d.type = tarfile.DIRTYPE; tf.addfile(d)
        _add(tf, "sub/test_af.pdb.gz", gzip.compress(af, mtime=0))
        _add(tf, "sub/inner/test.cif.gz", files["test.cif.gz"])
        ln = tarfile.TarInfo("link.pdb"); ln.type = tarfile.SYMTYPE; ln.linkname = "test.pdb"; tf.addfile(ln)
        if long_names:
            _add(tf, n99, af)
            _add(tf, n100, af)
            _add(tf, nlong, af)
        _add(tf, "README", b"not a structure\n")
        _add(tf, "multichain.pdb", files["multichain.pdb"])
    return path


def _plan(path):
    r = _host("plan-dump", str(path))
    assert r.returncode == 0, r.stderr
    rows = [l.split("\t") for l in r.stdout.splitlines() if "\t" in l]
    return [(n, int(ln)) for _, n, _, ln in rows], r.stderr


@pytest.mark.parametrize("gz", [False, True])
@pytest.mark.parametrize("fmt", [tarfile.GNU_FORMAT, tarfile.PAX_FORMAT, tarfile.USTAR_FORMAT])
def test_members_and_names(tmp_path, files, fmt, gz):
    """both hosts list the same members under the same names; those names are what the reference's run turns into output names"""
    p = _archive(str(tmp_path / ("in.tar.gz" if gz else "in.tar")), files, fmt, gz)
    listed, _ = _plan(p)
    py = [(n, len(d)) for n, d in iter_tar(p)]
    assert listed == py
    names = [n for n, _ in listed]
    assert "test.pdb" in names and "sub/test_af.pdb.gz" in names and "sub/inner/test.cif.gz" in names and "README" in names
    assert "sub" not in names and "link.pdb" not in names                       # a directory and a link are no files
    assert "a" * 95 + ".pdb" in names                                          # 99 characters: kept
    # 100 characters fill the name field without a terminator: microtar reads 99 of them (a GNU archive writes such a name as it is)
    if fmt == tarfile.GNU_FORMAT:
        assert "b" * 96 + ".pd" in names
        assert "deep/" * 18 + "c" * 40 + ".pdb" in names                      # the 'L' record's name
    if os.path.exists(REF):
        # the reference over the same archive: every structure member becomes <stem of its base name>.fcz in the output directory
        out = tmp_path / "ref_out"
        r = _ref("compress", p, str(out))
        assert r.returncode == 0
        want = set()
        for n, _ in listed:
            base = os.path.basename(n)
            if base == "README":
                continue
            stem, ext = file_parts(base)
            suffix = ".fcz" if (ext in ("pdb", "cif") or (ext == "gz" and file_parts(stem)[1] in ("pdb", "cif"))) else "." + ext
            if base == "multichain.pdb":
                want |= {"multichainA.fcz", "multichainB_0.fcz", "multichainB_1.fcz"}
            else:
                want.add(stem + suffix)
        assert set(os.listdir(out)) == want


def test_damaged_archives(tmp_path, files):
    """a truncated archive and a header that fails its checksum: the members before the damage, an [Error] line, no crash"""
    p = _archive(str(tmp_path / "in.tar"), files)
    raw = open(p, "rb").read()
    whole, _ = _plan(p)
    cut = tmp_path / "cut.tar"
    cut.write_bytes(raw[:512 + len(files["test.pdb"]) // 512 * 512 + 512 + 512 + 700])      # into the third member's bytes
    got, err = _plan(cut)
    assert got == whole[:len(got)] and 1 <= len(got) < len(whole) and "[Error]" in err
    assert [(n, len(d)) for n, d in iter_tar(str(cut))] == got[:len(list(iter_tar(str(cut))))]
    bad = bytearray(raw)
    second = 512 + (len(files["test.pdb"]) + 511) // 512 * 512                # the directory member's header
    bad[second + 10] ^= 0x55
    (tmp_path / "bad.tar").write_bytes(bytes(bad))
    got, err = _plan(tmp_path / "bad.tar")
    assert got == whole[:1] and "checksum" in err
    assert [(n, len(d)) for n, d in iter_tar(str(tmp_path / "bad.tar"))] == got
    (tmp_path / "empty.tar").write_bytes(b"\0" * 10240)
    assert _plan(tmp_path / "empty.tar")[0] == []


def test_sharded_ranges_of_a_tar(tmp_path, files):
    """--shard R/N cuts a plain tar's members into contiguous byte-balanced ranges; a gzipped tar is refused (one DEFLATE stream)"""
    p = _archive(str(tmp_path / "in.tar"), files)
    whole, _ = _plan(p)
    parts = []
    for r in range(3):
        out = _host("plan-dump", "--shard", f"{r}/3", p)
        assert out.returncode == 0, out.stderr
        parts += [(l.split("\t")[1], int(l.split("\t")[3])) for l in out.stdout.splitlines() if "\t" in l]
    assert parts == whole
    gzp = _archive(str(tmp_path / "in.tar.gz"), files, gz=True)
    out = _host("plan-dump", "--shard", "0/2", gzp)
    assert out.returncode != 0 and "gzipped tar" in out.stderr


def test_header_bytes(files):
    h = tar_header("test.fcz", 4485)
    assert len(h) == 512 and h[:8] == b"test.fcz" and h[100:104] == b"644\0" and h[108:110] == b"0\0" and h[124:130] == b"10605\0"
    assert h[136:138] == b"0\0" and h[156:157] == b"0" and h[257:263] == b"\0" * 6      # no ustar magic
    assert int(h[148:154], 8) == 256 + sum(h[:148]) + sum(h[156:]) and h[154:156] == b"\0 "


@needs_ref
def test_writer_equals_reference_archive(tmp_path, files):
    """tar-pack (the writer behind every -z output) over the reference's output FILES == the reference's output ARCHIVE of the same
    run, byte for byte apart from the records' uninitialised bytes"""
    src = tmp_path / "in"
    src.mkdir()
    (src / "a_test.pdb").write_bytes(files["test.pdb"]); (src / "b_test_af.pdb").write_bytes(files["test_af.pdb"]); (src / "c_test.cif.gz").write_bytes(files["test.cif.gz"])
    with tarfile.open(tmp_path / "in.tar", "w", format=tarfile.GNU_FORMAT) as tf:
        for n in sorted(os.listdir(src)):
            tf.add(src / n, arcname=n)
    assert _ref("compress", str(tmp_path / "in.tar"), str(tmp_path / "ref.tar")).returncode == 0
    assert _ref("compress", str(tmp_path / "in.tar"), str(tmp_path / "ref_dir")).returncode == 0
    r = _host("tar-pack", str(tmp_path / "ref_dir"), str(tmp_path / "mine.tar"))
    assert r.returncode == 0, r.stderr
    a, ta = _members((tmp_path / "ref.tar").read_bytes())
    b, tb = _members((tmp_path / "mine.tar").read_bytes())
    assert [m[0] for m in a] == [m[0] for m in b] == ["a_test.fcz", "b_test_af.fcz", "c_test.cif.fcz"]
    assert all(x[1] == y[1] and _mask(x[2]) == _mask(y[2]) for x, y in zip(a, b))
    assert ta == tb == b"\0" * 1024
    assert (tmp_path / "ref.tar").stat().st_size == (tmp_path / "mine.tar").stat().st_size
    for (n, h, d) in a:
        assert h == tar_header(n, len(d))


# ---- GPU: the runs themselves ----------------------------------------------------------------------------------------------------
def _fcz_dir(path):
    return {n: _mask(open(os.path.join(path, n), "rb").read()) for n in sorted(os.listdir(path))}


def _db(path):
    from foldcomp_amd.database import DatabaseReader
    r = DatabaseReader(str(path))
    out = [(r.name(i), r.data(i)) for i in range(len(r))]
    r.close()
    return out


@pytest.mark.gpu
@needs_ref
@pytest.mark.parametrize("gz", [False, True])
def test_compress_tar_equals_reference(tmp_path, files, gz):
    """a tar of .pdb / .pdb.gz / .cif.gz members (plain: its gzip members go to the device as they lie in the archive; gzipped: walked
    as it inflates) -> directory, database and archive, through both hosts == the reference's outputs of the same archive"""
    p = _archive(str(tmp_path / ("in.tar.gz" if gz else "in.tar")), files, gz=gz)
    assert _ref("compress", p, str(tmp_path / "ref_dir")).returncode == 0
    assert _ref("compress", p, str(tmp_path / "ref.tar")).returncode == 0
    assert _ref("compress", "-d", p, str(tmp_path / "ref_db")).returncode == 0
    want = _fcz_dir(tmp_path / "ref_dir")
    assert len(want) == 9                                                       # 6 single-chain files + 3 fragments of multichain.pdb
    want_tar = [(n, h, _mask(d)) for n, h, d in _members((tmp_path / "ref.tar").read_bytes())[0]]
    want_db = [(n, _mask(d)) for n, d in _db(tmp_path / "ref_db")]
    for tag, run in (("cpp", _host), ("py", _python)):
        r = run("compress", p, str(tmp_path / f"{tag}_dir"))
        assert r.returncode == 0, r.stderr
        assert "README" in r.stderr                                             # the member that is no structure is reported, not fatal
        assert _fcz_dir(tmp_path / f"{tag}_dir") == want, tag
        r = run("compress", p, str(tmp_path / f"{tag}.tar"))
        assert r.returncode == 0, r.stderr
        got, tail = _members((tmp_path / f"{tag}.tar").read_bytes())
        assert [(n, h, _mask(d)) for n, h, d in got] == want_tar and tail == b"\0" * 1024, tag
        r = run("compress", "-d", p, str(tmp_path / f"{tag}_db"))
        assert r.returncode == 0, r.stderr
        assert [(n, _mask(d)) for n, d in _db(tmp_path / f"{tag}_db")] == want_db, tag
    # -z without an output name: <input>.fcz.tar (src/main.cpp:359-360)
    r = _host("compress", "-z", p)
    assert r.returncode == 0 and os.path.exists(p + ".fcz.tar")
    assert [(n, _mask(d)) for n, _, d in _members(open(p + ".fcz.tar", "rb").read())[0]] == [(n, d) for n, _, d in want_tar]
    # the gzip members of a PLAIN tar never see zlib: inflated on the device
    r = _host("compress", "-d", "-y", "--json-stats", p, str(tmp_path / "stats_db"))
    st = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert st["records"] == 9
    if not gz:
        assert st["device_inflated_files"] == 2 and st["host_inflated_after_device_refusal"] == 0


@pytest.mark.gpu
@needs_ref
def test_decompress_extract_check_tar_equal_reference(tmp_path, files):
    """FCZ members of a tar (the reference's own archive) -> texts / pLDDT / sequences into a directory, an archive and a database,
    through both hosts == the reference's; a single FCZ file into an archive; check over an archive"""
    # (no member names near the 100 characters of a header's name field: the reference's writer copies a name into that field
    #  unchecked and is ended by the C library's overflow guard when <stem>.pdb reaches 100 -- this writer cuts at 99)
    p = _archive(str(tmp_path / "in.tar"), files, long_names=False)
    assert _ref("compress", p, str(tmp_path / "fcz.tar")).returncode == 0
    t = str(tmp_path / "fcz.tar")
    with tarfile.open(t) as tf:                                                 # (readable by any tar reader)
        assert len(tf.getnames()) == 6
    subprocess.run(["gzip", "-k", t], check=True)
    assert _ref("decompress", t, str(tmp_path / "ref_dir")).returncode == 0
    assert _ref("decompress", t, str(tmp_path / "ref.tar")).returncode == 0
    assert _ref("decompress", "-d", t, str(tmp_path / "ref_db")).returncode == 0
    assert _ref("decompress", "-a", t, str(tmp_path / "ref_alt.tar")).returncode == 0
    want_dir = {n: open(tmp_path / "ref_dir" / n, "rb").read() for n in sorted(os.listdir(tmp_path / "ref_dir"))}
    want_tar = (tmp_path / "ref.tar").read_bytes()
    want_db = _db(tmp_path / "ref_db")
    assert len(want_dir) == 6 and all(n.endswith(".pdb") for n in want_dir)
    for tag, run in (("cpp", _host), ("py", _python)):
        for src in (t, t + ".gz"):
            out = tmp_path / f"{tag}_dir{'_gz' if src.endswith('.gz') else ''}"
            r = run("decompress", src, str(out))
            assert r.returncode == 0, r.stderr
            assert {n: open(out / n, "rb").read() for n in sorted(os.listdir(out))} == want_dir, (tag, src)
        r = run("decompress", t, str(tmp_path / f"{tag}.tar"))
        assert r.returncode == 0, r.stderr
        assert (tmp_path / f"{tag}.tar").read_bytes() == want_tar, tag          # the whole archive, byte for byte
        r = run("decompress", "-a", t, str(tmp_path / f"{tag}_alt.tar"))
        assert (tmp_path / f"{tag}_alt.tar").read_bytes() == (tmp_path / "ref_alt.tar").read_bytes(), tag
        r = run("decompress", "-d", t, str(tmp_path / f"{tag}_db"))
        assert r.returncode == 0, r.stderr
        assert _db(tmp_path / f"{tag}_db") == want_db, tag
    # extract: merged file, archive, database; pLDDT digits and sequences
    for flags in (["--plddt"], ["--plddt", "-p", "3"], ["--fasta"]):
        k = "_".join(f.strip("-") for f in flags)
        assert _ref("extract", *flags, t, str(tmp_path / f"ref_{k}.txt")).returncode == 0
        assert _ref("extract", *flags, t, str(tmp_path / f"ref_{k}.tar")).returncode == 0
        assert _ref("extract", *flags, "-d", t, str(tmp_path / f"ref_{k}_db")).returncode == 0
        for tag, run in (("cpp", _host), ("py", _python)):
            assert run("extract", *flags, t, str(tmp_path / f"{tag}_{k}.txt")).returncode == 0
            assert (tmp_path / f"{tag}_{k}.txt").read_bytes() == (tmp_path / f"ref_{k}.txt").read_bytes(), (tag, k)
            assert run("extract", *flags, t, str(tmp_path / f"{tag}_{k}.tar")).returncode == 0
            assert (tmp_path / f"{tag}_{k}.tar").read_bytes() == (tmp_path / f"ref_{k}.tar").read_bytes(), (tag, k)
            assert run("extract", *flags, "-d", t, str(tmp_path / f"{tag}_{k}_db")).returncode == 0
            assert _db(tmp_path / f"{tag}_{k}_db") == _db(tmp_path / f"ref_{k}_db"), (tag, k)
    # one FCZ file into an archive: the member is <stem>.pdb (src/main.cpp:646-647)
    one = tmp_path / "ref_dir_fcz"
    assert _ref("compress", p, str(one)).returncode == 0
    assert _ref("decompress", str(one / "test_af.pdb.fcz"), str(tmp_path / "ref_one.tar")).returncode == 0
    for tag, run in (("cpp", _host), ("py", _python)):
        assert run("decompress", str(one / "test_af.pdb.fcz"), str(tmp_path / f"{tag}_one.tar")).returncode == 0
        assert (tmp_path / f"{tag}_one.tar").read_bytes() == (tmp_path / "ref_one.tar").read_bytes(), tag
    # check: every member is valid (the reference stays silent about valid entries, src/foldcomp.cpp:1534-1560: no [Error] line)
    rr = _ref("check", t)
    assert rr.returncode == 0 and "[Error]" not in rr.stderr
    for run in (_host, _python):
        r = run("check", t)
        assert r.returncode == 0 and r.stdout.count("is valid") == 6 and "[Error]" not in r.stderr


@pytest.mark.gpu
def test_large_tar_of_gz_members_on_the_device(tmp_path, golden):
    """an AFDB-shaped archive (3 000 .pdb.gz members in one plain tar) -> database: every member inflated on the device, the database
    equal to the one made from the same members as files of a directory"""
    from test_host_cpp import _pdb_text
    z, _ = golden
    srcs = ["pdb:test_af", "syn:len26", "syn:len129", "pdb:test", "syn:len350", "pdb:multichainA"]
    zs = [gzip.compress(_pdb_text(z, n).encode(), 6, mtime=0) for n in srcs]
    d = tmp_path / "dir"
    d.mkdir()
    with tarfile.open(tmp_path / "afdb.tar", "w", format=tarfile.GNU_FORMAT) as tf:
        for i in range(3000):
            name = f"AF-Q{i:07d}-F1-model_v4.pdb.gz"
            (d / name).write_bytes(zs[(i * 5) % len(zs)])
            _add(tf, name, zs[(i * 5) % len(zs)])
    r = _host("compress", "-d", "--json-stats", str(tmp_path / "afdb.tar"), str(tmp_path / "db_tar"))
    assert r.returncode == 0, r.stderr
    st = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert st["records"] == 3000 and st["device_inflated_files"] == 3000 and st["host_parsed_files"] == 0
    r = _host("compress", "-d", str(d), str(tmp_path / "db_dir"))
    assert r.returncode == 0, r.stderr
    for ext in ("", ".index", ".lookup", ".dbtype"):
        assert open(str(tmp_path / "db_tar") + ext, "rb").read() == open(str(tmp_path / "db_dir") + ext, "rb").read(), ext
    # the same archive gzipped as a whole: walked as zlib inflates it, the members still inflated on the device
    subprocess.run(["gzip", "-1", "-k", str(tmp_path / "afdb.tar")], check=True)
    r = _host("compress", "-d", "--json-stats", str(tmp_path / "afdb.tar.gz"), str(tmp_path / "db_tgz"))
    assert r.returncode == 0, r.stderr
    st = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert st["records"] == 3000 and st["device_inflated_files"] == 3000
    for ext in ("", ".index", ".lookup", ".dbtype"):
        assert open(str(tmp_path / "db_tgz") + ext, "rb").read() == open(str(tmp_path / "db_dir") + ext, "rb").read(), ext
