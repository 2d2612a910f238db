"""`python -m foldcomp_amd compress|decompress -d --gpus N` (foldcomp_amd/sharded_cli.py): the product driver of the sharded
database. Two REAL ranks (gloo: they share GPU 0 of the test box; on a node with a GPU per rank the
backend is nccl = RCCL) each run the pipelined C++ engine (`host/foldcomp-hip --shard R/N`) on their byte-balanced range, exchange
{records, bytes} in one all_gather and splice their partial databases into one (foldcomp_amd/shard.py); data, .index, .lookup and
.dbtype equal the single-process output byte for byte. N = 1 runs the same code in a 1-rank group. The engine's memory does not
grow with the shard (10x the database: the same peak resident set)."""
import json
import gzip
import os
import subprocess
import sys

import pytest

from test_host_cpp import _cif_text, _pdb_text

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cli(*args):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    return subprocess.run([sys.executable, "-m", "foldcomp_amd", *args], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)


def _stats(r):
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def _same_db(a, b):
    for ext in ("", ".index", ".lookup", ".dbtype"):
        assert open(str(a) + ext, "rb").read() == open(str(b) + ext, "rb").read(), ext


def test_two_ranks_compress_and_decompress_equal_the_single_process(tmp_path, golden):
    z, _ = golden
    d = tmp_path / "in"
    d.mkdir()
    srcs = ["pdb:test_af", "syn:len26", "syn:len129", "pdb:test", "syn:len350", "pdb:multichainA"]
    texts = {n: _pdb_text(z, n) for n in srcs}
    for i in range(600):
        (d / f"f{i:04d}.pdb").write_text(texts[srcs[(i * 5) % len(srcs)]])
    (d / "f0011.cif").write_text(_cif_text(z, "pdb:test_af"))
    (d / "f0013.pdb.gz").write_bytes(gzip.compress(texts["pdb:test"].encode()))
    (d / "f0019_mse.pdb").write_text(texts["pdb:test_af"].replace(" ALA ", " MSE ", 2))
    lines = texts["pdb:test_af"].splitlines()
    k = next(i for i, l in enumerate(lines) if l.startswith("ATOM"))
    (d / "f0007_sci.pdb").write_text("\n".join(lines[:k] + [lines[k][:30] + " 1.0e+01" + lines[k][38:]] + lines[k + 1:]) + "\n")
    # compress: single process (the unsharded Python driver), 1-rank group, 2 ranks
    r = _cli("compress", "-d", "-y", str(d), str(tmp_path / "c0"))
    assert r.returncode == 0, r.stderr
    r = _cli("compress", "-d", "-y", "--gpus", "1", str(d), str(tmp_path / "c1"))
    assert r.returncode == 0, r.stderr
    r = _cli("compress", "-d", "-y", "--gpus", "2", "--json-stats", str(d), str(tmp_path / "c2"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert '"world": 2' in r.stdout and '"records": 603' in r.stdout, r.stdout
    _same_db(tmp_path / "c0", tmp_path / "c1")
    _same_db(tmp_path / "c0", tmp_path / "c2")
    # ... and the C++ host writes the same database
    r = subprocess.run([os.path.join(ROOT, "host", "foldcomp-hip"), "compress", "-d", "-y", str(d), str(tmp_path / "c3")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    _same_db(tmp_path / "c0", tmp_path / "c3")
    # decompress the 603-record database, repeated to 2 412 entries through a second database
    from foldcomp_amd.database import DatabaseReader, DatabaseWriter
    rd = DatabaseReader(str(tmp_path / "c0"))
    w = DatabaseWriter(str(tmp_path / "big"))
    key = 0
    for rep in range(4):
        for i in range(len(rd)):
            w.append(rd.data(i), key, f"{rd.name(i)}_{rep}"); key += 1
    w.close(); rd.close()
    r = _cli("decompress", "-d", "-y", str(tmp_path / "big"), str(tmp_path / "d0"))
    assert r.returncode == 0, r.stderr
    # two ranks: counts exchanged BEFORE the writes (the engines' sizes pass), every record written once at its final offset of d2
    # itself -- no partial database exists at any time (a watcher thread lists the directory while the run is going)
    import threading
    seen, stop = set(), threading.Event()

    def watch():
        while not stop.is_set():
            seen.update(f for f in os.listdir(tmp_path) if ".part" in f)
            stop.wait(0.005)
    th = threading.Thread(target=watch); th.start()
    (tmp_path / "d2").write_bytes(b"x" * 10_000_000)          # what an earlier run left: longer than the new output, and gone afterwards
    r = _cli("decompress", "-d", "-y", "--gpus", "2", "--json-stats", str(tmp_path / "big"), str(tmp_path / "d2"))
    stop.set(); th.join()
    assert r.returncode == 0, r.stderr[-2000:]
    st = _stats(r)
    assert st["world"] == 2 and st["data_written_once"] is True and st["engine"].endswith("--place") and st["records"] == 2412
    assert not seen, seen
    _same_db(tmp_path / "d0", tmp_path / "d2")
    assert not [f for f in os.listdir(tmp_path) if ".index." in f or ".lookup." in f]
    # --check and an id list go through the placed run as well: the sizes pass leaves out exactly what the real pass leaves out
    bad = bytearray(open(tmp_path / "big", "rb").read())
    idx = [l.split("\t") for l in open(str(tmp_path / "big") + ".index").read().splitlines()]
    for k in (5, 1300):                                        # two records lose their magic: refused by --check / by the decoder
        bad[int(idx[k][1])] = ord("X")
    for ext in (".index", ".lookup", ".dbtype"):
        (tmp_path / ("bad" + ext)).write_bytes(open(str(tmp_path / "big") + ext, "rb").read())
    (tmp_path / "bad").write_bytes(bytes(bad))
    for flags in ((), ("--check",)):
        r = _cli("decompress", "-d", "-y", "--gpus", "1", *flags, str(tmp_path / "bad"), str(tmp_path / "b1"))
        assert r.returncode == 0, r.stderr[-2000:]
        r = _cli("decompress", "-d", "-y", "--gpus", "2", "--json-stats", *flags, str(tmp_path / "bad"), str(tmp_path / "b2"))
        assert r.returncode == 0, r.stderr[-2000:]
        assert _stats(r)["records"] == 2410
        _same_db(tmp_path / "b1", tmp_path / "b2")

    # --gpus without -d is refused
    r = _cli("compress", "-y", "--gpus", "2", str(d), str(tmp_path / "dir"))
    assert r.returncode != 0 and "add -d" in r.stderr


def test_shard_memory_is_independent_of_the_shard_size_and_databases_round_trip(tmp_path, golden):
    """configs[3] / [4] in the small: database in, database out, both directions, two ranks. A database 10x the size leaves the
    engines' peak resident set where it was (nothing per record is held: jobs stream through, the index is read line by line and
    written job by job); the PDB-text database goes back through `compress -d --gpus 2` (entries = file images under their lookup
    names, MMseqs NUL stripped, structure ingest on the device) and equals the single process's and the C++ host's database."""
    from foldcomp_amd.database import DatabaseReader, DatabaseWriter
    z, _ = golden
    recs = [z[f"{n}/fcz"].tobytes() for n in ("syn:len26", "syn:len64", "syn:len129")]
    host = os.path.join(ROOT, "host", "foldcomp-hip")

    def make(path, n):
        w = DatabaseWriter(str(path))
        for k in range(n):
            w.append(recs[k % 3], k, f"e{k:06d}")
        w.close()
    n_small = 12_288                                  # >= 2 workers x 2 full jobs of 2 048 entries per rank: the job buffers reach their size
    make(tmp_path / "small", n_small); make(tmp_path / "big", 10 * n_small)
    rss = {}
    for name in ("small", "big"):
        r = _cli("decompress", "-d", "-y", "--gpus", "2", "--json-stats", str(tmp_path / name), str(tmp_path / f"{name}_pdb"))
        assert r.returncode == 0, r.stderr[-2000:]
        st = _stats(r)
        assert st["world"] == 2 and st["records"] == (n_small if name == "small" else 10 * n_small) and len(st["records_per_rank"]) == 2
        assert abs(st["records_per_rank"][0] - st["records"] / 2) <= 2
        rss[name] = max(st["engine_max_rss_kb_per_rank"])
    assert rss["big"] < 1.15 * rss["small"] + 32_768, rss
    # the sharded output == the C++ host's single process, every entry the reference text + NUL
    r = subprocess.run([host, "decompress", "-d", "-y", str(tmp_path / "small"), str(tmp_path / "small_one")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    _same_db(tmp_path / "small_pdb", tmp_path / "small_one")
    rd = DatabaseReader(str(tmp_path / "small_pdb"))
    assert len(rd) == n_small and rd.name(4) == "e000004" and rd.data(4) == z["syn:len64/pdb0"].tobytes() + b"\0"
    rd.close()
    # ... and back: PDB-text database -> FCZ database, 2 ranks == 1 rank == the C++ host == the unsharded Python driver
    r = _cli("compress", "-d", "-y", "--gpus", "2", "--json-stats", str(tmp_path / "small_pdb"), str(tmp_path / "back2"))
    assert r.returncode == 0, r.stderr[-2000:]
    st = _stats(r)
    assert st["records"] == n_small and st["world"] == 2
    r = _cli("compress", "-d", "-y", "--gpus", "1", str(tmp_path / "small_pdb"), str(tmp_path / "back1"))
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([host, "compress", "-d", "-y", "--json-stats", str(tmp_path / "small_pdb"), str(tmp_path / "backh")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout.splitlines()[-1])["host_parsed_files"] == 0     # every entry went through the device ingest
    _same_db(tmp_path / "back2", tmp_path / "back1")
    _same_db(tmp_path / "back2", tmp_path / "backh")
    small = tmp_path / "tiny"
    w = DatabaseWriter(str(small)); src = DatabaseReader(str(tmp_path / "small_pdb"))
    for k in range(300):
        w.append(src.data(k), k, src.name(k))
    w.close(); src.close()
    r = _cli("compress", "-d", "-y", str(small), str(tmp_path / "tiny_py"))
    assert r.returncode == 0, r.stderr
    r = _cli("compress", "-d", "-y", "--gpus", "2", str(small), str(tmp_path / "tiny_2"))
    assert r.returncode == 0, r.stderr
    _same_db(tmp_path / "tiny_py", tmp_path / "tiny_2")


def test_more_ranks_than_work_and_mixed_input_lists(tmp_path, golden):
    """edge shapes of the range cut: three ranks over two files (one rank's engine finds an empty range and writes an empty partial
    database), and a `-f` list that names a database and a directory (the cut crosses the boundary between the two sources)"""
    from foldcomp_amd.database import DatabaseReader, DatabaseWriter
    z, _ = golden
    d = tmp_path / "two"
    d.mkdir()
    (d / "a.pdb").write_text(_pdb_text(z, "syn:len129"))
    (d / "b.pdb").write_text(_pdb_text(z, "pdb:test_af"))
    r = _cli("compress", "-d", "-y", "--gpus", "1", str(d), str(tmp_path / "t1"))
    assert r.returncode == 0, r.stderr[-2000:]
    r = _cli("compress", "-d", "-y", "--gpus", "3", "--json-stats", str(d), str(tmp_path / "t3"))
    assert r.returncode == 0, r.stderr[-2000:]
    st = _stats(r)
    assert st["world"] == 3 and st["records"] == 2 and sorted(st["records_per_rank"]) in ([0, 0, 2], [0, 1, 1])
    _same_db(tmp_path / "t1", tmp_path / "t3")
    assert not [f for f in os.listdir(tmp_path) if ".part" in f or ".index." in f or ".lookup." in f]
    # ... and back: three ranks over the two records (placed run: one rank's sizes pass finds nothing, it still takes part in the
    # exchange and writes nothing), the output of one rank
    r = _cli("decompress", "-d", "-y", "--gpus", "1", str(tmp_path / "t1"), str(tmp_path / "u1"))
    assert r.returncode == 0, r.stderr[-2000:]
    r = _cli("decompress", "-d", "-y", "--gpus", "3", "--json-stats", str(tmp_path / "t1"), str(tmp_path / "u3"))
    assert r.returncode == 0, r.stderr[-2000:]
    st = _stats(r)
    assert st["world"] == 3 and st["records"] == 2 and st["data_written_once"] is True and sorted(st["records_per_rank"]) in ([0, 0, 2], [0, 1, 1])
    _same_db(tmp_path / "u1", tmp_path / "u3")
    assert not [f for f in os.listdir(tmp_path) if ".part" in f or ".index." in f or ".lookup." in f]
    # a database of PDB texts + a directory of files, listed in one -f file
    w = DatabaseWriter(str(tmp_path / "texts"))
    for k, n in enumerate(["syn:len26", "syn:len64", "pdb:test", "syn:len350", "syn:len129"] * 8):
        w.append(_pdb_text(z, n).encode() + b"\0", k, f"entry{k:03d}")
    w.close()
    d2 = tmp_path / "more"
    d2.mkdir()
    for i in range(23):
        (d2 / f"m{i:02d}.pdb").write_text(_pdb_text(z, ["syn:len26", "pdb:test_af", "syn:len350"][i % 3]))
    (d2 / "m99.cif").write_text(_cif_text(z, "syn:len64"))
    lst = tmp_path / "inputs.txt"
    lst.write_text(f"{tmp_path / 'texts'}\n{d2}\n")
    for g in (1, 2, 3):
        r = _cli("compress", "-d", "-y", "-f", "--gpus", str(g), "--json-stats", str(lst), str(tmp_path / f"mix{g}"))
        assert r.returncode == 0, r.stderr[-2000:]
        assert _stats(r)["records"] == 40 + 24
    _same_db(tmp_path / "mix1", tmp_path / "mix2")
    _same_db(tmp_path / "mix1", tmp_path / "mix3")
    rd = DatabaseReader(str(tmp_path / "mix1"))
    assert rd.name(0) == "entry000" and rd.name(40) == "m00" and rd.name(63) == "m99"
    rd.close()
