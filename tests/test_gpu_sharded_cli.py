"""`python -m foldcomp_amd compress|decompress -d --gpus N` (foldcomp_amd/sharded_cli.py): the product driver of the sharded
database. Two REAL ranks (gloo: they share GPU 0 of the test box; on a node with a GPU per rank the
backend is nccl = RCCL) run the real codec on their byte-balanced ranges and write one database with shard.write_sharded_db;
data, .index, .lookup and .dbtype equal the single-process output byte for byte. N = 1 runs the same code in a 1-rank group."""
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
    r = _cli("decompress", "-d", "-y", "--gpus", "2", str(tmp_path / "big"), str(tmp_path / "d2"))
    assert r.returncode == 0, r.stderr[-2000:]
    _same_db(tmp_path / "d0", tmp_path / "d2")

    # --gpus without -d is refused
    r = _cli("compress", "-y", "--gpus", "2", str(d), str(tmp_path / "dir"))
    assert r.returncode != 0 and "add -d" in r.stderr
