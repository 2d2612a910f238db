"""CPU: the PYTHON side of the sharded drivers' protocol (foldcomp_amd/sharded_cli.py) with real ranks (gloo, world 2 and 3) around a
stand-in engine that needs no GPU (tests/_standin_engine.py speaks the C++ engine's side with made-up records):

  decompress  the engine's sizes line -> ONE all_gather before anything is written -> `key0 off0 total` on its stdin -> every rank
              writes once into the final file -> rank 0 joins the index lines: the database a single writer makes, no partial
              database at any time;
  compress    partial databases -> all_gather -> splice (`foldcomp-hip db-splice`, CPU work) -> the same database;
  failures    an engine that fails in its sizes pass, after the placement, or while writing, on any rank: every rank returns
              non-zero and NOTHING is left behind (a database without one rank's records looks complete).
The engines' own side of the protocol runs on the GPU box (tests/test_gpu_sharded_cli.py)."""
import argparse
import os
import socket
import subprocess
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "host", "foldcomp-hip")
STANDIN = os.path.join(ROOT, "tests", "_standin_engine.py")

pytestmark = pytest.mark.skipif(not os.path.exists(HOST), reason="host/foldcomp-hip not built (db-splice is the compress direction's file step)")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _args(mode, inp):
    return argparse.Namespace(mode=mode, input=inp, threads=2, brk=25, recursive=False, skip_discontinuous=False, alt=False, check=False,
                              id_list=None, id_mode=1, file_input=False, json_stats=False)


def _worker(rank, world, port, mode, inp, out, fail, rcs):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FCZ_SHARD_BACKEND="gloo")
    if fail:
        os.environ["FAIL_AT"] = fail
    import torch
    torch.cuda.is_available = lambda: True            # the driver refuses to run without a device; the stand-in engine needs none
    torch.cuda.device_count = lambda: 1
    from foldcomp_amd import sharded_cli
    sharded_cli.ENGINE = STANDIN
    rcs[rank] = sharded_cli.run(_args(mode, inp), [inp], out)


def _run(world, mode, inp, out, fail=""):
    mgr = mp.Manager()
    rcs = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), mode, inp, out, fail, rcs), nprocs=world, join=True)
    return [rcs.get(r) for r in range(world)]


def _items(n=41):
    return [(f"rec{i:03d}", 17 + (i * 37) % 211) for i in range(n)]


def _expected(items):
    data = b"".join((nm.encode() * (n // len(nm) + 1))[:n] for nm, n in items)
    idx, lk, o = "", "", 0
    for k, (nm, n) in enumerate(items):
        idx += f"{k}\t{o}\t{n}\n"; lk += f"{k}\t{nm}\t0\n"; o += n
    return data, idx, lk


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("mode", ["decompress", "compress"])
def test_ranks_build_the_single_writers_database(tmp_path, world, mode):
    items = _items()
    inp = tmp_path / "items.txt"
    inp.write_text("".join(f"{nm} {n}\n" for nm, n in items))
    out = tmp_path / "out"
    out.write_bytes(b"stale" * 100_000)                       # what an earlier run left: longer than the new output
    assert _run(world, mode, str(inp), str(out)) == [0] * world
    data, idx, lk = _expected(items)
    assert out.read_bytes() == data
    assert (tmp_path / "out.index").read_text() == idx and (tmp_path / "out.lookup").read_text() == lk
    assert (tmp_path / "out.dbtype").read_bytes() == (12).to_bytes(4, "little")
    assert sorted(os.listdir(tmp_path)) == ["items.txt", "out", "out.dbtype", "out.index", "out.lookup"]


@pytest.mark.parametrize("fail", ["sizes:0", "sizes:1", "place:1", "write:0", "write:1"])
def test_a_failure_at_any_stage_leaves_nothing_behind(tmp_path, fail):
    items = _items(23)
    inp = tmp_path / "items.txt"
    inp.write_text("".join(f"{nm} {n}\n" for nm, n in items))
    rcs = _run(2, "decompress", str(inp), str(tmp_path / "out"), fail=fail)
    assert all(rc not in (0, None) for rc in rcs), rcs
    assert sorted(os.listdir(tmp_path)) == ["items.txt"], os.listdir(tmp_path)


@pytest.mark.parametrize("fail", ["sizes:5", "write:0", "place:7"])
def test_world_8_a_failure_on_any_rank_leaves_nothing_behind(tmp_path, fail):
    """the shape of the driver's first real run (configs[3] / [4]: 8 ranks): rank 5 fails in its sizes pass, rank 0 while writing, the
    last rank after the placement -- all eight return non-zero, no file of any rank stays"""
    items = _items(53)
    inp = tmp_path / "items.txt"
    inp.write_text("".join(f"{nm} {n}\n" for nm, n in items))
    rcs = _run(8, "decompress", str(inp), str(tmp_path / "out"), fail=fail)
    assert len(rcs) == 8 and all(rc not in (0, None) for rc in rcs), rcs
    assert sorted(os.listdir(tmp_path)) == ["items.txt"], os.listdir(tmp_path)


@pytest.mark.parametrize("mode", ["decompress", "compress"])
def test_world_8_with_empty_ranges_on_three_ranks(tmp_path, mode):
    """five records over eight ranks: three ranks own nothing (the integer cut rule gives them an empty range) and still take part
    in both collectives; the database is the single writer's"""
    items = _items(5)
    inp = tmp_path / "items.txt"
    inp.write_text("".join(f"{nm} {n}\n" for nm, n in items))
    sys.path.insert(0, ROOT)
    from foldcomp_amd.shard import shard_cuts
    cuts = shard_cuts([n for _, n in items], 8)
    assert sum(1 for r in range(8) if cuts[r + 1] == cuts[r]) >= 3
    out = tmp_path / "out"
    assert _run(8, mode, str(inp), str(out)) == [0] * 8
    data, idx, lk = _expected(items)
    assert out.read_bytes() == data and (tmp_path / "out.index").read_text() == idx and (tmp_path / "out.lookup").read_text() == lk
    assert sorted(os.listdir(tmp_path)) == ["items.txt", "out", "out.dbtype", "out.index", "out.lookup"]


def test_preflight_ends_a_run_whose_group_does_not_answer(tmp_path):
    """a rank that never reaches the first collective: the others do not hang in it -- the pre-flight gives the group
    FCZ_PREFLIGHT_S seconds, says what it saw, kills its engine and leaves with status 3; nothing stays behind"""
    items = _items(11)
    inp = tmp_path / "items.txt"
    inp.write_text("".join(f"{nm} {n}\n" for nm, n in items))
    code = f"""
import os, sys, time, argparse
sys.path.insert(0, {ROOT!r})
rank = int(os.environ["RANK"])
import torch
torch.cuda.is_available = lambda: True
torch.cuda.device_count = lambda: 1
from foldcomp_amd import sharded_cli, shard
sharded_cli.ENGINE = {STANDIN!r}
if rank == 1:
    real = shard.preflight
    def late(*a, **k):
        time.sleep(30)          # this rank sits in front of its first collective: the group never answers in time
        os._exit(0)
    shard.preflight = late
a = argparse.Namespace(mode="decompress", input={str(inp)!r}, threads=2, brk=25, recursive=False, skip_discontinuous=False, alt=False, check=False,
                       id_list=None, id_mode=1, file_input=False, json_stats=False)
sys.exit(sharded_cli.run(a, [{str(inp)!r}], {str(tmp_path / "out")!r}))
"""
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FCZ_SHARD_BACKEND="gloo", FCZ_PREFLIGHT_S="4")
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stderr=subprocess.PIPE, text=True))
    err0 = procs[0].communicate(timeout=120)[1]
    assert procs[0].returncode == 3, (procs[0].returncode, err0[-500:])
    assert "pre-flight: no answer from the 2-rank group within 4 s" in err0 and "MASTER_ADDR" in err0
    procs[1].kill(); procs[1].wait()
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("out")) == []


def test_a_failed_compress_rank_leaves_nothing_behind(tmp_path):
    items = _items(19)
    inp = tmp_path / "items.txt"
    inp.write_text("".join(f"{nm} {n}\n" for nm, n in items))
    rcs = _run(2, "compress", str(inp), str(tmp_path / "out"), fail="write:1")
    assert all(rc not in (0, None) for rc in rcs), rcs
    assert sorted(os.listdir(tmp_path)) == ["items.txt"], os.listdir(tmp_path)
