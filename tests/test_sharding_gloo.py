"""CPU: the sharded-database path without a GPU -- the range cut every rank computes for itself (C++ engine's
InputPlan == shard.shard_cuts, database entries streamed from the index), and the exchange step with REAL ranks (gloo,
world_size 2 and 3): partial databases -> one all_gather of the counts -> splice == the database a single writer makes.
The partial databases come from `foldcomp-hip db-pack` here (the codec needs a GPU; the sharding logic does not)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "host", "foldcomp-hip")

pytestmark = pytest.mark.skipif(not os.path.exists(HOST), reason="host/foldcomp-hip not built")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _host(*args):
    r = subprocess.run([HOST, *args], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def _plan(inp, rank, world, *flags):
    """-> (items [(kind, name, off, len)], summary) of `foldcomp-hip plan-dump --shard rank/world`"""
    out = _host("plan-dump", "--shard", f"{rank}/{world}", *flags, inp).splitlines()
    items = [tuple(l.split("\t")) for l in out[:-1]]
    return [(int(k), n, int(o), int(ln)) for k, n, o, ln in items], json.loads(out[-1])


def _make_db(path, n, rng, keys=None, name=lambda k: f"n{k}"):
    data = b""; idx = []; lk = []
    keys = list(range(n)) if keys is None else keys
    for k in keys:
        e = bytes(rng.integers(0, 256, int(rng.integers(1, 400)), dtype=np.uint8))
        idx.append((k, len(data), len(e))); lk.append((k, name(k))); data += e
    open(path, "wb").write(data)
    open(path + ".index", "w").write("".join(f"{k}\t{o}\t{l}\n" for k, o, l in idx))
    open(path + ".lookup", "w").write("".join(f"{k}\t{nm}\t0\n" for k, nm in lk))
    open(path + ".dbtype", "wb").write((12).to_bytes(4, "little"))
    return idx, lk, data


def test_shard_range_balances():
    from foldcomp_amd.shard import shard_cuts, shard_range
    w = [10] * 100
    cuts = [shard_range(100, w, r, 8) for r in range(8)]
    assert cuts[0][0] == 0 and cuts[-1][1] == 100
    assert all(a[1] == b[0] for a, b in zip(cuts[:-1], cuts[1:]))
    assert max(hi - lo for lo, hi in cuts) - min(hi - lo for lo, hi in cuts) <= 1
    assert shard_range(0, [], 0, 2) == (0, 0)
    # integer rule: cut_r = the first item with ceil(total r / world) bytes before it; zero-weight items go with their successor
    assert shard_cuts([0, 0, 5, 0, 5, 0], 2) == [0, 3, 6] and shard_cuts([0, 0, 0], 3) == [0, 0, 0, 3]
    assert shard_cuts([7], 4) == [0, 1, 1, 1, 1]


def test_engine_plan_equals_shard_cuts(tmp_path):
    """the listing + range cut of the C++ engine (`--shard R/N`): identical to shard.shard_cuts on the same weights, contiguous,
    disjoint, complete -- for directories (sorted walk), databases (streamed; key order) and a `-f` list of both"""
    from foldcomp_amd.shard import shard_cuts
    rng = np.random.default_rng(5)
    d = tmp_path / "files"
    d.mkdir()
    sizes = {}
    for i in range(97):
        sizes[f"f{i:03d}.pdb"] = int(rng.integers(0, 5000))
        (d / f"f{i:03d}.pdb").write_bytes(b"x" * sizes[f"f{i:03d}.pdb"])
    idx, lk, data = _make_db(str(tmp_path / "db"), 9000, rng, keys=[3 * k + 1 for k in range(9000)])
    lst = tmp_path / "inputs.txt"
    lst.write_text(f"{tmp_path / 'db'}\n{d}\n")
    file_items = [(0, str(d / n), 0, sizes[n]) for n in sorted(sizes)]
    db_items = [(1, nm, o, l) for (k, o, l), (_, nm) in zip(idx, lk)]
    for inp, flags, want in ((str(d), (), file_items), (str(tmp_path / "db"), (), db_items), (str(lst), ("-f",), db_items + file_items)):
        for world in (1, 2, 3, 8):
            cuts = shard_cuts([it[3] for it in want], world)
            got_all = []
            for r in range(world):
                items, summ = _plan(inp, r, world, *flags)
                if world == 1:                     # (file sizes are only asked for when a cut needs them)
                    items = [(k, n, o, sizes[os.path.basename(n)] if k == 0 else ln) for k, n, o, ln in items]
                assert items == want[cuts[r]:cuts[r + 1]], (inp, world, r)
                assert summ["items"] == len(items) and summ["items_total"] == len(want) and summ["streamed_inputs"] is True
                got_all += items
            assert got_all == want


def test_engine_plan_falls_back_for_databases_it_cannot_stream(tmp_path):
    """a database whose index is not in key order, or read through --id-list, goes through the in-memory reader: same entries in key
    order (the reference's reader sorts by key, src/database_reader.cpp:109), `streamed_inputs` false"""
    rng = np.random.default_rng(9)
    keys = list(range(300))
    rng.shuffle(keys)
    idx, lk, data = _make_db(str(tmp_path / "db"), 300, rng, keys=keys)
    by_key = sorted(zip(idx, lk))
    want = [(1, nm, o, l) for (k, o, l), (_, nm) in by_key]
    got = []
    for r in range(3):
        items, summ = _plan(str(tmp_path / "db"), r, 3)
        assert summ["streamed_inputs"] is False
        got += items
    assert got == want
    (tmp_path / "ids.txt").write_text("n3\nn17\nmissing\nn5\n")
    items, summ = _plan(str(tmp_path / "db"), 0, 1, "-l", str(tmp_path / "ids.txt"))
    assert [it[1] for it in items] == ["n3", "n17", "n5"] and summ["streamed_inputs"] is False


def test_streamed_database_memory_does_not_grow_with_the_database(tmp_path):
    """the engine's view of a database 10x the size costs the same memory: the index is streamed, not loaded (the in-memory reader
    on the same files grows by its rows and names)"""
    rng = np.random.default_rng(11)

    def make(n, sub):
        p = str(tmp_path / sub)
        with open(p + ".index", "w") as fi, open(p + ".lookup", "w") as fl:
            off = 0
            for k in range(n):
                fi.write(f"{k}\t{off}\t{100 + k % 7}\n"); fl.write(f"{k}\tAF-Q{k:09d}-F1-model_v4\t0\n"); off += 100 + k % 7
        open(p, "wb").truncate(off)
        open(p + ".dbtype", "wb").write((12).to_bytes(4, "little"))
        return p
    small, big = make(100_000, "s"), make(1_000_000, "b")
    rss = {}
    for name, p in (("small", small), ("big", big)):
        out = _host("plan-dump", "--json-stats", "--shard", "1/4", p).splitlines()
        summ = json.loads(out[-1])
        assert summ["streamed_inputs"] is True and summ["items_total"] in (100_000, 1_000_000) and abs(summ["items"] - summ["items_total"] / 4) < 8
        rss[name] = summ["max_rss_kb"]
    assert rss["big"] < rss["small"] + 4096, rss          # 10x the entries: less than 4 MB more (marks of every 4 096th line)
    # the in-memory reader on the big one (forced through --id-list): every row and name resident
    (tmp_path / "one.txt").write_text("AF-Q000000005-F1-model_v4\n")
    out = _host("plan-dump", "--json-stats", "-l", str(tmp_path / "one.txt"), big).splitlines()
    assert json.loads(out[-1])["max_rss_kb"] > rss["big"] + 50_000


def _worker(rank, world, port, tmp):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from foldcomp_amd import shard
    files = sorted(os.listdir(os.path.join(tmp, "files")))
    cuts = shard.shard_cuts([os.path.getsize(os.path.join(tmp, "files", f)) for f in files], world)
    mine = os.path.join(tmp, f"in{rank}")
    os.mkdir(mine)
    for f in files[cuts[rank]:cuts[rank + 1]]:
        os.symlink(os.path.join(tmp, "files", f), os.path.join(mine, f))
    out = os.path.join(tmp, f"db{world}")
    part = out if rank == 0 else f"{out}.part{rank}"
    subprocess.run([HOST, "db-pack", mine, part], check=True)           # this rank's partial database: keys and offsets from 0
    n = cuts[rank + 1] - cuts[rank]
    key0, off0, failed, rows = shard.exchange_counts(n, os.path.getsize(part), False)
    assert not failed and key0 == cuts[rank] and [r[0] for r in rows] == [cuts[r + 1] - cuts[r] for r in range(world)]
    assert shard.splice(out, part, key0, off0)
    dist.barrier()
    if rank > 0:
        assert not any(os.path.exists(part + ext) for ext in ("", ".index", ".lookup", ".dbtype"))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_ranks_splice_their_partial_databases_into_the_single_writers(tmp_path, world):
    rng = np.random.default_rng(3)
    (tmp_path / "files").mkdir()
    for i in range(61):
        (tmp_path / "files" / f"e{i:03d}.fcz").write_bytes(bytes(rng.integers(0, 256, int(rng.integers(0, 3000)), dtype=np.uint8)))
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    _host("db-pack", str(tmp_path / "files"), str(tmp_path / "single"))
    for suffix in ("", ".index", ".lookup", ".dbtype"):
        assert open(str(tmp_path / f"db{world}") + suffix, "rb").read() == open(str(tmp_path / "single") + suffix, "rb").read(), suffix
    assert not [f for f in os.listdir(tmp_path) if ".part" in f or ".index." in f or ".lookup." in f]
    sys.path.insert(0, ROOT)
    from foldcomp_amd.database import DatabaseReader
    r = DatabaseReader(str(tmp_path / f"db{world}"))
    assert len(r) == 61 and r.data(7) == (tmp_path / "files" / "e007.fcz").read_bytes() and r.name(60) == "e060"
    r.close()


def test_a_failed_rank_leaves_no_database_behind(tmp_path):
    """exchange_counts carries the failure flag: every rank learns of it before anything is spliced"""
    port = _free_port()
    mp.spawn(_fail_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert not [f for f in os.listdir(tmp_path) if f.startswith("out")]


def _fail_worker(rank, world, port, tmp):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from foldcomp_amd import shard
    part = os.path.join(tmp, "out" if rank == 0 else "out.part1")
    for ext in ("", ".index", ".lookup", ".dbtype"):
        open(part + ext, "w").write("x")
    key0, off0, failed, rows = shard.exchange_counts(1, 1, rank == 1)
    assert failed
    shard.remove_db(part)
    dist.destroy_process_group()


def _placed_worker(rank, world, port, tmp):
    """the decompress shape of the exchange: counts FIRST, then every rank writes its records once at its final offset of the
    final file with final index lines (what `foldcomp-hip decompress --place` does after its sizes pass), rank 0 joins the lines"""
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from foldcomp_amd import shard
    files = sorted(os.listdir(os.path.join(tmp, "files")))
    cuts = shard.shard_cuts([os.path.getsize(os.path.join(tmp, "files", f)) for f in files], world)
    mine = files[cuts[rank]:cuts[rank + 1]]
    recs = [open(os.path.join(tmp, "files", f), "rb").read() for f in mine]
    out = os.path.join(tmp, f"placed{world}")
    if rank == 0:
        shard.remove_db(out)
    key0, off0, failed, rows = shard.exchange_counts(len(recs), sum(len(r) for r in recs), False)     # before any byte is written
    assert not failed and key0 == cuts[rank]
    assert not [f for f in os.listdir(tmp) if f.startswith(f"placed{world}")]
    dist.barrier()
    fd = os.open(out, os.O_CREAT | os.O_WRONLY, 0o666)
    tag = "" if rank == 0 else f".{rank}"
    with open(out + ".index" + tag, "w") as fi, open(out + ".lookup" + tag, "w") as fl:
        o = off0
        for k, (f, r) in enumerate(zip(mine, recs)):
            os.pwrite(fd, r, o)
            fi.write(f"{key0 + k}\t{o}\t{len(r)}\n"); fl.write(f"{key0 + k}\t{os.path.splitext(f)[0]}\t0\n"); o += len(r)
    os.close(fd)
    if rank == 0:
        open(out + ".dbtype", "wb").write((12).to_bytes(4, "little"))
    assert shard.join_lines(out)
    dist.barrier()
    assert not [f for f in os.listdir(tmp) if ".part" in f]                      # no partial database ever existed
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_placed_ranks_write_once_into_the_final_database(tmp_path, world):
    rng = np.random.default_rng(4)
    (tmp_path / "files").mkdir()
    for i in range(47):
        (tmp_path / "files" / f"e{i:03d}.fcz").write_bytes(bytes(rng.integers(0, 256, int(rng.integers(0, 3000)), dtype=np.uint8)))
    mp.spawn(_placed_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    _host("db-pack", str(tmp_path / "files"), str(tmp_path / "single"))
    for suffix in ("", ".index", ".lookup", ".dbtype"):
        assert open(str(tmp_path / f"placed{world}") + suffix, "rb").read() == open(str(tmp_path / "single") + suffix, "rb").read(), suffix
    assert not [f for f in os.listdir(tmp_path) if ".index." in f or ".lookup." in f]
