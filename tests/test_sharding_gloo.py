"""CPU, world_size 2 (gloo): the sharded-database path -- range partition, offset exchange, index gather,
per-rank pwrite -- produces the same database as a single writer. The per-record bytes come from the
oracle here (the GPU codec needs a GPU; the sharding logic does not)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, tmp, golden_path):
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from foldcomp_amd import shard
    z = np.load(golden_path)
    names = [n for n in bytes(z["index"]).decode().split("\n") if n.startswith("db:")]
    entries = [z[f"{n}/fcz"].tobytes() for n in names]
    lo, hi = shard.shard_range(len(entries), [len(e) for e in entries], rank, world)
    mine = entries[lo:hi]
    blob = b"".join(mine)
    shard.write_sharded_db(os.path.join(tmp, "db"), blob, np.asarray([len(e) for e in mine]), np.arange(lo, hi),
                           [bytes(z[f"{n}/name"]).decode() for n in names[lo:hi]])
    dist.destroy_process_group()


def test_pack_index_is_flat_arrays():
    from foldcomp_amd.shard import pack_index
    rec, blob = pack_index([10, 20, 30], [5, 6, 7], ["a", "bcd", ""])
    assert rec.dtype == np.int64 and rec.tolist() == [5, 10, 0, 6, 20, 1, 7, 30, 4]
    assert blob.dtype == np.uint8 and blob.tobytes() == b"abcd"
    rec, blob = pack_index([], [], [])
    assert rec.size == 0 and blob.size == 0


def test_pwrite_all_loops(tmp_path, monkeypatch):
    from foldcomp_amd import shard
    calls = []
    real = os.pwrite
    def short(fd, data, off):   # a kernel that takes 3 bytes at a time
        calls.append(off)
        return real(fd, bytes(data[:3]), off)
    monkeypatch.setattr(shard.os, "pwrite", short)
    p = tmp_path / "f"
    p.write_bytes(b"\0" * 12)
    fd = os.open(str(p), os.O_WRONLY)
    shard.pwrite_all(fd, b"0123456789", 2)
    os.close(fd)
    assert p.read_bytes() == b"\0\0" + b"0123456789" and calls == [2, 5, 8, 11]
    monkeypatch.setattr(shard.os, "pwrite", lambda fd, data, off: 0)
    fd = os.open(str(p), os.O_WRONLY)
    with pytest.raises(OSError):
        shard.pwrite_all(fd, b"xyz", 0)
    os.close(fd)


def test_shard_range_balances():
    from foldcomp_amd.shard import shard_range
    w = [10] * 100
    cuts = [shard_range(100, w, r, 8) for r in range(8)]
    assert cuts[0][0] == 0 and cuts[-1][1] == 100
    assert all(a[1] == b[0] for a, b in zip(cuts[:-1], cuts[1:]))
    assert max(hi - lo for lo, hi in cuts) - min(hi - lo for lo, hi in cuts) <= 1
    assert shard_range(0, [], 0, 2) == (0, 0)


def test_two_rank_sharded_db_equals_single_writer(tmp_path):
    golden_path = os.path.join(ROOT, "tests", "golden", "reference_vectors.npz")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), golden_path), nprocs=2, join=True)
    sys.path.insert(0, ROOT)
    from foldcomp_amd.database import DatabaseReader, DatabaseWriter
    z = np.load(golden_path)
    names = [n for n in bytes(z["index"]).decode().split("\n") if n.startswith("db:")]
    w = DatabaseWriter(str(tmp_path / "single"))
    for i, n in enumerate(names):
        w.append(z[f"{n}/fcz"].tobytes(), i, bytes(z[f"{n}/name"]).decode())
    w.close()
    for suffix in ("", ".index", ".lookup", ".dbtype"):
        assert open(str(tmp_path / "db") + suffix, "rb").read() == open(str(tmp_path / "single") + suffix, "rb").read(), suffix
    r = DatabaseReader(str(tmp_path / "db"))
    assert len(r) == 24 and r.data(7) == z[f"{names[7]}/fcz"].tobytes()
    r.close()


def test_sharded_cli_plan_covers_every_item_once(tmp_path):
    """the listing + range cut every rank of `python -m foldcomp_amd ... --gpus N` computes for itself: identical on every rank,
    contiguous, disjoint, complete, balanced by bytes -- for directories (sorted walk) and databases (key order, --id-list)"""
    from foldcomp_amd import shard, sharded_cli
    from foldcomp_amd.database import DatabaseWriter
    d = tmp_path / "files"
    d.mkdir()
    rng = np.random.default_rng(5)
    for i in range(97):
        (d / f"f{i:03d}.pdb").write_bytes(b"x" * int(rng.integers(1, 5000)))
    w = DatabaseWriter(str(tmp_path / "db"))
    for k in reversed(range(40)):
        w.append(b"y" * int(rng.integers(10, 3000)), k, f"n{k}")
    w.close()
    (tmp_path / "ids.txt").write_text("n3\nn17\nmissing\nn5\n")
    for inputs, id_list, n_exp in (([str(d)], None, 97), ([str(tmp_path / "db")], None, 40), ([str(tmp_path / "db"), str(d)], None, 137),
                                   ([str(tmp_path / "db")], str(tmp_path / "ids.txt"), 3)):
        items = sharded_cli.list_items(inputs, False, id_list, 1)
        assert len(items) == n_exp
        again = sharded_cli.list_items(inputs, False, id_list, 1)
        assert [it.name for it in items] == [it.name for it in again]
        for world in (1, 2, 3, 8):
            cuts = [shard.shard_range(len(items), [it.size for it in items], r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == len(items) and all(cuts[r][1] == cuts[r + 1][0] for r in range(world - 1))
            if world == 2 and n_exp > 20:
                tot = sum(it.size for it in items); half = sum(it.size for it in items[cuts[0][0]:cuts[0][1]])
                assert abs(half - tot / 2) <= max(it.size for it in items)
    if id_list := str(tmp_path / "ids.txt"):
        assert [it.name for it in sharded_cli.list_items([str(tmp_path / "db")], False, id_list, 1)] == ["n3", "n17", "n5"]
