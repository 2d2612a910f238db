"""`python -m foldcomp_amd compress|decompress -d --gpus N ...`: the database run sharded over the GPUs of one node
(reference driver loop src/input_processor.h:200-300, writer src/database_writer.cpp:59-73).

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI; "gloo" when several ranks have to share a device
or FCZ_SHARD_BACKEND=gloo says so). Structures are independent: the inputs -- the files of the input directories, or the
entries of the input databases in key order -- are listed identically on every rank, cut into `world` contiguous ranges
balanced by bytes (shard.shard_range), and every rank runs the codec on its own range:

    compress    plain PDB files through the structure ingest on the device (Codec.compress_pdb: text -> FCZ in HBM), what the
                device does not read or hands back (mmCIF, gzip, fields outside the fixed-column layout) through the host parser;
    decompress  FCZ entries -> PDB text with decode and formatting on the device (Codec.decompress_pdb).

The only exchange is the index: record counts (keys are numbered in input order over the records that made it), then
shard.write_sharded_db -- byte totals -> every rank pwrites its slice of the data file at its prefix offset, packed index rows
gathered on rank 0, which writes .index / .lookup / .dbtype as free_writer does. N = 1 runs the very same code in a 1-rank
group. Without a launcher in the environment the command starts its own N ranks (torch.distributed.run on 127.0.0.1).
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import List, Optional, Tuple

import numpy as np

PDB_EXT = (".pdb", ".ent")


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(argv: List[str], gpus: int) -> int:
    """start `gpus` ranks of this command line (one per GPU) and wait for them"""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "-m", "foldcomp_amd", *argv]
    return subprocess.run(cmd, env=env).returncode


# ---- the inputs, listed the same way on every rank -----------------------------------------------------------------------
class Item:
    __slots__ = ("name", "size", "src", "idx")

    def __init__(self, name, size, src, idx):
        self.name, self.size, self.src, self.idx = name, int(size), src, idx   # src: a path, or a DatabaseReader with entry idx


def list_items(inputs: List[str], recursive: bool, id_list: Optional[str], id_mode: int) -> List[Item]:
    from .database import DatabaseReader
    items: List[Item] = []
    for inp in inputs:
        if os.path.exists(inp + ".dbtype"):
            r = DatabaseReader(inp)
            ids = range(len(r))
            if id_list:
                want = [l.strip() for l in open(id_list) if l.strip()]
                ids = [r.id_of_key(int(w)) if id_mode == 0 else r.id_of_name(w) for w in want]
                ids = [i for i in ids if i >= 0]
            for i in ids:
                items.append(Item(r.name(i), r.lengths[i], r, i))
        elif os.path.isdir(inp):
            for root, dirs, files in os.walk(inp):
                dirs.sort()
                for f in sorted(files):
                    p = os.path.join(root, f)
                    items.append(Item(p, os.path.getsize(p), p, -1))
                if not recursive:
                    break
        else:
            items.append(Item(inp, os.path.getsize(inp), inp, -1))
    return items


def read_item(it: Item) -> bytes:
    if it.idx >= 0:
        return it.src.data(it.idx)
    with open(it.src, "rb") as f:
        return f.read()


# ---- one rank's share ------------------------------------------------------------------------------------------------------
def compress_items(items: List[Item], a, codec) -> List[Tuple[str, bytes]]:
    """-> [(database name, FCZ record)] of the items, in item order then fragment order"""
    from .__main__ import host_fragments
    from .structure import StructureError, build_batch
    per_file: List[List[Tuple[str, bytes]]] = [[] for _ in items]
    step = 2048
    for s0 in range(0, len(items), step):
        chunk = list(range(s0, min(len(items), s0 + step)))
        datas = {i: read_item(items[i]) for i in chunk}
        dev = [i for i in chunk if items[i].name.endswith(PDB_EXT)]
        host = [i for i in chunk if i not in set(dev)]
        if dev:
            names = [os.path.basename(items[i].name) for i in dev]
            r = codec.compress_pdb([datas[i] for i in dev], names, a.brk, a.skip_discontinuous)
            for c in range(len(r["status"])):
                i = dev[int(r["chain_file"][c])]
                if r["status"][c] != 0:
                    print(f"[Error] compressing {os.path.basename(items[i].name)}", file=sys.stderr); continue
                per_file[i].append((os.path.basename(items[i].name).rsplit(".", 1)[0], r["blob"][int(r["off"][c]):int(r["off"][c + 1])].tobytes()))
            for f, meta in r["refused"]:
                print(f"[Error] compressing {names[int(f)]}: fragment refused (reason {int(meta) >> 24})", file=sys.stderr)
            for k, st in enumerate(r["file_status"]):
                if st == 4:
                    print(f"[Error] No atoms found in the input file: {names[k]}", file=sys.stderr)
                elif st != 0:
                    host.append(dev[k])                    # handed back: the host parser takes the file
        pend = []                                           # (item, db name, chain)
        for i in sorted(host):
            try:
                for fname, dbname, ch in host_fragments(items[i].name, datas[i], a, "db", False, None):
                    pend.append((i, dbname, ch))
            except Exception as e:  # noqa: BLE001 - parse errors are reported and skipped like the reference
                print(f"[Error] {os.path.basename(items[i].name)}: {e}", file=sys.stderr)
        good = []
        for p in pend:
            try:
                build_batch([p[2]], a.brk); good.append(p)
            except StructureError as e:
                print(f"[Error] compressing {p[1]}: {e}", file=sys.stderr)
        if good:
            blob, off, st = codec.compress_batch(build_batch([p[2] for p in good], a.brk), strict=False)
            for q, (i, dbname, _) in enumerate(good):
                if st[q] != 0:
                    print(f"[Error] compressing {dbname}", file=sys.stderr); continue
                per_file[i].append((dbname, blob[int(off[q]):int(off[q + 1])].tobytes()))
    return [rec for recs in per_file for rec in recs]


def decompress_items(items: List[Item], a, codec) -> List[Tuple[str, bytes]]:
    """-> [(database name, PDB text + NUL)] of the items (src/main.cpp:656-664)"""
    from . import _lib
    out: List[Tuple[str, bytes]] = []
    step = 4096
    for s0 in range(0, len(items), step):
        chunk = items[s0:s0 + step]
        ents = [read_item(it) for it in chunk]
        if a.check:
            keep = []
            for it, e in zip(chunk, ents):
                arr = np.frombuffer(e, np.uint8)
                if len(arr) == 0 or _lib.load().fcz_check(arr.ctypes.data, len(arr)) != 0:
                    print(f"[Error] invalid FCZ entry skipped: {it.name}", file=sys.stderr); continue
                keep.append((it, e))
            chunk, ents = [k[0] for k in keep], [k[1] for k in keep]
        if not ents:
            continue
        off = np.zeros(len(ents) + 1, np.uint64)
        off[1:] = np.cumsum([len(e) for e in ents])
        texts, status = codec.decompress_pdb(np.frombuffer(b"".join(ents), np.uint8), off, alt_order=a.alt)
        for it, t, st in zip(chunk, texts, status):
            if st != 0:
                print(f"[Error] decompressing {it.name}", file=sys.stderr); continue
            out.append((os.path.basename(it.name).rsplit(".", 1)[0] if "." in os.path.basename(it.name) else os.path.basename(it.name), t + b"\0"))
    return out


def run(a, inputs: List[str], output: str) -> int:
    """this process's rank of the sharded run (a 1-rank group when no launcher set the environment)"""
    import torch
    import torch.distributed as dist
    from . import shard
    from .codec import Codec
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", str(rank)))
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n_dev <= 0:
        print("[Error] no HIP device (the codec has no CPU fallback)", file=sys.stderr); return 1
    backend = os.environ.get("FCZ_SHARD_BACKEND") or ("nccl" if world <= n_dev else "gloo")
    device_index = local % n_dev
    if backend == "nccl":
        torch.cuda.set_device(device_index)
    if "MASTER_ADDR" not in os.environ:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group(backend, rank=rank, world_size=world)
    tdev = torch.device("cuda", device_index) if backend == "nccl" else None
    try:
        items = list_items(inputs, a.recursive, a.id_list if a.mode != "compress" else None, a.id_mode)
        lo, hi = shard.shard_range(len(items), [it.size for it in items], rank, world)
        with Codec(device_index) as codec:
            recs = (compress_items if a.mode == "compress" else decompress_items)(items[lo:hi], a, codec)
        # keys: input order over the records that made it -> this rank's first key = the counts of the ranks before it
        n = torch.tensor([len(recs)], dtype=torch.int64, device=tdev)
        counts = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(counts, n)
        key0 = sum(int(c.item()) for c in counts[:rank])
        blob = b"".join(r[1] for r in recs)
        lengths = np.asarray([len(r[1]) for r in recs], np.int64)
        keys = np.arange(key0, key0 + len(recs), dtype=np.int64)
        shard.write_sharded_db(output, blob, lengths, keys, [r[0] for r in recs], tdev)
        if rank == 0 and getattr(a, "json_stats", False):
            import json
            print(json.dumps({"mode": a.mode, "world": world, "backend": backend, "items": len(items),
                              "records": int(sum(int(c.item()) for c in counts))}))
    finally:
        dist.destroy_process_group()
    return 0
