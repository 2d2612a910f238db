"""Sharding a database across the GPUs of one node.

Structures are independent, so the only cross-rank step is the *index*: every rank compresses (or
decompresses) a contiguous range of entries, then
  1. all ranks exchange their record byte lengths (all_gather of one int64 total -> exclusive prefix gives
     each rank's byte offset in the output data file, so ranks can pwrite their blob slices directly), and
  2. rank 0 gathers packed index rows -- int64 (key, length, name_off) records plus one uint8 blob of the names, two
     `gather`s of flat tensors, no pickled objects -- and writes `.index` / `.lookup` / `.dbtype` exactly like
     free_writer (reference src/database_writer.cpp:59-73).
`torch.distributed` is the transport: backend "nccl" (RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
Keys are assigned deterministically = input order (the reference's `key++` under `omp critical`,
src/main.cpp:514-518, is thread-schedule dependent, so per-entry bytes are the parity target).
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_items: int, weights: Sequence[int], rank: int, world: int) -> Tuple[int, int]:
    """contiguous [lo, hi) of items for `rank`, balanced by cumulative weight (bytes ~ residues)"""
    if n_items == 0:
        return 0, 0
    w = np.asarray(weights, np.float64)
    c = np.concatenate([[0.0], np.cumsum(w)])
    total = c[-1] if c[-1] > 0 else 1.0
    cuts = [int(np.searchsorted(c, total * r / world, side="left")) for r in range(world + 1)]
    cuts[0], cuts[-1] = 0, n_items
    for r in range(1, world + 1):
        cuts[r] = max(cuts[r], cuts[r - 1])
    return cuts[rank], cuts[rank + 1]


def exchange_offsets(local_bytes: int, device=None) -> Tuple[int, int]:
    """-> (byte offset of this rank's slice in the output data file, total bytes)"""
    world = dist.get_world_size()
    t = torch.tensor([int(local_bytes)], dtype=torch.int64, device=device)
    allv = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(allv, t)
    sizes = [int(v.item()) for v in allv]
    r = dist.get_rank()
    return sum(sizes[:r]), sum(sizes)


def _gather_var(t: torch.Tensor, device=None):
    """gather a 1-D tensor of rank-dependent length on rank 0 -> list of per-rank tensors there, None elsewhere.
    One all_gather of the lengths, then one gather of buffers padded to the longest."""
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([t.numel()], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    mx = max(counts + [1])
    pad = torch.zeros(mx, dtype=t.dtype, device=device)
    pad[:t.numel()] = t
    bufs = [torch.zeros(mx, dtype=t.dtype, device=device) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, bufs, dst=0)
    if rank != 0:
        return None
    return [bufs[r][:counts[r]] for r in range(world)]


def pack_index(lengths, keys, names: Sequence[str]):
    """this rank's index rows as flat arrays: int64 records (key, length, name_off) x n and one uint8 blob of the
    names back to back (name i = blob[name_off[i] : name_off[i+1]], the end of the last one = len(blob)). No Python
    objects cross the process group: at 214 M entries the rows are 5 GB of int64 and the names one byte array."""
    enc = [s.encode() for s in names]
    name_off = np.zeros(len(enc) + 1, np.int64)
    if enc:
        np.cumsum([len(e) for e in enc], out=name_off[1:])
    rec = np.stack([np.asarray(keys, np.int64).reshape(-1), np.asarray(lengths, np.int64).reshape(-1), name_off[:-1]], 1).reshape(-1)
    blob = np.frombuffer(b"".join(enc), np.uint8).copy() if enc else np.zeros(0, np.uint8)
    return rec, blob


def gather_index(lengths: np.ndarray, keys: np.ndarray, names: Sequence[str], device=None):
    """gather the packed index rows (pack_index) on rank 0; returns (keys, offsets, lengths, names) there, None elsewhere.
    Offsets are those of the data file the ranks wrote with write_sharded_db: rank slices back to back, records of a
    rank in its own order."""
    rec, blob = pack_index(lengths, keys, names)
    recs = _gather_var(torch.from_numpy(rec).to(device) if device is not None else torch.from_numpy(rec), device)
    blobs = _gather_var(torch.from_numpy(blob).to(device) if device is not None else torch.from_numpy(blob), device)
    if dist.get_rank() != 0:
        return None
    k_all, l_all, n_all = [], [], []
    for r in range(dist.get_world_size()):
        v = recs[r].cpu().numpy().reshape(-1, 3)
        nb = blobs[r].cpu().numpy().tobytes()
        ends = np.concatenate([v[1:, 2], [len(nb)]]) if len(v) else np.zeros(0, np.int64)
        k_all.append(v[:, 0]); l_all.append(v[:, 1])
        n_all += [nb[a:b].decode() for a, b in zip(v[:, 2].tolist(), ends.tolist())]
    k_all = np.concatenate(k_all); l_all = np.concatenate(l_all)
    offs = np.concatenate([[0], np.cumsum(l_all)[:-1]]) if len(l_all) else np.zeros(0, np.int64)
    return k_all, offs, l_all, n_all


def pwrite_all(fd: int, data, offset: int) -> None:
    """os.pwrite until every byte is on its way: one call moves at most 0x7ffff000 bytes on Linux and may write less"""
    view = memoryview(data).cast("B")
    while len(view):
        n = os.pwrite(fd, view[:1 << 30], offset)
        if n <= 0:
            raise OSError(f"pwrite returned {n} with {len(view)} bytes left at offset {offset}")
        view = view[n:]
        offset += n


def write_sharded_db(path: str, blob: bytes, lengths: np.ndarray, keys: np.ndarray, names: List[str], device=None):
    """every rank writes its blob slice at its prefix offset; rank 0 writes index/lookup/dbtype"""
    off, total = exchange_offsets(len(blob), device)
    rank = dist.get_rank()
    if rank == 0:
        with open(path, "wb") as f:
            f.truncate(total)
    dist.barrier()
    fd = os.open(path, os.O_WRONLY)
    try:
        pwrite_all(fd, blob, off)
    finally:
        os.close(fd)
    idx = gather_index(lengths, keys, names, device)
    if rank == 0:
        k, o, l, nm = idx
        order = np.argsort(k, kind="stable")
        with open(path + ".index", "w") as fi, open(path + ".lookup", "w") as fl:
            for i in order:
                fi.write("%d\t%d\t%d\n" % (k[i], o[i], l[i]))
                fl.write("%d\t%s\t0\n" % (k[i], nm[i]))
        with open(path + ".dbtype", "wb") as f:
            f.write((12).to_bytes(4, "little"))
    dist.barrier()
