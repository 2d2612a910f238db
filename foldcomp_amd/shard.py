"""Sharding a database across the GPUs of one node: the range cut and the one exchange step.

Structures are independent, so a run has no data-path collective. Every rank lists the inputs identically, takes the contiguous
range whose cumulative input bytes lie between rank/world and (rank+1)/world of the total (`shard_range`, the rule the C++
engine's InputPlan applies to the same listing, host/foldcomp_hip.cpp), and its engine writes a complete PARTIAL database with
keys and offsets counted from 0 (rank 0 straight into the final files). Then, once:

  1. `exchange_counts`: ONE all_gather of {records, data bytes, failed} over the process group (backend "nccl" = RCCL over xGMI
     with device tensors; "gloo" in the CPU tests) -> every rank knows key0 = records of the ranks before it and off0 = their
     bytes (the reference numbers keys in arrival order under `omp critical`, src/main.cpp:514-518: here input order);
  2. compress -- `splice`: rank r > 0 moves its data into the final file at off0 and rewrites its index / lookup lines with key0 /
     off0 added (`foldcomp-hip db-splice`: in-kernel copy, line streaming); after a barrier rank 0 appends those line files to its
     own (.index / .lookup sorted by key as free_writer leaves them, src/database_writer.cpp:59-73; .dbtype = int32 12). The data
     moved is the FCZ output (2.5 % of the bytes the run reads).
     decompress -- the exchange comes FIRST (the engines measure their ranges before they write, foldcomp_amd/sharded_cli.py), every
     rank writes its records once at off0 in the final file with final index lines, and `join_lines` only appends line files.

No rank holds a per-record Python object or another rank's rows: memory is O(one job of the engine) whatever the shard size.
"""
from __future__ import annotations

import os
import subprocess
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist

HOST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "host", "foldcomp-hip")


def shard_cuts(weights: Sequence[int], world: int) -> List[int]:
    """cut_r for r = 0..world: item g belongs to rank r iff cut_r <= g < cut_(r+1); cut_r = the first item with at least
    ceil(total x r / world) bytes before it (integer arithmetic: every rank and both implementations agree)"""
    n = len(weights)
    total = int(sum(int(w) for w in weights))
    cuts = [0] * (world + 1)
    cuts[world] = n
    g, before = 0, 0
    for r in range(1, world):
        t = (total * r + world - 1) // world
        while g < n and before < t:
            before += int(weights[g]); g += 1
        cuts[r] = g
    return cuts


def shard_range(n_items: int, weights: Sequence[int], rank: int, world: int) -> Tuple[int, int]:
    """contiguous [lo, hi) of items for `rank`, balanced by cumulative weight (bytes ~ residues)"""
    if n_items == 0:
        return 0, 0
    cuts = shard_cuts(list(weights)[:n_items], world)
    return cuts[rank], cuts[rank + 1]


PREFLIGHT_S = float(os.environ.get("FCZ_PREFLIGHT_S", "60"))


def preflight(device=None, timeout_s: float = None):
    """Fail fast instead of hanging: ONE 1-element all_gather of the ranks' numbers, given `timeout_s` seconds (FCZ_PREFLIGHT_S,
    default 60), before any work is generated. A collective that hangs cannot be cancelled, so it runs on a thread of its own and
    the caller decides what to kill. -> (ok, message); the message names what a maintainer checks first."""
    import threading
    world, rank = dist.get_world_size(), dist.get_rank()
    timeout_s = PREFLIGHT_S if timeout_s is None else timeout_s
    res = {}

    def go():
        try:
            t = torch.full((1,), rank, dtype=torch.int64, device=device)
            out = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(out, t)
            if device is not None:
                torch.cuda.synchronize(device)
            res["ranks"] = [int(x.item()) for x in out]
        except Exception as e:   # noqa: BLE001
            res["error"] = f"{type(e).__name__}: {e}"

    th = threading.Thread(target=go, daemon=True)
    th.start(); th.join(timeout_s)
    env = {k: os.environ.get(k) for k in ("MASTER_ADDR", "MASTER_PORT", "WORLD_SIZE", "RANK", "LOCAL_RANK", "HSA_ENABLE_IPC_MODE_LEGACY", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES")}
    where = f"rank {rank} of {world}, backend {dist.get_backend()}, device {device}, {env}"
    if th.is_alive():
        return False, (f"pre-flight: no answer from the {world}-rank group within {timeout_s:.0f} s ({where}). A rank that never joined, xGMI / "
                       "IPC between the GPUs (HSA_ENABLE_IPC_MODE_LEGACY=0 is needed on this driver), or a rendezvous address the ranks do not share")
    if "error" in res:
        return False, f"pre-flight: the first collective failed: {res['error']} ({where})"
    if res.get("ranks") != list(range(world)):
        return False, f"pre-flight: the group answered {res.get('ranks')}, expected 0..{world - 1} ({where})"
    return True, f"pre-flight: {world} ranks answered"


def exchange_counts(records: int, nbytes: int, failed: bool, device=None, extra: Sequence[int] = ()):
    """the run's only collective: all_gather of this rank's {records, data bytes, failed, *extra} (int64).
    -> (key0, off0, any_failed, rows) with rows[r] = the list rank r sent"""
    world, rank = dist.get_world_size(), dist.get_rank()
    t = torch.tensor([int(records), int(nbytes), 1 if failed else 0, *[int(x) for x in extra]], dtype=torch.int64, device=device)
    allv = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(allv, t)
    rows = [[int(x) for x in v.cpu()] for v in allv]
    key0 = sum(r[0] for r in rows[:rank]); off0 = sum(r[1] for r in rows[:rank])
    return key0, off0, any(r[2] for r in rows), rows


def remove_db(path: str) -> None:
    for ext in ("", ".index", ".lookup", ".dbtype"):
        try:
            os.remove(path + ext)
        except OSError:
            pass


def remove_rank_files(output: str, rank: int) -> None:
    """what rank `rank` may have left beside the output: its partial database and its index / lookup line files"""
    if rank > 0:
        remove_db(f"{output}.part{rank}")
        for ext in (".index", ".lookup"):
            try:
                os.remove(f"{output}{ext}.{rank}")
            except OSError:
                pass


def join_lines(output: str, device=None, host: str = HOST) -> bool:
    """file side of a PLACED run (decompress): every rank has already written its records into `output` at their final offsets and
    its index / lookup lines with final keys (rank 0: output.index, rank r: output.index.r). After a barrier rank 0 appends the
    line files in rank order -- the only bytes moved after the engines end. -> success on every rank"""
    world, rank = dist.get_world_size(), dist.get_rank()
    ok = True
    dist.barrier()
    if rank == 0 and world > 1:
        # (in this process: the lines are a few tens of bytes per record, starting the engine binary for them costs more than the copy)
        import shutil
        try:
            for ext in (".index", ".lookup"):
                with open(output + ext, "ab") as dst:
                    for r in range(1, world):
                        with open(f"{output}{ext}.{r}", "rb") as src:
                            shutil.copyfileobj(src, dst, 8 << 20)
            for ext in (".index", ".lookup"):
                for r in range(1, world):
                    os.remove(f"{output}{ext}.{r}")
        except OSError as e:
            print(f"[Error] joining the index lines: {e}", flush=True)
            ok = False
    if world == 1:
        return ok
    t = torch.tensor([0 if ok else 1], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item()) == 0


def splice(output: str, part: str, key0: int, off0: int, device=None, host: str = HOST) -> bool:
    """file side of the exchange (see the module text); every rank calls it once after exchange_counts. -> success on every rank"""
    world, rank = dist.get_world_size(), dist.get_rank()
    ok = True
    if rank > 0:
        r = subprocess.run([host, "db-splice", "--shard", f"{rank}/{world}", "--key0", str(key0), "--off0", str(off0), part, output])
        ok = r.returncode == 0
    dist.barrier()                                   # every rank's lines are on disk
    if rank == 0 and world > 1:
        r = subprocess.run([host, "db-splice", "--shard", f"0/{world}", output, output])
        ok = r.returncode == 0
    if world == 1:
        return ok
    t = torch.tensor([0 if ok else 1], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item()) == 0
