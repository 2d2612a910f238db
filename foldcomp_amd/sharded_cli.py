"""`python -m foldcomp_amd compress|decompress -d --gpus N ...`: the database run sharded over the GPUs of one node
(reference driver loop src/input_processor.h:200-300, writer src/database_writer.cpp:59-73).

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI; "gloo" when several ranks have to share a device
or FCZ_SHARD_BACKEND=gloo says so). A rank is two things:

    engine    `host/foldcomp-hip <mode> -d --shard R/N --device D ...`: the pipelined C++ host (read threads -> page-locked
              buffers -> structure ingest + codec on the device -> sequenced writes) on the rank's byte-balanced contiguous range
              of the inputs -- files of the input directories, or entries of the input databases streamed from their index --
              writing a complete partial database with keys and offsets from 0, its index appended job by job;
    exchange  this process: ONE all_gather of {records, bytes} (shard.exchange_counts).
              compress    after the engines: the partial databases are spliced into one (shard.splice: in-kernel data copy at the
                          prefix offset, index lines rebased, rank 0 appends them). The copied data is the FCZ output, 16.5 B per
                          residue against ~660 B per residue of input text read: 2.5 % of the run's bytes.
              decompress  BEFORE anything is written (world > 1): the output text is what this direction is bound by, so a partial
                          database copied afterwards would write seven eighths of it twice. The engine (`--place`) first walks its
                          range measuring only (records decoded on the device, nothing formatted or written), reports its counts,
                          gets `key0 off0 total` back and writes every record once at its final offset of the final data file with
                          final index lines; rank 0 then only concatenates the ranks' line files (shard.join_lines). Reference:
                          every record appended once, src/main.cpp:656-664, src/database_writer.cpp:36-58.

Memory of a rank does not grow with its shard: nothing per record is held by the engine (jobs stream through) or by this
process (two integers per rank cross the group). configs[3] (214 M records over 8 GPUs) is 27 M records and ~160 GB of FCZ per
rank: the engine's job buffers stay at a few hundred MB. N = 1 runs the very same code in a 1-rank group. Without a launcher
in the environment the command starts its own N ranks (torch.distributed.run on 127.0.0.1).
"""
from __future__ import annotations

import json
import os
import socket
import subprocess
import sys
import time
from typing import List

# (torch -- through foldcomp_amd.shard -- is imported only after the engine has been started: its import and the communicator's
# set-up take seconds, which the engine spends working)
ENGINE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "host", "foldcomp-hip")


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


def host_threads(world: int) -> int:
    """read / parse threads of one rank's engine: this process's CPUs (affinity, cut by a cgroup quota) shared by the ranks"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n // max(1, world))


def engine_command(a, rank: int, world: int, device_index: int, out_path: str, host: str = None, place: bool = False) -> List[str]:
    """the rank's engine: the C++ host on its range of the inputs (same option letters as the reference's command line)"""
    cmd = [host or ENGINE, a.mode, "-d", "-y", "--gpus", "1", "--device", str(device_index), "--device-mod", "--shard", f"{rank}/{world}", "--json-stats",
           "-t", str(a.threads if a.threads and a.threads > 1 else host_threads(world)), "-b", str(a.brk)]
    if a.recursive:
        cmd.append("-r")
    if a.mode == "compress" and a.skip_discontinuous:
        cmd.append("--skip-discontinuous")
    if a.mode == "decompress":
        if a.alt:
            cmd.append("-a")
        if a.check:
            cmd.append("--check")
        if a.id_list:
            cmd += ["-l", a.id_list, "-m", str(a.id_mode)]
    if a.file_input:
        cmd.append("-f")
    if place:
        cmd.append("--place")
    return cmd + [a.input.rstrip("/") if len(a.input) > 1 else a.input, out_path]


def run(a, inputs: List[str], output: str) -> int:
    """this process's rank of the sharded run (a 1-rank group when no launcher set the environment)"""
    t_start = time.perf_counter()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if not os.path.exists(ENGINE):
        print(f"[Error] the engine {ENGINE} is not built (make -C host)", file=sys.stderr); return 1
    for inp in inputs:
        if inp.endswith((".tar", ".tar.gz", ".tgz")):
            print("[Error] --gpus shards directories and databases; unpack tar inputs first", file=sys.stderr); return 1
    # decompress with more than one rank: counts first, one write (see the module text); everything else: partial database + splice
    place = a.mode == "decompress" and world > 1
    part = output if (rank == 0 or place) else f"{output}.part{rank}"
    env = {k: v for k, v in os.environ.items() if k != "OMP_NUM_THREADS"}     # (a launcher exports OMP_NUM_THREADS=1: the engine gets -t)
    t_spawn = time.perf_counter()
    # (the engine takes device LOCAL_RANK modulo the device count: this process must not touch HIP before torch does -- the library
    # links the system's runtime, torch brings its own, and whichever initialises second finds no device)
    proc = subprocess.Popen(engine_command(a, rank, world, local, part, place=place), env=env, stdout=subprocess.PIPE, stdin=subprocess.PIPE if place else None, text=True)
    joined = False
    rc = 1
    try:
        # ---- beside the running engine: torch, the process group, the communicator ----
        import torch
        import torch.distributed as dist
        from . import shard
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        no_device = n_dev <= 0
        backend = os.environ.get("FCZ_SHARD_BACKEND") or ("nccl" if (not no_device and world <= n_dev) else "gloo")
        device_index = local % n_dev if n_dev else 0
        if backend == "nccl":
            torch.cuda.set_device(device_index)
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(_free_port())
        # (a rank without a device still joins the group and reports its failure through the exchange: the others must not wait in
        # the rendezvous for a rank that left)
        dist.init_process_group(backend, rank=rank, world_size=world)
        joined = True
        tdev = torch.device("cuda", device_index) if backend == "nccl" else None
        t_group = time.perf_counter()
        # RCCL builds its rings on the first collective (~1 s): done here it stays off the path between the engines' end and the
        # exchange -- and it is the PRE-FLIGHT: a group that does not answer within FCZ_PREFLIGHT_S (60 s) ends the run now, with what
        # to look at, instead of hanging in the exchange after the engines' work (a hung collective cannot be cancelled: the engine is
        # killed and the process leaves)
        ok, msg = shard.preflight(tdev)
        if not ok:
            print(f"[Error] {msg}", file=sys.stderr); sys.stderr.flush()
            proc.kill()
            if place and rank == 0:
                shard.remove_db(output)
            shard.remove_db(part); shard.remove_rank_files(output, rank)
            os._exit(3)
        t_comm = time.perf_counter()
        if no_device:
            proc.kill()
            print("[Error] no HIP device (the codec has no CPU fallback)", file=sys.stderr)
        sizes = {}
        t_sizes = t_comm
        if place:
            # the engine's sizes pass ends with one line; the exchange comes BEFORE the writes
            while True:
                line = proc.stdout.readline()
                if not line:
                    break
                if line.startswith("{"):
                    try:
                        sizes = json.loads(line)
                    except ValueError:
                        sizes = {}                    # a garbled line is this rank's failure in the exchange, not an exception that leaves the group
                        break
                    if sizes.get("phase") == "sizes":
                        break
            t_sizes = time.perf_counter()
            failed = no_device or not sizes or bool(sizes.get("failed"))
            key0, off0, any_failed, rows0 = shard.exchange_counts(sizes.get("records", 0), sizes.get("data_bytes", 0), failed, tdev)
            if not any_failed:
                # the ranks write INTO the final file: what an earlier run left (the database, line files of a larger world) goes
                # before any of them opens it -- and only now that every rank's inputs have passed their sizes pass: a run that fails
                # at once leaves the previous database alone
                if rank == 0:
                    shard.remove_db(output)
                    import glob as _glob
                    for stale in _glob.glob(output + ".index.*") + _glob.glob(output + ".lookup.*"):
                        try:
                            os.remove(stale)
                        except OSError:
                            pass
                dist.barrier()
            try:
                proc.stdin.write("abort\n" if any_failed else f"{key0} {off0} {sum(r_[1] for r_ in rows0)}\n"); proc.stdin.flush(); proc.stdin.close()
            except (BrokenPipeError, OSError):
                pass
            proc.stdin = None                         # (communicate() below must not flush a closed pipe)
        # (one mechanism for both phases: lines to EOF -- communicate() would read the raw descriptor past what readline buffered)
        st = {}
        for line in proc.stdout:
            if line.startswith("{"):
                try:
                    st = json.loads(line)
                except ValueError:
                    st = {}
        proc.wait()
        failed = no_device or proc.returncode != 0 or not st
        t_engine = time.perf_counter()
        extra = [st.get("residues", 0), int(st.get("wall_s", 0.0) * 1e6), int(st.get("ctx_ready_s", 0.0) * 1e6), st.get("max_rss_kb", 0),
                 st.get("items", st.get("files", 0)), st.get("input_bytes", st.get("fcz_bytes", 0)), int(st.get("sizes_pass_s", 0.0) * 1e6)]
        key0, off0, any_failed, rows = shard.exchange_counts(st.get("records", 0), st.get("data_bytes", 0), failed, tdev, extra)
        t_exchanged = time.perf_counter()             # (a collective: it returns when the SLOWEST rank's engine has ended)
        if any_failed:
            # a database without one rank's records looks complete: nothing is left behind
            shard.remove_db(part)
            shard.remove_rank_files(output, rank)
            if place:
                dist.barrier()
                if rank == 0:
                    shard.remove_db(output)
            if rank == 0:
                print(f"[Error] the run failed: {output} was not written", file=sys.stderr)
            rc = 1
        else:
            ok = shard.join_lines(output, tdev) if place else shard.splice(output, part, key0, off0, tdev)
            if not ok:
                # an incomplete output looks like a database: every rank removes what it left, rank 0 the output itself
                shard.remove_rank_files(output, rank)
                dist.barrier()
                if rank == 0:
                    shard.remove_db(output)
                    print(f"[Error] joining the ranks' databases failed: {output} was not written", file=sys.stderr)
                rc = 1
            else:
                rc = 0
        t_done = time.perf_counter()
        if rank == 0 and getattr(a, "json_stats", False):
            res = sum(r_[3] for r_ in rows)
            eng_wall = max(r_[4] for r_ in rows) / 1e6
            steady = max((r_[4] - r_[5]) for r_ in rows) / 1e6
            print(json.dumps({"mode": a.mode, "world": world, "backend": backend, "engine": "host/foldcomp-hip --shard R/N" + (" --place" if place else ""),
                              "host_threads_per_rank": (a.threads if a.threads and a.threads > 1 else host_threads(world)), "cpus_of_this_process": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count(),
                              "items": sum(r_[7] for r_ in rows), "records": sum(r_[0] for r_ in rows), "data_bytes": sum(r_[1] for r_ in rows),
                              "residues": res, "input_bytes": sum(r_[8] for r_ in rows),
                              "records_per_rank": [r_[0] for r_ in rows], "bytes_per_rank": [r_[1] for r_ in rows],
                              "engine_max_rss_kb_per_rank": [r_[6] for r_ in rows],
                              "data_written_once": bool(place or world == 1),
                              "sizes_pass_s_max": round(max(r_[9] for r_ in rows) / 1e6, 4) if place else None,
                              "counts_exchanged_s_after_spawn": round(t_sizes - t_spawn, 4) if place else None,
                              "wall_s": round(t_done - t_start, 4), "torch_and_group_s_beside_engine": round(t_group - t_spawn, 4), "communicator_s_beside_engine": round(t_comm - t_group, 4),
                              "engine_s": round(t_engine - t_spawn, 4), "engine_wall_s_max": round(eng_wall, 4),
                              "engine_steady_s_max": round(steady, 4), "exchange_and_splice_s": round(t_done - t_engine, 4),
                              # ... of which: rank 0 waiting in the all_gather for the slowest rank's engine, and the file work after it
                              "exchange_wait_for_slowest_rank_s": round(t_exchanged - t_engine, 4), "file_work_after_exchange_s": round(t_done - t_exchanged, 4),
                              "residues_per_s": round(res / (t_done - t_start), 1) if t_done > t_start else None,
                              # the steady rate: without the group's and the engines' start-up (HIP context), WITH the exchange
                              "steady_residues_per_s": round(res / (steady + (t_done - t_engine)), 1) if steady + (t_done - t_engine) > 0 else None}))
    finally:
        # whatever went wrong above (rendezvous, a collective, a broken pipe): the engine does not outlive this process
        if proc.poll() is None:
            proc.kill()
        try:
            proc.wait(timeout=30)
        except Exception:
            pass
        if joined:
            import torch.distributed as dist
            dist.destroy_process_group()
    return rc
