#!/usr/bin/env python3
"""A DRAM-side look at the codec kernels' memory traffic.

rocprofv3's FETCH_SIZE / WRITE_SIZE (and the TCC_EA0_*_DRAM* counters gfx950 offers) count requests at the L2's fabric side: hits
in the 256 MB Infinity Cache are included, so a kernel whose working set lives in that cache (k_backbone's per-wavefront ring)
looks as if it moved those bytes through HBM. The only memory-controller-side signal this stack exposes is the driver's
`mem_busy_percent` (amdgpu sysfs: the SMU's activity level of the memory controllers). This tool turns it into bytes:

  1. calibration: device copies at full rate and at duty cycles of it (copy, then a spin kernel of matching length), the achieved
     bytes/s from HIP events against the median mem_busy_percent -> bytes/s per percent (and how linear that is);
  2. probes: a loop of ONE codec stage at the headline batch (FCZ_PROFILE_STAGES leaves the other decompress stages out; compress
     as a whole) for a few seconds, median mem_busy_percent -> HBM bytes/s -> bytes per residue of that stage.

It is a coarse instrument (integer percent, firmware-averaged) and says so in its output; it complements the counters, it does not
replace them. The stage mask is only read by a library built with -DFCZ_PROFILING (the product library ignores the variable):
  hipcc <flags of foldcomp_amd/csrc/Makefile> -DFCZ_PROFILING -shared -o build/libfcz_prof.so foldcomp_amd/csrc/fcz_abi.hip
usage (GPU box): FCZ_HIP_LIB=$PWD/build/libfcz_prof.so python tools/hbm_busy_probe.py [--chains 1000000] [--seconds 3] [--probes k_backbone] > gpurun_out/hbm_busy.json
"""
from __future__ import annotations

import argparse
import ctypes
import glob
import json
import os
import statistics
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def find_busy_file():
    """the card of THIS process's GPU: a node shows every GPU of the host in sysfs, only one belongs to the container -- the one
    whose memory gets busy while a device copy runs here"""
    import torch
    files = []
    for f in sorted(glob.glob("/sys/class/drm/card*/device/mem_busy_percent")):
        try:
            int(open(f).read()); files.append(f)
        except (OSError, ValueError):
            continue
    if len(files) <= 1:
        return files[0] if files else None
    a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda:0"); b = torch.empty_like(a)
    acc = {f: [] for f in files}
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.5:
        for _ in range(8):
            b.copy_(a)
        for f in files:
            try:
                acc[f].append(int(open(f).read()))
            except (OSError, ValueError):
                pass
    torch.cuda.synchronize()
    del a, b
    best = max(files, key=lambda f: sum(acc[f][len(acc[f]) // 2:]) / max(len(acc[f]) // 2, 1))
    return best


class Sampler:
    def __init__(self, path, period=0.002):
        self.path, self.period, self.samples, self._stop = path, period, [], False
        self.t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop:
            try:
                with open(self.path) as fh:
                    self.samples.append((time.perf_counter(), int(fh.read())))
            except (OSError, ValueError):
                pass
            time.sleep(self.period)

    def __enter__(self):
        self.t.start(); return self

    def __exit__(self, *a):
        self._stop = True; self.t.join()

    def window(self, t0, t1):
        v = [b for t, b in self.samples if t0 <= t <= t1]
        return v


def summarise(v):
    if not v:
        return {"samples": 0}
    return {"samples": len(v), "median": statistics.median(v), "mean": round(statistics.fmean(v), 2), "min": min(v), "max": max(v)}


def stage_child(args):
    """runs in a child process (FCZ_PROFILE_STAGES is process-wide): loop one stage, print its window and rate"""
    import torch
    import bench
    from foldcomp_amd.codec import Codec
    dev = "cuda:0"
    d = bench.generate_resident(args.chains, 350, 25, 32768, dev, seed_base=0)
    codec = Codec(0)
    w = bench.Workload(codec, d, dev)
    w.compress(); w.decompress(); codec.synchronize()          # every stage once: the hand-over arrays of the loop below are valid
    os.environ["FCZ_PROFILE_STAGES"] = args.mask               # (the library reads it at every decompress batch call)
    fn = {"compress": w.compress, "decompress": w.decompress}[args.stage_fn]
    fn(); codec.synchronize()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < args.seconds:
        for _ in range(4):
            fn()
        codec.synchronize(); n += 4
    t1 = time.perf_counter()
    print(json.dumps({"t0": t0, "t1": t1, "calls": n, "ms_per_call": (t1 - t0) / n * 1e3, "residues": w.R, "atoms": w.M, "fcz_bytes": w.fcz_bytes}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=1_000_000)
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--stage-fn", default=None)
    ap.add_argument("--mask", default="7")
    ap.add_argument("--probes", default="", help="comma-separated subset of the probes (default: all)")
    args = ap.parse_args()
    if args.stage_fn:
        return stage_child(args)
    path = find_busy_file()
    out = {"sysfs": path, "what": "amdgpu mem_busy_percent (memory-controller activity level reported by the SMU), sampled every 2 ms"}
    if path is None:
        out["failed"] = "no readable /sys/class/drm/card*/device/mem_busy_percent on this box"
        print(json.dumps(out)); return
    import torch
    dev = "cuda:0"
    with Sampler(path) as S:
        time.sleep(1.0)
        t_idle = (time.perf_counter() - 0.8, time.perf_counter())
        # ---- calibration: copies (read + write bytes) at duty cycles ----
        nbytes = 1 << 30
        a = torch.empty(nbytes, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
        b.copy_(a); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); b.copy_(a); e1.record(); torch.cuda.synchronize()
        copy_ms = e0.elapsed_time(e1)
        clock_khz = 2_400_000          # torch.cuda._sleep counts device clock cycles (approximately: the duty cycles are measured, not assumed)
        cal = []
        for duty in (1.0, 0.5, 0.25):
            spin_cycles = int(copy_ms * 1e-3 * clock_khz * 1e3 * (1.0 / duty - 1.0))
            torch.cuda.synchronize()
            t0 = time.perf_counter(); n = 0
            while time.perf_counter() - t0 < args.seconds:
                for _ in range(8):
                    b.copy_(a)
                    if spin_cycles:
                        torch.cuda._sleep(spin_cycles)
                torch.cuda.synchronize(); n += 8
            t1 = time.perf_counter()
            rate = 2.0 * nbytes * n / (t1 - t0)
            time.sleep(0.3)
            cal.append({"duty_requested": duty, "bytes_per_s": rate, **summarise(S.window(t0 + 0.3, t1 - 0.1))})
        del a, b
        torch.cuda.empty_cache()
        out["idle"] = summarise(S.window(*t_idle))
        out["calibration"] = cal
        full = cal[0]
        per_pct = full["bytes_per_s"] / max(full.get("median", 0) or 1, 1)
        out["bytes_per_s_per_percent"] = per_pct
        out["linearity"] = [round((c["bytes_per_s"] / per_pct) / max(c.get("median", 0) or 1, 1), 3) for c in cal]
        # ---- probes: one stage per child process ----
        probes = {}
        for name, fnname, mask in (("k_backbone", "decompress", "1"), ("k_res_index", "decompress", "2"), ("k_sidechain", "decompress", "4"),
                                   ("decompress_all", "decompress", "7"), ("compress_all", "compress", "7")):
            if args.probes and name not in args.probes.split(","):
                continue
            env = {k: v for k, v in os.environ.items() if k != "FCZ_PROFILE_STAGES"}
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--stage-fn", fnname, "--mask", mask, "--chains", str(args.chains), "--seconds", str(args.seconds)],
                               env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                probes[name] = {"failed": (r.stderr or r.stdout)[-300:]}; continue
            st = json.loads(line[-1])
            busy = summarise(S.window(st["t0"] + 0.3, st["t1"] - 0.1))
            rate = per_pct * (busy.get("median", 0) or 0)
            probes[name] = {"ms_per_call": round(st["ms_per_call"], 3), "calls": st["calls"], "mem_busy_percent": busy,
                            "hbm_bytes_per_s_estimate": rate, "hbm_bytes_per_call_estimate": rate * st["ms_per_call"] * 1e-3,
                            "hbm_bytes_per_residue_estimate": round(rate * st["ms_per_call"] * 1e-3 / st["residues"], 1),
                            "note": "decompress calls include the sizes pass (~0.9 ms of residue-word reads) beside the named stage" if fnname == "decompress" and mask != "7" else None}
            time.sleep(0.5)
        out["probes"] = probes
    out["caveat"] = ("mem_busy_percent is an integer activity level averaged by firmware, not a byte counter: the estimates carry the calibration's "
                     "linearity error and +-1 percent (~1 % of the copy rate) of resolution")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
