#!/usr/bin/env python3
"""bench.py -- residues/s of the FCZ hot path (compress + decompress) on synthetic 350-residue chains.

    python bench.py --gpus N --steps K --warmup W            (N=1: plain python; N>1: one rank per GPU
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   under torchrun)

One "step" = one pass of the hot path over the rank's resident batch: compress every chain to FCZ, then
decompress every FCZ record back to SoA atoms (round trip). Workload at N=1 = BASELINE.json configs[1]:
1 000 000 synthetic 350-residue chains on one MI355X (inputs resident in HBM before the clock starts).
N>1: every rank owns its own 1M-chain shard (weak scaling); the data path has no collective, the ranks
only exchange the output index (per-record lengths -> global offsets) over RCCL after compressing.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (dominant kernel,
HIP-event time measured here) and `cpu_baseline` (the reference's own CPU path, oracle/_ref, timed on a
bounded sample of the same chains on this host).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from foldcomp_amd import _lib, synthetic  # noqa: E402
from foldcomp_amd.codec import Codec  # noqa: E402
from foldcomp_amd.structure import CAtomsOut, CChainBatch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--chains", type=int, default=1_000_000, help="chains per GPU")
    ap.add_argument("--residues", type=int, default=350)
    ap.add_argument("--anchor", type=int, default=25)
    ap.add_argument("--gen-chunk", type=int, default=32768)
    ap.add_argument("--cpu-sample", type=int, default=32768, help="chains timed on the CPU baseline (0 = skip)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--parity-chains", type=int, default=65536,
                    help="chains of the GPU result compared with the oracle bit for bit after the timed region (FCZ bytes and coordinates)")
    ap.add_argument("--seed-base", type=int, default=0,
                    help="first chain id of rank 0's batch (chain c of rank r is seeded with seed-base + r x chains + c): another value = another batch")
    ap.add_argument("--parity-chunk", type=int, default=65536, help="chains per oracle call of the parity check (bounds its host memory)")
    ap.add_argument("--pdb-sample", type=int, default=65536,
                    help="chains rendered to PDB text on the device after the timed region (SURVEY §8 f2 leg; 0 = skip)")
    ap.add_argument("--mixed-chains", type=int, default=542_000,
                    help="chains per GPU of the secondary legs (decompress_only = BASELINE configs[2] shape, mixed = configs[4] shape: "
                         "log-normal lengths, anchor -b 25); 0 = skip")
    ap.add_argument("--mixed-steps", type=int, default=3)
    ap.add_argument("--e2e-files", type=int, default=8192,
                    help="PDB files of the end_to_end leg (disk -> FCZ database through host/foldcomp-hip, N=1 only; 0 = skip)")
    ap.add_argument("--e2e-passes", type=int, default=24,
                    help="times every end_to_end run of the GPU host walks the files (steady state: 24 x 8192 = 196 608 file reads per run; "
                         "the reference's loops, which have no start-up, walk them min(passes, 4) times)")
    ap.add_argument("--e2e-workers", type=int, default=0, help="--workers-per-gpu of the end_to_end compress runs (0 = the host's default, 2)")
    ap.add_argument("--host-chains", type=int, default=65536,
                    help="chains pushed through the host-pointer entry points for the PCIe-inclusive rate (0 = skip)")
    ap.add_argument("--numerics", choices=("exact", "fast"), default="exact",
                    help="decompress numerics of the TIMED steps: exact = float32 coordinates bit-identical to the reference (the headline), "
                         "fast = FCZ_NUMERICS_FAST (plain float arithmetic, parallel backbone). The other mode is always measured "
                         "once after the timed region and reported as `alt_numerics`.")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch/rendezvous check only: ranks are started and meet in a process group (gloo when no GPU is "
                         "present), no codec call is made and the JSON line carries value null; used by the CPU tests")
    ap.add_argument("--mixed", action="store_true",
                    help="BASELINE configs[4] stand-in: log-normal chain lengths (mu = ln 250, sigma = 0.6, clipped to [16, 2700]) "
                         "instead of the fixed --residues; not the headline workload")
    return ap.parse_args()


def generate_resident(n_chains, n_res, anchor, chunk, device, seed_base, mixed=False):
    """build the rank's batch on the GPU chunk by chunk -> dict of device tensors (fcz_chain_batch layout)"""
    parts = []
    done = 0
    if mixed:
        # The generator is dense in [chains, longest chain of the chunk]: chains are drawn with log-normal lengths, dealt into
        # chunks of similar length (so that a chunk costs its own residues, not 2 700 per chain) and the chunks are laid out in
        # random order: lengths vary freely across the batch, stretches of up to 65 536 chains are of similar length.
        lens_all = np.sort(synthetic.mixed_lengths(n_chains, seed=seed_base + 7))
        bounds = [0]                              # greedy groups of similar length: chains x longest chain <= 6 M residue slots
        while bounds[-1] < n_chains:
            a = bounds[-1]; b = min(n_chains, a + 65536)
            while b - a > 1 and (b - a) * int(lens_all[b - 1]) > 6_000_000:
                b = a + max(1, (b - a) // 2)
            bounds.append(b)
        order = np.random.default_rng(seed_base + 11).permutation(len(bounds) - 1)
        for ci in order:
            lens = lens_all[bounds[ci]:bounds[ci + 1]]
            parts.append(synthetic.generate(len(lens), lens, seed=0xF01DC0DE, device=device, anchor_threshold=anchor,
                                            first_chain_id=seed_base + done))
            done += len(lens)
    while done < n_chains:
        c = min(chunk, n_chains - done)
        parts.append(synthetic.generate(c, n_res, seed=0xF01DC0DE, device=device, anchor_threshold=anchor,
                                        first_chain_id=seed_base + done))
        done += c
    if len(parts) == 1:
        return parts[0]
    out = {"anchor_threshold": anchor}
    for k in ("x", "y", "z", "atom_code", "res_code", "bfac_ca", "first_res_index", "first_atom_index", "chain_id", "titles"):
        out[k] = torch.cat([p[k] for p in parts])
    for k, unit in (("res_off", "res_off"), ("atom_off", "atom_off"), ("title_off", "title_off")):
        acc = []; base = 0
        for p in parts:
            v = p[k].to(torch.int64)
            acc.append(v[:-1] + base); base += int(v[-1])
        if base >= 1 << 32:
            # the ABI counts residues and atoms in uint32 (carried in torch int32 here): refuse instead of wrapping
            raise SystemExit(f"bench.py: {k} total {base} does not fit the uint32 counts of fcz_chain_batch; use fewer --chains per GPU")
        acc.append(torch.tensor([base], dtype=torch.int64, device=device))
        out[k] = torch.cat(acc).to(torch.int32)
    return out


def c_batch(d) -> CChainBatch:
    s = CChainBatch()
    s.n_chains = d["res_off"].numel() - 1
    s.n_residues = int(d["res_off"][-1]) & 0xFFFFFFFF
    s.n_atoms = int(d["atom_off"][-1]) & 0xFFFFFFFF      # uint32 carried in a torch int32
    s.anchor_threshold = int(d["anchor_threshold"])
    for k in ("res_off", "atom_off", "x", "y", "z", "atom_code", "res_code", "bfac_ca", "first_res_index", "first_atom_index",
              "chain_id", "titles", "title_off"):
        setattr(s, k, d[k].data_ptr())
    return s


def _u32(t):
    return int(t) & 0xFFFFFFFF            # the ABI's uint32 counts are carried in torch int32


def host_slice(d, c0, c1):
    """chains [c0, c1) of the device batch as a host ChainBatch (offsets rebased to the slice)"""
    C = d["res_off"].numel() - 1
    c0 = max(0, min(c0, C)); c1 = max(c0, min(c1, C))
    r0, r1 = _u32(d["res_off"][c0]), _u32(d["res_off"][c1])
    a0, a1 = _u32(d["atom_off"][r0]), _u32(d["atom_off"][r1])
    t0, t1 = _u32(d["title_off"][c0]), _u32(d["title_off"][c1])
    sub = {k: d[k][a0:a1] for k in ("x", "y", "z", "atom_code")}
    sub.update({k: d[k][r0:r1] for k in ("res_code", "bfac_ca")})
    sub.update({k: d[k][c0:c1] for k in ("first_res_index", "first_atom_index", "chain_id")})
    sub["res_off"] = (d["res_off"][c0:c1 + 1].to(torch.int64) & 0xFFFFFFFF) - r0
    sub["atom_off"] = (d["atom_off"][r0:r1 + 1].to(torch.int64) & 0xFFFFFFFF) - a0
    sub["titles"] = d["titles"][t0:t1]
    sub["title_off"] = (d["title_off"][c0:c1 + 1].to(torch.int64) & 0xFFFFFFFF) - t0
    sub["anchor_threshold"] = d["anchor_threshold"]
    return synthetic.to_chain_batch(sub)


def host_sample(d, n_sample):
    """first n_sample chains of the device batch as a host ChainBatch"""
    return host_slice(d, 0, n_sample)


def effective_cores():
    """CPUs this process may actually use: the scheduler affinity, cut by the cgroup's CFS quota when there is one
    (cpu.max 'quota period'). The GPU box shows 256 hardware threads under a 16-CPU quota: more runnable threads than the
    quota only thrash (both this repo's host code and the reference's OpenMP loop get slower beyond it)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0]); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / p + 0.5)))
            break
        except Exception:
            continue
    return n


def cpu_baseline(hb, anchor, gpu=None):
    """the reference's CPU path on this host (oracle/_ref, OpenMP over chains like `foldcomp -t`);
    falls back to the C port (oracle/) only as a *baseline*, never as part of the product path.
    `gpu` (the product's results for the same chains: records, offsets, decoded arrays): the LIVE reference's records and decoded
    atoms are compared with them chain by chain (64-bit FNV-1a of the pad-masked record and of the x / y / z / B-factor bit patterns,
    computed by the checker library for both sides) -> `live_reference` in the returned dict."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _harness as H
    from foldcomp_amd._aa_tables import ATOM_NAMES, RES3
    cores = effective_cores()
    R = hb.n_residues
    sample = f"first {hb.n_chains} chains of the GPU workload ({R} residues), codec only (objects in, FCZ, objects out)"
    if H.have_ref():
        lib = H.load_ref()
        an = np.zeros((37, 4), np.uint8); rn = np.zeros((24, 4), np.uint8)
        for i, n in enumerate(ATOM_NAMES): an[i, :len(n)] = np.frombuffer(n.encode(), np.uint8)
        for i, n in enumerate(RES3): rn[i, :3] = np.frombuffer(n.encode(), np.uint8)
        lib.ref_bench_roundtrip.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 10 + [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 11
        tc = ctypes.c_double(); td = ctypes.c_double(); fb = ctypes.c_ulonglong(); ao = ctypes.c_ulonglong()
        n = hb.n_chains
        h_fcz = np.zeros(n, np.uint64); h_xyz = np.zeros(n, np.uint64)
        titles = np.ascontiguousarray(hb.titles, dtype=np.uint8); toff = np.ascontiguousarray(hb.title_off, dtype=np.uint32)
        fres = np.ascontiguousarray(hb.first_res_index, dtype=np.int32); fatom = np.ascontiguousarray(hb.first_atom_index, dtype=np.int32)
        cid = np.ascontiguousarray(hb.chain_id, dtype=np.uint8)
        fail = lib.ref_bench_roundtrip(n, hb.res_off.ctypes.data, hb.atom_off.ctypes.data, hb.x.ctypes.data, hb.y.ctypes.data,
                                       hb.z.ctypes.data, hb.atom_code.ctypes.data, hb.res_code.ctypes.data, hb.bfac_ca.ctypes.data,
                                       an.ctypes.data, rn.ctypes.data, anchor, cores, ctypes.byref(tc), ctypes.byref(td),
                                       ctypes.byref(fb), ctypes.byref(ao), titles.ctypes.data, toff.ctypes.data, fres.ctypes.data, fatom.ctypes.data,
                                       cid.ctypes.data, h_fcz.ctypes.data, h_xyz.ctypes.data)
        out = {"value": R / (tc.value + td.value), "unit": "residues/s", "cores": cores, "hardware_threads": os.cpu_count(), "kind": "reference", "sample": sample,
               "compress_residues_per_s": R / tc.value, "decompress_residues_per_s": R / td.value, "failed_chains": int(fail)}
        if gpu is not None:
            # the product's records and decoded atoms of the same chains, hashed by the same checker functions
            lib.ref_hash_records.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
            lib.ref_hash_atoms.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_long, ctypes.c_void_p]
            g_fcz = np.zeros(n, np.uint64); g_xyz = np.zeros(n, np.uint64)
            lib.ref_hash_records(gpu["blob"].ctypes.data, gpu["off"].ctypes.data, n, g_fcz.ctypes.data)
            lib.ref_hash_atoms(gpu["x"].ctypes.data, gpu["y"].ctypes.data, gpu["z"].ctypes.data, gpu["atom_off"].ctypes.data,
                               gpu["bfac_res"].ctypes.data, gpu["res_off"].ctypes.data, n, g_xyz.ctypes.data)
            ne_f = np.nonzero(g_fcz != h_fcz)[0]; ne_x = np.nonzero(g_xyz != h_xyz)[0]
            out["live_reference"] = {"chains": int(n), "records_equal": int(n - len(ne_f)), "coords_equal": int(n - len(ne_x)),
                                     "first_record_mismatch_chain": int(ne_f[0]) if len(ne_f) else None,
                                     "first_coords_mismatch_chain": int(ne_x[0]) if len(ne_x) else None,
                                     "digest_records": f"{int(np.bitwise_xor.reduce(h_fcz)):016x}", "digest_coords": f"{int(np.bitwise_xor.reduce(h_xyz)):016x}",
                                     "what": "Foldcomp::compress + writeStream / read + decompress of oracle/_ref on these chains (own titles, numbering, chain ids) "
                                             "against the GPU's records (header bytes 14, 15, 22, 23 masked) and decoded x / y / z / B-factor bits, chain by chain"}
        return out
    t0 = time.perf_counter(); blob, off, st = H.oracle_compress(hb, n_threads=cores)
    t1 = time.perf_counter(); H.oracle_decompress(blob, off, n_threads=cores); t2 = time.perf_counter()
    return {"value": R / (t2 - t0), "unit": "residues/s", "cores": cores, "kind": "port", "sample": sample,
            "compress_residues_per_s": R / (t1 - t0), "decompress_residues_per_s": R / (t2 - t1)}


KERNEL_SOURCES = ("fcz_kernels.h", "fcz_math.h", "fcz_compress.h", "fcz_sidechain.h", "aa_tables.inc")   # the codec kernels' device code (fcz_abi.hip holds their launches among much host code: not hashed)


def _code_only(text: bytes) -> bytes:
    """a kernel source without what the compiler does not see of it: full-line and trailing // comments (not on lines that hold a
    string literal), trailing blanks, empty lines"""
    out = []
    for line in text.split(b"\n"):
        if b'"' not in line:
            k = line.find(b"//")
            if k >= 0:
                line = line[:k]
        line = line.rstrip()
        if line:
            out.append(line)
    return b"\n".join(out)


def csrc_sha16():
    """what the codec kernels are compiled from, hashed: a counter profile describes the kernels of ONE state of these files (their
    code: a reworded comment does not make a profile stale, a changed token does)"""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "foldcomp_amd", "csrc", f), "rb") as fh:
            h.update(f.encode()); h.update(_code_only(fh.read()))
    return h.hexdigest()[:16]


def traffic_profile():
    """profiles/traffic.json if it describes the kernels this run timed (the hash of their sources, written when the passes were
    collected by tools/pmc_summary.py, equals the tree's) -> (dict, None); else (None, why): a stale profile yields `traffic: null`,
    never a number of other kernels"""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            t = json.load(fh)
    except (OSError, ValueError):
        return None, "profiles/traffic.json missing"
    now = csrc_sha16()
    if t.get("csrc_sha16") != now:
        return None, f"profiles/traffic.json describes csrc {t.get('csrc_sha16')}, this tree is {now}: counter passes have to be re-collected (tools/profile_gpu.sh)"
    return t, None


def measured_traffic(kernel, n_residues, n_res_per_chain, section="kernels"):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/traffic.json, written by
    tools/pmc_summary.py from separate `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` runs of this same workload,
    gfx950 corrections applied). PMC collection serialises dispatches and cannot run inside the timed region, so
    the per-residue figure measured there is scaled to this launch; null when no profile of THESE kernels (source hash) and
    this workload shape exists. section: "kernels" (uniform chains of residues_per_chain) or "mixed" (the mixed-length generator)."""
    t, why = traffic_profile()
    if t is None:
        return None, why
    try:
        sec = t if section == "kernels" else t[section]
        k = sec["kernels"][kernel]
        if section == "kernels" and int(t["residues_per_chain"]) != int(n_res_per_chain):
            return None, f"profiles/traffic.json holds {t['residues_per_chain']}-residue chains"
        return (k["fetch_bytes_per_residue"] + k["write_bytes_per_residue"]) * n_residues, sec.get("source")
    except (KeyError, ValueError):
        return None, f"no counter pass of {kernel} ({section}) in profiles/traffic.json"


def controller_side_traffic(kernel, n_residues, n_res_per_chain):
    """the same kernel's bytes per launch as the MEMORY CONTROLLERS saw them (profiles/traffic.json `controller_side`, from
    tools/hbm_busy_probe.py: the driver's mem_busy_percent during a seconds-long loop of that one kernel, calibrated on device
    copies); (bytes, fabric-side bytes of the same scope, source) or (None, None, None)"""
    try:
        t, _why = traffic_profile()
        if t is None or int(t["residues_per_chain"]) != int(n_res_per_chain):
            return None, None, None
        c = t["controller_side"]
        return c["bytes_per_residue"][kernel] * n_residues, c["fabric_side_counters_same_scope"][kernel] * n_residues, c["source"]
    except (OSError, KeyError, ValueError):
        return None, None, None


def copy_ceiling_f4(codec, nbytes=1 << 30, reps=10):
    """device copy by the kernel MI355X_MICROARCH.md quotes its 6.29 TB/s with (float4 per lane, grid-stride, a persistent grid):
    read + write bytes per second, timed by HIP events on the ctx stream (fcz_selftest_copy)"""
    gbs = ctypes.c_double(0.0)
    rc = codec.lib.fcz_selftest_copy(codec.ctx, ctypes.c_uint64(nbytes), int(reps), ctypes.byref(gbs))
    return gbs.value if rc == 0 else None


def copy_ceiling(dev, nbytes=1 << 30, reps=5):
    """torch's device-to-device copy_ (rounds 1-5's calibration figure; kept beside the float4 kernel's): read + write bytes"""
    a = torch.empty(nbytes, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
    b.copy_(a); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del a, b
    return 2 * nbytes / (ms * 1e-3) / 1e9


def parity_check(d, w, n, chunk=65536, threads=None):
    """GPU results of the rank's first n chains against the oracle, bit for bit (checker only, outside the timed region):
    FCZ offsets and bytes of `w.blob_dev`, then x / y / z / per-residue B-factors of `w.out_t` (the default atom order) against the
    oracle's decode of ITS OWN records. In chunks of `chunk` chains, so that the host never holds more than one chunk of atoms
    (the full 1 M-chain batch is 38 GB in, 36 GB out). -> dict with the two flags and the first differing chain of each."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _harness as H
    threads = threads or effective_cores()
    n = min(n, w.C)
    ok_c = ok_d = True
    first_c = first_d = None
    for c0 in range(0, n, max(1, chunk)):
        c1 = min(n, c0 + max(1, chunk))
        hb = host_slice(d, c0, c1)
        oblob, ooff, ost = H.oracle_compress(hb, n_threads=threads)
        goff = w.off_dev[c0:c1 + 1].cpu().numpy().astype(np.int64)
        b0, b1 = int(goff[0]), int(goff[-1])
        got = w.blob_dev[b0:b1].cpu().numpy()
        same = np.array_equal(goff - b0, ooff.astype(np.int64)) and got.tobytes() == oblob.tobytes()
        if not same and first_c is None:
            k = 0
            if np.array_equal(goff - b0, ooff.astype(np.int64)):
                diff = np.nonzero(got != np.asarray(oblob))[0]
                k = int(np.searchsorted(ooff, diff[0], side="right") - 1) if len(diff) else 0
            else:
                k = int(np.nonzero((goff - b0) != ooff.astype(np.int64))[0][0]) - 1
            first_c = c0 + max(k, 0)
        ok_c = ok_c and bool(same)
        o = H.oracle_decompress(oblob, ooff, n_threads=threads)
        a0, a1 = _u32(w.atom_off_dev[c0]), _u32(w.atom_off_dev[c1])
        r0, r1 = _u32(w.res_off_dev[c0]), _u32(w.res_off_dev[c1])
        good = (a1 - a0) == int(o["atom_off"][-1]) and (r1 - r0) == int(o["res_off"][-1])
        bad_atom = None
        if good:
            for k_ in ("x", "y", "z"):
                g = w.out_t[k_][a0:a1].cpu().numpy().view(np.uint32)
                ne = np.nonzero(g != o[k_].view(np.uint32))[0]
                if len(ne):
                    good = False; bad_atom = int(ne[0]) if bad_atom is None else min(bad_atom, int(ne[0]))
            g = w.out_t["bfac_res"][r0:r1].cpu().numpy().view(np.uint32)
            if not np.array_equal(g, o["bfac_res"].view(np.uint32)):
                good = False
        if not good and first_d is None:
            first_d = c0 + (int(np.searchsorted(o["atom_off"], bad_atom, side="right") - 1) if bad_atom is not None else 0)
        ok_d = ok_d and bool(good)
        del hb, oblob, o, got
    return {"chains_checked": n, "fcz_bit_exact": bool(ok_c), "coords_bit_exact": bool(ok_d),
            "first_fcz_mismatch_chain": first_c, "first_coords_mismatch_chain": first_d}


class Workload:
    """one rank's resident batch plus every output buffer of the round trip; compress() / decompress() only enqueue on the
    codec's stream (device pointers in, device pointers out)"""

    def __init__(self, codec, d, dev):
        self.codec, self.lib, self.d, self.dev = codec, codec.lib, d, dev
        self.C = d["res_off"].numel() - 1
        self.R, self.M = int(d["res_off"][-1]) & 0xFFFFFFFF, int(d["atom_off"][-1]) & 0xFFFFFFFF
        self.cb = c_batch(d)
        C, R, M = self.C, self.R, self.M
        self.off_dev = torch.zeros(C + 1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        _lib.check(self.lib.fcz_compress_sizes_dev(codec.ctx, ctypes.byref(self.cb), self.off_dev.data_ptr()), "sizes")
        codec.synchronize()
        self.fcz_bytes = int(self.off_dev[-1])          # exact: Foldcomp::getSize on the device
        self.blob_dev = torch.zeros(self.fcz_bytes, dtype=torch.uint8, device=dev)
        self.status_dev = torch.zeros(C, dtype=torch.int32, device=dev)
        self.res_off_dev = torch.zeros(C + 1, dtype=torch.int32, device=dev)
        self.atom_off_dev = torch.zeros(C + 1, dtype=torch.int32, device=dev)
        self.out_t = {k: torch.zeros(M, dtype=torch.float32, device=dev) for k in ("x", "y", "z")}
        self.out_t["bfac_res"] = torch.zeros(R, dtype=torch.float32, device=dev)
        self.out_t["res_code"] = torch.zeros(R, dtype=torch.uint8, device=dev)
        o = self.out_t
        self.cout = CAtomsOut(o["x"].data_ptr(), o["y"].data_ptr(), o["z"].data_ptr(), o["bfac_res"].data_ptr(), o["res_code"].data_ptr(), None)
        torch.cuda.synchronize()

    def compress(self):
        """SoA atoms -> FCZ records"""
        _lib.check(self.lib.fcz_compress_sizes_dev(self.codec.ctx, ctypes.byref(self.cb), self.off_dev.data_ptr()), "sizes")
        _lib.check(self.lib.fcz_compress_batch_dev(self.codec.ctx, ctypes.byref(self.cb), self.off_dev.data_ptr(), self.blob_dev.data_ptr(),
                                                   self.status_dev.data_ptr()), "compress")

    def decompress(self, alt_order=0):
        """FCZ records -> SoA atoms"""
        tr = ctypes.c_uint32(); ta = ctypes.c_uint32()
        _lib.check(self.lib.fcz_decompress_sizes_dev(self.codec.ctx, self.blob_dev.data_ptr(), self.off_dev.data_ptr(), self.C,
                                                     self.res_off_dev.data_ptr(), self.atom_off_dev.data_ptr(), ctypes.byref(tr), ctypes.byref(ta)), "dsizes")
        assert tr.value == self.R and ta.value == self.M, (tr.value, self.R, ta.value, self.M)
        _lib.check(self.lib.fcz_decompress_batch_dev(self.codec.ctx, self.blob_dev.data_ptr(), self.off_dev.data_ptr(), self.C,
                                                     self.res_off_dev.data_ptr(), self.atom_off_dev.data_ptr(), alt_order, ctypes.byref(self.cout)), "decompress")

    def checksum(self):
        nb8 = (self.fcz_bytes // 8) * 8
        return int(self.blob_dev[:nb8].view(torch.int64).sum()) if nb8 else 0

    def round_trip_deviation(self):
        """decode(encode(x)) against x over the whole batch, atoms in the input's order (`-a`): (rmsd, max deviation) in A"""
        self.decompress(alt_order=1)
        self.codec.synchronize()
        sq = 0.0; mx = 0.0; step_n = 1 << 28
        for a in range(0, self.M, step_n):
            b_ = min(self.M, a + step_n)
            dsq = (self.out_t["x"][a:b_] - self.d["x"][a:b_]) ** 2
            dsq += (self.out_t["y"][a:b_] - self.d["y"][a:b_]) ** 2
            dsq += (self.out_t["z"][a:b_] - self.d["z"][a:b_]) ** 2
            sq += float(dsq.sum(dtype=torch.float64)); mx = max(mx, float(dsq.max()))
            del dsq
        return (sq / self.M) ** 0.5, mx ** 0.5

    def kernel_bytes(self):
        """algorithmic bytes per launch of each kernel: the compress side reads
        13 B per atom + 9 B per residue and writes the record, the decompress side reads the record and writes 12 B per atom + 4 B
        per residue; hand-over arrays between kernels (angles, blended backbone, per-residue index) are not algorithmic traffic"""
        A = self.M / self.R; f = self.fcz_bytes / self.R; R = self.R
        sc = (A - 3.0) + 1.0                          # side-chain torsion bytes + the B-factor byte, per residue
        return {"k_compress_angles_w": (13 * A + (A - 3.0)) * R, "k_compress_index": 1.0 * R, "k_compress_pack": (9 + f - (A - 3.0)) * R,
                "k_backbone": (f - sc) * R, "k_res_index": (sc + 4) * R, "k_sidechain": 12 * A * R}


KERNEL_SPANS = {"k_compress_angles_w": "compress_angles", "k_compress_index": "compress_index", "k_compress_pack": "compress_pack",
                "k_backbone": "decompress_backbone", "k_res_index": "decompress_index", "k_sidechain": "decompress_sidechain"}
ALL_SPANS = ("compress_sizes", "compress_index", "compress_angles", "compress_pack", "decompress_sizes", "decompress_backbone",
             "decompress_index", "decompress_sidechain")


def span_ms(codec):
    out = {}
    for name in ALL_SPANS:
        ms, n = codec.kernel_time(name)
        out[name] = (ms / n) if n else 0.0
    return out


def cif_from_pdb_text(pdb: bytes, entry_id: str) -> bytes:
    """the ATOM records of a PDB text as an mmCIF file of the shape AFDB ships (one data_ block, _entry.id, an _atom_site loop of
    21 columns, one row per line): input of the end_to_end mmCIF leg"""
    cols = ["group_PDB", "id", "type_symbol", "label_atom_id", "label_alt_id", "label_comp_id", "label_asym_id", "label_entity_id", "label_seq_id",
            "pdbx_PDB_ins_code", "Cartn_x", "Cartn_y", "Cartn_z", "occupancy", "B_iso_or_equiv", "pdbx_formal_charge", "auth_seq_id", "auth_comp_id",
            "auth_asym_id", "auth_atom_id", "pdbx_PDB_model_num"]
    rows, seqres = [], []
    for l in pdb.decode("latin-1").split("\n"):
        if l.startswith("ATOM"):
            name, res, ch, seq = l[12:16].strip(), l[17:20].strip(), l[21], l[22:26].strip()
            rows.append(f"ATOM {l[6:11].strip()} {(l[76:78].strip() or name[:1])} {name} . {res} {ch} 1 {seq} ? {l[30:38].strip()} {l[38:46].strip()} "
                        f"{l[46:54].strip()} 1.0 {l[60:66].strip()} ? {seq} {res} {ch} {name} 1")
            if name == "CA":
                seqres.append((seq, res, l[60:66].strip()))
    # what stands before the atoms in an AFDB file: items, a loop with quoted strings, text fields, and three per-residue tables
    # (about as many lines again as the chain has residues x 3)
    one3 = dict(zip("ALA ARG ASN ASP CYS GLN GLU GLY HIS ILE LEU LYS MET PHE PRO SER THR TRP TYR VAL".split(), "ARNDCQEGHILKMFPSTWYV"))
    one = "".join(one3.get(r, "X") for _, r, _ in seqres)
    out = ["data_" + entry_id, "#", "_entry.id " + entry_id, "#", "loop_", "_audit_author.name", "_audit_author.pdbx_ordinal",
           '"Jumper, John" 1', '"Evans, Richard" 2', "'Hassabis, Demis' 3", "#",
           "_entity_poly.entity_id 1", "_entity_poly.type polypeptide(L)", "_entity_poly.pdbx_seq_one_letter_code"] + \
          [(";" if k == 0 else "") + one[k:k + 80] for k in range(0, max(len(one), 1), 80)] + [";", "#", "loop_", "_entity_poly_seq.entity_id",
           "_entity_poly_seq.hetero", "_entity_poly_seq.mon_id", "_entity_poly_seq.num"] + [f"1 n {r} {q}" for q, r, _ in seqres] + \
          ["#", "loop_", "_ma_qa_metric_local.label_asym_id", "_ma_qa_metric_local.label_comp_id", "_ma_qa_metric_local.label_seq_id",
           "_ma_qa_metric_local.metric_id", "_ma_qa_metric_local.metric_value", "_ma_qa_metric_local.model_id", "_ma_qa_metric_local.ordinal_id"] + \
          [f"A {r} {q} 2 {b} 1 {q}" for q, r, b in seqres] + \
          ["#", "loop_", "_pdbx_poly_seq_scheme.asym_id", "_pdbx_poly_seq_scheme.auth_seq_num", "_pdbx_poly_seq_scheme.entity_id", "_pdbx_poly_seq_scheme.hetero",
           "_pdbx_poly_seq_scheme.mon_id", "_pdbx_poly_seq_scheme.pdb_ins_code", "_pdbx_poly_seq_scheme.pdb_mon_id", "_pdbx_poly_seq_scheme.pdb_seq_num",
           "_pdbx_poly_seq_scheme.pdb_strand_id", "_pdbx_poly_seq_scheme.seq_id"] + [f"A {q} 1 n {r} . {r} {q} A {q}" for q, r, _ in seqres] + \
          ["#", "loop_"] + ["_atom_site." + c for c in cols] + rows + ["#"]
    return ("\n".join(out) + "\n").encode("latin-1")


ARCHIVE_STYLES = ("plain", "plain", "plain", "plain", "plain", "plain", "plain", "plain", "plain", "plain", "plain", "plain",
                  "chains", "chains", "chains", "icode", "icode", "quoted", "quoted", "models")     # 60 / 15 / 10 / 10 / 5 %


def cif_archive_from_pdb_text(pdb: bytes, entry_id: str, style: str) -> bytes:
    """the ATOM records of a PDB text as an mmCIF file of the shape the PDB ARCHIVE ships (not a predicted structure): _cell /
    _symmetry items, an _atom_site loop of 21 columns -- and, by `style`, what makes archive files differ from AFDB's:
      chains   multi-character auth_asym_id ("AA"; label_asym_id stays "A")
      icode    insertion codes: every 25th residue repeats its predecessor's number with pdbx_PDB_ins_code A (24, 24A, 25 ...)
      quoted   a hetero group behind the chain whose atom names carry primes and therefore quotes ("O5'", "C1'" ...)
      models   the chain twice, pdbx_PDB_model_num 1 and 2
    input of the end_to_end mmcif_archive leg (how much of such a corpus the device ingest takes, how much goes to the host reader)"""
    rows = []
    atoms = [l for l in pdb.decode("latin-1").split("\n") if l.startswith("ATOM")]
    asym = "AA" if style == "chains" else "A"
    def emit(model, serial0):
        out, serial, shift, last_seq, ins_of = [], serial0, 0, None, {}
        for l in atoms:
            name, res, seq = l[12:16].strip(), l[17:20].strip(), int(l[22:26])
            ins = "?"
            if style == "icode":
                if seq != last_seq:
                    last_seq = seq
                    if seq % 25 == 0:
                        shift += 1; ins_of[seq] = "A"
                num = seq - shift
                ins = ins_of.get(seq, "?")
            else:
                num = seq
            serial += 1
            out.append(f"ATOM {serial} {(l[76:78].strip() or name[:1])} {name} . {res} A 1 {seq} {ins} {l[30:38].strip()} {l[38:46].strip()} "
                       f"{l[46:54].strip()} 1.00 {l[60:66].strip()} ? {num} {res} {asym} {name} {model}")
        if style == "quoted":
            x, y, z = atoms[-1][30:38].strip(), atoms[-1][38:46].strip(), atoms[-1][46:54].strip()
            for k, (nm, el) in enumerate((("\"O5'\"", "O"), ("\"C5'\"", "C"), ("\"C4'\"", "C"), ("\"O4'\"", "O"), ("\"C1'\"", "C"), ("N9", "N"), ("PA", "P"))):
                serial += 1
                out.append(f"HETATM {serial} {el} {nm} . ATP B 2 . ? {x} {y} {z} 1.00 30.00 ? 900 ATP {asym if style == 'chains' else 'A'} {nm} {model}")
        return out, serial
    rows, last = emit(1, 0)
    if style == "models":
        rows2, _ = emit(2, 0)
        rows += rows2
    cols = ["group_PDB", "id", "type_symbol", "label_atom_id", "label_alt_id", "label_comp_id", "label_asym_id", "label_entity_id", "label_seq_id",
            "pdbx_PDB_ins_code", "Cartn_x", "Cartn_y", "Cartn_z", "occupancy", "B_iso_or_equiv", "pdbx_formal_charge", "auth_seq_id", "auth_comp_id",
            "auth_asym_id", "auth_atom_id", "pdbx_PDB_model_num"]
    out = ["data_" + entry_id, "#", "_entry.id " + entry_id, "#", "_cell.entry_id " + entry_id, "_cell.length_a 61.240", "_cell.length_b 61.240", "_cell.length_c 153.320",
           "_cell.angle_alpha 90.00", "_cell.angle_beta 90.00", "_cell.angle_gamma 120.00", "_cell.Z_PDB 6", "#", "_symmetry.entry_id " + entry_id,
           "_symmetry.space_group_name_H-M 'P 32 2 1'", "_symmetry.Int_Tables_number 154", "#", "loop_", "_entity.id", "_entity.type", "_entity.pdbx_description",
           "1 polymer 'a protein of the bench generator'", "2 non-polymer \"ADENOSINE-5'-TRIPHOSPHATE\"", "#", "_struct.entry_id " + entry_id,
           "_struct.title", ";A structure rendered in the PDB archive's mmCIF shape", "(bench.py cif_archive_from_pdb_text)", ";", "#",
           "loop_"] + ["_atom_site." + c for c in cols] + rows + ["#"]
    return ("\n".join(out) + "\n").encode("latin-1")


class Comm:
    """what the ranks exchange outside the data path: reductions of clocks and check flags, the barrier, and a wait that does not
    spin (the store). `dist` None = no group (a single process). Backend "nccl" = RCCL with device tensors; "gloo" (test mode:
    FCZ_BENCH_BACKEND=gloo, several ranks sharing one GPU, or the CPU dry run) goes through host tensors."""

    def __init__(self, dist, dev, backend, rank, world):
        self.dist, self.dev, self.backend, self.rank, self.world = dist, dev, backend, rank, world
        self.tdev = dev if backend == "nccl" else "cpu"

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def reduce(self, values, op):
        """element-wise max / min of a list of floats over the ranks"""
        if self.dist is None or not len(values):
            return [float(v) for v in values]
        t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=self.tdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.MIN)
        return [float(v) for v in t.cpu()]

    def all_true(self, flags):
        """{name: bool} -> {name: bool}: true where every rank says true (all_reduce MIN of 0 / 1); None stays None"""
        keys = [k for k, v in flags.items() if v is not None]
        red = self.reduce([1.0 if flags[k] else 0.0 for k in keys], "min")
        out = dict(flags)
        out.update({k: bool(v >= 0.5) for k, v in zip(keys, red)})
        return out

    def wait_for_rank0(self, key, work=None):
        """rank 0 runs `work` while the other ranks SLEEP on the rendezvous store (a barrier on the GPU would spin their host
        threads on the cores the work is being timed on); -> work's result on rank 0, None elsewhere"""
        if self.dist is None or self.world == 1:
            return work() if work else None
        store = self.dist.distributed_c10d._get_default_store()
        if self.rank == 0:
            try:
                return work() if work else None
            finally:
                store.set(key, "1")
        store.wait([key])
        return None


def timed(fn, steps, comm):
    """`steps` calls of fn bracketed by barrier + synchronize on both sides; seconds, max over ranks"""
    comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    comm.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return comm.reduce([dt], "max")[0]


def end_to_end_leg(args, codec, w, dev):
    """Disk to disk through the C++ host, both directions, next to the reference's own driver loops on the same inputs
    (oracle/_ref, `omp parallel for`, best thread count):
      compress    PDB files -> FCZ database (`host/foldcomp-hip compress -d`): the structure ingest runs on the device (the host
                  threads read the files into page-locked buffers, one DMA per job, parse -> fragments -> batch -> FCZ in HBM);
                  `--host-parse` (round 2's pipeline: parse threads -> SoA batch -> GPU) is timed beside it;
      decompress  FCZ database -> PDB-text database (`host/foldcomp-hip decompress -d`, BASELINE configs[2] as the reference runs
                  it: Foldcomp::read + decompress + writeAtomCoordinatesToPDB per entry, src/main.cpp:612-689).
    The input files are the first --e2e-files chains of the headline workload rendered to PDB text by the device formatter; every
    run walks them --e2e-passes times (a `-f` list that names the directory that often), so that the wall time is many times the
    HIP start-up; that start-up (`ctx_ready_s`) is reported and the steady rate excludes it. The GPU is one stage of several
    here: `gpu_call_share` = time inside the codec calls / (workers x steady wall)."""
    import shutil
    import subprocess
    import tempfile
    n = min(w.C, args.e2e_files)
    passes = max(1, args.e2e_passes)
    host = os.path.join(ROOT, "host", "foldcomp-hip")
    if not os.path.exists(host):
        return {"skipped": "host/foldcomp-hip not built"}
    lib = codec.lib
    text_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    _lib.check(lib.fcz_pdb_sizes_dev(codec.ctx, w.blob_dev.data_ptr(), w.off_dev.data_ptr(), n, w.res_off_dev.data_ptr(),
                                     w.atom_off_dev.data_ptr(), ctypes.byref(w.cout), text_off.data_ptr()), "pdb sizes")
    codec.synchronize()
    tbytes = int(text_off[-1])
    text_dev = torch.empty(tbytes, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    _lib.check(lib.fcz_pdb_format_dev(codec.ctx, w.blob_dev.data_ptr(), w.off_dev.data_ptr(), n, w.res_off_dev.data_ptr(),
                                      w.atom_off_dev.data_ptr(), ctypes.byref(w.cout), 0, text_off.data_ptr(), text_dev.data_ptr()), "pdb format")
    codec.synchronize()
    text = text_dev.cpu().numpy(); toff = text_off.cpu().numpy()
    del text_dev
    tmp = tempfile.mkdtemp(prefix="fcz_e2e_", dir=os.environ.get("TMPDIR") or "/tmp")
    try:
        src = os.path.join(tmp, "pdb"); os.mkdir(src)
        paths = []
        for i in range(n):
            pth = os.path.join(src, f"s{i:07d}.pdb")
            with open(pth, "wb") as fh:
                fh.write(text[toff[i]:toff[i + 1]].tobytes())
            paths.append(pth)
        lst = os.path.join(tmp, "inputs.txt")
        with open(lst, "w") as fh:
            fh.write((src + "\n") * passes)
        n_res = int(w.res_off_dev[n])
        eff = effective_cores(); cores = os.cpu_count() or 1
        out = {"files": n, "passes": passes, "input_bytes": int(toff[n]) * passes, "residues": n_res * passes, "host_cores": eff, "hardware_threads": cores}
        tcounts = sorted({t for t in (max(1, eff // 2), eff, 2 * eff) if 1 <= t <= cores})

        def run_host(cmd, timeout=900):
            t0 = time.perf_counter()
            r = subprocess.run([host, *cmd], capture_output=True, text=True, timeout=timeout)
            wall = time.perf_counter() - t0
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                raise RuntimeError((r.stderr or r.stdout)[-400:])
            st = json.loads(line[-1]); st["process_wall_s"] = round(wall, 4)
            return st

        def summarise(runs, bytes_key, what):
            best = min(runs, key=lambda x: x["wall_s"])
            steady = max(best["wall_s"] - best["ctx_ready_s"], 1e-9)
            return {"command": what, "wall_s": best["wall_s"], "process_wall_s": best["process_wall_s"], "ctx_ready_s": best["ctx_ready_s"],
                    "steady_wall_s": round(steady, 4), "workers": best["workers"], "host_threads": best.get("host_threads"),
                    "records": best["records"], "residues_per_s": round(best["residues"] / best["wall_s"]),
                    "steady_residues_per_s": round(best["residues"] / steady),
                    "steady_text_GB_per_s": round(best[bytes_key] / steady / 1e9, 2),
                    "codec_call_s_sum": best["codec_call_s_sum"], "gpu_call_share": round(best["codec_call_s_sum"] / (best["workers"] * steady), 3),
                    "read_or_parse_s": best.get("parse_s"), "host_parsed_files": best.get("host_parsed_files"),
                    "page_locked_blocks": best.get("pinned_blocks"),
                    **({"queue_wait_s_sum": best["queue_wait_s_sum"], "write_s_sum": best["write_s_sum"], "buffer_alloc_s_sum": best.get("buffer_alloc_s_sum"),
                        "all_queued_s": best.get("all_queued_s")} if "write_s_sum" in best else {}),
                    "wall_s_by_threads": {str(r_.get("host_threads")): r_["wall_s"] for r_ in runs}}

        # ---- compress: device ingest, and round 2's host-parse pipeline beside it ----
        comp = {}
        wpg = ["--workers-per-gpu", str(args.e2e_workers)] if args.e2e_workers else []
        runs = [run_host(["compress", "-d", "-y", "-t", str(t), "--gpus", "1", *wpg, "--json-stats", "-f", lst, os.path.join(tmp, f"db{t}")]) for t in tcounts]
        comp["gpu_host"] = summarise(runs, "input_bytes", "host/foldcomp-hip compress -d -t <threads> --gpus 1 -f <list> <db>   (structure ingest on the device)")
        comp["gpu_host"]["link_GB_per_s"] = comp["gpu_host"]["steady_text_GB_per_s"]     # every text byte crosses the link once; the FCZ bytes coming back are 2.5 % of it
        runs_h = [run_host(["compress", "-d", "-y", "-t", str(eff), "--host-parse", "--gpus", "1", "--json-stats", "-f", lst, os.path.join(tmp, "dbh")])]
        comp["gpu_host_parse"] = summarise(runs_h, "input_bytes", "... --host-parse   (parse threads -> SoA batch -> GPU: round 2's pipeline)")
        comp["device_ingest_over_host_parse"] = round(comp["gpu_host"]["steady_residues_per_s"] / max(comp["gpu_host_parse"]["steady_residues_per_s"], 1), 2)
        same = True
        for ext in ("", ".index"):
            same = same and open(os.path.join(tmp, f"db{tcounts[0]}") + ext, "rb").read() == open(os.path.join(tmp, "dbh") + ext, "rb").read()
        comp["databases_identical"] = same
        out["compress"] = comp
        # ---- the same chains as mmCIF text (the format AFDB ships): structure ingest of mmCIF on the device (k_ingest_parse_cif) beside
        #      the host reader on the same files ----
        try:
            n_cif = min(n, max(64, args.e2e_files // 4))
            cdir = os.path.join(tmp, "cif"); os.mkdir(cdir)
            cif_bytes = 0
            for i in range(n_cif):
                cb = cif_from_pdb_text(text[toff[i]:toff[i + 1]].tobytes(), f"S{i:07d}")
                cif_bytes += len(cb)
                with open(os.path.join(cdir, f"s{i:07d}.cif"), "wb") as fh:
                    fh.write(cb)
            clst = os.path.join(tmp, "cifs.txt")
            with open(clst, "w") as fh:
                fh.write((cdir + "\n") * passes)
            runs_c = [run_host(["compress", "-d", "-y", "-t", str(eff), "--gpus", "1", *wpg, "--json-stats", "-f", clst, os.path.join(tmp, "dbc")])]
            runs_ch = [run_host(["compress", "-d", "-y", "-t", str(eff), "--host-parse", "--gpus", "1", "--json-stats", "-f", clst, os.path.join(tmp, "dbch")])]
            cc = {"files": n_cif, "passes": passes, "text_bytes_per_pass": cif_bytes,
                  "gpu_host": summarise(runs_c, "input_bytes", "host/foldcomp-hip compress -d -f <list of .cif> <db>   (mmCIF parsed on the device)"),
                  "gpu_host_parse": summarise(runs_ch, "input_bytes", "... --host-parse   (the host's mmCIF reader)")}
            cc["device_ingest_over_host_parse"] = round(cc["gpu_host"]["steady_residues_per_s"] / max(cc["gpu_host_parse"]["steady_residues_per_s"], 1), 2)
            cc["databases_identical"] = all(open(os.path.join(tmp, "dbc") + ext, "rb").read() == open(os.path.join(tmp, "dbch") + ext, "rb").read() for ext in ("", ".index", ".lookup"))
            comp["mmcif"] = cc
            shutil.rmtree(cdir, ignore_errors=True)
        except (RuntimeError, subprocess.TimeoutExpired, OSError) as e:
            comp["mmcif"] = {"failed": str(e)[-300:]}
        # ---- mmCIF as the PDB ARCHIVE ships it: the same chains rendered with multi-character chain names, insertion
        #      codes, quoted atom names on a hetero group, two models (ARCHIVE_STYLES: 60 % none of these, 15 / 10 / 10 / 5 %) + the
        #      reference's own mmCIF fixture: how much of such a corpus the device ingest takes, how much goes back to the host reader ----
        try:
            n_arc = min(n, 1000)
            adir = os.path.join(tmp, "cif_archive"); os.mkdir(adir)
            styles = {}
            for i in range(n_arc):
                st_ = ARCHIVE_STYLES[i % len(ARCHIVE_STYLES)]
                styles[st_] = styles.get(st_, 0) + 1
                with open(os.path.join(adir, f"a{i:06d}.cif"), "wb") as fh:
                    fh.write(cif_archive_from_pdb_text(text[toff[i]:toff[i + 1]].tobytes(), f"A{i:06d}", st_))
            try:
                import gzip as _gz
                fx = np.load(os.path.join(ROOT, "tests", "golden", "reference_ingest.npz"))
                with open(os.path.join(adir, "ref_test.cif"), "wb") as fh:
                    fh.write(_gz.decompress(fx["file:test.cif.gz"].tobytes()))
                styles["reference fixture test.cif"] = 1
            except (OSError, KeyError):
                pass
            alst = os.path.join(tmp, "arc.txt")
            with open(alst, "w") as fh:
                fh.write((adir + "\n") * passes)
            run_a = run_host(["compress", "-d", "-y", "-t", str(eff), "--gpus", "1", *wpg, "--json-stats", "-f", alst, os.path.join(tmp, "dba")])
            run_ah = run_host(["compress", "-d", "-y", "-t", str(eff), "--host-parse", "--gpus", "1", "--json-stats", "-f", alst, os.path.join(tmp, "dbah")])
            nfa = sum(styles.values()) * passes
            comp["mmcif_archive"] = {"files_per_pass": sum(styles.values()), "passes": passes, "styles": styles,
                                     "gpu_host": summarise([run_a], "input_bytes", "host/foldcomp-hip compress -d -f <list of PDB-archive-style .cif> <db>"),
                                     "gpu_host_parse": summarise([run_ah], "input_bytes", "... --host-parse"),
                                     "host_parsed_files": run_a.get("host_parsed_files"), "hand_back_rate": round((run_a.get("host_parsed_files") or 0) / max(nfa, 1), 4),
                                     "databases_identical": all(open(os.path.join(tmp, "dba") + ext, "rb").read() == open(os.path.join(tmp, "dbah") + ext, "rb").read() for ext in ("", ".index", ".lookup"))}
            shutil.rmtree(adir, ignore_errors=True)
        except (RuntimeError, subprocess.TimeoutExpired, OSError) as e:
            comp["mmcif_archive"] = {"failed": str(e)[-300:]}
        # ---- gzipped input (AFDB ships .pdb.gz / .cif.gz): the gzip members cross the link as they are and are inflated on the
        #      device (k_inflate, round 6; reference: zlib in gemmi::MaybeGzipped / uncompressBuffer src/structure_reader.cpp:156-203),
        #      then parsed and compressed there. The first 4 096 files gzipped at level 6, both formats; `--host-inflate` (zlib on the
        #      reader threads, round 5's route) beside it; the reference's loop on the same files is timed in the cpu_reference block ----
        gz_sets = {}
        try:
            import gzip as _gzip
            n_gz = min(n, 4096)
            gzo = {}
            for kind in ("pdb", "cif"):
                gdir = os.path.join(tmp, "gz_" + kind); os.mkdir(gdir)
                def make_gz(i, kind=kind, gdir=gdir):                     # (zlib releases the GIL: the files are made on all host cores)
                    tb_ = text[toff[i]:toff[i + 1]].tobytes()
                    if kind == "cif":
                        tb_ = cif_from_pdb_text(tb_, f"S{i:07d}")
                    zb = _gzip.compress(tb_, 6)
                    gp = os.path.join(gdir, f"s{i:07d}.{kind}.gz")
                    with open(gp, "wb") as fh:
                        fh.write(zb)
                    return gp, len(tb_), len(zb)
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(max(1, eff)) as ex:
                    made = list(ex.map(make_gz, range(n_gz)))
                gpaths = [m_[0] for m_ in made]; raw_b = sum(m_[1] for m_ in made); gz_b = sum(m_[2] for m_ in made)
                glst = os.path.join(tmp, f"gz_{kind}.txt")
                with open(glst, "w") as fh:
                    fh.write((gdir + "\n") * passes)
                runs_g = [run_host(["compress", "-d", "-y", "-t", str(t), "--gpus", "1", *wpg, "--json-stats", "-f", glst, os.path.join(tmp, f"dbgz_{kind}_{t}")]) for t in tcounts]
                sm = summarise(runs_g, "input_bytes", f"host/foldcomp-hip compress -d -f <list of .{kind}.gz> <db>   (inflate + parse + codec on the device)")
                res_pass = int(w.res_off_dev[n_gz]) & 0xFFFFFFFF
                gzo[kind + "_gz"] = {"files": n_gz, "passes": passes, "gz_bytes_per_pass": gz_b, "text_bytes_per_pass": raw_b, "residues_per_pass": res_pass,
                                     "gpu_host": sm, "records_per_pass": runs_g[0]["records"] // passes,
                                     "steady_inflated_text_GB_per_s": round(raw_b * passes / max(sm["steady_wall_s"], 1e-9) / 1e9, 2)}
                # the same list with zlib on the reader threads (round 5's route), and the records of both routes compared
                run_hi = run_host(["compress", "-d", "-y", "-t", str(eff), "--host-inflate", "--gpus", "1", *wpg, "--json-stats", "-f", glst, os.path.join(tmp, f"dbgz_{kind}_hi")])
                hi = summarise([run_hi], "input_bytes", "... --host-inflate   (zlib on the reader threads, parse + codec on the device)")
                gzo[kind + "_gz"]["host_inflate"] = {"steady_residues_per_s": hi["steady_residues_per_s"], "wall_s": hi["wall_s"], "host_threads": hi["host_threads"]}
                gzo[kind + "_gz"]["device_over_host_inflate"] = round(sm["steady_residues_per_s"] / max(hi["steady_residues_per_s"], 1), 2)
                gzo[kind + "_gz"]["device_inflated_files"] = runs_g[0].get("device_inflated_files"); gzo[kind + "_gz"]["host_inflated_after_device_refusal"] = runs_g[0].get("host_inflated_after_device_refusal")
                gzo[kind + "_gz"]["databases_identical"] = all(open(os.path.join(tmp, f"dbgz_{kind}_{tcounts[0]}") + ext, "rb").read() == open(os.path.join(tmp, f"dbgz_{kind}_hi") + ext, "rb").read() for ext in ("", ".index", ".lookup"))
                gz_sets[kind] = gpaths
                # k_inflate alone on the same members, resident in HBM (HIP events on the ctx stream; every member's status; a sample of
                # the texts against zlib's)
                try:
                    import zlib as _zlib
                    nk = min(n_gz, 2560)
                    mem = [open(g_, "rb").read() for g_ in gpaths[:nk]]
                    goff = np.zeros(nk + 1, np.uint64); goff[1:] = np.cumsum([len(m_) for m_ in mem])
                    rawz = np.frombuffer(b"".join(mem), np.uint8)
                    tzo = np.zeros(nk + 1, np.uint64)
                    _lib.check(lib.fcz_inflate_sizes(rawz.ctypes.data, goff.ctypes.data, nk, None, tzo.ctypes.data), "fcz_inflate_sizes")
                    d_raw = torch.from_numpy(rawz.copy()).to(dev); d_go = torch.from_numpy(goff.view(np.int64)).to(dev); d_to = torch.from_numpy(tzo.view(np.int64)).to(dev)
                    d_tx = torch.empty(int(tzo[nk]) + 64, dtype=torch.uint8, device=dev); d_st = torch.zeros(nk, dtype=torch.int32, device=dev)
                    torch.cuda.synchronize()
                    call_ = lambda: _lib.check(lib.fcz_inflate_dev(codec.ctx, d_raw.data_ptr(), d_go.data_ptr(), nk, None, d_to.data_ptr(), d_tx.data_ptr(), d_st.data_ptr()), "fcz_inflate_dev")
                    call_(); codec.synchronize(); codec.enable_timing(True); codec.reset_timing()
                    for _ in range(5):
                        call_()
                    codec.synchronize()
                    ims, iln = codec.kernel_time("inflate"); ims /= max(iln, 1)
                    got_ = d_tx[: int(tzo[nk])].cpu().numpy()
                    same_ = all(got_[int(tzo[i_]):int(tzo[i_ + 1])].tobytes() == _zlib.decompress(mem[i_], 31) for i_ in range(0, nk, max(1, nk // 32)))
                    gzo[kind + "_gz"]["inflate_kernel"] = {"members": nk, "gz_bytes": int(goff[nk]), "text_bytes": int(tzo[nk]), "avg_launch_ms": round(ims, 3),
                                                           "inflated_text_GB_per_s": round(int(tzo[nk]) / ims / 1e6, 1) if ims else None,
                                                           "residues_per_s": round((int(w.res_off_dev[nk]) & 0xFFFFFFFF) / ims * 1e3) if ims else None,
                                                           "refused_members": int((d_st != 0).sum()), "sampled_texts_equal_zlib": bool(same_),
                                                           "what": "k_inflate (one wavefront per gzip member, CRC-32 + ISIZE verified on the device) on members resident in HBM; "
                                                                   "bytes moved = gz in + text out: not a bandwidth kernel (DESIGN.md 6.2)"}
                    del d_raw, d_tx, d_st, d_go, d_to
                except Exception as e_:   # noqa: BLE001
                    gzo[kind + "_gz"]["inflate_kernel"] = {"failed": str(e_)[-200:]}
            # ---- the same .pdb.gz members as ONE plain tar archive -- how AFDB ships a proteome -- through `compress -d <tar> <db>`: the
            #      members are byte ranges of the archive, read into the job buffers and inflated on the device like the files of
            #      the directory above; beside it the reference's own command line on the same archive (oracle/_ref/foldcomp_ref =
            #      src/main.cpp + lib/microtar compiled as they lie; TarProcessor reads the members under `omp critical`) ----
            try:
                import tarfile as _tarfile
                tarp = os.path.join(tmp, "afdb_like.tar")
                with _tarfile.open(tarp, "w", format=_tarfile.GNU_FORMAT) as tf_:
                    for gp_ in gz_sets["pdb"]:
                        tf_.add(gp_, arcname="afdb/" + os.path.basename(gp_))
                tlst = os.path.join(tmp, "tars.txt")
                with open(tlst, "w") as fh:
                    fh.write((tarp + "\n") * passes)
                run_t = run_host(["compress", "-d", "-y", "-t", str(eff), "--gpus", "1", *wpg, "--json-stats", "-f", tlst, os.path.join(tmp, "dbtar")])
                smt = summarise([run_t], "input_bytes", "host/foldcomp-hip compress -d -f <list naming one tar of .pdb.gz members> <db>   (members read from the archive, inflate + parse + codec on the device)")
                tr = {"archive_bytes": os.path.getsize(tarp), "members": len(gz_sets["pdb"]), "passes": passes, "gpu_host": smt,
                      "device_inflated_files": run_t.get("device_inflated_files"),
                      "steady_over_directory_of_the_same_members": round(smt["steady_residues_per_s"] / max(gzo["pdb_gz"]["gpu_host"]["steady_residues_per_s"], 1), 3),
                      "databases_identical_to_the_directory_run": all(open(os.path.join(tmp, "dbtar") + ext, "rb").read() == open(os.path.join(tmp, f"dbgz_pdb_{tcounts[0]}") + ext, "rb").read() for ext in ("", ".index", ".lookup"))}
                refcli = os.path.join(ROOT, "oracle", "_ref", "foldcomp_ref")
                if os.path.exists(refcli):
                    best_ = None
                    for t_ in tcounts:
                        for ext in ("", ".index", ".lookup", ".dbtype"):
                            if os.path.exists(os.path.join(tmp, "dbtar_ref") + ext):
                                os.remove(os.path.join(tmp, "dbtar_ref") + ext)
                        t0_ = time.perf_counter()
                        r_ = subprocess.run([refcli, "compress", "-t", str(t_), "-d", tarp, os.path.join(tmp, "dbtar_ref")], capture_output=True, text=True, timeout=900)
                        dt_ = time.perf_counter() - t0_
                        if r_.returncode == 0 and (best_ is None or dt_ < best_[0]):
                            best_ = (dt_, t_)
                    if best_:
                        res_pass_ = gzo["pdb_gz"]["residues_per_pass"]
                        tr["cpu_reference"] = {"what": "oracle/_ref/foldcomp_ref compress -t <threads> -d <the same tar> <db> (the reference's own command line, one walk, process wall time, best thread count)",
                                               "wall_s": round(best_[0], 4), "threads": best_[1], "residues_per_s": round(res_pass_ / best_[0])}
                        tr["steady_speedup_vs_cpu_reference"] = round(smt["steady_residues_per_s"] / max(tr["cpu_reference"]["residues_per_s"], 1), 2)
                        # every record of the reference's database (by name; its key order is thread-schedule dependent) == ours, pad bytes masked
                        try:
                            from foldcomp_amd.database import DatabaseReader as _DR
                            ra_, rb_ = _DR(os.path.join(tmp, "dbtar_ref")), _DR(os.path.join(tmp, "dbtar"))
                            def mk_(b_):
                                a_ = bytearray(b_)
                                for i_ in (14, 15, 22, 23):
                                    a_[i_] = 0
                                return bytes(a_)
                            mine_ = {rb_.name(i_): mk_(rb_.data(i_)) for i_ in range(len(gz_sets["pdb"]))}     # the first walk
                            eq_ = sum(1 for i_ in range(len(ra_)) if mine_.get(ra_.name(i_)) == mk_(ra_.data(i_)))
                            tr["records_equal_reference"] = f"{eq_}/{len(ra_)}"
                            ra_.close(); rb_.close()
                        except Exception as e_:   # noqa: BLE001
                            tr["records_equal_reference"] = "failed: " + str(e_)[-120:]
                gzo["pdb_gz"]["tar"] = tr
                os.remove(tarp)
            except (RuntimeError, subprocess.TimeoutExpired, OSError, KeyError) as e:
                gzo["pdb_gz"]["tar"] = {"failed": str(e)[-300:]}
            plain = comp["gpu_host"]["steady_residues_per_s"]
            gzo["pdb_gz"]["steady_over_plain_pdb"] = round(gzo["pdb_gz"]["gpu_host"]["steady_residues_per_s"] / max(plain, 1), 3)
            if isinstance(comp.get("mmcif"), dict) and "gpu_host" in comp["mmcif"]:
                gzo["cif_gz"]["steady_over_plain_cif"] = round(gzo["cif_gz"]["gpu_host"]["steady_residues_per_s"] / max(comp["mmcif"]["gpu_host"]["steady_residues_per_s"], 1), 3)
            comp["gz"] = gzo
        except (RuntimeError, subprocess.TimeoutExpired, OSError, KeyError) as e:
            comp["gz"] = {"failed": str(e)[-300:]}
        # one pass as the decompress leg's input
        db1 = os.path.join(tmp, "db_one")
        run_host(["compress", "-d", "-y", "-t", str(eff), "--gpus", "1", "--json-stats", src, db1])
        dlist = os.path.join(tmp, "dbs.txt")
        dpasses = max(1, passes // 4)
        with open(dlist, "w") as fh:
            fh.write((db1 + "\n") * dpasses)
        dec = {"entries": n * dpasses}
        runs_d = []
        for t in tcounts:
            for ext in ("", ".index", ".lookup", ".dbtype"):       # (a run does not pay for the truncation of the previous run's output)
                if os.path.exists(os.path.join(tmp, "pdbdb") + ext):
                    os.remove(os.path.join(tmp, "pdbdb") + ext)
            runs_d.append(run_host(["decompress", "-d", "-y", "-t", str(t), "--gpus", "1", *wpg, "--json-stats", "-f", dlist, os.path.join(tmp, "pdbdb")]))
        dec["gpu_host"] = summarise(runs_d, "text_bytes", "host/foldcomp-hip decompress -d --gpus 1 -f <list> <db>   (decode + PDB text on the device)")
        dec["gpu_host"]["link_GB_per_s"] = dec["gpu_host"]["steady_text_GB_per_s"]
        out["decompress"] = dec

        # ---- the sharded driver at N = 1 (what `--gpus 8` runs per rank): `python -m foldcomp_amd <mode> -d --gpus 1`
        #      = a 1-rank RCCL group around the same engine (`foldcomp-hip --shard 0/1`) + the count exchange + the splice, on the
        #      same inputs as gpu_host above: what the process group, the second process and the exchange cost beside the bare engine
        def run_sharded(mode, inp_list, out_db, ranks=1):
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "OMP_NUM_THREADS")}
            env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
            t0 = time.perf_counter()
            r = subprocess.run([sys.executable, "-m", "foldcomp_amd", mode, "-d", "-y", "--gpus", str(ranks), "-t", str(max(1, eff // ranks)), "--json-stats", "-f", inp_list, out_db],
                               capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
            wall = time.perf_counter() - t0
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                raise RuntimeError((r.stderr or r.stdout)[-400:])
            st = json.loads(line[-1]); st["process_wall_s"] = round(wall, 4)
            return st
        try:
            sh = {}
            for mode, lst_, ref_db, ref_leg in (("compress", lst, os.path.join(tmp, f"db{tcounts[0]}"), comp["gpu_host"]), ("decompress", dlist, os.path.join(tmp, "pdbdb"), dec["gpu_host"])):
                st = run_sharded(mode, lst_, os.path.join(tmp, f"sharded_{mode}"))
                same = all(open(os.path.join(tmp, f"sharded_{mode}") + ext, "rb").read() == open(ref_db + ext, "rb").read() for ext in (".index", ".lookup", ".dbtype"))
                same = same and os.path.getsize(os.path.join(tmp, f"sharded_{mode}")) == os.path.getsize(ref_db)
                with open(os.path.join(tmp, f"sharded_{mode}"), "rb") as fa, open(ref_db, "rb") as fb:
                    same = same and fa.read(1 << 24) == fb.read(1 << 24)
                sh[mode] = {"command": f"python -m foldcomp_amd {mode} -d --gpus 1 -f <list> <db>   (1-rank {st['backend']} group; engine = host/foldcomp-hip --shard 0/1)",
                            "world": st["world"], "records": st["records"], "wall_s": st["wall_s"], "process_wall_s": st["process_wall_s"],
                            "torch_and_group_s_beside_engine": st["torch_and_group_s_beside_engine"], "engine_s": st["engine_s"], "engine_steady_s": st["engine_steady_s_max"],
                            "exchange_and_splice_s": st["exchange_and_splice_s"], "residues_per_s": st["residues_per_s"],
                            "steady_residues_per_s": st["steady_residues_per_s"], "engine_max_rss_kb": st["engine_max_rss_kb_per_rank"],
                            "steady_over_gpu_host": round(st["steady_residues_per_s"] / max(ref_leg["steady_residues_per_s"], 1), 3),
                            "database_equals_gpu_host": bool(same)}
                os.remove(os.path.join(tmp, f"sharded_{mode}"))
            # ---- TEST MODE: two ranks on this ONE GPU (gloo group; on a node every rank has its own GPU and the group is RCCL). What it
            #      shows is the file side of the sharded decompress: the ranks exchange {records, text bytes} BEFORE they write (sizes
            #      pre-pass of the engine), every job is written once at its final offset of the final data file, and what is left
            #      after the engines end is the concatenation of the ranks' index lines (reference: every record
            #      appended once, src/main.cpp:656-664, src/database_writer.cpp:36-58)
            try:
                st = run_sharded("decompress", dlist, os.path.join(tmp, "sharded_two"), ranks=2)
                same = all(open(os.path.join(tmp, "sharded_two") + ext, "rb").read() == open(os.path.join(tmp, "pdbdb") + ext, "rb").read() for ext in (".index", ".lookup", ".dbtype"))
                same = same and os.path.getsize(os.path.join(tmp, "sharded_two")) == os.path.getsize(os.path.join(tmp, "pdbdb"))
                with open(os.path.join(tmp, "sharded_two"), "rb") as fa, open(os.path.join(tmp, "pdbdb"), "rb") as fb:
                    while same:
                        ba, bb_ = fa.read(1 << 26), fb.read(1 << 26)
                        same = ba == bb_
                        if not ba:
                            break
                sh["decompress_two_ranks"] = {"mode": "TEST MODE: 2 ranks share the one GPU of this box (gloo group, half the host threads each)",
                                              "world": st["world"], "backend": st["backend"], "records": st["records"], "records_per_rank": st["records_per_rank"],
                                              "wall_s": st["wall_s"], "engine_s": st["engine_s"], "sizes_pass_s": st.get("sizes_pass_s_max"),
                                              "exchange_and_splice_s": st["exchange_and_splice_s"],
                                              "exchange_and_splice_over_engine": round(st["exchange_and_splice_s"] / max(st["engine_s"], 1e-9), 4),
                                              "exchange_wait_for_slowest_rank_s": st.get("exchange_wait_for_slowest_rank_s"),
                                              "file_work_after_exchange_s": st.get("file_work_after_exchange_s"),
                                              "before_this_change": {"exchange_and_splice_s": 1.5273, "engine_s": 1.1591, "note": "round 4's partial database + splice on the same box class (gpurun_out/r5_ring/e2e_before.json, kept as profiles/r5_sharded_decompress_before_after.json)"},
                                              "data_written_once": st.get("data_written_once"),
                                              "steady_residues_per_s": st["steady_residues_per_s"], "database_equals_gpu_host": bool(same)}
                os.remove(os.path.join(tmp, "sharded_two"))
            except (RuntimeError, subprocess.TimeoutExpired, OSError, KeyError) as e:
                sh["decompress_two_ranks"] = {"failed": str(e)[-400:]}
            out["sharded"] = sh
        except (RuntimeError, subprocess.TimeoutExpired, OSError, KeyError) as e:
            out["sharded"] = {"failed": str(e)[-400:]}

        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _harness as H
        if H.have_ref():
            rl = H.load_ref()
            rl.ref_compress_files.restype = ctypes.c_int
            rl.ref_compress_files.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p]
            rpasses = min(passes, 4)
            blob = b"".join(p_.encode() + b"\0" for p_ in paths) * rpasses
            secs = ctypes.c_double(); rres = ctypes.c_ulonglong(); rbytes = ctypes.c_ulonglong(); flen = ctypes.c_long()
            first = ctypes.create_string_buffer(1 << 20)
            ref_runs = {}
            file_hash = np.zeros(n * rpasses, np.uint64)          # per file: the live reference's record(s), pad-masked, hashed
            for t in tcounts:
                fail = rl.ref_compress_files(blob, n * rpasses, t, args.anchor, ctypes.byref(secs), ctypes.byref(rres), ctypes.byref(rbytes), first, 1 << 20, ctypes.byref(flen),
                                             file_hash.ctypes.data)
                ref_runs[t] = (secs.value, int(fail))
            bt = min(ref_runs, key=lambda k: ref_runs[k][0])
            comp["cpu_reference"] = {"what": "the reference's driver loop on the same files (oracle/_ref: StructureReader + Foldcomp::compress, omp parallel for), best thread count",
                                     "cores": bt, "passes": rpasses, "wall_s": round(ref_runs[bt][0], 4), "residues_per_s": round(rres.value / ref_runs[bt][0]) if ref_runs[bt][0] else None,
                                     "failed_files": ref_runs[bt][1], "fcz_bytes": int(rbytes.value),
                                     "wall_s_by_threads": {str(k): round(v[0], 4) for k, v in ref_runs.items()}}
            first_ref_record = first.raw[:flen.value]              # (the calls below reuse the buffer)
            # the reference's loop on the gzipped sets (one walk each)
            for kind, gpaths in gz_sets.items():
                gblob = b"".join(p_.encode() + b"\0" for p_ in gpaths)
                gr = {}
                for t in tcounts:
                    fail = rl.ref_compress_files(gblob, len(gpaths), t, args.anchor, ctypes.byref(secs), ctypes.byref(rres), ctypes.byref(rbytes), first, 1 << 20, ctypes.byref(flen), None)
                    gr[t] = (secs.value, int(fail), int(rres.value))
                bt_ = min(gr, key=lambda k: gr[k][0])
                g_ = comp["gz"][kind + "_gz"]
                g_["cpu_reference"] = {"what": "the reference's driver loop on the same .gz files (oracle/_ref: inflate + StructureReader + Foldcomp::compress, omp parallel for), best thread count",
                                       "cores": bt_, "wall_s": round(gr[bt_][0], 4), "residues_per_s": round(gr[bt_][2] / gr[bt_][0]) if gr[bt_][0] else None, "failed_files": gr[bt_][1],
                                       "wall_s_by_threads": {str(k): round(v[0], 4) for k, v in gr.items()}}
                if g_["cpu_reference"]["residues_per_s"]:
                    g_["steady_speedup_vs_cpu_reference"] = round(g_["gpu_host"]["steady_residues_per_s"] / g_["cpu_reference"]["residues_per_s"], 2)
            from foldcomp_amd.database import DatabaseReader
            rd = DatabaseReader(os.path.join(tmp, f"db{tcounts[0]}"))
            e0 = bytearray(rd.data(0)); rd.close()
            for k in (14, 15, 22, 23):
                e0[k] = 0
            comp["first_record_equals_reference"] = bytes(e0) == first_ref_record
            # every record of the GPU database against the live reference's record of the same file (by lookup name = the file's stem;
            # the 4 uninitialised header bytes masked), hashed on both sides by the checker's FNV-1a
            rl.ref_hash_records.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
            rd = DatabaseReader(os.path.join(tmp, f"db{tcounts[0]}"))
            nrec = len(rd)
            recs = [rd.data(i) for i in range(nrec)]
            names = [rd.name(i) for i in range(nrec)]
            rd.close()
            roff = np.zeros(nrec + 1, np.uint64); roff[1:] = np.cumsum([len(r_) for r_ in recs], dtype=np.uint64)
            rblob = np.frombuffer(b"".join(recs), np.uint8)
            gh = np.zeros(nrec, np.uint64)
            rl.ref_hash_records(rblob.ctypes.data, roff.ctypes.data, nrec, gh.ctypes.data)
            M64 = (1 << 64) - 1
            def chain1(h):                      # ref_compress_files chains a file's fragment hashes; these files hold one fragment
                x = 0xcbf29ce484222325
                for wd in (h & 0xFFFFFFFF, h >> 32):
                    x = ((x ^ wd) * 0x100000001b3) & M64
                return x
            want = {os.path.splitext(os.path.basename(p_))[0]: int(file_hash[i]) for i, p_ in enumerate(paths)}
            equal = sum(1 for nm_, h in zip(names, gh) if want.get(nm_) == chain1(int(h)))
            comp["records_equal_reference"] = f"{equal}/{nrec}"
            comp["records_equal_reference_all"] = bool(equal == nrec and nrec == n * passes)
            comp["speedup_vs_cpu_reference"] = round(comp["gpu_host"]["residues_per_s"] / comp["cpu_reference"]["residues_per_s"], 2)
            comp["steady_speedup_vs_cpu_reference"] = round(comp["gpu_host"]["steady_residues_per_s"] / comp["cpu_reference"]["residues_per_s"], 2)
            comp["speedup_note"] = (f"per-residue rates: the GPU host walks the files {passes} times, the reference's loop {rpasses} times (it has no start-up to amortise); "
                                    "speedup_vs_cpu_reference compares full walls (HIP start-up inside), the steady ratio leaves the GPU host's start-up (ctx_ready_s) out")
            # decompress: the reference's loop over the same database
            rl.ref_decompress_db.restype = ctypes.c_int
            rl.ref_decompress_db.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
            tb = ctypes.c_ulonglong(); first_t = ctypes.create_string_buffer(1 << 22)
            dref = {}
            for t in tcounts:
                fail = rl.ref_decompress_db(db1.encode(), (db1 + ".index").encode(), t, 0, dpasses, os.path.join(tmp, "refpdbdb").encode(), ctypes.byref(secs),
                                            ctypes.byref(rres), ctypes.byref(tb), first_t, 1 << 22, ctypes.byref(flen))
                dref[t] = (secs.value, int(fail))
            bt = min(dref, key=lambda k: dref[k][0])
            dec["cpu_reference"] = {"what": "the reference's decompress loop on the same database into a database (oracle/_ref: Foldcomp::read + decompress + writeAtomCoordinatesToPDB, "
                                            "omp for, writer_append under omp critical, free_writer), best thread count",
                                    "cores": bt, "wall_s": round(dref[bt][0], 4), "residues_per_s": round(rres.value / dref[bt][0]) if dref[bt][0] else None,
                                    "text_GB_per_s": round(tb.value / dref[bt][0] / 1e9, 3) if dref[bt][0] else None, "failed_entries": dref[bt][1],
                                    "wall_s_by_threads": {str(k): round(v[0], 4) for k, v in dref.items()}}
            rd = DatabaseReader(os.path.join(tmp, "pdbdb"))
            dec["first_text_equals_reference"] = rd.data(0) == first_t.raw[:flen.value]
            # every text of the GPU database against the reference writer's database (its last run above wrote refpdbdb), by lookup name
            rr = DatabaseReader(os.path.join(tmp, "refpdbdb"))
            ref_by_name = {}
            for i in range(len(rr)):
                ref_by_name.setdefault(rr.name(i), i)
            equal_t = 0
            n_cmp = min(len(rd), n)                 # one walk over the input database = every distinct entry (the later walks repeat it)
            for i in range(n_cmp):
                j = ref_by_name.get(rd.name(i))
                equal_t += 1 if (j is not None and rd.data(i) == rr.data(j)) else 0
            dec["texts_equal_reference"] = f"{equal_t}/{n_cmp}"
            dec["texts_equal_reference_all"] = bool(equal_t == n_cmp and n_cmp == n and len(rd) == n * dpasses)
            rr.close()
            rd.close()
            dec["speedup_vs_cpu_reference"] = round(dref[bt][0] / dec["gpu_host"]["wall_s"], 2)
            dec["steady_speedup_vs_cpu_reference"] = round(dref[bt][0] / dec["gpu_host"]["steady_wall_s"], 2)
        return out
    except (RuntimeError, subprocess.TimeoutExpired) as e:
        return {"failed": str(e)[-400:]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def host_boundary_leg(args, codec, hb):
    """The PCIe-inclusive rate: the same codec through the HOST-pointer entry points (fcz_compress_batch /
    fcz_decompress_batch, include/fcz_hip.h), page-locked buffers on both sides (fcz_pinned_alloc, what the C++ host
    stages through). Every byte crosses the link twice per round trip (objects in, FCZ out; FCZ in, objects out), so this is
    bounded by PCIe, not by the kernels; it is reported beside the resident `value`, never as it. Two figures: one batch in
    flight on one ctx, and `overlapped`: two ctxs on the GPU, each with its own page-locked buffers, driven by two host
    threads -- one ctx's transfers run beside the other's kernels and, the link being full duplex, beside its transfers in
    the other direction (the arrangement of the C++ host's two workers per GPU)."""
    import ctypes
    import threading
    from foldcomp_amd._lib import CAtomsOut
    from foldcomp_amd.structure import batch_as_c
    lib = codec.lib
    held = []

    def pinned(arr):
        arr = np.ascontiguousarray(arr)
        nb = max(arr.nbytes, 1)
        p = lib.fcz_pinned_alloc(nb)
        if not p:
            raise MemoryError("fcz_pinned_alloc")
        held.append(p)
        view = np.ctypeslib.as_array((ctypes.c_ubyte * nb).from_address(p))[:arr.nbytes].view(arr.dtype)
        view[...] = arr.reshape(-1)
        return view

    def make_set(cdc):
        """page-locked copies of the batch and of every output, and the two calls on ctx `cdc`"""
        import copy
        pb = copy.copy(hb)
        for k in ("res_off", "atom_off", "x", "y", "z", "atom_code", "res_code", "bfac_ca", "first_res_index",
                  "first_atom_index", "chain_id", "titles", "title_off"):
            setattr(pb, k, pinned(getattr(hb, k)))
        C, R, M = pb.n_chains, pb.n_residues, pb.n_atoms
        off = pinned(cdc.compress_sizes(pb))
        blob = pinned(np.zeros(int(off[-1]), np.uint8))
        st = pinned(np.zeros(C, np.int32))
        cb = batch_as_c(pb)
        x = pinned(np.zeros(M, np.float32)); y = pinned(np.zeros(M, np.float32)); z = pinned(np.zeros(M, np.float32))
        bf = pinned(np.zeros(R, np.float32)); rc = pinned(np.zeros(R, np.uint8)); ac = pinned(np.zeros(M, np.uint8))
        out = CAtomsOut(x.ctypes.data, y.ctypes.data, z.ctypes.data, bf.ctypes.data, rc.ctypes.data, ac.ctypes.data)

        def compress():
            r = lib.fcz_compress_batch(cdc.ctx, ctypes.byref(cb), off.ctypes.data, blob.ctypes.data, st.ctypes.data)
            if r != 0:
                raise RuntimeError(f"fcz_compress_batch -> {r}")

        compress()
        info, d_res_off, d_atom_off = cdc.decompress_sizes(blob, off)

        def decompress():
            r = lib.fcz_decompress_batch(cdc.ctx, blob.ctypes.data, off.ctypes.data, C, d_res_off.ctypes.data,
                                         d_atom_off.ctypes.data, 0, ctypes.byref(out))
            if r != 0:
                raise RuntimeError(f"fcz_decompress_batch -> {r}")

        decompress()
        return dict(pb=pb, off=off, blob=blob, x=x, compress=compress, decompress=decompress, keep=(st, cb, y, z, bf, rc, ac, out, info, d_res_off, d_atom_off))

    second = None
    try:
        A = make_set(codec)
        pb, off, blob, x = A["pb"], A["off"], A["blob"], A["x"]
        C, R, M = pb.n_chains, pb.n_residues, pb.n_atoms
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps): A["compress"]()
        t1 = time.perf_counter()
        for _ in range(reps): A["decompress"]()
        t2 = time.perf_counter()
        tc, td = (t1 - t0) / reps, (t2 - t1) / reps
        in_c = sum(getattr(pb, k).nbytes for k in ("res_off", "atom_off", "x", "y", "z", "atom_code", "res_code", "bfac_ca",
                                                   "first_res_index", "first_atom_index", "chain_id", "titles", "title_off"))
        fcz = int(off[-1])
        out_d = 12 * M + 4 * R + R + M
        res = {"chains": C, "residues": R, "buffers": "page-locked (fcz_pinned_alloc)",
               "compress_ms": round(tc * 1e3, 3), "decompress_ms": round(td * 1e3, 3),
               "residues_per_s": round(R / (tc + td)),
               "compress_link_GBs": round((in_c + fcz + 4 * C) / tc / 1e9, 2),
               "decompress_link_GBs": round((fcz + out_d) / td / 1e9, 2),
               "bytes_over_link_per_residue": round((in_c + 2 * fcz + out_d) / R, 1),
               "fcz_sha": __import__("hashlib").sha1(blob.tobytes()).hexdigest()[:16],
               "coords_filled": bool(np.isfinite(x).all() and float(np.abs(x).max()) > 0)}
        # ---- two ctxs, two host threads, alternating page-locked buffers ----
        second = Codec(codec.device)
        B = make_set(second)
        oreps = 4
        errs = []

        def worker(S, order):
            try:
                for _ in range(oreps):
                    for k in order:
                        S[k]()
            except Exception as e:   # noqa: BLE001
                errs.append(repr(e))

        def both(orders):
            th = [threading.Thread(target=worker, args=(S_, o_)) for S_, o_ in zip((A, B), orders)]
            t0_ = time.perf_counter()
            for t_ in th: t_.start()
            for t_ in th: t_.join()
            return time.perf_counter() - t0_
        w_c = both((("compress",), ("compress",)))                      # both ctxs compress: copies of one beside kernels of the other
        w_d = both((("decompress",), ("decompress",)))
        w_rt = both((("compress", "decompress"), ("decompress", "compress")))   # opposite directions at the same time: the link is full duplex
        if errs:
            res["overlapped"] = {"failed": errs[0]}
        else:
            nb = 2 * oreps                                              # batches per direction and measurement
            res["overlapped"] = {"ctxs": 2, "host_threads": 2, "batches_per_run": nb,
                                 "compress_ms_per_batch": round(w_c / nb * 1e3, 3), "decompress_ms_per_batch": round(w_d / nb * 1e3, 3),
                                 "compress_link_GBs": round((in_c + fcz + 4 * C) * nb / w_c / 1e9, 2),
                                 "decompress_link_GBs": round((fcz + out_d) * nb / w_d / 1e9, 2),
                                 "round_trip_ms_per_batch": round(w_rt / nb * 1e3, 3),
                                 "residues_per_s": round(R * nb / w_rt),
                                 "round_trip_link_GBs_both_directions": round((in_c + 2 * fcz + out_d + 4 * C) * nb / w_rt / 1e9, 2),
                                 "fcz_equals_single_ctx": bool(B["blob"].tobytes() == blob.tobytes()),
                                 "over_one_in_flight": round((tc + td) / (w_rt / nb), 2)}
        return res
    finally:
        if second is not None:
            second.close()
        for p in held:
            lib.fcz_pinned_free(p)


def alt_numerics_leg(args, codec, w, dev, comm, timed_mode, compress_ms):
    """the decompress side once more in the OTHER numerics mode (see --numerics): time, per-kernel times, roofline fraction of
    its longest kernel, and the deviation of its coordinates from the timed mode's on the first 65 536 chains. Leaves the
    outputs and the ctx in the timed mode."""
    other = "fast" if timed_mode == "exact" else "exact"
    ns = min(w.C, 65536)
    na = int(w.atom_off_dev[ns]) & 0xFFFFFFFF
    ref = {k: w.out_t[k][:na].clone() for k in ("x", "y", "z")}
    codec.set_numerics(other == "fast")
    w.decompress(); codec.synchronize()
    codec.reset_timing()
    steps = max(1, min(args.steps, 5))
    world = comm.world
    dt = timed(lambda: (w.decompress(), codec.synchronize()), steps, comm)
    km = span_ms(codec)
    # per-atom deviation = the largest of |dx|, |dy|, |dz|. In pieces of 2^27 atoms (one stacked tensor of 3 x 547 M floats -- 65 536
    # chains of 1 000 residues -- made torch's own reduction kernel fault), the order statistics on a strided sample of <= 2^24 atoms
    piece = 1 << 27; stride = max(1, na // (1 << 24))
    mx_dev = 0.0; n_above = 0; samp = []
    for a0 in range(0, na, piece):
        a1 = min(na, a0 + piece)
        dv = (w.out_t["x"][a0:a1] - ref["x"][a0:a1]).abs()
        dv = torch.maximum(dv, (w.out_t["y"][a0:a1] - ref["y"][a0:a1]).abs())
        dv = torch.maximum(dv, (w.out_t["z"][a0:a1] - ref["z"][a0:a1]).abs())
        if dv.numel():
            mx_dev = max(mx_dev, float(dv.max())); n_above += int((dv > 1e-2).sum())
            first = (-a0) % stride
            samp.append(dv[first::stride].clone())
        del dv
    devv = torch.cat(samp) if samp else torch.zeros(1, device=dev)
    stats = {"sample_chains": ns, "sample_atoms": na, "order_statistics_on_atoms": int(devv.numel()), "median_A": float(devv.median()),
             "p999_A": float(torch.quantile(devv[::max(1, devv.numel() // 4_000_000)].float(), 0.999)), "max_A": mx_dev,
             "frac_above_1e-2_A": n_above / max(na, 1)}
    rmsd, mx = w.round_trip_deviation()
    kb = w.kernel_bytes()
    names = ("k_backbone", "k_res_index", "k_sidechain")
    dom = max(names, key=lambda k: km[KERNEL_SPANS[k]])
    ms = km[KERNEL_SPANS[dom]]
    dec_ms = dt / steps * 1e3
    A = w.M / w.R; f = w.fcz_bytes / w.R
    out = {"mode": other, "timed_mode": timed_mode,
           "what": "decompress in the other numerics mode (fast = plain float arithmetic, parallel rigid-transform backbone; "
                   "exact = float32 coordinates bit-identical to the reference), same device-resident records",
           "steps": steps, "decompress_ms": round(dec_ms, 3), "decompress_residues_per_s": round(w.R * world / (dec_ms * 1e-3)),
           "round_trip_ms_with_compress": round(compress_ms + dec_ms, 3),
           "round_trip_residues_per_s": round(w.R * world / ((compress_ms + dec_ms) * 1e-3)) if compress_ms else None,
           "decompress_algorithmic_GBs": round((f + 12 * A + 4) * w.R / (dec_ms * 1e-3) / 1e9, 1),
           "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(kb[dom] / (ms * 1e-3) / 1e9, 1) if ms else None, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(kb[dom] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms else None, "avg_launch_ms": round(ms, 4)},
           "kernel_ms": {k: round(v, 4) for k, v in km.items() if v and k.startswith("decompress")},
           "deviation_from_timed_mode": stats, "all_atom_rmsd_vs_input_A": round(rmsd, 4)}
    codec.set_numerics(timed_mode == "fast")
    w.decompress(); codec.synchronize()
    del ref, devv, samp
    return out


def secondary_legs(args, codec, dev, comm):
    """BASELINE configs[2] and configs[4] at their shape (the datasets themselves cannot be fetched): `--mixed-chains` chains
    per GPU with log-normal lengths (AFDB Swiss-Prot has 542 k structures), anchor -b 25.
      decompress_only  device-resident FCZ records -> SoA atoms (sizes pass + batch), the configs[2] operation
      mixed            compress + decompress of the same ragged batch, the configs[4] operation
    Each with its own time, residues/s (all ranks), per-kernel times, roofline fraction of its longest kernel, a 256-chain
    oracle sample (bit-exact bar) and the full-batch round-trip deviation. Outside the headline `value`."""
    Cm = args.mixed_chains
    rank, world = comm.rank, comm.world
    d = generate_resident(Cm, 0, args.anchor, args.gen_chunk, dev, seed_base=(1 << 30) + args.seed_base + rank * Cm, mixed=True)
    note(f"secondary legs: generated {Cm} mixed-length chains")
    w = Workload(codec, d, dev)
    w.compress(); w.decompress(); codec.synchronize()            # warm-up; leaves the FCZ records resident
    steps = max(1, args.mixed_steps)
    kb = w.kernel_bytes()

    def leg(fn, names, nbytes):
        codec.reset_timing()
        dt = timed(lambda: (fn(), codec.synchronize()), steps, comm)
        km = span_ms(codec)
        dom = max(names, key=lambda k: km[KERNEL_SPANS[k]])
        ms = km[KERNEL_SPANS[dom]]
        ach = kb[dom] / (ms * 1e-3) / 1e9 if ms else 0.0
        return {"chains_per_gpu": Cm, "residues_per_gpu": w.R, "mean_residues_per_chain": round(w.R / Cm, 1), "longest_chain": int((d["res_off"][1:] - d["res_off"][:-1]).max()),
                "steps": steps, "ms_per_step": round(dt / steps * 1e3, 3), "residues_per_s": round(w.R * world * steps / dt),
                "algorithmic_GBs": round(nbytes / (dt / steps) / 1e9, 1), "frac_of_hbm_peak": round(nbytes / (dt / steps) / 1e9 / HBM_PEAK_GBS, 4),
                "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(ach / HBM_PEAK_GBS, 4), "avg_launch_ms": round(ms, 4),
                             **dict(zip(("traffic", "traffic_source"), measured_traffic(dom, w.R, -1, "mixed")))},
                "kernel_ms": {k: round(v, 4) for k, v in km.items() if v}}

    A = w.M / w.R; f = w.fcz_bytes / w.R
    dec = leg(w.decompress, ("k_backbone", "k_res_index", "k_sidechain"), (f + 12 * A + 4) * w.R)
    mix = leg(lambda: (w.compress(), w.decompress()), tuple(KERNEL_SPANS), (13 * A + 9 + f + f + 12 * A + 4) * w.R)
    if not args.no_parity:
        # every rank checks its own batch; the flags are AND-ed over the ranks, the deviations are the worst of any rank
        n = min(256, Cm)
        pc = parity_check(d, w, n, chunk=args.parity_chunk)
        rmsd, mx = w.round_trip_deviation()
        flags = comm.all_true({"fcz_bit_exact": pc["fcz_bit_exact"], "coords_bit_exact": pc["coords_bit_exact"],
                               "all_status_ok": int((w.status_dev != 0).sum()) == 0,
                               "residue_counts_round_trip": bool(torch.equal(w.res_off_dev, d["res_off"].to(torch.int32)))})
        rmsd, mx = comm.reduce([rmsd, mx], "max")
        par = {"chains_checked": n * world, "ranks_checked": world, **flags,
               "all_atom_rmsd_A": round(rmsd, 4), "max_atom_deviation_A": round(mx, 3)}
        dec["parity"] = par; mix["parity"] = par
    del w, d
    torch.cuda.empty_cache()
    return dec, mix


def self_launch_command(args_gpus, argv, env):
    """--gpus N > 1 outside a torch.distributed launcher: the command that starts N ranks of this script on this node
    (one per GPU, RCCL rendezvous on 127.0.0.1). None when no launch is needed. A launcher whose WORLD_SIZE disagrees with
    --gpus is an error: a silent 1-GPU run labelled as N GPUs (or the reverse) must not happen."""
    ws = env.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != int(args_gpus):
            raise SystemExit(f"bench.py: --gpus {args_gpus} but the launcher set WORLD_SIZE={ws}; refusing to run a mislabelled job")
        return None
    if args_gpus <= 1:
        return None
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(args_gpus)}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def dry_run(args, world, rank):
    """rendezvous + the post-clock plumbing of an N-rank line (see --dry-run), without a codec call: the ranks meet in a process
    group, every rank runs its own check and the flags are AND-ed exactly as the real run does it (Comm.all_true), rank 0 times the
    CPU baseline while the others sleep on the store (Comm.wait_for_rank0). With no GPU result to compare, a rank's check is the
    oracle against itself on 64 small chains of ITS OWN seed range (records decode, and re-encode to the same bytes)."""
    import torch.distributed as dist
    have_gpu = torch.cuda.is_available()
    backend = os.environ.get("FCZ_BENCH_BACKEND") or ("nccl" if have_gpu else "gloo")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = f"cuda:{local}" if (have_gpu and backend == "nccl") else "cpu"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend)
        comm = Comm(dist, dev, backend, rank, world)
        from foldcomp_amd import shard as _shard
        ok, pre = _shard.preflight(torch.device(dev) if backend == "nccl" else None)
        if not ok:
            print(f"bench.py: {pre}", file=sys.stderr); sys.stderr.flush()
            os._exit(3)
        seen = int(comm.reduce([1.0], "max")[0] * dist.get_world_size())
        dist.barrier()
    else:
        comm = Comm(None, dev, backend, 0, 1)
        seen = 1
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _harness as H
    n = 64
    d = synthetic.generate(n, 120, seed=0xF01DC0DE, device="cpu", anchor_threshold=args.anchor, first_chain_id=args.seed_base + rank * n)
    hb = synthetic.to_chain_batch(d)
    blob, off, st = H.oracle_compress(hb, n_threads=1)
    o = H.oracle_decompress(blob, off, n_threads=1)
    blob2, off2, _ = H.oracle_compress(hb, n_threads=2)
    flags = comm.all_true({"fcz_bit_exact": bool((st == 0).all()) and blob.tobytes() == blob2.tobytes() and np.array_equal(off, off2),
                           "coords_bit_exact": int(o["res_off"][-1]) == hb.n_residues and bool(np.isfinite(o["x"]).all())})
    cpu = comm.wait_for_rank0("cpu_baseline_done", (lambda: cpu_baseline(hb, args.anchor)) if args.cpu_sample else None)
    if rank == 0:
        emit_line({"metric": "residues/sec compress+decompress, 350-aa chains; bit-exact FCZ; 1/2/4/8 GPUs", "value": None,
                   "n_gpus": seen, "steps": args.steps, "warmup": args.warmup, "dry_run": True, "cpu_baseline": cpu,
                   "preflight": "answered" if world > 1 else None, "host_cpus": effective_cores(), "host_threads_per_rank": max(1, effective_cores() // max(1, world)),
                   "parity": {"chains_checked": n * world, "ranks_checked": world, **flags,
                              "kind": "dry run: no GPU result exists; every rank checked the oracle against itself on its own seed range"}})
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


_T0 = time.perf_counter()
_JSON_FD = None


def emit_line(obj):
    """the JSON line, on the process's original stdout (see main)"""
    data = (json.dumps(obj) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(data.decode()); sys.stdout.flush()
    else:
        os.write(_JSON_FD, data)


def note(msg):
    """progress + wall time to stderr (stdout carries exactly one JSON line)"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def main():
    args = parse_args()
    cmd = self_launch_command(args.gpus, sys.argv[1:], os.environ)
    if cmd is not None:
        import subprocess
        if not args.dry_run and torch.cuda.is_available() and torch.cuda.device_count() < args.gpus and os.environ.get("FCZ_BENCH_BACKEND") != "gloo":
            raise SystemExit(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) are visible")
        raise SystemExit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries the ONE JSON line and nothing else: native libraries (RCCL prints a version banner) write to file
    # descriptor 1 behind Python's back, so descriptor 1 is pointed at stderr and the line goes to a private copy of the original
    sys.stdout.flush()
    global _JSON_FD
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)
    if args.dry_run:
        return dry_run(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the FCZ hot path has no CPU fallback")
    # FCZ_BENCH_BACKEND=gloo is the TEST mode of the N > 1 code path on a box with fewer GPUs than ranks: the ranks share devices
    # (RCCL refuses two ranks on one GPU) and the index exchange goes through host tensors; the line says so and is not a scaling point.
    backend = os.environ.get("FCZ_BENCH_BACKEND") or "nccl"
    if backend == "gloo":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    # One code path for every N: also a single GPU runs in a (1-rank) RCCL process group, so that the step timed at N = 1 --
    # compress, synchronise, gather of the record lengths for the index, decompress -- is the step timed at N = 2, 4, 8.
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world == 1 and "MASTER_PORT" not in os.environ:
        import socket
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
    group_note = None
    try:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
            group_note = f"TEST MODE: {world} ranks over {backend} sharing {torch.cuda.device_count()} GPU(s); record lengths gathered through host tensors"
        world = dist.get_world_size()   # what RCCL actually formed; n_gpus below reports this, not the flag
        assert world == args.gpus, (world, args.gpus)
    except Exception as e:   # noqa: BLE001
        if world > 1:
            raise
        dist = None; group_note = f"no 1-rank RCCL group ({type(e).__name__}): the index exchange of the step is skipped at N = 1"
    comm = Comm(dist, dev, backend, rank, world)
    preflight_note = None
    if dist is not None:
        # fail fast: one 1-element all_gather with a deadline BEFORE anything is generated (an 8-rank line that hangs in its first
        # collective after minutes of generation tells nobody anything)
        from foldcomp_amd import shard as _shard
        ok, preflight_note = _shard.preflight(torch.device(dev) if backend == "nccl" else None)
        if not ok:
            print(f"bench.py: {preflight_note}", file=sys.stderr); sys.stderr.flush()
            os._exit(3)

    C, n_res = args.chains, args.residues
    d = generate_resident(C, n_res, args.anchor, args.gen_chunk, dev, seed_base=args.seed_base + rank * C, mixed=args.mixed)
    note(f"generated {C} chains")
    codec = Codec(local)
    codec.set_numerics(args.numerics == "fast")
    lib = codec.lib
    w = Workload(codec, d, dev)
    R, M, fcz_bytes = w.R, w.M, w.fcz_bytes
    off_dev, blob_dev, status_dev, res_off_dev, atom_off_dev, out_t, cout = (w.off_dev, w.blob_dev, w.status_dev, w.res_off_dev,
                                                                             w.atom_off_dev, w.out_t, w.cout)
    lengths_dev = torch.zeros(C, dtype=torch.int64, device=dev)
    gdev = comm.tdev
    gathered = [torch.zeros(C, dtype=torch.int64, device=gdev) for _ in range(world)] if (dist is not None and rank == 0) else None
    torch.cuda.synchronize()

    def step():
        w.compress()
        if dist is not None:
            # the only exchange of the sharded job: per-record lengths -> rank 0 builds the global index
            codec.synchronize()
            torch.sub(off_dev[1:], off_dev[:-1], out=lengths_dev)
            dist.gather(lengths_dev if backend == "nccl" else lengths_dev.cpu(), gathered, dst=0)
        w.decompress()

    for _ in range(args.warmup):
        step()
    codec.synchronize(); torch.cuda.synchronize()
    warm_csum = w.checksum() if (args.warmup and not args.no_parity) else None
    codec.enable_timing(True); codec.reset_timing()
    # the timed region: exactly --steps steps between barrier + synchronize pairs, max over ranks
    comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    codec.synchronize(); torch.cuda.synchronize()
    comm.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = comm.reduce([dt], "max")[0]

    # per-kernel device time (HIP events on the codec's own stream)
    ktime = span_ms(codec)
    note(f"headline timed: {dt / args.steps * 1e3:.2f} ms/step")
    alt = alt_numerics_leg(args, codec, w, dev, comm, args.numerics,
                           ktime["compress_sizes"] + ktime["compress_index"] + ktime["compress_angles"] + ktime["compress_pack"])
    note(f"alt numerics ({alt['mode']}) done: decompress {alt['decompress_ms']} ms")
    # ---- BASELINE configs[2] / configs[4] at their shape: every rank runs them (their clocks are max-over-ranks too) ----
    legs = secondary_legs(args, codec, dev, comm) if (args.mixed_chains and not args.mixed) else None
    note("secondary legs done")
    # ---- size-independent properties of the FULL batch (outside the timed region) ----
    props = None
    if not args.no_parity:
        # every rank checks its own batch; flags AND-ed over the ranks, deviations = the worst of any rank
        # (1) determinism: the blob after the timed steps has the checksum it had after the warm-up steps
        csum = w.checksum()
        # (2) decode(encode(x)) ~ x: decompress once more in the input's atom order (`-a`) and compare every atom
        rmsd, mxdev = w.round_trip_deviation()
        # (3) sizes: decompress counts == input counts, every chain compressed with status OK
        pf = comm.all_true({"residue_counts_round_trip": bool(torch.equal(res_off_dev, d["res_off"].to(torch.int32))),
                            "deterministic": (csum == warm_csum) if warm_csum is not None else None})
        rmsd, mxdev = comm.reduce([rmsd, mxdev], "max")
        props = {"all_atom_rmsd_A": round(rmsd, 4), "max_atom_deviation_A": round(mxdev, 3), **pf,
                 "blob_checksum_rank0": csum, "ranks_checked": world}
        # restore the default-order outputs the parity check and the PDB leg below read
        w.decompress()
        codec.synchronize()

    note("properties done")
    # ---- §8 f2 leg, outside the timed region: PDB text of the first chains, formatted on the device ----
    pdb = None
    if args.pdb_sample and rank == 0:
        npdb = min(C, args.pdb_sample)
        text_off = torch.zeros(npdb + 1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        _lib.check(lib.fcz_pdb_sizes_dev(codec.ctx, blob_dev.data_ptr(), off_dev.data_ptr(), npdb, res_off_dev.data_ptr(),
                                         atom_off_dev.data_ptr(), ctypes.byref(cout), text_off.data_ptr()), "pdb sizes")
        codec.synchronize()
        tbytes = int(text_off[-1])
        text_dev = torch.empty(tbytes, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        codec.reset_timing()
        for _ in range(3):
            _lib.check(lib.fcz_pdb_sizes_dev(codec.ctx, blob_dev.data_ptr(), off_dev.data_ptr(), npdb, res_off_dev.data_ptr(),
                                             atom_off_dev.data_ptr(), ctypes.byref(cout), text_off.data_ptr()), "pdb sizes")
            _lib.check(lib.fcz_pdb_format_dev(codec.ctx, blob_dev.data_ptr(), off_dev.data_ptr(), npdb, res_off_dev.data_ptr(),
                                              atom_off_dev.data_ptr(), ctypes.byref(cout), 0, text_off.data_ptr(), text_dev.data_ptr()), "pdb format")
        codec.synchronize()
        ms_s, n_s = codec.kernel_time("pdb_sizes"); ms_f, n_f = codec.kernel_time("pdb_format")
        ms_s /= max(n_s, 1); ms_f /= max(n_f, 1)
        n_at = int(atom_off_dev[npdb]) & 0xFFFFFFFF
        # parity of the first chain against the host restatement of the reference writer (oracle/host_text.py, a checker)
        from foldcomp_amd import fczfile
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from host_text import pdb_from_result as _pdb_from_result
        e0 = blob_dev[:int(off_dev[1])].cpu().numpy().tobytes()
        a1, r1 = int(atom_off_dev[1]), int(res_off_dev[1])
        d0 = {"atom_off": np.asarray([0, a1]), "res_off": np.asarray([0, r1]), "res_code": out_t["res_code"][:r1].cpu().numpy(),
              "bfac_res": out_t["bfac_res"][:r1].cpu().numpy(), "x": out_t["x"][:a1].cpu().numpy(), "y": out_t["y"][:a1].cpu().numpy(),
              "z": out_t["z"][:a1].cpu().numpy()}
        rec = fczfile.parse(e0)
        from foldcomp_amd._aa_tables import RES_ALT_SLOT, RES_ATOMS, RES_NATOMS
        ac = np.concatenate([[RES_ATOMS[c][j] for j in range(RES_NATOMS[c])] for c in d0["res_code"]] + [[36]] * (a1 - int(sum(RES_NATOMS[c] for c in d0["res_code"]))))
        d0["atom_code"] = np.asarray(ac, np.uint8)
        ok_pdb = text_dev[:int(text_off[1])].cpu().numpy().tobytes() == _pdb_from_result(rec, d0, 0, False).encode("latin-1")
        pdb = {"chains": npdb, "atoms": n_at, "text_bytes": tbytes, "ms_sizes": round(ms_s, 4), "ms_format": round(ms_f, 4),
               "write_GBs": round(tbytes / (ms_f * 1e-3) / 1e9, 1) if ms_f else None,
               "frac_of_hbm_peak": round(tbytes / (ms_f * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms_f else None,
               "atoms_per_s": round(n_at / ((ms_s + ms_f) * 1e-3)) if ms_f else None, "first_chain_equals_host_writer": bool(ok_pdb)}
        # ---- §8 f3 leg: the same text back through the structure ingest on the device (PDB text -> fcz_chain_batch in HBM) ----
        try:
            from foldcomp_amd._lib import CIngestResult
            nm = [f"s{i:07d}.pdb".encode() for i in range(npdb)]
            name_off = torch.from_numpy(np.arange(npdb + 1, dtype=np.int64) * len(nm[0])).to(torch.int32).to(dev)
            names_dev = torch.from_numpy(np.frombuffer(b"".join(nm), np.uint8).copy()).to(dev)
            stem_len = torch.full((npdb,), len(nm[0]) - 4, dtype=torch.int32, device=dev)
            file_off = text_off.to(torch.int64)
            res = CIngestResult()
            torch.cuda.synchronize()
            codec.reset_timing()
            reps_i = 3
            for _ in range(reps_i):
                _lib.check(lib.fcz_ingest_pdb_dev(codec.ctx, text_dev.data_ptr(), file_off.data_ptr(), npdb, tbytes, names_dev.data_ptr(), name_off.data_ptr(),
                                                  stem_len.data_ptr(), args.anchor, 0, ctypes.byref(res)), "fcz_ingest_pdb_dev")
            codec.synchronize()
            ims = {k: codec.kernel_time(k)[0] / reps_i for k in ("ingest_parse", "ingest_frags", "ingest_fill")}
            tot_ms = sum(ims.values())
            n_r = int(res_off_dev[npdb]) & 0xFFFFFFFF
            ingest = {"files": npdb, "text_bytes": tbytes, "ms": {k: round(v, 4) for k, v in ims.items()},
                      "text_GBs": round(tbytes / (tot_ms * 1e-3) / 1e9, 1) if tot_ms else None,
                      "frac_of_hbm_peak": round(tbytes / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if tot_ms else None,
                      "residues_per_s": round(n_r / (tot_ms * 1e-3)) if tot_ms else None,
                      "counts_equal_decoded": bool(res.batch.n_chains == npdb and res.batch.n_residues == n_r and res.batch.n_atoms == n_at),
                      "refused": int(res.n_refused)}
            pdb["ingest_of_the_same_text"] = ingest
            # ---- ... and as mmCIF text (k_ingest_parse_cif): the first chains rendered as AFDB-shaped mmCIF on the host, tiled to the
            #      same number of files on the device ----
            n_src = min(npdb, 1024)
            th = text_dev[:int(text_off[n_src])].cpu().numpy(); to = text_off[:n_src + 1].cpu().numpy()
            cifs = [cif_from_pdb_text(th[to[i]:to[i + 1]].tobytes(), f"S{i:07d}") for i in range(n_src)]
            reps_t = max(1, npdb // n_src)
            one = np.frombuffer(b"".join(cifs), np.uint8)
            lens = np.asarray([len(c) for c in cifs] * reps_t, np.int64)
            coff = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)])).to(dev)
            ctext = torch.from_numpy(one.copy()).to(dev).repeat(reps_t)
            nf = n_src * reps_t
            nmc = [f"s{i:07d}.cif".encode() for i in range(nf)]
            cname_off = torch.from_numpy(np.arange(nf + 1, dtype=np.int64) * len(nmc[0])).to(torch.int32).to(dev)
            cnames = torch.from_numpy(np.frombuffer(b"".join(nmc), np.uint8).copy()).to(dev)
            cstem = torch.full((nf,), len(nmc[0]) - 4, dtype=torch.int32, device=dev)
            resc = CIngestResult()
            torch.cuda.synchronize()
            _lib.check(lib.fcz_ingest_pdb_dev(codec.ctx, ctext.data_ptr(), coff.data_ptr(), nf, int(ctext.numel()), cnames.data_ptr(), cname_off.data_ptr(),
                                              cstem.data_ptr(), args.anchor, 0, ctypes.byref(resc)), "fcz_ingest_pdb_dev (mmCIF)")
            codec.synchronize(); codec.reset_timing()
            for _ in range(reps_i):
                _lib.check(lib.fcz_ingest_pdb_dev(codec.ctx, ctext.data_ptr(), coff.data_ptr(), nf, int(ctext.numel()), cnames.data_ptr(), cname_off.data_ptr(),
                                                  cstem.data_ptr(), args.anchor, 0, ctypes.byref(resc)), "fcz_ingest_pdb_dev (mmCIF)")
            codec.synchronize()
            cms = {k: codec.kernel_time(k)[0] / reps_i for k in ("ingest_parse", "ingest_parse_cif", "ingest_rows_cif", "ingest_frags", "ingest_fill")}
            ctot = sum(cms.values())
            n_rc = int(res_off_dev[n_src]) * reps_t
            pdb["ingest_of_the_same_chains_as_mmcif"] = {
                "files": nf, "distinct_files": n_src, "text_bytes": int(ctext.numel()), "ms": {k: round(v, 4) for k, v in cms.items()},
                "text_GBs": round(int(ctext.numel()) / (ctot * 1e-3) / 1e9, 1) if ctot else None,
                "frac_of_hbm_peak": round(int(ctext.numel()) / (ctot * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ctot else None,
                "residues_per_s": round(n_rc / (ctot * 1e-3)) if ctot else None,
                "counts_equal_decoded": bool(resc.batch.n_chains == nf and resc.batch.n_residues == n_rc), "refused": int(resc.n_refused)}
            del ctext
        except Exception as e:   # noqa: BLE001
            pdb.setdefault("ingest_of_the_same_text", {"failed": repr(e)[:300]})
            pdb["ingest_of_the_same_chains_as_mmcif"] = pdb.get("ingest_of_the_same_chains_as_mmcif") or {"failed": repr(e)[:300]}
        del text_dev
    # ---- §8 f4 leg: `extract --plddt -p 2` of every record straight from the FCZ bytes ----
    ext = None
    if not args.no_parity:
        data_off = torch.zeros(C + 1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        _lib.check(lib.fcz_extract_sizes_dev(codec.ctx, blob_dev.data_ptr(), off_dev.data_ptr(), C, 0, 2, data_off.data_ptr()), "extract sizes")
        codec.synchronize()
        dbytes = int(data_off[-1])
        data_dev = torch.empty(max(dbytes, 1), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        codec.reset_timing()
        for _ in range(3):
            _lib.check(lib.fcz_extract_dev(codec.ctx, blob_dev.data_ptr(), off_dev.data_ptr(), C, 0, 2, data_off.data_ptr(), data_dev.data_ptr()), "extract")
        codec.synchronize()
        ms_e, n_e = codec.kernel_time("extract"); ms_e /= max(n_e, 1)
        from foldcomp_amd import fczfile as _ff
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from host_text import extract_plddt as _extract_plddt
        e0 = blob_dev[:int(off_dev[1])].cpu().numpy().tobytes()
        ok_e = data_dev[:int(data_off[1])].cpu().numpy().tobytes() == _extract_plddt(_ff.parse(e0), 2).encode()
        ext = {"mode": "plddt -p 2", "records": C, "data_bytes": dbytes, "ms": round(ms_e, 4),
               "residues_per_s": round(R / (ms_e * 1e-3)) if ms_e else None,
               # algorithmic bytes: header (84 B) + one B-factor byte per residue in, the characters out
               "algorithmic_GBs": round((84 * C + R + dbytes) / (ms_e * 1e-3) / 1e9, 1) if ms_e else None,
               **comm.all_true({"first_record_equals_host": bool(ok_e)}), "ranks_checked": world}
        del data_dev
    e2e = None
    if rank == 0 and world == 1 and args.e2e_files and not args.mixed:
        # a secondary leg (disk space, a missing tool) must never cost the headline its line
        try:
            e2e = end_to_end_leg(args, codec, w, dev)
        except Exception as e:   # noqa: BLE001
            e2e = {"failed": repr(e)[-400:]}
        note("end-to-end leg done")
    codec.enable_timing(False)
    note("pdb / extract legs done")
    bad_status = int((status_dev != 0).sum())
    # ---- parity: EVERY rank compares the first --parity-chains chains of its own batch with the oracle (FCZ bytes and
    #      coordinates, bit for bit); the flags are AND-ed over the ranks ----
    parity = None
    if not args.no_parity:
        pc = parity_check(d, w, args.parity_chains, chunk=args.parity_chunk)
        flags = comm.all_true({"fcz_bit_exact": pc["fcz_bit_exact"], "coords_bit_exact": pc["coords_bit_exact"]})
        none = float(1 << 62)
        gid = lambda c: none if c is None else float(args.seed_base + rank * C + c)   # noqa: E731 - the chain's generator seed offset
        firsts = comm.reduce([gid(pc["first_fcz_mismatch_chain"]), gid(pc["first_coords_mismatch_chain"])], "min")
        parity = {"chains_checked": pc["chains_checked"] * world, "chains_checked_per_rank": pc["chains_checked"], "ranks_checked": world, **flags,
                  "bad_status": int(comm.reduce([bad_status], "max")[0]), "seed_base": args.seed_base,
                  "first_fcz_mismatch_chain_id": None if firsts[0] >= none else int(firsts[0]),
                  "first_coords_mismatch_chain_id": None if firsts[1] >= none else int(firsts[1])}
        if args.numerics == "fast":
            parity["coords_bit_exact_note"] = "timed in FCZ_NUMERICS_FAST: coordinates are not expected to be bit-identical; see alt_numerics.deviation_from_timed_mode"
        note(f"parity check done: {parity['chains_checked']} chains, fcz {parity['fcz_bit_exact']}, coords {parity['coords_bit_exact']}")
    # ---- CPU baseline: rank 0, at any world size, with an explicit thread count (a launcher exports OMP_NUM_THREADS=1; the
    #      reference loop takes num_threads(cores) itself); the other ranks sleep on the store meanwhile, off the host cores ----
    def gpu_sample(n):
        """the product's records and decoded arrays of the rank's first n chains, on the host (default atom order)"""
        n = min(n, w.C)
        goff = w.off_dev[:n + 1].cpu().numpy().astype(np.uint64)
        a1 = _u32(w.atom_off_dev[n]); r1 = _u32(w.res_off_dev[n])
        return {"blob": np.ascontiguousarray(w.blob_dev[:int(goff[-1])].cpu().numpy()), "off": np.ascontiguousarray(goff),
                "x": w.out_t["x"][:a1].cpu().numpy(), "y": w.out_t["y"][:a1].cpu().numpy(), "z": w.out_t["z"][:a1].cpu().numpy(),
                "atom_off": np.ascontiguousarray(w.atom_off_dev[:n + 1].cpu().numpy().view(np.uint32)),
                "bfac_res": w.out_t["bfac_res"][:r1].cpu().numpy(), "res_off": np.ascontiguousarray(w.res_off_dev[:n + 1].cpu().numpy().view(np.uint32))}
    cpu = comm.wait_for_rank0("cpu_baseline_done", (lambda: cpu_baseline(host_sample(d, args.cpu_sample), args.anchor,
                                                                         gpu=None if (args.no_parity or args.numerics == "fast") else gpu_sample(args.cpu_sample))) if args.cpu_sample else None)
    note("cpu baseline done")
    if parity is not None and cpu and cpu.get("live_reference"):
        lr = cpu["live_reference"]
        parity["live_reference_chains"] = lr["chains"]
        parity["live_reference_equal"] = bool(lr["records_equal"] == lr["chains"] and lr["coords_equal"] == lr["chains"])

    if rank == 0:
        A = M / R                                   # atoms per residue
        fcz_per_res = fcz_bytes / R
        # algorithmic bytes: compress reads 13A+9, writes fcz; decompress reads fcz, writes 12A+4
        bytes_compress = (13 * A + 9 + fcz_per_res) * R
        bytes_decompress = (fcz_per_res + 12 * A + 4) * R
        dec_ms = ktime["decompress_backbone"] + ktime["decompress_index"] + ktime["decompress_sidechain"]
        # The dominant kernel = the single longest launch. Decompress is two launches (k_backbone -> bb scratch ->
        # k_sidechain); SURVEY's decompress bytes belong to the pair, each kernel is charged its own share:
        # k_backbone reads the header/anchors/words (fcz minus side-chain and B-factor bytes), k_res_index reads the
        # side-chain and B-factor bytes and writes the B-factors, k_sidechain writes the atoms. The hand-over arrays
        # between the three (bb, per-residue index) are not algorithmic traffic.
        sc_bytes = (A - 3.0) + 1.0
        ktime["compress"] = ktime["compress_index"] + ktime["compress_angles"] + ktime["compress_pack"]
        # compress is three launches; the angle kernel reads the atoms and writes the side-chain bytes, the pack kernel
        # reads codes/B-factors/offsets and writes the rest of the record ([6][R] angle scratch = hand-over, not counted)
        kern = {"k_compress_angles_w": ((13 * A + (A - 3.0)) * R, ktime["compress_angles"]),
                "k_compress_index": (1 * R, ktime["compress_index"]),
                "k_compress_pack": ((9 + fcz_per_res - (A - 3.0)) * R, ktime["compress_pack"]),
                "k_backbone": ((fcz_per_res - sc_bytes) * R, ktime["decompress_backbone"]),
                "k_res_index": ((sc_bytes + 4) * R, ktime["decompress_index"]),
                "k_sidechain": (12 * A * R, ktime["decompress_sidechain"])}
        dom = max(kern, key=lambda k: kern[k][1])
        by, ms = kern[dom]
        ach = by / (ms * 1e-3) / 1e9 if ms else 0.0
        traffic, traffic_src = measured_traffic(dom, R, -1 if args.mixed else n_res)
        # the bound that actually holds (SURVEY.md section 8d "secondary bound to report honestly"): VALU issue. Instruction counts per
        # residue and the VALU-active share of a wavefront's cycles come from the committed PMC passes (they cannot be collected
        # inside the timed region); the SIMD-cycles per VALU wave-instruction use this run's own kernel time.
        valu = None
        try:
            tk = (traffic_profile()[0] or {})["kernels"][dom]
            occ = {"k_backbone": 2, "k_compress_angles_w": 3, "k_compress_pack": 3, "k_sidechain": 4}.get(dom)
            n_simd = 4 * torch.cuda.get_device_properties(dev).multi_processor_count
            valu = {"bound": "VALU issue", "valu_wave_insts_per_residue": tk["valu_wave_insts_per_residue"],
                    "simd_cycles_per_valu_wave_inst": round(ms * 1e-3 * 2.4e9 * n_simd / (tk["valu_wave_insts_per_residue"] * R), 2) if ms else None,
                    "issue_cost_of_the_mix_cycles": "2.9 (f32 add/mul/fma) ... 4.6 (f64, compares, 3-operand integer) ... 16 (f64 rsq): profiles/r3_valu_rates.txt",
                    "valu_active_share_of_simd_cycles": round(tk["valu_active_share_of_wave_cycles"] * occ, 3) if occ else None,
                    "resident_wavefronts_per_simd": occ, "source": "profiles/traffic.json (PMC passes) + this run's HIP-event time; 2.4 GHz"}
        except (OSError, KeyError, ValueError, ZeroDivisionError):
            pass
        ctl, ctl_fabric, ctl_src = controller_side_traffic(dom, R, -1 if args.mixed else n_res)
        roofline = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                    "traffic_controller_side": ctl, "traffic_controller_side_source": ctl_src,
                    "traffic_note": "`traffic` = FETCH_SIZE x 2 + WRITE_SIZE as MI355X_MICROARCH.md prescribes (requests leaving the L2 towards the fabric; the exact-unit "
                                    "TCC_EA0_*_DRAM_32B counters of the same passes give the same bytes). `traffic_controller_side` = what the memory controllers "
                                    "saw of the same launch (mem_busy_percent during a loop of this one kernel, calibrated on copies): for k_res_index, k_sidechain "
                                    "and the compress kernels the two agree within 1 %; for k_backbone the controllers see 165-168 of the 198 B/residue -- the Infinity "
                                    "Cache absorbs a sixth of its ring traffic, the rest is real HBM traffic (forward atoms and torsion trig of a segment, written "
                                    "once and read back once by the reverse pass: 120 B/residue by construction of the two-sweep algorithm, 17x the algorithmic bytes)",
                    "secondary_bound": valu,
                    "algorithmic_bytes_per_launch": by, "avg_launch_ms": ms,
                    "kernel_ms": {k: round(v, 4) for k, v in ktime.items()},
                    "per_kernel_GBs": {k: round(b / (t * 1e-3) / 1e9, 1) if t else None for k, (b, t) in kern.items()},
                    "decompress_pair_GBs": round(bytes_decompress / (dec_ms * 1e-3) / 1e9, 1) if dec_ms else None,
                    "traffic_profile_valid_for_these_kernels": traffic_profile()[0] is not None, "csrc_sha16": csrc_sha16(),
                    "hbm_copy_measured_GBs": round(copy_ceiling_f4(codec) or 0.0, 1),
                    "hbm_copy_note": "float4 grid-stride copy (fcz_selftest_copy: the kernel MI355X_MICROARCH.md measures 6.29 TB/s with), read + write bytes; "
                                     "rounds 1-5 quoted torch's copy_ of the same 1 GiB, kept as hbm_copy_torch_GBs",
                    "hbm_copy_torch_GBs": round(copy_ceiling(dev), 1)}
        hostb = None
        if args.host_chains and world == 1 and not args.mixed:
            try:
                hostb = host_boundary_leg(args, codec, host_sample(d, args.host_chains))
                # the same chains compressed on the resident path must give the same bytes
                n_h = hostb["chains"]; e_h = int(off_dev[n_h])
                hostb["fcz_equals_resident_path"] = (
                    __import__("hashlib").sha1(blob_dev[:e_h].cpu().numpy().tobytes()).hexdigest()[:16] == hostb.pop("fcz_sha"))
            except Exception as e:   # noqa: BLE001
                hostb = {"failed": repr(e)[-400:]}
            note("host-pointer (PCIe-inclusive) leg done")
        total_res = R * world * args.steps
        line = {
            "metric": "residues/sec compress+decompress, 350-aa chains; bit-exact FCZ; 1/2/4/8 GPUs",
            "value": total_res / dt, "unit": "residues/s (round trip: each residue compressed and decompressed once)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32/f64 + u8 bitpack",
            "data": "synthetic",
            "config": {"workload": (f"{C} synthetic mixed-length chains per GPU (log-normal, mean {R / C:.0f} residues), "
                                    if args.mixed else f"{C} synthetic {n_res}-residue chains per GPU, ")
                                   + f"compress+decompress, anchor -b {args.anchor}",
                       "chains_per_gpu": C, "residues_per_chain": round(R / C, 1) if args.mixed else n_res, "atoms_per_residue": round(A, 3),
                       "fcz_bytes_per_residue": round(fcz_per_res, 3), "parallelism": f"chain-sharded x{world}, no data-path collective",
                       "index_exchange": group_note or f"record lengths gathered on rank 0 over RCCL inside every step ({world}-rank group)",
                       "preflight": preflight_note, "host_cpus": effective_cores(), "host_threads_per_rank": max(1, effective_cores() // max(1, world)),
                       "seed_base": args.seed_base, "backend": backend,
                       # how to read a --gpus N line of this bench, and of the product drivers, on a CPU quota that does not grow with N
                       "expected_bound_at_n_gpus": ("this step: inputs resident in HBM, no host stage, one small gather per step -> weak scaling, per-rank rate = the 1-GPU rate; "
                                                    "the CPU baseline and the parity checkers run on rank 0's host cores only. The product drivers (end_to_end / "
                                                    "python -m foldcomp_amd --gpus N) divide the host's CPU quota between the ranks' reader threads (sharded_cli.host_threads): "
                                                    "one engine needs ~16 threads to fill one GPU's link, so on a 16-CPU quota an 8-rank directory -> database run is "
                                                    "reader-thread-bound (2 threads per engine), not GPU- or xGMI-bound")},
            "compress_residues_per_s": R / (ktime["compress"] * 1e-3) if ktime["compress"] else None,
            "decompress_residues_per_s": R / (dec_ms * 1e-3) if dec_ms else None,
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "properties": props,
            "numerics": args.numerics, "alt_numerics": alt,
            "decompress_only": legs[0] if legs else None, "mixed": legs[1] if legs else None, "pdb_text": pdb, "extract": ext, "end_to_end": e2e,
            "host_boundary": hostb,
        }
        emit_line(line)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    codec.close()


if __name__ == "__main__":
    main()
