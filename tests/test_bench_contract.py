"""CPU: the pieces of bench.py that do not need a GPU -- argument defaults the driver relies on, and the measured-traffic
table (profiles/traffic.json) naming the kernels bench.py reports on."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    import importlib
    return importlib.import_module("bench")


def test_defaults_are_the_headline_workload(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse_args()
    assert a.gpus == 1 and a.chains == 1_000_000 and a.residues == 350 and a.anchor == 25 and not a.mixed
    assert a.steps >= 1 and a.warmup >= 1


def test_traffic_table_covers_the_reported_kernels():
    b = _bench()
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert t["residues_per_chain"] == 350
    for k in ("k_compress_angles_w", "k_backbone", "k_sidechain", "k_compress_pack", "k_res_index"):
        assert k in t["kernels"], k
        traffic, src = b.measured_traffic(k, 350_000_000, 350)
        assert traffic and traffic > 0 and "pmc" in src
    # other chain lengths have no measured profile: null (and why), never a made-up number
    assert b.measured_traffic("k_compress_angles_w", 1000, 123)[0] is None
    assert b.measured_traffic("no_such_kernel", 1000, 350)[0] is None
    # the mixed-length workload has counter passes of its own (round 6): per residue of THAT batch
    mt, msrc = b.measured_traffic("k_backbone", 162_000_000, -1, "mixed")
    assert mt and mt > 0 and "mixed" in msrc


def test_traffic_is_only_quoted_for_the_kernels_it_was_measured_on(monkeypatch):
    """the counter profile carries the hash of the kernel sources it was collected on (tools/pmc_summary.py): bench.py quotes its
    figures only while the tree's hash is that one -- a kernel change makes `traffic` null until the passes are re-collected, it
    never goes stale silently"""
    b = _bench()
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert t["csrc_sha16"] == b.csrc_sha16(), "profiles/traffic.json describes other kernel sources than this tree: re-collect (tools/profile_gpu.sh + tools/pmc_summary.py + tools/merge_traffic.py)"
    assert {"k_backbone", "k_compress_pack", "k_compress_pack_rows_4", "k_res_index_rows"} <= set(t["kernel_set"])
    monkeypatch.setattr(b, "csrc_sha16", lambda: "0" * 16)
    traffic, why = b.measured_traffic("k_backbone", 350_000_000, 350)
    assert traffic is None and "re-collected" in why
    assert b.controller_side_traffic("k_backbone", 350_000_000, 350) == (None, None, None)


def test_gpus_flag_and_launcher_must_agree():
    b = _bench()
    import pytest
    # inside a launcher with the matching world size: no re-launch
    assert b.self_launch_command(2, ["--gpus", "2"], {"WORLD_SIZE": "2"}) is None
    assert b.self_launch_command(1, [], {}) is None
    # --gpus N without a launcher: N ranks of this script under torch.distributed.run on 127.0.0.1
    cmd = b.self_launch_command(4, ["--gpus", "4", "--steps", "3"], {})
    assert "torch.distributed.run" in cmd and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    # a launcher whose world size disagrees with the flag: refuse (a mislabelled run is worse than none)
    with pytest.raises(SystemExit):
        b.self_launch_command(8, ["--gpus", "8"], {"WORLD_SIZE": "1"})
    with pytest.raises(SystemExit):
        b.self_launch_command(1, [], {"WORLD_SIZE": "2"})


def test_plain_python_gpus_2_starts_two_ranks():
    """`python bench.py --gpus 2` (no torchrun) must become a 2-rank job: the dry run meets in a process group (gloo here)
    and reports the world size the group saw"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2
    # an N > 1 line must be creditable: the CPU baseline is timed on rank 0 at any world size (with more than the one OpenMP
    # thread a launcher exports) and EVERY rank's check enters the parity flags
    assert line["cpu_baseline"] is not None and line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["kind"] in ("reference", "port")
    assert line["parity"]["ranks_checked"] == 2 and line["parity"]["chains_checked"] == 2 * 64
    assert line["parity"]["fcz_bit_exact"] is True and line["parity"]["coords_bit_exact"] is True
    # and a real (non-dry) run on a box without GPUs fails loudly instead of printing a number
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_dry_run_of_the_eight_rank_line():
    """the driver's first N = 8 run has no second try: the rendezvous, the pre-flight, every rank's check AND-ed into the parity flags,
    the CPU baseline on rank 0 while seven ranks sleep on the store, ONE JSON line -- all of it at world 8 on the CPU (gloo; the codec
    call itself is what the dry run leaves out)"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--cpu-sample", "64"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["parity"]["ranks_checked"] == 8 and line["parity"]["chains_checked"] == 8 * 64
    assert line["parity"]["fcz_bit_exact"] is True and line["parity"]["coords_bit_exact"] is True
    assert line["preflight"] == "answered" and line["host_threads_per_rank"] >= 1 and line["host_cpus"] >= line["host_threads_per_rank"]
    assert line["cpu_baseline"] is not None and line["cpu_baseline"]["value"] > 0


def test_bench_mmcif_rendering_holds_the_same_atoms():
    """the mmCIF files of bench.py's mmCIF legs are the chains of its PDB legs: cif_from_pdb_text of a PDB text reads back (host
    reader, mmCIF rules) as the atoms the PDB text reads back as (host reader, PDB rules), title = the entry id"""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    b = _bench()
    from foldcomp_amd.structure import parse_pdb_gemmi, parse_structure_gemmi
    from test_host_cpp import _pdb_text
    z = np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"))
    for name in ("syn:len26", "syn:len129", "pdb:test_af"):
        pdb = _pdb_text(z, name).encode()
        cif = b.cif_from_pdb_text(pdb, "ENTRY1")
        tp, _ = parse_pdb_gemmi(pdb)
        tc, title = parse_structure_gemmi(cif)
        assert title == "ENTRY1" and len(tc) == len(tp) > 0
        assert list(tc.atom) == list(tp.atom) and list(tc.residue) == list(tp.residue) and list(tc.chain) == list(tp.chain)
        assert np.array_equal(np.asarray(tc.res_index), np.asarray(tp.res_index)) and np.array_equal(np.asarray(tc.atom_index), np.asarray(tp.atom_index))
        assert np.array_equal(tc.xyz.view(np.uint32), tp.xyz.view(np.uint32)) and np.array_equal(np.asarray(tc.bfac, np.float32).view(np.uint32), np.asarray(tp.bfac, np.float32).view(np.uint32))
