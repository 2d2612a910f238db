"""The C restatement (oracle/fcz_oracle.c) against the LIVE reference (oracle/_ref = the reference's own sources compiled here) on
fresh chains of the bench generator -- the uniform 350-residue shape of configs[1] and the mixed-length shape of configs[4] --
record by record and atom by atom: every large-scale bit-exact claim of the GPU path is made against the restatement, this pins
the restatement on the workload itself.

Default size: 600 + 600 chains (seconds; the mixed-length generator is dense in chains x longest chain on the CPU). FCZ_SLOW_CHAINS=100000 runs the full-size check (minutes; the line it prints is kept
under profiles/ per round)."""
import json
import os
import sys

import numpy as np
import pytest

import _harness as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref not built (needs /root/reference)")


@pytest.mark.parametrize("mixed", [False, True])
def test_restatement_equals_live_reference_on_bench_chains(mixed):
    import bench
    n = int(os.environ.get("FCZ_SLOW_CHAINS", "1200")) // 2
    seed_base = int(os.environ.get("FCZ_SLOW_SEED", "777000"))
    d = bench.generate_resident(n, 350, 25, 8192, "cpu", seed_base, mixed=mixed)
    hb = bench.host_slice(d, 0, n)
    threads = bench.effective_cores()
    blob, off, st = H.oracle_compress(hb, n_threads=threads)
    assert (st == 0).all()
    o = H.oracle_decompress(blob, off, n_threads=threads)
    side = {"blob": np.ascontiguousarray(blob), "off": np.ascontiguousarray(off.astype(np.uint64)), "x": o["x"], "y": o["y"], "z": o["z"],
            "atom_off": np.ascontiguousarray(o["atom_off"].astype(np.uint32)), "bfac_res": o["bfac_res"],
            "res_off": np.ascontiguousarray(o["res_off"].astype(np.uint32))}
    r = bench.cpu_baseline(hb, 25, gpu=side)
    lr = r["live_reference"]
    print(json.dumps({"what": "oracle/fcz_oracle.c vs oracle/_ref, chain by chain", "shape": "mixed-length (log-normal 16..2700)" if mixed else "350 residues",
                      "seed_base": seed_base, "chains": lr["chains"], "residues": hb.n_residues, "records_equal": lr["records_equal"], "coords_equal": lr["coords_equal"],
                      "first_record_mismatch_chain": lr["first_record_mismatch_chain"], "first_coords_mismatch_chain": lr["first_coords_mismatch_chain"],
                      "digest_records": lr["digest_records"], "digest_coords": lr["digest_coords"], "reference_failed_chains": r["failed_chains"]}))
    assert r["failed_chains"] == 0
    assert lr["records_equal"] == n, lr
    assert lr["coords_equal"] == n, lr


@pytest.mark.parametrize("sigma", [0.02, 0.3, 1.5])
def test_restatement_equals_live_reference_on_distorted_chains(sigma):
    """the same pin on chains whose bond lengths and angles are not ideal (every atom moved by N(0, sigma), three decimals):
    what tests/test_gpu_edge_cases.py::test_distorted_geometry holds the GPU path to"""
    import bench
    from _cases import distorted_batch
    n = 256
    hb = distorted_batch(n, sigma, seed=int(sigma * 1000) + 11)
    threads = bench.effective_cores()
    blob, off, st = H.oracle_compress(hb, n_threads=threads)
    assert (st == 0).all()
    o = H.oracle_decompress(blob, off, n_threads=threads)
    side = {"blob": np.ascontiguousarray(blob), "off": np.ascontiguousarray(off.astype(np.uint64)), "x": o["x"], "y": o["y"], "z": o["z"],
            "atom_off": np.ascontiguousarray(o["atom_off"].astype(np.uint32)), "bfac_res": o["bfac_res"],
            "res_off": np.ascontiguousarray(o["res_off"].astype(np.uint32))}
    r = bench.cpu_baseline(hb, 25, gpu=side)
    lr = r["live_reference"]
    assert r["failed_chains"] == 0
    assert lr["records_equal"] == n and lr["coords_equal"] == n, lr


def test_restatement_equals_live_reference_on_input_variants():
    """and on every input variant of the differential fuzz (_cases.input_variants: what tests/test_gpu_parity_fuzz.py holds the GPU
    path to the restatement on) -- the variants the reference compresses (it refuses one-residue chains)"""
    import bench
    from _cases import input_variants
    rng = np.random.default_rng(20261001)
    threads = bench.effective_cores()
    checked = 0
    for name, hb in input_variants(rng, 12):
        blob, off, st = H.oracle_compress(hb, n_threads=threads)
        if not (st == 0).all():
            continue
        o = H.oracle_decompress(blob, off, n_threads=threads)
        side = {"blob": np.ascontiguousarray(blob), "off": np.ascontiguousarray(off.astype(np.uint64)), "x": o["x"], "y": o["y"], "z": o["z"],
                "atom_off": np.ascontiguousarray(o["atom_off"].astype(np.uint32)), "bfac_res": o["bfac_res"],
                "res_off": np.ascontiguousarray(o["res_off"].astype(np.uint32))}
        r = bench.cpu_baseline(hb, 25, gpu=side)
        lr = r["live_reference"]
        assert r["failed_chains"] == 0, name
        assert lr["records_equal"] == hb.n_chains and lr["coords_equal"] == hb.n_chains, (name, lr)
        checked += 1
    assert checked > 75, checked
