"""bench.py as the driver runs it (a subprocess, one JSON line on stdout), at a size that finishes in seconds: every object the
contract and DESIGN.md §6 name must be present, parity flags true, the host-pointer leg byte-identical to the resident path."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def line():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--chains", "16384", "--steps", "2", "--warmup", "1",
           "--cpu-sample", "512", "--parity-chains", "512", "--pdb-sample", "1024", "--mixed-chains", "6000", "--mixed-steps", "1",
           "--e2e-files", "0", "--host-chains", "4096"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(out) == 1, r.stdout[-2000:]
    return json.loads(out[0])


def test_contract_fields(line):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1 and line["vs_baseline"] is None
    assert line["numerics"] == "exact" and "workload" in line["config"] and "model" not in line["config"]
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert rf["kernel"] in rf["per_kernel_GBs"]
    cpu = line["cpu_baseline"]
    assert cpu["kind"] in ("reference", "port") and cpu["value"] > 0 and cpu["cores"] >= 1


def test_parity_and_properties(line):
    p = line["parity"]
    assert p["fcz_bit_exact"] and p["coords_bit_exact"] and p["bad_status"] == 0 and p["chains_checked"] == 512


def test_secondary_objects(line):
    assert line["decompress_only"]["residues_per_s"] > 0
    assert line["mixed"]["residues_per_s"] > 0 and line["mixed"]["chains_per_gpu"] == 6000
    alt = line["alt_numerics"]
    assert alt["mode"] == "fast" and alt["decompress_ms"] > 0


def test_host_pointer_leg_matches_the_resident_path(line):
    h = line["host_boundary"]
    assert h["chains"] == 4096 and h["fcz_equals_resident_path"] and h["coords_filled"]
    assert h["residues_per_s"] > 0 and h["compress_link_GBs"] > 0 and h["decompress_link_GBs"] > 0
    # over the link every residue costs its objects twice and its record twice: far above the resident path's time
    assert h["residues_per_s"] < line["value"]
