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
           "--e2e-files", "384", "--e2e-passes", "8", "--host-chains", "4096"]
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


def test_overlapped_host_boundary(line):
    o = line["host_boundary"]["overlapped"]
    assert o["ctxs"] == 2 and o["fcz_equals_single_ctx"] and o["residues_per_s"] > 0 and o["round_trip_link_GBs_both_directions"] > 0


def test_end_to_end_both_directions(line):
    """disk -> FCZ database with the structure ingest on the device (== the host-parse pipeline's database, first record == the
    reference's), and FCZ database -> PDB-text database (first text == the reference's); each beside the reference's own loop"""
    e = line["end_to_end"]
    assert e["files"] == 384 and e["passes"] == 8
    c, d = e["compress"], e["decompress"]
    assert c["databases_identical"] and c["gpu_host"]["records"] == 384 * 8 and c["gpu_host"]["host_parsed_files"] == 0
    assert c["gpu_host"]["page_locked_blocks"] > 0 and c["gpu_host_parse"]["page_locked_blocks"] > 0
    assert c["gpu_host"]["steady_residues_per_s"] > 0 and d["gpu_host"]["steady_residues_per_s"] > 0
    if "cpu_reference" in c:      # oracle/_ref travels to the GPU box
        # EVERY record / text against the live reference's of the same name (pad bytes masked), not totals
        assert c["first_record_equals_reference"] and c["records_equal_reference_all"] and c["cpu_reference"]["failed_files"] == 0
        assert c["records_equal_reference"] == f"{384 * 8}/{384 * 8}"
        assert d["first_text_equals_reference"] and d["texts_equal_reference_all"] and d["cpu_reference"]["failed_entries"] == 0
        assert d["texts_equal_reference"] == "384/384"


def test_live_reference_hashes_in_the_line(line):
    """the chains the CPU baseline runs through the reference are compared with the GPU's, record by record and atom by atom"""
    if line["cpu_baseline"].get("kind") != "reference":
        pytest.skip("oracle/_ref not on this box")
    lr = line["cpu_baseline"]["live_reference"]
    assert lr["chains"] > 0 and lr["records_equal"] == lr["chains"] and lr["coords_equal"] == lr["chains"]
    assert line["parity"]["live_reference_equal"] is True and line["parity"]["live_reference_chains"] == lr["chains"]


def test_gz_leg_and_two_rank_decompress(line):
    """the measurement-only .gz rows, and the two-rank sharded decompress (TEST MODE) that writes its output once"""
    g = line["end_to_end"]["compress"]["gz"]
    assert "failed" not in g, g
    for k in ("pdb_gz", "cif_gz"):
        assert g[k]["gpu_host"]["steady_residues_per_s"] > 0 and g[k]["records_per_pass"] == g[k]["files"]
    t = line["end_to_end"]["sharded"]["decompress_two_ranks"]
    assert "failed" not in t, t
    assert t["world"] == 2 and t["data_written_once"] is True and t["database_equals_gpu_host"] and sum(t["records_per_rank"]) == t["records"]


def test_mmcif_leg(line):
    """the same chains as mmCIF files: parsed on the device, the database the host reader's pipeline writes"""
    c = line["end_to_end"]["compress"]["mmcif"]
    assert "failed" not in c, c
    assert c["databases_identical"] and c["gpu_host"]["host_parsed_files"] == 0 and c["gpu_host"]["records"] == c["files"] * c["passes"]


def test_sharded_driver_leg(line):
    """the sharded driver at N = 1 (1-rank group + engine + exchange) writes the database the bare engine writes, both directions"""
    sh = line["end_to_end"]["sharded"]
    assert "failed" not in sh, sh
    for mode in ("compress", "decompress"):
        assert sh[mode]["world"] == 1 and sh[mode]["database_equals_gpu_host"] and sh[mode]["steady_residues_per_s"] > 0
    assert sh["compress"]["records"] == 384 * 8


def test_resident_ingest_leg(line):
    i = line["pdb_text"]["ingest_of_the_same_text"]
    assert i["files"] == 1024 and i["counts_equal_decoded"] and i["refused"] == 0 and i["text_GBs"] > 0


def test_two_rank_line_is_creditable():
    """The N > 1 code path with two REAL ranks (gloo test mode: both on the box's one GPU, RCCL refuses that): the line carries a
    CPU baseline timed on rank 0, both ranks' parity checks enter the flags, their batches differ (seed ranges), the value counts
    both ranks' residues"""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["FCZ_BENCH_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--chains", "8192", "--steps", "2", "--warmup", "1",
           "--cpu-sample", "512", "--parity-chains", "700", "--parity-chunk", "256", "--pdb-sample", "0", "--mixed-chains", "3000",
           "--mixed-steps", "1", "--e2e-files", "0", "--host-chains", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(out) == 1, r.stdout[-2000:]
    ln = json.loads(out[0])
    assert ln["n_gpus"] == 2 and ln["config"]["backend"] == "gloo"
    assert ln["cpu_baseline"] is not None and ln["cpu_baseline"]["value"] > 0 and ln["cpu_baseline"]["cores"] >= 1
    p = ln["parity"]
    assert p["ranks_checked"] == 2 and p["chains_checked"] == 1400 and p["fcz_bit_exact"] and p["coords_bit_exact"] and p["bad_status"] == 0
    assert ln["properties"]["ranks_checked"] == 2 and ln["properties"]["deterministic"] and ln["properties"]["residue_counts_round_trip"]
    assert ln["mixed"]["parity"]["ranks_checked"] == 2 and ln["mixed"]["parity"]["fcz_bit_exact"] and ln["mixed"]["parity"]["coords_bit_exact"]
    assert abs(ln["value"] - 2 * 8192 * 350 / (ln["ms_per_step"] * 1e-3)) / ln["value"] < 1e-6
