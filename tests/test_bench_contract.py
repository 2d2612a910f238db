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
    for k in ("k_compress_angles", "k_backbone", "k_sidechain", "k_compress_pack", "k_res_index"):
        assert k in t["kernels"], k
        traffic, src = b.measured_traffic(k, 350_000_000, 350)
        assert traffic and traffic > 0 and "pmc" in src
    # other chain lengths have no measured profile: null, never a made-up number
    assert b.measured_traffic("k_compress_angles", 1000, 123) == (None, None)
    assert b.measured_traffic("no_such_kernel", 1000, 350) == (None, None)
