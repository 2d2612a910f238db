"""GPU: the reference's own known-answer pins (build.sh:25-38) through this repo's hosts and the HIP codec:
    compress test/test.pdb    -> decompress     -> all-atom RMSD vs the input = 0.0826751 (+- 0.001)
    compress test/test.cif.gz -> decompress -a  -> all-atom RMSD vs the input = 0.130284  (+- 0.001)
(atoms paired in file order, no superposition: the reference's `rmsd` subcommand), plus bit-exactness of the mmCIF case against
the goldens minted from the real reference (tests/golden/reference_ingest.npz)."""
import os
import subprocess

import numpy as np
import pytest

from foldcomp_amd.__main__ import load_structure
from foldcomp_amd.structure import Chain, build_batch, identify_chains, identify_discontinuous, remove_alternative_position

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "host", "foldcomp-hip")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ing():
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_ingest.npz"))


def _one_fragment_batch(fn, data):
    t, title = load_structure(fn, data)
    t = remove_alternative_position(t)
    chains = identify_chains(t)
    frags = identify_discontinuous(t, chains[0])
    assert len(chains) == 1 and len(frags) == 1
    stem = fn.rsplit(".", 1)[0]
    return t, build_batch([Chain(stem if title == fn else title, t.take(frags[0]))], 25)


@pytest.mark.parametrize("fn,alt,pin", [("test.pdb", False, 0.0826751), ("test.cif.gz", True, 0.130284)])
def test_reference_rmsd_pins(codec, ing, fn, alt, pin):
    t, b = _one_fragment_batch(fn, ing[f"file:{fn}"].tobytes())
    blob, off, st = codec.compress_batch(b)
    assert st[0] == 0
    d = codec.decompress_batch(blob, off, alt_order=alt)
    assert len(d["x"]) == len(t)
    got = np.stack([d["x"], d["y"], d["z"]], 1).astype(np.float64)
    rmsd = float(np.sqrt(((got - t.xyz.astype(np.float64)) ** 2).sum(1).mean()))
    assert abs(rmsd - pin) < 1e-3, (rmsd, pin)


def test_mmcif_case_bit_exact_against_the_reference(codec, ing):
    """test.cif.gz: FCZ bytes == Foldcomp::compress of the reference's own atom table, coordinates == Foldcomp::decompress"""
    t, b = _one_fragment_batch("test.cif.gz", ing["file:test.cif.gz"].tobytes())
    blob, off, st = codec.compress_batch(b)
    assert blob.tobytes() == ing["cif:test/fcz"].tobytes()
    for alt in (0, 1):
        d = codec.decompress_batch(blob, off, alt_order=bool(alt))
        got = np.stack([d["x"], d["y"], d["z"]], 1)
        assert np.array_equal(got.view(np.uint32), ing[f"cif:test/xyz{alt}"].view(np.uint32)), alt


def test_cpp_cli_rmsd_pins(ing, tmp_path):
    """the same two pins through the C++ command line: compress -> decompress [-a] -> rmsd (column 6 of its output, as build.sh cuts it)"""
    for fn, alt, pin in (("test.pdb", False, 0.0826751), ("test.cif.gz", True, 0.130284)):
        src = tmp_path / fn
        src.write_bytes(ing[f"file:{fn}"].tobytes())
        fcz, out = tmp_path / (fn + ".fcz"), tmp_path / (fn + ".out.pdb")
        r = subprocess.run([BIN, "compress", "-y", str(src), str(fcz)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and fcz.exists(), r.stderr
        r = subprocess.run([BIN, "decompress", "-y"] + (["-a"] if alt else []) + [str(fcz), str(out)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and out.exists(), r.stderr
        r = subprocess.run([BIN, "rmsd", str(src), str(out)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        rmsd = float(r.stdout.strip().split("\t")[5])
        assert abs(rmsd - pin) < 1e-3, (fn, rmsd, pin)
