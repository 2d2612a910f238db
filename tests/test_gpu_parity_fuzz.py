"""Differential fuzz of the codec against the C restatement on inputs the generator alone does not make (_cases.input_variants:
distorted geometry, translations, scalings, coarse precision, signed zeros, denormals, odd B-factors, every short length, one
residue type per batch, numbering): statuses, record bytes and the decoded coordinates of both atom orders, bit for bit.
Round 6: the first run of this fuzz found a decoded 0.0 with the wrong sign bit (vdiv3, foldcomp_amd/csrc/fcz_math.h).
tools/dbg/parity_fuzz.py runs the same at any size and seed (512 chains x 86 variants x seeds 2, 3: no difference)."""
import numpy as np
import pytest

import _harness as H
from _cases import input_variants

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _tool(name):
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location(name, os.path.join(root, "tools", "dbg", name + ".py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("seed", [20261001])
def test_input_variants_bit_exact(codec, seed):
    """statuses, record bytes, decoded coordinates of both atom orders; anchor thresholds 25 and, for every third variant, 200 and 7
    (tools/dbg/parity_fuzz.py holds the comparison: chains whose anchor count does not fit the header are refused here, see there)"""
    m = _tool("parity_fuzz")
    rng = np.random.default_rng(seed)
    n = 0; bad = 0
    for name, b in input_variants(rng, 64):
        for thr in ((25,) if n % 3 else (25, 200, 7)):
            bad += m.compare(codec, f"{name} (-b {thr})", b, thr)
        n += 1
    assert n > 80 and bad == 0


def test_pdb_text_and_extract_on_input_variants(codec):
    """k_pdb_format / k_extract on the same variants (columns that overflow, negative and huge B-factors, numbering beyond the
    columns): device text == the host restatement of the reference's writer (pinned to the live reference in
    tests/test_host_formats.py), pLDDT strings of 1-4 digits == the host's (tools/dbg/pdb_text_fuzz.py holds the loop)"""
    n, bad = _tool("pdb_text_fuzz").run(6, 20261001, codec)
    assert n > 80 and bad == 0


def test_ingest_of_rendered_input_variants(codec):
    """the variants as TEXT -- PDB (overflowing columns and all), AFDB-shaped and archive-shaped mmCIF -- through the device ingest:
    every file is read into the batch the host reader builds or handed back (tools/dbg/ingest_variants_fuzz.py)"""
    n, bad = _tool("ingest_variants_fuzz").run(4, 20261001, codec)
    assert n > 80 and bad == 0


def test_records_with_any_quantiser_parameters_decode_like_the_restatement(codec):
    """the twelve angle parameters of a record's header set to anything -- huge, tiny, negative, zero, NaN, infinite (angles of
    thousands of radians: glibc's large-argument sine / cosine, NaN and infinite angles): the device decodes every record to the
    restatement's bits, both atom orders (tools/dbg/param_fuzz.py; round 6: such angles used to decode to NaN here)"""
    n, bad = _tool("param_fuzz").run(20261001, codec, per_record=8)
    assert n > 400 and bad == 0


def test_inflate_of_rendered_variants_and_noise(codec):
    """k_inflate on texts the fixtures do not hold (the variants as PDB / mmCIF text, composite files, noise, runs, every size around
    the kernel's windows), each member made with a random level, strategy, window and memory level: zlib's bytes, none refused
    (tools/dbg/inflate_fuzz.py)"""
    n, bad = _tool("inflate_fuzz").run(2, 20261001, codec)
    assert n > 150 and bad == 0
