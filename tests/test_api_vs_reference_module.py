"""The Python surface, call by call, against the reference's OWN extension module (oracle/_ref/pymod/foldcomp*.so: foldcomp/foldcomp.cxx
compiled from where it lies by oracle/build_ref.sh -- test infrastructure only). Both modules are called `foldcomp`, so each runs in
a subprocess of its own with tests/_api_probe.py: the same calls, every result (FCZ bytes with the four uninitialised header bytes
masked, floats bit for bit) and every exception (type and text) as JSON; the two documents must agree."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PYMOD = os.path.join(ROOT, "oracle", "_ref", "pymod")

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not glob.glob(os.path.join(PYMOD, "foldcomp*.so")), reason="oracle/_ref/pymod not built")]


def _probe(path_first, tmp_path, tag, extra=None, fork=False):
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env["PYTHONPATH"] = path_first
    work = tmp_path / tag
    work.mkdir()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_api_probe.py"), os.path.join(ROOT, "tests", "golden", "reference_ingest.npz"), str(work)] + ([str(extra)] if extra else []) + (["fork"] if fork else []),
                       capture_output=True, text=True, timeout=900, env=env, cwd=str(work))
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout[r.stdout.index("{"):]), r.stderr


# what the drop-in does differently ON PURPOSE (each stated where it is implemented):
KNOWN = {
    # split_pdb_by_chain lives in the reference's pure-Python wrapper (foldcomp/util.py), not in the extension probed here
    "compress_chains",
    # a residue without its N: the reference counts backbone atoms / 3 and writes a record of 27 "residues" whose atoms are shifted by
    # one against their residues; the contract here refuses what does not compress to a meaningful record
    "compress_missing_backbone_atom",
}


def test_every_call_agrees_with_the_reference_module(tmp_path):
    ref, _ = _probe(PYMOD, tmp_path, "ref")
    mine, _ = _probe(ROOT, tmp_path, "mine")
    assert sorted(ref) == sorted(mine)
    diff = {}
    for k in ref:
        if k in KNOWN:
            continue
        a, b = ref[k], mine[k]
        if a[0] == "raised" and b[0] == "raised":
            if a[1] != b[1]:
                diff[k] = (a, b)                                              # the exception's type; its text is compared where the reference's is its own
            elif a[1] in ("foldcomp.error", "KeyError", "IndexError") and a[2] != b[2]:
                diff[k] = (a, b)
        elif a != b:
            diff[k] = (str(a)[:300], str(b)[:300])
    assert not diff, json.dumps(diff, indent=1)[:6000]
    assert mine["compress_missing_backbone_atom"][:2] == ["raised", "foldcomp.error"] and ref["compress_missing_backbone_atom"][0] == "ok"


def test_rendered_variants_agree_with_the_reference_module(tmp_path):
    """the module's own reader (ATOM lines of a string, foldcomp/foldcomp.cxx) and the codec behind compress / decompress / get_data
    on the input variants of the differential fuzz written as PDB text (_cases.input_variants: "-0.000", columns that overflow or
    run together, odd B-factors, short chains, UNK, long chains ...) and on composite files (several chains, gaps, alternative
    locations, insertion codes, waters): every result and every exception equal to the reference module's"""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import host_text
    from _cases import composite_pdb, input_variants, _variant_base
    rng = np.random.default_rng(20261001)
    files = {}
    for vi, (name, b) in enumerate(input_variants(rng, 4)):
        if name.startswith("side chains: extra"):
            continue
        res_of_atom = np.repeat(np.arange(b.n_residues), np.diff(b.atom_off.astype(np.int64)))
        for c in range(min(b.n_chains, 2)):
            r0, r1 = int(b.res_off[c]), int(b.res_off[c + 1])
            if r1 - r0 > 700:
                continue
            sl = slice(int(b.atom_off[r0]), int(b.atom_off[r1]))
            text = host_text.format_pdb(f"V{vi}", b.atom_code[sl], b.res_code[res_of_atom[sl]], int(b.first_res_index[c]) + res_of_atom[sl] - r0,
                                        chr(b.chain_id[c]) if 32 < b.chain_id[c] < 127 else "A", int(b.first_atom_index[c]), b.x[sl], b.y[sl], b.z[sl], b.bfac_ca[res_of_atom[sl]])
            files[f"v{vi:03d}_{c}"] = np.frombuffer(text.encode("latin-1"), np.uint8)
    pool = _variant_base(rng, 30, 5, 120)
    for i in range(30):
        files[f"w{i:02d}"] = np.frombuffer(composite_pdb(rng, pool, f"COMPOSITE {i}"), np.uint8)
    assert len(files) > 150
    np.savez(tmp_path / "extra.npz", **files)
    ref, _ = _probe(PYMOD, tmp_path, "ref", tmp_path / "extra.npz", fork=True)
    mine, _ = _probe(ROOT, tmp_path, "mine", tmp_path / "extra.npz")
    # texts on which the reference's module ends the interpreter (std::stof on a field that is no number: columns that overflowed):
    # the drop-in must answer with an exception, nothing more can be compared
    ended = {k.split(":", 1)[1] for k, v in ref.items() if k.startswith("x_compress:") and v[0] == "process ended"}
    for n in ended:
        assert mine["x_compress:" + n][0] == "raised", (n, mine["x_compress:" + n][:2])
    split = {n: (ref["x_compress:" + n][:3] if ref["x_compress:" + n][0] != "ok" else "ok", mine["x_compress:" + n][:3] if mine["x_compress:" + n][0] != "ok" else "ok")
             for n in files if n not in ended and (ref["x_compress:" + n][0] == "ok") != (mine["x_compress:" + n][0] == "ok")}
    assert not split, (len(split), json.dumps(dict(list(split.items())[:8]), indent=1)[:3000])
    keys = [k for k in ref if k.startswith("x_") and k.split(":", 1)[1] not in ended]
    assert sorted(keys) == sorted(k for k in mine if k.startswith("x_") and k.split(":", 1)[1] not in ended) and len(keys) > 400, (len(keys), len(ended))
    diff = {}
    for k in keys:
        a, b = ref[k], mine[k]
        if a[0] == "raised" and b[0] == "raised":
            if a[1] != b[1]:
                diff[k] = (a, b)
        elif a != b:
            diff[k] = (str(a)[:200], str(b)[:200])
    # the stated difference (KNOWN above): what does not make a meaningful record is refused here -- residues
    # with a second N / CA / C (insertion codes and alternative chains read by number only), a residue without its backbone --
    # where the reference computes on the shifted atoms. Only the composite files have such residues
    refused = {k for k, (a, b) in diff.items() if "'ok'" in a[:8] and ("StructureError" in b or "foldcomp.error" in b)}
    assert all(k.split(":", 1)[1].startswith("w") for k in refused), sorted(refused)[:10]
    rest = {k: v for k, v in diff.items() if k not in refused}
    print({"inputs": len(files), "reference process ended on": len(ended), "results compared": len(keys), "refused here by contract (composite files)": len(refused), "differences": len(rest)})
    assert not rest, (len(rest), json.dumps(dict(list(rest.items())[:6]), indent=1)[:4000])
