#!/usr/bin/env python3
"""PDB text written on the device (k_pdb_format) against the host restatement of the reference's writer (oracle/host_text.py, pinned
to the live reference in tests/test_host_formats.py) on the input variants of the differential fuzz (_cases.input_variants: huge
coordinates that overflow their columns, negative and huge B-factors, numbering beyond the columns, every chain id ...).
usage (GPU box): python tools/dbg/pdb_text_fuzz.py [chains per variant] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import _harness as H
from _cases import input_variants
from foldcomp_amd import fczfile
from foldcomp_amd.codec import Codec
from host_text import pdb_from_result, extract_plddt

def run(N, seed, codec=None):
    rng = np.random.default_rng(seed)
    own = codec is None
    if own: codec = Codec(0)
    bad = 0; n = 0
    if True:
        for name, b in input_variants(rng, N):
            n += 1
            blob, off, st = codec.compress_batch(b, strict=False)
            keep = [i for i in range(b.n_chains) if st[i] == 0]
            if not keep:
                continue
            entries = [blob[off[i]:off[i + 1]].tobytes() for i in keep]
            eoff = np.zeros(len(entries) + 1, np.uint64); eoff[1:] = np.cumsum([len(e) for e in entries])
            eblob = np.frombuffer(b"".join(entries), np.uint8).copy()
            for digits in (1, 2, 3, 4):
                got = codec.extract(eblob, eoff, mode=0, digits=digits)
                for i, e in enumerate(entries):
                    try:
                        exp_p = extract_plddt(fczfile.parse(e), digits)
                    except ValueError:                       # (a NaN B-factor: the reference converts it to a char, which C leaves undefined)
                        continue
                    if got[i].decode("latin-1") != exp_p:
                        print(f"[{name}] extract -p {digits} chain {i}: device {got[i][:60]!r} host {exp_p[:60]!r}"); bad += 1; break
            for alt in (False, True):
                texts, status = codec.decompress_pdb(eblob, eoff, alt_order=alt)
                o = H.oracle_decompress(eblob, eoff, alt_order=alt, n_threads=16)
                for i, (t, e) in enumerate(zip(texts, entries)):
                    exp = pdb_from_result(fczfile.parse(e), o, i, alt).encode("latin-1")
                    if status[i] != 0 or t != exp:
                        k = next((k for k in range(min(len(t), len(exp))) if t[k] != exp[k]), None)
                        print(f"[{name}] alt={alt} chain {i}: status {status[i]}, first difference at byte {k}: device {t[max(0, (k or 0) - 40):(k or 0) + 40]!r} host {exp[max(0, (k or 0) - 40):(k or 0) + 40]!r}")
                        bad += 1
                        break
    if own: codec.close()
    print(f"{n} variants, {bad} differences")
    return n, bad



if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 24, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
