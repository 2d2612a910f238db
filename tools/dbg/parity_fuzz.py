#!/usr/bin/env python3
"""Differential fuzz of the codec on the GPU against the C restatement (oracle/, itself pinned to the live reference): chains of the
generator put through transformations real inputs have and the generator does not -- distortions, translations, raw float
coordinates, odd B-factors, truncated side chains, unknown residues, tiny chains. Reports every difference (status, record bytes,
decoded coordinates in both atom orders). usage (GPU box): python tools/dbg/parity_fuzz.py [chains per variant] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _harness as H
from foldcomp_amd import synthetic
from foldcomp_amd.codec import Codec

bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 384
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1


from _cases import input_variants


def compare(codec, name, b, thr=25):
    b.anchor_threshold = thr
    try:
        blob, off, st = codec.compress_batch(b, strict=False)
    except Exception as e:
        print(f"[{name}] GPU compress raised {type(e).__name__}: {e}"); return 1
    oblob, ooff, ost = H.oracle_compress(b, n_threads=16)
    bad = 0
    if not np.array_equal(st, ost):
        w = np.flatnonzero(np.asarray(st) != np.asarray(ost)); print(f"[{name}] status differs for {len(w)} chains, first {w[:4]}: gpu {np.asarray(st)[w[:4]]} oracle {np.asarray(ost)[w[:4]]}"); bad += 1
    if not np.array_equal(off, ooff) or blob.tobytes() != oblob.tobytes():
        w = [c for c in range(b.n_chains) if not np.array_equal(off[c:c + 2] - off[c], ooff[c:c + 2] - ooff[c]) or blob[off[c]:off[c + 1]].tobytes() != oblob[ooff[c]:ooff[c + 1]].tobytes()]
        print(f"[{name}] records differ for {len(w)} chains of {b.n_chains}, first {w[:6]}"); bad += 1
        return bad
    for alt in (False, True):
        d = codec.decompress_batch(blob, off, alt_order=alt)
        o = H.oracle_decompress(oblob, ooff, alt_order=alt, n_threads=16)
        for k in ("x", "y", "z", "bfac_res"):
            m = ~((bits(d[k]) == bits(o[k])) | (np.isnan(d[k]) & np.isnan(o[k])))
            if m.any():
                i = np.flatnonzero(m)
                print(f"[{name}] alt={alt} {k}: {len(i)} values differ, first at {i[:4]}: gpu {d[k][i[:4]]} oracle {o[k][i[:4]]}"); bad += 1
    return bad


def main():
    rng = np.random.default_rng(SEED)
    total = 0; n = 0
    with Codec(0) as codec:
        for name, b in input_variants(rng, N):
            for thr in ((25,) if n % 3 else (25, 200, 7)):
                total += compare(codec, name + f" (-b {thr})", b, thr)
            n += 1
    print(f"{n} variants, {total} differences")


if __name__ == "__main__":
    main()
