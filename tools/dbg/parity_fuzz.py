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


from _cases import input_variants


def compare(codec, name, b, thr=25):
    b.anchor_threshold = thr
    try:
        blob, off, st = codec.compress_batch(b, strict=False)
    except Exception as e:
        print(f"[{name}] GPU compress raised {type(e).__name__}: {e}"); return 1
    oblob, ooff, ost = H.oracle_compress(b, n_threads=16)
    bad = 0
    # the one deliberate difference (tests/test_gpu_edge_cases.py::test_chain_beyond_header_counts_is_refused): a chain whose anchor
    # count n / thr + 2 does not fit the header's uint8 (or whose residue count does not fit its uint16) is refused here with a
    # zero-filled record; the reference wraps the counts and writes a record nobody can read. Those chains: refused, nothing else compared
    nres = np.diff(np.asarray(b.res_off).astype(np.int64))
    wraps = (nres // thr + 2 > 255) | (nres > 65535)
    st = np.asarray(st); ost = np.asarray(ost)
    if not np.array_equal(st[~wraps], ost[~wraps]) or not (st[wraps] == -1).all():
        w = np.flatnonzero((st != ost) & ~wraps); print(f"[{name}] status differs for {len(w)} chains, first {w[:4]}: gpu {st[w[:4]]} oracle {ost[w[:4]]}; wrapped chains not refused: {int((st[wraps] != -1).sum())}"); bad += 1
    if not np.array_equal(off, ooff):
        print(f"[{name}] record sizes differ"); return bad + 1
    w = [c for c in range(b.n_chains) if not wraps[c] and blob[off[c]:off[c + 1]].tobytes() != oblob[ooff[c]:ooff[c + 1]].tobytes()]
    if w:
        print(f"[{name}] records differ for {len(w)} chains of {b.n_chains}, first {w[:6]}"); return bad + 1
    if wraps.any():
        keep = np.flatnonzero(~wraps & (st == 0))
        if len(keep) == 0:
            return bad
        ent = [blob[off[c]:off[c + 1]].tobytes() for c in keep]
        off = np.zeros(len(ent) + 1, np.uint64); off[1:] = np.cumsum([len(e) for e in ent]); ooff = off
        blob = oblob = np.frombuffer(b"".join(ent), np.uint8).copy()
    elif int(off[-1]) == 0:
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
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 384
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    total = 0; n = 0
    with Codec(0) as codec:
        for name, b in input_variants(rng, N):
            if os.environ.get("FUZZ_ONLY") and os.environ["FUZZ_ONLY"] not in name:
                n += 1; continue
            for thr in ((25,) if n % 3 else (25, 200, 7)):
                total += compare(codec, name + f" (-b {thr})", b, thr)
            n += 1
    print(f"{n} variants, {total} differences")


if __name__ == "__main__":
    main()
