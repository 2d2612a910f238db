#!/usr/bin/env python3
"""FCZ records whose angle quantiser parameters (the twelve floats of the header: minimum and step of phi, psi, omega and the three
bond angles) -- and, for a third of them, anchor coordinates and the OXT atom -- are not what a compressor writes -- huge, tiny, negative, zero, NaN, infinite: angles of thousands of radians
(glibc's sinf / cosf switch to their large-argument reduction at |x| >= 120), NaN and infinite angles. Device decode against the
restatement (libm on the host), both atom orders. usage (GPU box): python tools/dbg/param_fuzz.py [seed]"""
import os, struct, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import _harness as H
from _cases import entries_blob, golden_records
from foldcomp_amd.codec import Codec

VALUES = [0.0, -0.0, 1.0, -1.0, 0.5, 7.0, -7.0, 57.3, 180.0, -180.0, 360.0, 1e3, -1e3, 6875.5, 1e4, 1e5, -1e5, 1e6, 1e8, 1e12, 1e20, 3e38, -3e38, 1e-3, 1e-10, 1e-30, 1e-38, 1e-45,
          float("nan"), -float("nan"), float("inf"), -float("inf")]


def mutated(records, rng, per_record=24):
    out = []
    for e in records:
        for _ in range(per_record):
            b = bytearray(e)
            n_anchor = b[12]; tl = struct.unpack_from("<I", b, 24)[0]; o_anchor = 76 + 4 * n_anchor + tl
            if rng.random() < 0.35:                                                # anchor coordinates and the OXT atom: any float as well
                for _ in range(int(rng.integers(1, 4))):
                    k = int(rng.integers(0, 9 * n_anchor + 3))
                    at = o_anchor + 4 * k if k < 9 * n_anchor else o_anchor + 36 * n_anchor + 1 + 4 * (k - 9 * n_anchor)
                    struct.pack_into("<f", b, at, VALUES[int(rng.integers(0, len(VALUES)))])
                if rng.random() < 0.5:
                    out.append(bytes(b)); continue
            for _ in range(int(rng.integers(1, 4))):
                q = int(rng.integers(0, 12))                                   # mins at 28 + 4q (q < 6), steps at 52 + 4(q - 6)
                v = VALUES[int(rng.integers(0, len(VALUES)))] if rng.random() < 0.7 else float(np.float32(rng.normal(0, 1) * 10.0 ** rng.integers(-6, 9)))
                struct.pack_into("<f", b, 28 + 4 * q, v)
            out.append(bytes(b))
    return out


def run(seed, codec=None, per_record=24):
    z = np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"))
    index = bytes(z["index"]).decode().split("\n")
    recs = golden_records((z, index))
    rng = np.random.default_rng(seed)
    entries = mutated(recs, rng, per_record)
    own = codec is None
    if own: codec = Codec(0)
    bad = 0
    bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
    for alt in (False, True):
        blob, off = entries_blob(entries)
        d = codec.decompress_batch(blob, off, alt_order=alt)
        o = H.oracle_decompress(blob, off, alt_order=alt, n_threads=16)
        sd = [d["info"][i].status for i in range(len(entries))]; so = [o["info"][i].status for i in range(len(entries))]
        if sd != so:
            print("statuses differ", [(i, a, b) for i, (a, b) in enumerate(zip(sd, so)) if a != b][:5]); bad += 1; continue
        aoff = np.asarray(o["atom_off"]).astype(np.int64)
        for k in ("x", "y", "z"):
            m = ~((bits(d[k]) == bits(o[k])) | (np.isnan(d[k]) & np.isnan(o[k])))
            if m.any():
                i = np.flatnonzero(m); ent = np.searchsorted(aoff, i, side="right") - 1
                print(f"alt={alt} {k}: {len(i)} values differ in {len(np.unique(ent))} entries; first entry {ent[0]}: params", np.frombuffer(entries[ent[0]], np.float32, 12, 28),
                      "gpu", d[k][i[:3]], "oracle", o[k][i[:3]]); bad += 1
    if own: codec.close()
    print(f"{len(entries)} records, {bad} differences")
    return len(entries), bad


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
