import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import _harness as H
from _cases import distorted_batch
from foldcomp_amd.codec import Codec
bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
with Codec(0) as codec:
    for sigma in (0.02, 0.1, 0.3, 1.5):
        b = distorted_batch(768, sigma, seed=int(sigma * 1000) + 11)
        blob, off, st = codec.compress_batch(b)
        oblob, ooff, ost = H.oracle_compress(b, n_threads=8)
        print(sigma, "status eq", np.array_equal(st, ost), "bad status", int((st != 0).sum()), "blob eq", blob.tobytes() == oblob.tobytes())
        for alt in (False, True):
            d = codec.decompress_batch(blob, off, alt_order=alt)
            o = H.oracle_decompress(oblob, ooff, alt_order=alt, n_threads=8)
            for k in ("x", "y", "z", "bfac_res"):
                bad = ~((bits(d[k]) == bits(o[k])) | (np.isnan(d[k]) & np.isnan(o[k])))
                if bad.any():
                    idx = np.flatnonzero(bad)
                    if k != "bfac_res":
                        aoff = np.asarray(o["atom_off"]).astype(np.int64); roff = np.asarray(o["res_off"]).astype(np.int64)
                        res = np.searchsorted(aoff, idx, side="right") - 1
                        ch = np.searchsorted(roff, res, side="right") - 1
                        within = idx - aoff[res]
                        print(" ", sigma, alt, k, "mismatches", len(idx), "chains", len(np.unique(ch)), "first", idx[:5], "chain", ch[:5], "res in chain", (res - roff[ch])[:5], "of", (roff[ch + 1] - roff[ch])[:5], "atom in res", within[:5],
                              "gpu", d[k][idx[:3]], "oracle", o[k][idx[:3]], "maxdiff", float(np.nanmax(np.abs(d[k][idx] - o[k][idx]))))
                        u, cnt = np.unique(within, return_counts=True); print("    atom-in-res histogram", dict(zip(u.tolist(), cnt.tolist())))
                    else:
                        print(" ", sigma, alt, k, "mismatches", len(idx))
