#!/usr/bin/env python3
"""k_inflate alone: N gzipped PDB (or mmCIF) texts of the bench generator's chains resident on the device, kernel time by HIP events.

  python tools/inflate_bench.py [--files 2048] [--residues 350] [--level 6] [--reps 5] [--cif]
prints one JSON line: inflated GB/s, compressed GB/s, residues/s of the inflate stage, refused members (must be 0)."""
import argparse
import ctypes
import gzip
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=2048)
    ap.add_argument("--residues", type=int, default=350)
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--distinct", type=int, default=256, help="distinct chains (the rest are repeats of them)")
    ap.add_argument("--cif", action="store_true")
    ap.add_argument("--stats", action="store_true", help="symbol statistics of member 0 (plain-Python decoder, seconds)")
    a = ap.parse_args()
    import torch
    torch.cuda.init()
    from foldcomp_amd import _lib, synthetic
    from foldcomp_amd.codec import Codec
    lib = _lib.load()
    nd = min(a.distinct, a.files)
    b = synthetic.to_chain_batch(synthetic.generate(nd, [a.residues] * nd, seed=11))
    with Codec(0) as codec:
        blob, off, st = codec.compress_batch(b)
        texts, _ = codec.decompress_pdb(blob, off)
        texts = [t if isinstance(t, bytes) else t.encode() for t in texts]
        if a.cif:
            sys.path.insert(0, ROOT)
            from bench import cif_from_pdb_text
            texts = [cif_from_pdb_text(t, f"S{i:07d}") for i, t in enumerate(texts)]
        with ThreadPoolExecutor(os.cpu_count() or 4) as ex:
            gz = list(ex.map(lambda t: gzip.compress(t, a.level), texts))
        members = [gz[i % nd] for i in range(a.files)]
        n = len(members)
        goff = np.zeros(n + 1, np.uint64); goff[1:] = np.cumsum([len(m) for m in members])
        raw = np.frombuffer(b"".join(members), np.uint8)
        toff = np.zeros(n + 1, np.uint64)
        _lib.check(lib.fcz_inflate_sizes(raw.ctypes.data, goff.ctypes.data, n, None, toff.ctypes.data), "sizes")
        d_raw = torch.from_numpy(raw.copy()).cuda(); d_goff = torch.from_numpy(goff.view(np.int64)).cuda(); d_toff = torch.from_numpy(toff.view(np.int64)).cuda()
        d_text = torch.empty(int(toff[n]) + 64, dtype=torch.uint8, device="cuda"); d_st = torch.zeros(n, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        codec.enable_timing(True)
        call = lambda: _lib.check(lib.fcz_inflate_dev(codec.ctx, d_raw.data_ptr(), d_goff.data_ptr(), n, None, d_toff.data_ptr(), d_text.data_ptr(), d_st.data_ptr()), "inflate")
        call(); codec.synchronize(); codec.reset_timing()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            call()
        codec.synchronize()
        wall = (time.perf_counter() - t0) / a.reps
        ms, launches = codec.kernel_time("inflate")
        ms /= max(launches, 1)
        refused = int((d_st != 0).sum())
        got = d_text[: int(toff[n])].cpu().numpy().tobytes()
        ok = all(got[int(toff[i]):int(toff[i + 1])] == texts[i % nd] for i in range(0, n, max(1, n // 64)))
        extra = {}
        if a.stats:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from inflate_stats import inflate_stats
            st0, dh, _, _ = inflate_stats(members[0])
            sym = st0["literals"] + st0["matches"]
            extra = {"member0": {"symbols": sym, "literals": st0["literals"], "matches": st0["matches"], "blocks": st0["blocks_type2"],
                                 "bytes_per_symbol": round(st0["out_bytes"] / sym, 2), "matches_beyond_8K": sum(v for k, v in dh.items() if k >= 14),
                                 "matches_beyond_16K": sum(v for k, v in dh.items() if k >= 15)},
                     "ns_per_symbol_per_wave_if_all_resident": round(ms * 1e6 / sym, 1)}
        print(json.dumps({**extra, "files": n, "residues_per_file": a.residues, "level": a.level, "format": "cif" if a.cif else "pdb",
                          "gz_bytes": int(goff[n]), "text_bytes": int(toff[n]), "ratio": round(int(toff[n]) / int(goff[n]), 2),
                          "kernel_ms": round(ms, 3), "wall_ms": round(wall * 1e3, 3), "inflated_GB_per_s": round(int(toff[n]) / ms / 1e6, 1),
                          "compressed_GB_per_s": round(int(goff[n]) / ms / 1e6, 1), "residues_per_s": round(n * a.residues / ms * 1e3),
                          "refused": refused, "sampled_texts_equal": ok}))


if __name__ == "__main__":
    main()
