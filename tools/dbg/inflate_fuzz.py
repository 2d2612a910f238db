#!/usr/bin/env python3
"""k_inflate on texts the fixtures do not hold: the input variants of the differential fuzz rendered as PDB and mmCIF text, composite
files, binary noise, long runs, texts of every size around the kernel's windows -- each compressed with a random level (0-9),
strategy (default / filtered / Huffman only / RLE / fixed), window (9-15 bits) and memory level (1-9: many small blocks) -- against
zlib: the member inflates to zlib's bytes or is refused (none of these should be). usage (GPU box): python tools/dbg/inflate_fuzz.py [n] [seed]"""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import bench
from _cases import composite_pdb, input_variants, _variant_base
from foldcomp_amd.codec import Codec


def gz(data, rng):
    level = int(rng.integers(0, 10)); strategy = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
    c = zlib.compressobj(level, zlib.DEFLATED, 16 + int(rng.integers(9, 16)), int(rng.integers(1, 10)), strategy)
    return c.compress(data) + c.flush()


def run(N, seed, codec=None):
    rng = np.random.default_rng(seed)
    own = codec is None
    if own: codec = Codec(0)
    texts = []
    for name, b in input_variants(rng, N):
        blob, off, st = codec.compress_batch(b, strict=False)
        keep = [i for i in range(b.n_chains) if st[i] == 0][:N]
        if not keep:
            continue
        ent = [blob[off[i]:off[i + 1]].tobytes() for i in keep]
        eoff = np.zeros(len(ent) + 1, np.uint64); eoff[1:] = np.cumsum([len(e) for e in ent])
        t, s = codec.decompress_pdb(np.frombuffer(b"".join(ent), np.uint8).copy(), eoff)
        for i, x in enumerate(t):
            if s[i] == 0:
                texts.append(x)
                if i % 3 == 0:
                    try: texts.append(bench.cif_from_pdb_text(x, "V"))
                    except Exception: pass
    pool = _variant_base(rng, 30, 4, 160)
    texts += [composite_pdb(rng, pool) for _ in range(4 * N)]
    texts += [rng.integers(0, 256, int(n), dtype=np.uint8).tobytes() for n in rng.integers(1, 200000, 2 * N)]                      # noise: stored blocks, long codes
    texts += [bytes([int(rng.integers(0, 256))]) * int(n) for n in rng.integers(1, 300000, N)]                                     # one byte repeated: distance 1, length 258
    texts += [(b"ATOM  %5d" % k) * int(n) for k, n in enumerate(rng.integers(1, 20000, N))]
    texts += [texts[0][:n] for n in (0, 1, 2, 3, 63, 64, 65, 255, 256, 257, 4095, 4096, 4097, 8191, 8192, 8193, 32767, 32768, 32769, 65535, 65536, 65537)]
    members = [gz(t, rng) for t in texts]
    got, st = codec.inflate(members)
    bad = 0
    for i, (t, g, s) in enumerate(zip(texts, got, st)):
        if s != 0 or g != t:
            k = next((j for j in range(min(len(g), len(t))) if g[j] != t[j]), min(len(g), len(t)))
            print(f"member {i}: status {s}, {len(t)} bytes of text, first difference at {k}"); bad += 1
    if own: codec.close()
    print(f"{len(members)} members, {sum(map(len, texts))} bytes of text, {bad} differences or refusals")
    return len(members), bad


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 6, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
