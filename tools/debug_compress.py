import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from _cases import golden_batch
from foldcomp_amd.codec import Codec
z = np.load(os.path.join(ROOT, "tests/golden/reference_vectors.npz"))
c = Codec(0)
for name in sys.argv[1:] or ["pdb:test_af"]:
    b = golden_batch(z, name)
    blob, off, st = c.compress_batch(b)
    exp = z[f"{name}/fcz"].tobytes(); got = blob.tobytes()
    n = int.from_bytes(exp[4:6], "little"); na = exp[12]; tl = int.from_bytes(exp[24:28], "little"); nsc = int.from_bytes(exp[16:20], "little")
    o_words = 76 + 4 * na + tl + 36 * na + 13
    regions = [("magic+hdr", 0, 28), ("mins", 28, 52), ("contfs", 52, 76), ("aidx", 76, 76 + 4 * na), ("title", 76 + 4 * na, 76 + 4 * na + tl),
               ("anchors", 76 + 4 * na + tl, o_words - 13), ("oxt", o_words - 13, o_words), ("words", o_words, o_words + 8 * n),
               ("sc", o_words + 8 * n, o_words + 8 * n + nsc), ("tmp", o_words + 8 * n + nsc, len(exp))]
    print(name, "status", st, "len", len(got), len(exp))
    for nm, a, e in regions:
        d = [i for i in range(a, min(e, len(got))) if got[i] != exp[i]]
        print("  %-10s %5d diffs %s" % (nm, len(d), [(i - a, got[i], exp[i]) for i in d[:6]]))
