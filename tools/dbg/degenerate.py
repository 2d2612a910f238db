"""degenerate coordinates through compress: device against the oracle (run on the GPU box)"""
import sys, numpy as np
sys.path[:0] = ['/root/repo', '/root/repo/tests', '/root/repo/oracle']
import _harness as H
from foldcomp_amd import synthetic
from foldcomp_amd.codec import Codec
c = Codec(0)
def run(tag, mut, lens=(40, 350, 90), seed=3):
    b = synthetic.to_chain_batch(synthetic.generate(len(lens), list(lens), seed=seed))
    x, y, z = b.x.copy(), b.y.copy(), b.z.copy()
    mut(b, x, y, z)
    b.x, b.y, b.z = x, y, z
    blob, off, st = c.compress_batch(b, strict=False) if 'strict' in c.compress_batch.__code__.co_varnames else c.compress_batch(b)
    oblob, ooff, ost = H.oracle_compress(b, n_threads=4)
    same = blob.tobytes() == oblob.tobytes()
    bad = [k for k in range(b.n_chains) if blob[off[k]:off[k+1]].tobytes() != oblob[ooff[k]:ooff[k+1]].tobytes()]
    print(tag, "status", st.tolist(), ost.tolist(), "same", same, "chains differing", bad)
    for k in bad[:2]:
        a_, o_ = blob[off[k]:off[k+1]], oblob[ooff[k]:ooff[k+1]]
        d = np.nonzero(a_ != o_)[0]
        print("   first diffs at", d[:12].tolist(), "n", len(d), "dev", a_[d[:8]].tolist(), "ora", o_[d[:8]].tolist())
def atoms_of(b, chain, res):
    r = b.res_off[chain] + res
    a0 = b.atom_off[r]
    return a0
def ca_eq_n(b, x, y, z):
    for ch in range(b.n_chains):
        a0 = atoms_of(b, ch, 5); x[a0+1], y[a0+1], z[a0+1] = x[a0], y[a0], z[a0]
def ca_eq_n_first(b, x, y, z):
    for ch in range(b.n_chains):
        a0 = atoms_of(b, ch, 0); x[a0+1], y[a0+1], z[a0+1] = x[a0], y[a0], z[a0]
def collinear(b, x, y, z):
    for ch in range(b.n_chains):
        a0 = atoms_of(b, ch, 7)
        x[a0+1], y[a0+1], z[a0+1] = x[a0] + 1.0, y[a0], z[a0]
        x[a0+2], y[a0+2], z[a0+2] = x[a0] + 2.0, y[a0], z[a0]
def all_zero_res(b, x, y, z):
    for ch in range(b.n_chains):
        a0 = atoms_of(b, ch, 3); a1 = atoms_of(b, ch, 4)
        x[a0:a1] = 0; y[a0:a1] = 0; z[a0:a1] = 0
def huge(b, x, y, z):
    for ch in range(b.n_chains):
        a0 = atoms_of(b, ch, 9); x[a0+1] = 1e30
def random_coords(b, x, y, z):
    rng = np.random.default_rng(1)
    x[:] = rng.normal(0, 10, len(x)).astype(np.float32).round(3); y[:] = rng.normal(0, 10, len(x)).astype(np.float32).round(3); z[:] = rng.normal(0, 10, len(x)).astype(np.float32).round(3)
def lattice(b, x, y, z):
    rng = np.random.default_rng(2)
    x[:] = rng.integers(-3, 4, len(x)).astype(np.float32); y[:] = rng.integers(-3, 4, len(x)).astype(np.float32); z[:] = rng.integers(-3, 4, len(x)).astype(np.float32)
for tag, m in (("none", lambda *a: None), ("CA==N res5", ca_eq_n), ("CA==N res0", ca_eq_n_first), ("collinear", collinear), ("zero residue", all_zero_res), ("huge", huge), ("random", random_coords), ("lattice", lattice)):
    try: run(tag, m)
    except Exception as e: print(tag, "EXC", repr(e)[:300])
