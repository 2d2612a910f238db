"""GPU: the HIP path (through the C-ABI) against the oracle and the reference-minted goldens.
Bar: FCZ bytes bit-exact; decompressed float32 coordinates bit-exact (the device restates glibc's
sinf/cosf and rounds acos exactly as the host does, see fcz_math.h)."""
import numpy as np
import pytest

import _harness as H
from _cases import compress_cases, concat_batches, db_cases, entries_blob, golden_batch

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _first_diff(a: bytes, b: bytes):
    n = min(len(a), len(b))
    d = [i for i in range(n) if a[i] != b[i]]
    return len(d), d[:16]


def test_compress_golden_bit_exact(codec, golden):
    z, index = golden
    for name in compress_cases(index):
        b = golden_batch(z, name)
        blob, off, st = codec.compress_batch(b)
        assert st[0] == 0, name
        exp = z[f"{name}/fcz"].tobytes()
        assert blob.tobytes() == exp, (name, _first_diff(blob.tobytes(), exp))


def test_compress_golden_as_one_batch(codec, golden):
    """all golden chains with threshold 25 in a single launch (ragged batch)"""
    z, index = golden
    names = [n for n in compress_cases(index) if int(z[f"{n}/in/anchor_threshold"][0]) == 25]
    b = concat_batches([golden_batch(z, n) for n in names])
    blob, off, st = codec.compress_batch(b)
    assert (st == 0).all()
    for i, n in enumerate(names):
        assert blob[off[i]:off[i + 1]].tobytes() == z[f"{n}/fcz"].tobytes(), n


def test_decompress_golden_bit_exact(codec, golden):
    z, index = golden
    names = compress_cases(index) + db_cases(index)
    for alt in (0, 1):
        use = [n for n in names if f"{n}/xyz{alt}" in z]
        blob, off = entries_blob([z[f"{n}/fcz"].tobytes() for n in use])
        d = codec.decompress_batch(blob, off, alt_order=bool(alt))
        worst = 0.0
        for i, n in enumerate(use):
            assert d["info"][i].status == 0, n
            a0, a1 = d["atom_off"][i], d["atom_off"][i + 1]
            exp = z[f"{n}/xyz{alt}"]
            assert a1 - a0 == len(exp), n
            got = np.stack([d["x"][a0:a1], d["y"][a0:a1], d["z"][a0:a1]], 1)
            worst = max(worst, float(np.abs(got - exp).max()))
            assert np.array_equal(_bits(got), _bits(exp)), (n, alt, float(np.abs(got - exp).max()))
            r0, r1 = d["res_off"][i], d["res_off"][i + 1]
            # EVERY residue's B-factor equals the reference's per-atom B-factor of that residue's atoms (the chain's OXT
            # carries the last residue's, src/foldcomp.cpp:893-900)
            per_atom = z[f"{n}/bfac"]
            nat = np.asarray([codec.lib.fcz_res_code_natoms(int(c)) for c in d["res_code"][r0:r1]])
            rep = np.repeat(d["bfac_res"][r0:r1], nat)
            if len(per_atom) == len(rep) + 1:
                rep = np.append(rep, d["bfac_res"][r1 - 1])
            assert len(rep) == len(per_atom), n
            assert np.array_equal(_bits(rep), _bits(per_atom)), n


def test_synthetic_batch_vs_oracle(codec):
    """seeded ragged synthetic batch: HIP == oracle for compress bytes and decompress bits"""
    from foldcomp_amd import synthetic
    lens = [350] * 40 + [2, 3, 17, 25, 26, 63, 64, 65, 128, 129, 200, 511, 512, 513, 900, 1500]
    b = synthetic.to_chain_batch(synthetic.generate(len(lens), lens, seed=20260926))
    blob, off, st = codec.compress_batch(b)
    oblob, ooff, ost = H.oracle_compress(b, n_threads=8)
    assert (st == 0).all() and (ost == 0).all()
    assert np.array_equal(off, ooff)
    if blob.tobytes() != oblob.tobytes():
        bad = [c for c in range(b.n_chains) if blob[off[c]:off[c + 1]].tobytes() != oblob[off[c]:off[c + 1]].tobytes()]
        c = bad[0]
        raise AssertionError((bad, _first_diff(blob[off[c]:off[c + 1]].tobytes(), oblob[off[c]:off[c + 1]].tobytes())))
    d = codec.decompress_batch(blob, off)
    o = H.oracle_decompress(oblob, ooff, n_threads=8)
    assert np.array_equal(d["atom_off"], o["atom_off"]) and np.array_equal(d["res_off"], o["res_off"])
    for k in ("x", "y", "z", "bfac_res"):
        assert np.array_equal(_bits(d[k]), _bits(o[k])), (k, float(np.abs(d[k] - o[k]).max()))
    assert np.array_equal(d["res_code"], o["res_code"]) and np.array_equal(d["atom_code"], o["atom_code"])
    # round-trip quality: the reference pins an all-atom RMSD of ~0.08 A on real structures; synthetic
    # chains with random side-chain torsions stay well below 0.5 A
    # (atoms matched by (residue, atom name): the decoder's default order is the canonical one, the input has its own)
    assert len(d["x"]) == b.n_atoms
    def by_name(res_starts, codes, n_atoms):
        res_of = np.searchsorted(np.asarray(res_starts[1:], np.int64), np.arange(n_atoms), side="right")
        return np.lexsort((codes, res_of))
    # same residue -> atom partition on both sides (a chain's OXT closes its last residue in the input and in the output)
    oi = by_name(np.asarray(b.atom_off, np.int64), b.atom_code, b.n_atoms)
    gi = by_name(np.asarray(b.atom_off, np.int64), d["atom_code"], b.n_atoms)
    assert np.array_equal(np.asarray(b.atom_code)[oi], d["atom_code"][gi])
    dx = np.stack([d["x"][gi] - b.x[oi], d["y"][gi] - b.y[oi], d["z"][gi] - b.z[oi]], 1)
    rmsd = float(np.sqrt((dx.astype(np.float64) ** 2).sum(1).mean()))
    assert rmsd < 0.5, rmsd


def test_bad_entries_are_skipped_not_crashing(codec, golden):
    z, index = golden
    good = z["pdb:test_af/fcz"].tobytes()
    entries = [good, b"XXXX" + good[4:], good[:90], good]
    blob, off = entries_blob(entries)
    d = codec.decompress_batch(blob, off)
    st = [d["info"][i].status for i in range(4)]
    assert st == [0, -4, -5, 0]
    n = len(z["pdb:test_af/xyz0"])
    assert list(d["atom_off"]) == [0, n, n, n, 2 * n]
    assert np.array_equal(_bits(d["x"][:n]), _bits(d["x"][n:]))


def test_invalid_chain_reports_status(codec, golden):
    z, _ = golden
    b = golden_batch(z, "pdb:test_af")
    b.res_code = b.res_code.copy(); b.res_code[3] = 21   # GLX: the reference aborts on it
    blob, off, st = codec.compress_batch(b, strict=False)
    assert st[0] == -6
    assert not blob.any()


def test_device_resident_entry_points_vs_oracle(codec):
    """the *_dev entry points on device buffers (the calls bench.py times): sizes -> compress -> sizes -> decompress,
    mixed lengths incl. chains longer than the register path of the pack kernel; results against the oracle"""
    import ctypes
    import torch
    from foldcomp_amd import _lib, synthetic
    from foldcomp_amd.structure import CAtomsOut
    import bench
    lens = [350] * 70 + [2, 5, 64, 65, 383, 384, 385, 700, 1100, 40, 41, 16] + list(synthetic.mixed_lengths(60, seed=3))
    d = synthetic.generate(len(lens), lens, seed=777, device="cuda:0")
    C = len(lens)
    R, M = int(d["res_off"][-1]), int(d["atom_off"][-1])
    lib = codec.lib
    cb = bench.c_batch(d)
    off_dev = torch.zeros(C + 1, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    _lib.check(lib.fcz_compress_sizes_dev(codec.ctx, ctypes.byref(cb), off_dev.data_ptr()), "sizes")
    codec.synchronize()
    blob_dev = torch.zeros(int(off_dev[-1]), dtype=torch.uint8, device="cuda:0")
    st_dev = torch.full((C,), -99, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    _lib.check(lib.fcz_compress_batch_dev(codec.ctx, ctypes.byref(cb), off_dev.data_ptr(), blob_dev.data_ptr(), st_dev.data_ptr()), "compress")
    codec.synchronize()
    hb = synthetic.to_chain_batch(d)
    oblob, ooff, ost = H.oracle_compress(hb, n_threads=8)
    assert (st_dev.cpu().numpy() == 0).all() and (ost == 0).all()
    assert np.array_equal(off_dev.cpu().numpy().astype(np.uint64), ooff)
    assert blob_dev.cpu().numpy().tobytes() == oblob.tobytes()
    for rep in range(2):   # second pass: sizes cache of the ctx is reused / refreshed
        res_off = torch.zeros(C + 1, dtype=torch.int32, device="cuda:0"); atom_off = torch.zeros(C + 1, dtype=torch.int32, device="cuda:0")
        out = {k: torch.zeros(M + 1, dtype=torch.float32, device="cuda:0") for k in ("x", "y", "z")}
        bf = torch.zeros(R, dtype=torch.float32, device="cuda:0"); rcod = torch.zeros(R, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        tr = ctypes.c_uint32(); ta = ctypes.c_uint32()
        _lib.check(lib.fcz_decompress_sizes_dev(codec.ctx, blob_dev.data_ptr(), off_dev.data_ptr(), C, res_off.data_ptr(), atom_off.data_ptr(),
                                                ctypes.byref(tr), ctypes.byref(ta)), "dsizes")
        assert tr.value == R and ta.value == M
        cout = CAtomsOut(out["x"].data_ptr(), out["y"].data_ptr(), out["z"].data_ptr(), bf.data_ptr(), rcod.data_ptr(), None)
        _lib.check(lib.fcz_decompress_batch_dev(codec.ctx, blob_dev.data_ptr(), off_dev.data_ptr(), C, res_off.data_ptr(), atom_off.data_ptr(),
                                                0, ctypes.byref(cout)), "decompress")
        codec.synchronize()
        o = H.oracle_decompress(oblob, ooff, n_threads=8)
        for k in ("x", "y", "z"):
            assert np.array_equal(_bits(out[k][:M].cpu().numpy()), _bits(o[k])), (rep, k)
        assert np.array_equal(_bits(bf.cpu().numpy()), _bits(o["bfac_res"]))
        assert np.array_equal(rcod.cpu().numpy(), o["res_code"])
