"""GPU: the device-resident path at the shape of BASELINE configs[2] / configs[4] (the datasets cannot be fetched): 100 000 chains
with log-normal lengths (16 ... 2 700 residues, anchor -b 25) through compress + decompress in one batch.
 * a 4 096-chain sample against the oracle, bit for bit (FCZ bytes and coordinates);
 * size-independent properties of the whole batch: every chain OK, sizes pass == input counts, decode(encode(x)) within the
   reference's RMSD regime, deterministic blob, decompress-only repeatable."""
import os
import sys

import numpy as np
import pytest

import _harness as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mixed_workload(codec):
    import torch
    sys.path.insert(0, ROOT)
    import bench
    C = 100_000
    d = bench.generate_resident(C, 0, 25, 2048, "cuda:0", seed_base=977, mixed=True)
    w = bench.Workload(codec, d, "cuda:0")
    w.compress(); w.decompress(); codec.synchronize()
    yield bench, w, d, torch
    del w, d
    torch.cuda.empty_cache()


def test_mixed_100k_oracle_sample_bit_exact(mixed_workload):
    bench, w, d, torch = mixed_workload
    lens = (d["res_off"][1:] - d["res_off"][:-1]).cpu().numpy()
    assert len(lens) == 100_000 and lens.min() >= 16 and lens.max() > 1024     # the split long-chain path is exercised
    n = 4096                       # 1.2 M residues, 6.6 M side-chain torsion bytes: rare mis-rounded values would show
    pc = bench.parity_check(d, w, n, chunk=1024)
    assert pc["chains_checked"] == n
    assert pc["fcz_bit_exact"], f"FCZ bytes of the first 4096 chains differ from the oracle (first: chain {pc['first_fcz_mismatch_chain']})"
    assert pc["coords_bit_exact"], f"coordinates of the first 4096 chains differ from the oracle (first: chain {pc['first_coords_mismatch_chain']})"
    # the longest chains sit anywhere in the batch: check the 8 longest against the oracle too
    big = np.argsort(lens)[-8:]
    off = w.off_dev.cpu().numpy().astype(np.int64)
    ro = d["res_off"].cpu().numpy().astype(np.int64); ao = d["atom_off"].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    from foldcomp_amd import synthetic
    for c in big:
        r0, r1 = int(ro[c]), int(ro[c + 1]); a0, a1 = int(ao[r0]), int(ao[r1]); t0, t1 = int(d["title_off"][c]), int(d["title_off"][c + 1])
        sub = {k: d[k][a0:a1] for k in ("x", "y", "z", "atom_code")}
        sub.update({k: d[k][r0:r1] for k in ("res_code", "bfac_ca")})
        sub.update({k: d[k][c:c + 1] for k in ("first_res_index", "first_atom_index", "chain_id")})
        sub["res_off"] = d["res_off"][c:c + 2] - r0
        sub["atom_off"] = (d["atom_off"][r0:r1 + 1].to(torch.int64) & 0xFFFFFFFF) - a0
        sub["titles"] = d["titles"][t0:t1]; sub["title_off"] = d["title_off"][c:c + 2] - t0
        sub["anchor_threshold"] = d["anchor_threshold"]
        hb1 = synthetic.to_chain_batch(sub)
        oblob, ooff, ost = H.oracle_compress(hb1, n_threads=1)
        assert ost[0] == 0
        got = w.blob_dev[int(off[c]):int(off[c + 1])].cpu().numpy().tobytes()
        assert got == oblob.tobytes(), f"chain {c} ({lens[c]} residues)"


def test_mixed_100k_full_batch_properties(mixed_workload, codec):
    bench, w, d, torch = mixed_workload
    assert int((w.status_dev != 0).sum()) == 0
    assert torch.equal(w.res_off_dev, d["res_off"].to(torch.int32))
    assert int(w.atom_off_dev[-1]) & 0xFFFFFFFF == w.M
    c0 = w.checksum()
    x0 = w.out_t["x"].clone()
    w.compress(); w.decompress(); codec.synchronize()
    assert w.checksum() == c0, "compress is not deterministic"
    assert torch.equal(w.out_t["x"].view(torch.int32), x0.view(torch.int32)), "decompress is not deterministic"
    # decompress-only (the configs[2] operation) from the resident records, twice: same bits
    w.decompress(); codec.synchronize()
    assert torch.equal(w.out_t["x"].view(torch.int32), x0.view(torch.int32))
    rmsd, mx = w.round_trip_deviation()
    assert rmsd < 0.2, rmsd            # the reference pins ~0.08 A on real structures (build.sh:35)
    assert np.isfinite(mx)


def test_max_deviation_is_the_codecs_own(mixed_workload):
    """decode(encode(x)) is farthest from x where the synthetic side chains are ill-conditioned; on the
    oracle sample the farthest atom and its distance are the reference algorithm's own: same atom, same float."""
    bench, w, d, torch = mixed_workload
    n = 4096
    hb = bench.host_sample(d, n)
    oblob, ooff, ost = H.oracle_compress(hb, n_threads=8)
    o = H.oracle_decompress(oblob, ooff, alt_order=True, n_threads=8)
    w.decompress(alt_order=1); w.codec.synchronize()
    m = hb.n_atoms
    g = {k: w.out_t[k][:m].cpu().numpy() for k in ("x", "y", "z")}
    dev_g = np.sqrt((g["x"] - hb.x) ** 2 + (g["y"] - hb.y) ** 2 + (g["z"] - hb.z) ** 2)
    dev_o = np.sqrt((o["x"][:m] - hb.x) ** 2 + (o["y"][:m] - hb.y) ** 2 + (o["z"][:m] - hb.z) ** 2)
    assert int(dev_g.argmax()) == int(dev_o.argmax()) and float(dev_g.max()) == float(dev_o.max())
    assert np.array_equal(dev_g.view(np.uint32), dev_o.view(np.uint32))
    w.decompress(); w.codec.synchronize()


def test_configs2_full_size_decompress_only(codec):
    """BASELINE configs[2] at its size: 542 000 mixed-length records (the afdb_swissprot_v4 count; the dataset itself cannot be
    fetched) decompress-only from device-resident FCZ. A 4 096-chain sample == the oracle's decode bit for bit; every record OK;
    counts == the encoder's; repeatable; decode(encode(x)) in the reference's RMSD regime."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    C = 542_000
    d = bench.generate_resident(C, 0, 25, 4096, "cuda:0", seed_base=31337, mixed=True)
    w = bench.Workload(codec, d, "cuda:0")
    w.compress(); codec.synchronize()
    assert int((w.status_dev != 0).sum()) == 0
    for _ in range(2):                                      # the configs[2] operation proper: decompress only, twice
        w.decompress()
    codec.synchronize()
    assert w.R > 150_000_000 and torch.equal(w.res_off_dev, d["res_off"].to(torch.int32))
    x0 = w.out_t["x"].clone()
    n = 4096
    pc = bench.parity_check(d, w, n)
    assert pc["fcz_bit_exact"] and pc["coords_bit_exact"], f"the 4 096-chain sample differs from the oracle: {pc}"
    w.decompress(); codec.synchronize()
    assert torch.equal(w.out_t["x"].view(torch.int32), x0.view(torch.int32)), "decompress-only is not repeatable"
    rmsd, mx = w.round_trip_deviation()
    assert rmsd < 0.2 and np.isfinite(mx), (rmsd, mx)
    del w, d, x0
    torch.cuda.empty_cache()


def test_configs2_database_to_pdb_text_equals_reference(codec, tmp_path):
    """BASELINE configs[2] as the reference runs it (src/main.cpp:612-689): an FCZ database -> PDB text per entry through
    `foldcomp-hip decompress -d`; EVERY entry of the output == the reference's text for that record (oracle/_ref), NUL included."""
    if not H.have_ref():
        pytest.skip("oracle/_ref is not built")
    import subprocess
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from foldcomp_amd.database import DatabaseReader, DatabaseWriter
    C = 3000
    d = bench.generate_resident(C, 0, 25, 1024, "cuda:0", seed_base=4242, mixed=True)
    w = bench.Workload(codec, d, "cuda:0")
    w.compress(); codec.synchronize()
    off = w.off_dev.cpu().numpy().astype(np.int64)
    blob = w.blob_dev[:int(off[-1])].cpu().numpy()
    wr = DatabaseWriter(str(tmp_path / "fczdb"))
    for i in range(C):
        wr.append(blob[off[i]:off[i + 1]].tobytes(), i, f"rec{i:05d}")
    wr.close()
    r = subprocess.run([os.path.join(ROOT, "host", "foldcomp-hip"), "decompress", "-d", "-y", "--gpus", "1", str(tmp_path / "fczdb"), str(tmp_path / "pdbdb")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    rd = DatabaseReader(str(tmp_path / "pdbdb"))
    assert len(rd) == C
    for i in range(C):
        assert rd.name(i) == f"rec{i:05d}"
        assert rd.data(i) == H.ref_decompress_pdb(blob[off[i]:off[i + 1]].tobytes()).encode("latin-1") + b"\0", i
    rd.close()
    del w, d
    torch.cuda.empty_cache()


def test_a_remembered_sizes_pass_is_forgotten_when_another_batch_runs_in_between():
    """fcz_decompress_sizes_dev leaves totals, the length order and the residue-code array for the batch call that follows on the SAME
    records; a batch call on OTHER records in between recomputes them for those -- the remembered ones must not be used afterwards"""
    import ctypes
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from foldcomp_amd import _lib
    from foldcomp_amd.codec import Codec
    dev = "cuda:0"
    codec = Codec(0)
    ws = []
    for seed, n_res in ((0, 120), (5000, 77)):
        d = bench.generate_resident(700, n_res, 25, 4096, dev, seed_base=seed)
        w = bench.Workload(codec, d, dev)
        w.compress(); w.decompress(); codec.synchronize()
        ws.append(w)
    a, b = ws
    want = {k: a.out_t[k].clone() for k in ("x", "y", "z", "bfac_res")}
    for k in ("x", "y", "z", "bfac_res"):
        a.out_t[k].zero_()
    lib = codec.lib
    tr = ctypes.c_uint32(); ta = ctypes.c_uint32()
    _lib.check(lib.fcz_decompress_sizes_dev(codec.ctx, a.blob_dev.data_ptr(), a.off_dev.data_ptr(), a.C, a.res_off_dev.data_ptr(), a.atom_off_dev.data_ptr(),
                                            ctypes.byref(tr), ctypes.byref(ta)), "sizes a")
    _lib.check(lib.fcz_decompress_batch_dev(codec.ctx, b.blob_dev.data_ptr(), b.off_dev.data_ptr(), b.C, b.res_off_dev.data_ptr(), b.atom_off_dev.data_ptr(), 0,
                                            ctypes.byref(b.cout)), "batch b")
    _lib.check(lib.fcz_decompress_batch_dev(codec.ctx, a.blob_dev.data_ptr(), a.off_dev.data_ptr(), a.C, a.res_off_dev.data_ptr(), a.atom_off_dev.data_ptr(), 0,
                                            ctypes.byref(a.cout)), "batch a")
    codec.synchronize()
    for k in ("x", "y", "z", "bfac_res"):
        assert torch.equal(a.out_t[k].view(torch.int32), want[k].view(torch.int32)), k
    codec.close()
