"""GPU: shapes that stress the kernels' tiling and batching logic, HIP vs oracle (bit-exact):
long chains (several metadata super-blocks), chains around the 64-residue tile edges, residues with explicit
hydrogens (more than 16 atoms per residue, staging-capacity shrink path), many tiny chains, large anchor
thresholds (one long segment), alt atom order, and size-independent round-trip properties at scale."""
import numpy as np
import pytest

import _harness as H
from foldcomp_amd import synthetic
from foldcomp_amd.structure import ChainBatch

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _check(codec, b, alt=False):
    blob, off, st = codec.compress_batch(b)
    oblob, ooff, ost = H.oracle_compress(b, n_threads=8)
    assert (st == 0).all() and (ost == 0).all()
    assert np.array_equal(off, ooff)
    if blob.tobytes() != oblob.tobytes():
        bad = [c for c in range(b.n_chains) if blob[off[c]:off[c + 1]].tobytes() != oblob[off[c]:off[c + 1]].tobytes()]
        raise AssertionError(("compress differs for chains", bad[:8], len(bad)))
    d = codec.decompress_batch(blob, off, alt_order=alt)
    o = H.oracle_decompress(oblob, ooff, alt_order=alt, n_threads=8)
    for k in ("x", "y", "z", "bfac_res"):
        assert np.array_equal(_bits(d[k]), _bits(o[k])), k
    assert np.array_equal(d["atom_code"], o["atom_code"])
    return blob, off, d


def test_tile_edges_and_long_chains(codec):
    lens = [2, 3, 62, 63, 64, 65, 66, 127, 128, 129, 191, 192, 193, 511, 512, 513, 514, 575, 576, 577, 1023, 1025, 2700]
    b = synthetic.to_chain_batch(synthetic.generate(len(lens), lens, seed=99))
    _check(codec, b)
    _check(codec, b, alt=True)


def test_atom_richest_tiles(codec):
    """chains of TRP only (14 atoms per residue): 256-residue tiles of the side-chain stage exceed its staging buffer and go
    through the device list to the 128-residue launch; the wavefront tiles of the compress stage (63 * 14 atoms) exceed theirs
    and go to the block-tile kernel. Mixed with ordinary chains so that both routes run in one batch."""
    rich = synthetic.to_chain_batch(synthetic.generate(5, [700, 300, 64, 513, 2], seed=41, res_code=17))
    _check(codec, rich)
    _check(codec, rich, alt=True)
    both = synthetic.to_chain_batch(synthetic.generate(7, [350, 90, 900, 257, 600, 2, 300], seed=42, res_code=[-1, -1, 17, 17, -1, 17, -1]))
    _check(codec, both)


@pytest.mark.parametrize("thr", [1, 2, 7, 200, 5000])
def test_anchor_thresholds(codec, thr):
    # n / thr + 2 anchors must fit the header's uint8 (next test). -b 1: an anchor per residue and two more, every interval 0 --
    # the reference writes and reads such records (tests/test_api_vs_reference_module.py: compress_b1)
    lens = [30, 64, 350, 700 if thr > 2 else 506] if thr > 1 else [2, 3, 16, 30, 64, 129, 253]
    b = synthetic.to_chain_batch(synthetic.generate(len(lens), lens, seed=5, anchor_threshold=thr))
    _check(codec, b)


def test_chain_beyond_header_counts_is_refused(codec):
    """nAnchor is a uint8 and nResidue a uint16 in the FCZ header (src/foldcomp.h:120-125): the reference wraps them silently
    and writes a record that cannot be read back; here such a chain gets FCZ_E_INVALID_ARG and a zero-filled record while its
    neighbours compress as usual"""
    lens = [64, 700, 350]
    b = synthetic.to_chain_batch(synthetic.generate(len(lens), lens, seed=5, anchor_threshold=2))   # 700 / 2 + 2 = 352 anchors
    blob, off, st = codec.compress_batch(b, strict=False)
    assert list(st) == [0, -1, 0]
    assert not blob[off[1]:off[2]].any()
    oblob, ooff, ost = H.oracle_compress(b, n_threads=2)
    assert np.array_equal(off, ooff)
    for c in (0, 2):
        assert blob[off[c]:off[c + 1]].tobytes() == oblob[off[c]:off[c + 1]].tobytes()


def _with_hydrogens(b: ChainBatch, per_res: int, seed=3) -> ChainBatch:
    """insert `per_res` extra atoms named 'other' (code 255) after the backbone of every residue"""
    rng = np.random.default_rng(seed)
    x, y, z, code, aoff = [], [], [], [], [0]
    for r in range(b.n_residues):
        s, e = int(b.atom_off[r]), int(b.atom_off[r + 1])
        ins = s + 3
        hx = rng.normal(0, 5, per_res).astype(np.float32)
        x += [b.x[s:ins], hx, b.x[ins:e]]; y += [b.y[s:ins], hx + 1, b.y[ins:e]]; z += [b.z[s:ins], hx - 1, b.z[ins:e]]
        code += [b.atom_code[s:ins], np.full(per_res, 255, np.uint8), b.atom_code[ins:e]]
        aoff.append(aoff[-1] + (e - s) + per_res)
    return ChainBatch(res_off=b.res_off, atom_off=np.asarray(aoff, np.uint32), x=np.concatenate(x), y=np.concatenate(y),
                      z=np.concatenate(z), atom_code=np.concatenate(code), res_code=b.res_code, bfac_ca=b.bfac_ca,
                      first_res_index=b.first_res_index, first_atom_index=b.first_atom_index, chain_id=b.chain_id,
                      titles=b.titles, title_off=b.title_off, anchor_threshold=b.anchor_threshold)


@pytest.mark.parametrize("per_res", [6, 12, 30])
def test_explicit_hydrogens(codec, per_res):
    """14-44 atoms per residue: more than 16 atoms per residue and tiles that exceed the staging capacity"""
    b0 = synthetic.to_chain_batch(synthetic.generate(5, [40, 64, 65, 130, 350], seed=11))
    _check(codec, _with_hydrogens(b0, per_res))


def test_many_tiny_chains(codec):
    lens = np.random.default_rng(1).integers(2, 40, 3000)
    b = synthetic.to_chain_batch(synthetic.generate(len(lens), lens, seed=17))
    _check(codec, b)


def test_length_classes_in_one_batch_with_refused_neighbours(codec):
    """The compress side sorts a batch's chains into length classes handled by different kernels (round 5): 2..16 residues four
    to a wavefront in one round of 16 lanes, 17..32 / 33..64 / 65..128 four to a wavefront in 2 / 4 / 8 rounds (compress_pack_rows: a
    16-lane group per chain), ..256 / ..384 a wavefront each with 4 / 6 rounds of 64, longer ones in two passes -- and the decompress side's
    k_res_index / k_res_index_rows likewise. Every length from 2 to 70 and the class edges beyond, in random order, so that the
    groups of one wavefront hold chains of different lengths, partly filled wavefronts occur, and refused chains (a residue name
    the reference cannot process, a NaN B-factor, a NaN coordinate) sit in groups next to good ones: statuses as expected, refused
    records zero, every other record and its decode equal to the oracle's. Both anchor thresholds 25 and 3 (many anchors per
    chain: the per-lane anchor loop of a group takes several rounds)."""
    rng = np.random.default_rng(77)
    lens = list(range(2, 71)) * 3 + list(range(71, 131, 3)) + [127, 128, 129, 255, 256, 257, 383, 384, 385, 600] + [16] * 37 + [32] * 21 + [17, 33, 64, 65, 128] * 5
    rng.shuffle(lens)
    for thr in (25, 3):
        b = synthetic.to_chain_batch(synthetic.generate(len(lens), lens, seed=4242, anchor_threshold=thr))
        _check(codec, b)
        # refuse every 9th chain one way or another
        rc, bf, x = b.res_code.copy(), b.bfac_ca.copy(), b.x.copy()
        want = np.zeros(len(lens), np.int32)
        for c in range(0, len(lens), 9):
            r0, n = int(b.res_off[c]), int(b.res_off[c + 1] - b.res_off[c])
            how = (c // 9) % 3
            if how == 0:
                rc[r0 + n // 2] = 21; want[c] = -6                       # FCZ_E_RESIDUE
            elif how == 1:
                bf[r0 + n - 1] = np.nan; want[c] = -9                    # FCZ_E_NONFINITE (CA B-factor)
            else:
                x[int(b.atom_off[r0 + n // 2])] = np.inf; want[c] = -9   # FCZ_E_NONFINITE (coordinate of the residue's N)
        b2 = ChainBatch(res_off=b.res_off, atom_off=b.atom_off, x=x, y=b.y, z=b.z, atom_code=b.atom_code, res_code=rc, bfac_ca=bf,
                        first_res_index=b.first_res_index, first_atom_index=b.first_atom_index, chain_id=b.chain_id, titles=b.titles,
                        title_off=b.title_off, anchor_threshold=b.anchor_threshold)
        good_blob, good_off, _ = codec.compress_batch(b)
        blob, off, st = codec.compress_batch(b2, strict=False)
        assert np.array_equal(st, want), (thr, np.nonzero(st != want)[0][:10], st[st != want][:10])
        for c in range(len(lens)):                                  # (a changed residue name changes that record's size: own offsets)
            rec = blob[off[c]:off[c + 1]].tobytes()
            assert rec == (bytes(len(rec)) if want[c] else good_blob[good_off[c]:good_off[c + 1]].tobytes()), (thr, c, lens[c])


def test_roundtrip_properties_at_scale(codec):
    """size-independent properties on 20k chains: idempotence (compress(decompress(compress(x))) keeps the
    words' residue codes, side-chain byte count and sizes), determinism, and round-trip accuracy"""
    C = 20000
    b = synthetic.to_chain_batch(synthetic.generate(C, 350, seed=123))
    blob, off, st = codec.compress_batch(b)
    assert (st == 0).all()
    blob2, off2, _ = codec.compress_batch(b)
    assert np.array_equal(blob, blob2) and np.array_equal(off, off2)          # deterministic
    d = codec.decompress_batch(blob, off)
    assert np.array_equal(d["atom_off"], b.atom_off[b.res_off]) and len(d["x"]) == b.n_atoms
    # input atoms are listed in `-a` order, the canonical output order differs only by the O/CB swap & co:
    # compare per-residue sorted coordinates' centroid and the overall RMSD of matched atom names
    dd = codec.decompress_batch(blob, off, alt_order=True)
    same = dd["atom_code"] == b.atom_code
    assert same.mean() > 0.999
    err = np.sqrt((dd["x"] - b.x) ** 2 + (dd["y"] - b.y) ** 2 + (dd["z"] - b.z) ** 2)[same]
    rmsd = float(np.sqrt(np.mean(err.astype(np.float64) ** 2)))
    assert rmsd < 0.5, rmsd      # synthetic side chains have random torsions quantised to 1.4 degrees
    bb = same & (b.atom_code < 3)
    errb = np.sqrt((dd["x"] - b.x) ** 2 + (dd["y"] - b.y) ** 2 + (dd["z"] - b.z) ** 2)[bb]
    assert float(np.sqrt(np.mean(errb.astype(np.float64) ** 2))) < 0.15


def test_batch_beyond_32_bit_atom_offsets_is_refused(codec):
    """residue / atom offsets are 32-bit (include/fcz_hip.h): a decompress batch whose atoms reach 2^32 is refused by the sizes pass
    (FCZ_E_INVALID_ARG), not wrapped. 4 700 copies of one 65 535-residue all-TRP record = 4.3 G atoms."""
    import ctypes
    import torch
    from foldcomp_amd import _lib, synthetic
    n = 65535
    b = synthetic.to_chain_batch(synthetic.generate(1, [n], seed=99, anchor_threshold=300, res_code=17))   # 65 535 x TRP, -b 300: 220 anchors
    blob, off, st = codec.compress_batch(b)
    assert st[0] == 0
    per_atoms = int(codec.decompress_sizes(blob, off)[2][-1])
    copies = (1 << 32) // per_atoms + 2
    dev = "cuda:0"
    one = torch.from_numpy(blob).to(dev)
    big = one.repeat(copies)
    offs = (torch.arange(copies + 1, dtype=torch.int64, device=dev) * len(blob))
    res_off = torch.zeros(copies + 1, dtype=torch.int32, device=dev); atom_off = torch.zeros(copies + 1, dtype=torch.int32, device=dev)
    tr, ta = ctypes.c_uint32(0), ctypes.c_uint32(0)
    torch.cuda.synchronize()
    rc = codec.lib.fcz_decompress_sizes_dev(codec.ctx, big.data_ptr(), offs.data_ptr(), copies, res_off.data_ptr(), atom_off.data_ptr(), ctypes.byref(tr), ctypes.byref(ta))
    assert rc == -1, rc                                   # FCZ_E_INVALID_ARG
    # one copy fewer than the limit is fine
    ok = (1 << 32) // per_atoms - 1
    rc = codec.lib.fcz_decompress_sizes_dev(codec.ctx, big.data_ptr(), offs.data_ptr(), ok, res_off.data_ptr(), atom_off.data_ptr(), ctypes.byref(tr), ctypes.byref(ta))
    assert rc == 0 and ta.value == ok * per_atoms
    del big, one
    torch.cuda.empty_cache()


def test_degenerate_coordinates(codec):
    """zero-length bonds, straight lines, coinciding atoms, overflow, every atom at one point (_cases.degenerate_cases; the oracle
    is pinned to the live reference on the same inputs in test_oracle_vs_golden.py): records bit-exact, decoded coordinates
    bit-exact where they are numbers and NaN where the oracle's are NaN"""
    from _cases import degenerate_batch, degenerate_cases
    for name, mutate in degenerate_cases():
        b = degenerate_batch(mutate)
        blob, off, st = codec.compress_batch(b)
        oblob, ooff, ost = H.oracle_compress(b, n_threads=4)
        assert (st == 0).all() and (ost == 0).all(), name
        assert np.array_equal(off, ooff) and blob.tobytes() == oblob.tobytes(), name
        for alt in (False, True):
            d = codec.decompress_batch(blob, off, alt_order=alt)
            o = H.oracle_decompress(oblob, ooff, alt_order=alt, n_threads=4)
            for k in ("x", "y", "z", "bfac_res"):
                assert np.all((_bits(d[k]) == _bits(o[k])) | (np.isnan(d[k]) & np.isnan(o[k]))), (name, alt, k)


@pytest.mark.parametrize("sigma", [0.02, 0.1, 0.3, 1.5])
def test_distorted_geometry(codec, sigma):
    """real models do not have ideal bond lengths: every atom of the generator's chains moved by N(0, sigma) and written at PDB
    precision (_cases.distorted_batch; the restatement is held to the live reference on the same chains in
    test_oracle_vs_live_reference.py): records bit-exact, decoded coordinates bit-exact in both atom orders"""
    from _cases import distorted_batch
    b = distorted_batch(768, sigma, seed=int(sigma * 1000) + 11)
    blob, off, st = codec.compress_batch(b)
    oblob, ooff, ost = H.oracle_compress(b, n_threads=8)
    assert np.array_equal(st, ost)
    assert np.array_equal(off, ooff) and blob.tobytes() == oblob.tobytes()
    for alt in (False, True):
        d = codec.decompress_batch(blob, off, alt_order=alt)
        o = H.oracle_decompress(oblob, ooff, alt_order=alt, n_threads=8)
        for k in ("x", "y", "z", "bfac_res"):
            assert np.all((_bits(d[k]) == _bits(o[k])) | (np.isnan(d[k]) & np.isnan(o[k]))), (sigma, alt, k)


def test_atom_rich_last_tile(codec):
    """the batch's LAST side-chain tile has fewer than 256 residues: when nearly all of them are TRP its atoms fit the staging
    buffer (up to 164 x 14) while its work items (10 per TRP) do not fit the item list -- such a tile goes to the 128-residue
    launch like a tile whose atoms do not fit (round 6: it did not, and its side chains decoded to garbage; found by the
    differential fuzz). Every size around the two limits, the last chain starting on a tile boundary and inside a tile"""
    for rc, sizes in ((17, list(range(136, 170)) + [2, 64, 128, 200, 255, 256]), (18, range(150, 200, 7)), (1, range(160, 257, 12))):
        for last in sizes:
            for pre in ([256], [100, 193]):
                lens = pre + [int(last)]
                b = synthetic.to_chain_batch(synthetic.generate(len(lens), lens, seed=rc * 1000 + int(last), res_code=rc))
                _check(codec, b, alt=bool(last & 1))
