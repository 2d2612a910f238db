"""Seeded synthetic protein chains in the SoA layout of `fcz_chain_batch`.

Chains are built with NeRF in float64 from sampled internal coordinates:
  * residue types i.i.d. uniform over the 20 standard codes;
  * (phi, psi) from a 3-component mixture (alpha 45 %, beta 35 %, coil 20 %), omega ~ N(180, 5),
    backbone bond angles N-CA-C ~ N(111.0, 2.5), CA-C-N ~ N(116.6, 1.5), C-N-CA ~ N(121.4, 1.8),
    bond lengths N-CA 1.4581 (1.353 after PRO, as the codec assumes), CA-C 1.5281, C-N 1.3311;
  * side chains grown with the ideal table geometry, every side-chain torsion ~ U(-180, 180);
  * pLDDT per residue ~ U(30, 100) rounded to 2 decimals; OXT on every chain (as in AFDB);
  * coordinates rounded to 3 decimals (what a PDB file would hold) and stored as float32;
  * atoms listed in the usual PDB/AFDB order (N, CA, C, CB, O, ...), i.e. the `-a` order.
torch is used only as an array library (CPU for tests, the GPU for the 1M-chain bench input).
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import numpy as np
import torch

from ._aa_tables import RES_ALT_SLOT, RES_ATOMS, RES_NATOMS
from .structure import ChainBatch

_DEG = math.pi / 180.0


def _geometry_tables():
    """(prev slots, bond length, bond angle) per (res code, slot) from the generated C tables."""
    import os, re
    inc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "aa_tables.inc")
    txt = open(inc).read()

    def table(name):
        m = re.search(r"FCZ_T\(%s\)\[FCZ_N_RES_CODES\]\[FCZ_MAX_RES_ATOMS\] = \{(.*?)\n\};" % name, txt, re.S)
        rows = re.findall(r"\{([^}]*)\}", m.group(1))
        return np.array([[int(v.strip().rstrip("u"), 0) for v in r.split(",")] for r in rows], np.int64)

    prev = table("res_prev")
    blen = table("res_blen_bits").astype(np.uint32).view(np.float32).astype(np.float64)
    bang = table("res_bang_bits").astype(np.uint32).view(np.float32).astype(np.float64)
    return prev, blen, bang


def _place(a, b, c, L, ang, tor):
    """NeRF (float64, radians): place d from a, b, c."""
    bc = c - b
    bcn = bc / bc.norm(dim=-1, keepdim=True)
    n = torch.cross(b - a, bcn, dim=-1)
    n = n / n.norm(dim=-1, keepdim=True)
    m = torch.cross(n, bcn, dim=-1)
    d2 = torch.stack((-L * torch.cos(ang), L * torch.cos(tor) * torch.sin(ang), L * torch.sin(tor) * torch.sin(ang)), -1)
    return c + bcn * d2[..., 0:1] + m * d2[..., 1:2] + n * d2[..., 2:3]


def generate(n_chains: int, lengths, seed: int = 0xF01DC0DE, device: str = "cpu", anchor_threshold: int = 25,
             first_chain_id: int = 0, res_code=None):
    """-> dict of torch tensors laid out as fcz_chain_batch (plus 'anchor_threshold').

    lengths: int (all chains) or a sequence/array of per-chain residue counts.
    res_code: None = residue types i.i.d. uniform over the 20 standard codes; an int = every residue of that type; a
    per-chain sequence = that type for the chain, -1 = uniform (tests of the atom-richest tiles: 17 = TRP, 14 heavy atoms).
    """
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed) + int(first_chain_id))
    C = int(n_chains)
    if isinstance(lengths, int):
        lens = torch.full((C,), int(lengths), dtype=torch.int64, device=dev)
    else:
        lens = torch.as_tensor(np.asarray(lengths, np.int64), device=dev)
    nmax = int(lens.max().item())
    f64 = torch.float64

    def randn(*s): return torch.randn(*s, generator=g, device=dev, dtype=f64)
    def rand(*s): return torch.rand(*s, generator=g, device=dev, dtype=f64)

    rc = torch.randint(0, 20, (C, nmax), generator=g, device=dev)
    if res_code is not None:
        per_chain = torch.as_tensor(np.broadcast_to(np.asarray(res_code, np.int64), (C,)).copy(), device=dev)[:, None]
        rc = torch.where(per_chain >= 0, per_chain.expand_as(rc), rc)
    # backbone internal coordinates
    comp = rand(C, nmax)
    phi = torch.where(comp < 0.45, -63 + 15 * randn(C, nmax), torch.where(comp < 0.80, -120 + 20 * randn(C, nmax), -180 + 360 * rand(C, nmax)))
    psi = torch.where(comp < 0.45, -43 + 15 * randn(C, nmax), torch.where(comp < 0.80, 135 + 20 * randn(C, nmax), -180 + 360 * rand(C, nmax)))
    omg = 180 + 5 * randn(C, nmax)
    a_nca = (111.0 + 2.5 * randn(C, nmax)) * _DEG
    a_can = (116.6 + 1.5 * randn(C, nmax)) * _DEG
    a_cna = (121.4 + 1.8 * randn(C, nmax)) * _DEG
    phi, psi, omg = phi * _DEG, psi * _DEG, omg * _DEG

    P = torch.zeros(C, nmax, 14, 3, dtype=f64, device=dev)
    # first residue in a fixed frame
    N0 = torch.zeros(C, 3, dtype=f64, device=dev)
    CA0 = torch.tensor([1.4581, 0.0, 0.0], dtype=f64, device=dev).expand(C, 3)
    ang0 = a_nca[:, 0]
    C0 = CA0 + 1.5281 * torch.stack((-torch.cos(ang0), torch.sin(ang0), torch.zeros_like(ang0)), -1)
    P[:, 0, 0], P[:, 0, 1], P[:, 0, 2] = N0, CA0, C0
    is_pro = rc == 14
    for k in range(1, nmax):
        a, b, c = P[:, k - 1, 0], P[:, k - 1, 1], P[:, k - 1, 2]
        Lc = torch.full((C,), 1.3311, dtype=f64, device=dev)
        N = _place(a, b, c, Lc, a_can[:, k - 1], psi[:, k - 1])
        Ln = torch.where(is_pro[:, k - 1], torch.full_like(Lc, 1.353), torch.full_like(Lc, 1.4581))
        CA = _place(b, c, N, Ln, a_cna[:, k - 1], omg[:, k - 1])
        Cc = _place(c, N, CA, torch.full_like(Lc, 1.5281), a_nca[:, k], phi[:, k])
        P[:, k, 0], P[:, k, 1], P[:, k, 2] = N, CA, Cc
    # side chains, all residues at once, slot by slot
    prev, blen, bang = _geometry_tables()
    prev_t = torch.as_tensor(prev, device=dev); blen_t = torch.as_tensor(blen, device=dev); bang_t = torch.as_tensor(bang, device=dev)
    natoms_t = torch.as_tensor(np.asarray(RES_NATOMS, np.int64), device=dev)
    na = natoms_t[rc]                                # (C, nmax)
    Pf = P.view(C * nmax, 14, 3)
    rcf = rc.reshape(-1)
    idx = torch.arange(C * nmax, device=dev)
    for j in range(3, 14):
        pk = prev_t[rcf, j]
        p0, p1, p2 = pk & 15, (pk >> 4) & 15, (pk >> 8) & 15
        tor = (-180 + 360 * rand(C * nmax)) * _DEG
        L = blen_t[rcf, j]; A = bang_t[rcf, j] * _DEG
        valid = (j < natoms_t[rcf])
        L = torch.where(valid, L, torch.ones_like(L)); A = torch.where(valid, A, torch.full_like(A, 1.9))
        d = _place(Pf[idx, p0], Pf[idx, p1], Pf[idx, p2], L, A, tor)
        Pf[:, j] = torch.where(valid[:, None], d, torch.zeros_like(d))
    # OXT on the last residue: placed like O with the torsion flipped by 180 degrees
    last = (lens - 1)
    ar = torch.arange(C, device=dev)
    Pl = P[ar, last]
    oxt = _place(Pl[:, 0], Pl[:, 1], Pl[:, 2], torch.full((C,), 1.25, dtype=f64, device=dev),
                 torch.full((C,), 118.0 * _DEG, dtype=f64, device=dev), (-180 + 360 * rand(C)) * _DEG)

    # ---- compact to SoA in the PDB/AFDB atom order ----
    alt = np.full((24, 14), 0, np.int64); code = np.full((24, 14), 255, np.int64)
    for r in range(24):
        for j, s in enumerate(RES_ALT_SLOT[r]):
            alt[r, j] = s; code[r, j] = RES_ATOMS[r][s]
    alt_t = torch.as_tensor(alt, device=dev); code_t = torch.as_tensor(code, device=dev)
    res_valid = torch.arange(nmax, device=dev)[None, :] < lens[:, None]          # (C, nmax)
    slot_valid = (torch.arange(14, device=dev)[None, None, :] < na[:, :, None]) & res_valid[:, :, None]
    ordered = torch.gather(P, 2, alt_t[rc][..., None].expand(C, nmax, 14, 3))    # positions in output order
    codes = code_t[rc]                                                          # (C, nmax, 14)
    # append OXT as a 15th slot of the last residue
    ordered = torch.cat((ordered, torch.zeros(C, nmax, 1, 3, dtype=f64, device=dev)), 2)
    codes = torch.cat((codes, torch.full((C, nmax, 1), 36, dtype=torch.int64, device=dev)), 2)
    slot_valid = torch.cat((slot_valid, torch.zeros(C, nmax, 1, dtype=torch.bool, device=dev)), 2)
    ordered[ar, last, 14] = oxt
    slot_valid[ar, last, 14] = True
    flat_valid = slot_valid.reshape(-1)
    xyz = ordered.reshape(-1, 3)[flat_valid]
    xyz = (torch.round(xyz * 1000.0) / 1000.0).to(torch.float32)
    atom_code = codes.reshape(-1)[flat_valid].to(torch.uint8)
    per_res = slot_valid.sum(-1)                                                # atoms per residue incl. OXT
    per_res_flat = per_res[res_valid]
    atom_off = torch.zeros(per_res_flat.numel() + 1, dtype=torch.int64, device=dev)
    atom_off[1:] = torch.cumsum(per_res_flat, 0)
    res_off = torch.zeros(C + 1, dtype=torch.int64, device=dev)
    res_off[1:] = torch.cumsum(lens, 0)
    res_code = rc[res_valid].to(torch.uint8)
    plddt = (torch.round((30 + 70 * rand(C, nmax)) * 100.0) / 100.0).to(torch.float32)[res_valid]
    first_atom_index = torch.ones(C, dtype=torch.int32, device=dev)
    first_res_index = torch.ones(C, dtype=torch.int32, device=dev)
    chain_id = torch.full((C,), ord("A"), dtype=torch.uint8, device=dev)
    titles = "".join("synth_%010d" % (first_chain_id + i) for i in range(C)).encode() if C <= 200000 else None
    if titles is None:
        # vectorised title bytes for very large batches
        ids = torch.arange(first_chain_id, first_chain_id + C, device=dev, dtype=torch.int64)
        digs = torch.stack([(ids // (10 ** (9 - d))) % 10 + 48 for d in range(10)], 1).to(torch.uint8)
        pre = torch.as_tensor(np.frombuffer(b"synth_", np.uint8).copy(), device=dev)[None, :].expand(C, 6)
        titles_t = torch.cat((pre, digs), 1).reshape(-1).contiguous()
    else:
        titles_t = torch.as_tensor(np.frombuffer(titles, np.uint8).copy(), device=dev)
    title_off = (torch.arange(C + 1, device=dev, dtype=torch.int64) * 16).to(torch.int32)
    return dict(
        res_off=res_off.to(torch.int32), atom_off=atom_off.to(torch.int32),
        x=xyz[:, 0].contiguous(), y=xyz[:, 1].contiguous(), z=xyz[:, 2].contiguous(),
        atom_code=atom_code.contiguous(), res_code=res_code.contiguous(), bfac_ca=plddt.contiguous(),
        first_res_index=first_res_index, first_atom_index=first_atom_index, chain_id=chain_id,
        titles=titles_t, title_off=title_off, anchor_threshold=anchor_threshold)


def to_chain_batch(d: dict) -> ChainBatch:
    """torch dict (any device) -> host numpy ChainBatch"""
    def h(k, dt):
        return np.ascontiguousarray(d[k].detach().cpu().numpy()).astype(dt, copy=False)
    return ChainBatch(
        res_off=h("res_off", np.uint32), atom_off=h("atom_off", np.uint32), x=h("x", np.float32), y=h("y", np.float32),
        z=h("z", np.float32), atom_code=h("atom_code", np.uint8), res_code=h("res_code", np.uint8),
        bfac_ca=h("bfac_ca", np.float32), first_res_index=h("first_res_index", np.int32),
        first_atom_index=h("first_atom_index", np.int32), chain_id=h("chain_id", np.uint8), titles=h("titles", np.uint8),
        title_off=h("title_off", np.uint32), anchor_threshold=int(d["anchor_threshold"]))


def mixed_lengths(n_chains: int, seed: int = 7, mu: float = math.log(250.0), sigma: float = 0.6, lo: int = 16, hi: int = 2700):
    """log-normal chain lengths clipped to [lo, hi] (BASELINE config 5 stand-in)"""
    r = np.random.default_rng(seed)
    return np.clip(np.round(np.exp(r.normal(mu, sigma, n_chains))), lo, hi).astype(np.int64)
