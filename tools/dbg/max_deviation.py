#!/usr/bin/env python3
"""Where decode(encode(x)) is farthest from x on the synthetic workload, and that the oracle decodes the same chain to the same
bits (so the distance is the codec's, not this implementation's): tools/dbg/max_deviation.py [chains]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, _harness as H
from foldcomp_amd.codec import Codec
from foldcomp_amd import synthetic
from foldcomp_amd._aa_tables import ATOM_NAMES, RES3

C = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
codec = Codec(0)
d = bench.generate_resident(C, 350, 25, 32768, "cuda:0", seed_base=0xF01DC0DE)
w = bench.Workload(codec, d, "cuda:0")
w.compress(); w.decompress(alt_order=1); codec.synchronize()
dev = torch.sqrt((w.out_t["x"] - d["x"]) ** 2 + (w.out_t["y"] - d["y"]) ** 2 + (w.out_t["z"] - d["z"]) ** 2)
top = torch.topk(dev, 12)
ao = (d["atom_off"].to(torch.int64) & 0xFFFFFFFF)
ro = d["res_off"].to(torch.int64)
print("rmsd", float(torch.sqrt((dev.double() ** 2).mean())), "quantiles", [float(torch.quantile(dev[:8_000_000].float(), q)) for q in (0.5, 0.99, 0.9999)])
for v, a in zip(top.values.tolist(), top.indices.tolist()):
    r = int(torch.searchsorted(ao, torch.tensor([a], device=ao.device), right=True)[0]) - 1
    c = int(torch.searchsorted(ro, torch.tensor([r], device=ro.device), right=True)[0]) - 1
    k = r - int(ro[c]); n = int(ro[c + 1] - ro[c])
    print(f"dev {v:.3f} A  chain {c} residue {k}/{n} {RES3[int(d['res_code'][r])]} atom {ATOM_NAMES[int(d['atom_code'][a])]} (atom {a - int(ao[r])} of its residue)")
# the chain of the worst atom through the oracle: same bits?
a = int(top.indices[0]); r = int(torch.searchsorted(ao, torch.tensor([a], device=ao.device), right=True)[0]) - 1
c = int(torch.searchsorted(ro, torch.tensor([r], device=ro.device), right=True)[0]) - 1
r0, r1 = int(ro[c]), int(ro[c + 1]); a0, a1 = int(ao[r0]), int(ao[r1]); t0, t1 = int(d["title_off"][c]), int(d["title_off"][c + 1])
sub = {k: d[k][a0:a1] for k in ("x", "y", "z", "atom_code")}
sub.update({k: d[k][r0:r1] for k in ("res_code", "bfac_ca")})
sub.update({k: d[k][c:c + 1] for k in ("first_res_index", "first_atom_index", "chain_id")})
sub["res_off"] = d["res_off"][c:c + 2] - r0
sub["atom_off"] = (d["atom_off"][r0:r1 + 1].to(torch.int64) & 0xFFFFFFFF) - a0
sub["titles"] = d["titles"][t0:t1]; sub["title_off"] = d["title_off"][c:c + 2] - t0
sub["anchor_threshold"] = d["anchor_threshold"]
hb = synthetic.to_chain_batch(sub)
oblob, ooff, ost = H.oracle_compress(hb)
o = H.oracle_decompress(oblob, ooff, alt_order=True)
gx = w.out_t["x"][a0:a1].cpu().numpy()
print("oracle decodes the worst chain to the same bits:", np.array_equal(o["x"].view(np.uint32)[:len(gx)], gx.view(np.uint32)))
odev = np.sqrt((o["x"][:len(gx)] - hb.x) ** 2 + (o["y"][:len(gx)] - hb.y) ** 2 + (o["z"][:len(gx)] - hb.z) ** 2)
print("oracle's own max deviation on that chain:", float(odev.max()), "at atom", int(odev.argmax()), "GPU:", float(top.values[0]), "at", a - a0)
# context: the residue's input atoms and the decoded ones
k = r - r0
s0, s1 = int(hb.atom_off[k]), int(hb.atom_off[k + 1])
for i in range(s0, s1):
    print(f"  {ATOM_NAMES[int(hb.atom_code[i])]:4s} in ({hb.x[i]:8.3f} {hb.y[i]:8.3f} {hb.z[i]:8.3f})  out ({o['x'][i]:8.3f} {o['y'][i]:8.3f} {o['z'][i]:8.3f})  d {odev[i]:.3f}")
