import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from foldcomp_amd.codec import Codec
torch.cuda.init()
codec = Codec(0)
d = bench.generate_resident(100_000, 0, 25, 2048, "cuda:0", seed_base=4242, mixed=True)
w = bench.Workload(codec, d, "cuda:0")
w.compress(); w.decompress(); codec.synchronize()
exact = {k: w.out_t[k].clone() for k in ("x", "y", "z")}
codec.set_numerics(True); w.decompress(); codec.synchronize(); codec.set_numerics(False)
dev = torch.stack([(w.out_t[k] - exact[k]).abs() for k in ("x", "y", "z")]).max(0).values
print("max", float(dev.max()), "rms", float((dev.double() ** 2).mean().sqrt()), "frac>1e-3", float((dev > 1e-3).double().mean()), "frac>1e-2", float((dev > 1e-2).double().mean()))
ao = w.atom_off_dev.to(torch.int64) & 0xFFFFFFFF
ro = w.res_off_dev.to(torch.int64)
bad = torch.nonzero(dev > 0.02).flatten()
print("n bad atoms", bad.numel())
if bad.numel():
    ch = torch.searchsorted(ao, bad, right=True) - 1
    uniq = torch.unique(ch)
    print("bad chains", uniq.numel(), uniq[:20].tolist())
    lens = (ro[1:] - ro[:-1])
    for c in uniq[:10].tolist():
        a0, a1 = int(ao[c]), int(ao[c + 1])
        dd = dev[a0:a1]
        nz = torch.nonzero(dd > 0.02).flatten()
        print("chain", c, "len", int(lens[c]), "atoms", a1 - a0, "first bad atom", int(nz[0]), "last", int(nz[-1]), "max", float(dd.max()),
              "coord max", float(exact["x"][a0:a1].abs().max()))
    # distribution of max dev per chain vs length
percs = torch.quantile(dev[::97].float(), torch.tensor([0.5, 0.9, 0.99, 0.999], device="cuda:0"))
print("quantiles 50/90/99/99.9:", percs.tolist())

for part in ("1", "2"):
    os.environ["FCZ_DEBUG_FAST_PARTS"] = part
    codec.set_numerics(True); w.decompress(); codec.synchronize(); codec.set_numerics(False)
    dv = torch.stack([(w.out_t[k] - exact[k]).abs() for k in ("x", "y", "z")]).max(0).values
    print("parts", part, "max", float(dv.max()), "n>0.02", int((dv > 0.02).sum()), "n>1e-3", int((dv > 1e-3).sum()))
