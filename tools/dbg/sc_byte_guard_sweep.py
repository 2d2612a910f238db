"""CPU study for DESIGN.md section 9 item 2: how wide must the guard band be if the COSINE of a side-chain dihedral is taken in
float (ip * rsq(p)) instead of through the exact double path, before the float acos and the byte quantiser?

Model (numpy, float32 arithmetic where the device has it):
  exact:  ct = float32(float64(ip) / sqrt(float64(p)))            -- reference getCosineTheta, src/float3d.h:38-43
  approx: ct~ = float32(ip * r), r = float32(1/sqrt(p)) perturbed by -1, 0, +1 ulp (v_rsq_f32 is accurate to 1 ulp)
  angle:  exact byte from float32(acos(float64(ct)) * 180 / pi); approx angle from the same formula on ct~ plus +-1e-4 degrees
          (the bound of acos_deg_f32, tests/test_device_math.py)
  byte:   trunc((angle + 180) * float32(255 / 360))               -- FixedAngleDiscretizer(255), src/discretizer.h:89-106
Reports, for guard(theta) = g0 + g1 / sin(theta) (in bins), the number of samples where the approximate byte differs from the
exact one although the approximate value is farther than the guard from every bin edge (must be 0), and the share of samples
inside the guard (they would take the exact path on the device). Inputs: random side-chain-like geometries."""
import sys
import numpy as np

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(float(sys.argv[2])) if len(sys.argv) > 2 else 4_000_000
f32 = np.float32


def dihedral_inputs(n):
    # four atoms with bond lengths 1.3..1.6 and random angles, coordinates rounded to 3 decimals like PDB input
    a = rng.normal(size=(n, 3)) * 20
    d = [rng.normal(size=(n, 3)) for _ in range(3)]
    pts = [a]
    for v in d:
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        pts.append(pts[-1] + v * rng.uniform(1.3, 1.6, size=(n, 1)))
    return [np.round(p, 3).astype(f32) for p in pts]


def cross(u, v):
    return np.stack([u[:, 1] * v[:, 2] - u[:, 2] * v[:, 1], u[:, 2] * v[:, 0] - u[:, 0] * v[:, 2], u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0]], 1).astype(f32)


def dot(u, v):   # float, left to right
    return ((u[:, 0] * v[:, 0]).astype(f32) + (u[:, 1] * v[:, 1]).astype(f32)).astype(f32) + (u[:, 2] * v[:, 2]).astype(f32)


A, B, C, D = dihedral_inputs(N)
d1, d2, d3 = (B - A).astype(f32), (C - B).astype(f32), (D - C).astype(f32)
u1, u2 = cross(d1, d2), cross(d2, d3)
ip = dot(u1, u2).astype(f32)
p = (dot(u1, u1).astype(f32) * dot(u2, u2).astype(f32)).astype(f32)
neg = dot(u1, cross(u2, d2)).astype(f32) < 0
ok = p > 0
ip, p, neg = ip[ok], p[ok], neg[ok]
ct = (ip.astype(np.float64) / np.sqrt(p.astype(np.float64))).astype(f32)
good = np.abs(ct) < 1
ip, p, neg, ct = ip[good], p[good], neg[good], ct[good]
disc = f32(255.0) / f32(360.0)


def byte_of(theta_deg, negm):
    v = np.where(negm, -theta_deg, theta_deg).astype(f32)
    return ((v - f32(-180.0)).astype(f32) * disc).astype(f32)


th = (np.arccos(ct.astype(np.float64)) * 180.0 / np.pi).astype(f32)
f_exact = byte_of(th, neg)
q_exact = np.trunc(f_exact).astype(np.int64)
r0 = (1.0 / np.sqrt(p.astype(np.float64))).astype(f32)
worst = np.zeros(len(ct))
bad_at = {}
for g0, g1 in ((3e-4, 0.0), (3e-4, 1.5e-5), (4e-4, 3e-5), (6e-4, 6e-5)):
    viol = 0
    inside = np.zeros(len(ct), bool)
    for du in (-1, 0, 1):
        r = (r0.view(np.int32) + du).view(f32)
        cta = (ip * r).astype(f32)
        cta = np.clip(cta, -1, 1)
        for dacos in (-1e-4, 0.0, 1e-4):
            tha = ((np.arccos(cta.astype(np.float64)) * 180.0 / np.pi) + dacos).astype(f32)
            fa = byte_of(tha, neg)
            qa = np.trunc(fa).astype(np.int64)
            sin_t = np.sqrt(np.maximum(1.0 - cta.astype(np.float64) ** 2, 1e-30))
            guard = g0 + g1 / sin_t
            certain = np.abs(fa - np.rint(fa)) > guard
            viol += int(np.sum(certain & (qa != q_exact)))
            inside |= ~certain
    print(f"guard = {g0:g} + {g1:g}/sin(theta) bins: samples {len(ct)}, wrong-but-certain {viol}, inside the guard {100.0 * inside.mean():.3f} %"
          f" (a wavefront round of 64 items hits one with probability {100.0 * (1 - (1 - inside.mean()) ** 64):.1f} %)")
