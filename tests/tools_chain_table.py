"""ChainBatch chain -> AtomTable (to drive the reference shim from SoA inputs)."""
import numpy as np

from foldcomp_amd._aa_tables import ATOM_NAMES, RES3
from foldcomp_amd.structure import AtomTable


def chain_table(b, c):
    r0, r1 = b.res_off[c], b.res_off[c + 1]; a0, a1 = b.atom_off[r0], b.atom_off[r1]
    xyz = np.stack([b.x, b.y, b.z], 1)
    atom = [ATOM_NAMES[k] if k < 37 else "H" for k in b.atom_code[a0:a1]]
    residx = np.zeros(a1 - a0, np.int32); res = []; bf = np.zeros(a1 - a0, np.float32)
    for r in range(r0, r1):
        s, e = b.atom_off[r] - a0, b.atom_off[r + 1] - a0
        residx[s:e] = b.first_res_index[c] + (r - r0)
        res += [RES3[b.res_code[r]]] * (e - s)
        bf[s:e] = b.bfac_ca[r]
    return AtomTable(atom, res, [chr(b.chain_id[c])] * (a1 - a0), np.arange(a1 - a0, dtype=np.int32) + b.first_atom_index[c],
                     residx, xyz[a0:a1].copy(), bf)
