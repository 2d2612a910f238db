"""PDB text writer, a vectorised restatement of writeAtomCoordinatesToPDB (reference
src/atom_coordinate.cpp:220-291) and fast_ftoa<T,P> (:185-218): numbers are formatted by
`r = n ± 0.5f/T` (float32 add, sign of n), `I = (int)r`, `D = (int)((r - (float)I) * T)`, printed as
`[-]|I|.|D|` with D zero-padded to P digits."""
from __future__ import annotations

import numpy as np

from ._aa_tables import ATOM_NAMES, RES1, RES3


def fast_ftoa(values: np.ndarray, T: int, P: int):
    v = np.ascontiguousarray(values, np.float32)
    half = np.float32(0.5) / np.float32(T)
    neg = v < 0
    r = v + np.where(neg, -half, half).astype(np.float32)
    I = r.astype(np.int32)                                  # C truncation
    D = ((r - I.astype(np.float32)) * np.float32(T)).astype(np.int32)
    I = np.abs(I); D = np.abs(D)
    return ["%s%d.%0*d" % ("-" if ng else "", i, P, d) for ng, i, d in zip(neg.tolist(), I.tolist(), D.tolist())]


def title_lines(title: str) -> str:
    if title == "":
        return ""
    out = ["TITLE     %s\n" % title[:70]]
    rest, k = title[70:], 2
    while rest:
        out.append("TITLE  % 3d%s\n" % (k, rest[:70]))
        rest = rest[70:]; k += 1
    return "".join(out)


def format_pdb(title: str, atom_code: np.ndarray, res_code_per_atom: np.ndarray, res_index_per_atom: np.ndarray,
               chain: str, first_atom_index: int, x, y, z, bfac_per_atom, res_name_override=None) -> str:
    n = len(atom_code)
    xs, ys, zs = fast_ftoa(x, 1000, 3), fast_ftoa(y, 1000, 3), fast_ftoa(z, 1000, 3)
    bs = fast_ftoa(bfac_per_atom, 100, 2)
    lines = [title_lines(title)]
    for i in range(n):
        name = ATOM_NAMES[atom_code[i]]
        res = res_name_override[i] if res_name_override is not None else RES3[res_code_per_atom[i]]
        an = ("%-4s" % name) if len(name) == 4 else (" %-3s" % name)
        lines.append("ATOM  %5d %s %3s %s%4d    %8s%8s%8s  1.00%6s          %2s  \n" % (
            first_atom_index + i, an, res, chain, res_index_per_atom[i], xs[i], ys[i], zs[i], bs[i], name[0]))
    if n:
        res = res_name_override[n - 1] if res_name_override is not None else RES3[res_code_per_atom[n - 1]]
        lines.append("TER   %5d      %3s %s%4d\n" % (first_atom_index + n, res, chain, res_index_per_atom[n - 1]))
    return "".join(lines)


def three_letter_from_one(ch: str) -> str:
    i = RES1.find(ch)
    return RES3[i] if i >= 0 else "UNK"
