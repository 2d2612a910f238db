"""Host restatements of the reference's TEXT outputs -- test infrastructure, like the rest of oracle/.

  * the PDB writer: writeAtomCoordinatesToPDB (reference src/atom_coordinate.cpp:220-291) and fast_ftoa<T,P> (:185-218):
    numbers are formatted by `r = n +- 0.5f/T` (float32 add, sign of n), `I = (int)r`, `D = (int)((r - (float)I) * T)`,
    printed as `[-]|I|.|D|` with D zero-padded to P digits;
  * `foldcomp extract --plddt`: Foldcomp::extract type 0 (src/foldcomp.cpp:1262-1325).

Pinned: equal to the reference's own text for every golden (tests/test_host_formats.py) and, where oracle/_ref exists, to the
real reference on column-overflow cases the goldens do not contain. Only tests/ and bench.py's parity legs import this
module: the product formats PDB text and extract strings on the GPU (foldcomp_amd/csrc/fcz_pdb.h, fcz_extract.h).
"""
from __future__ import annotations

import struct

import numpy as np

from foldcomp_amd import fczfile
from foldcomp_amd._aa_tables import ATOM_NAMES, RES1, RES3, RES_NATOMS as _NATOMS


def fast_ftoa(values: np.ndarray, T: int, P: int):
    v = np.ascontiguousarray(values, np.float32)
    half = np.float32(0.5) / np.float32(T)
    neg = v < 0
    r = v + np.where(neg, -half, half).astype(np.float32)
    # (int) of a NaN, an infinity or a float beyond int is INT_MIN on x86-64, for the integer and then for the decimals; std::abs
    # leaves it, and itoa_pos_only (:172-183) stops after one character for a negative number: '0' + INT_MIN % 10 = '('
    odd = ~(np.abs(r) < np.float32(2147483648.0))
    rs = np.where(odd, np.float32(0), r)
    I = rs.astype(np.int32)                                 # C truncation
    D = ((rs - I.astype(np.float32)) * np.float32(T)).astype(np.int32)
    I = np.abs(I); D = np.abs(D)
    return [("%s(.%s(" % ("-" if ng else "", "0" * (P - 1))) if od else "%s%d.%0*d" % ("-" if ng else "", i, P, d)
            for ng, i, d, od in zip(neg.tolist(), I.tolist(), D.tolist(), odd.tolist())]


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


def pdb_from_result(rec: fczfile.FczRecord, d, i: int, alt_order: bool) -> str:
    a0, a1 = int(d["atom_off"][i]), int(d["atom_off"][i + 1])
    r0, r1 = int(d["res_off"][i]), int(d["res_off"][i + 1])
    n_at = a1 - a0
    rc = d["res_code"][r0:r1]
    ac = d["atom_code"][a0:a1]
    natoms = np.asarray([len_ for len_ in map(lambda c: _NATOMS[c], rc)], np.int64)
    res_of_atom = np.repeat(np.arange(r1 - r0), natoms)
    has_oxt = n_at == int(natoms.sum()) + 1
    bf = d["bfac_res"][r0:r1][res_of_atom]
    resnum = rec.first_res_index + res_of_atom
    rcode_atom = rc[res_of_atom]
    names = None
    if has_oxt:
        # the OXT record carries header.nResidue as residue number and header.lastResidue as name
        # (Foldcomp::read, src/foldcomp.cpp:960-963)
        bf = np.concatenate([bf, d["bfac_res"][r1 - 1:r1]])
        resnum = np.concatenate([resnum, [rec.n_residues]])
        rcode_atom = np.concatenate([rcode_atom, rc[-1:]])
        names = [RES3[c] for c in rcode_atom[:-1]] + [three_letter_from_one(rec.last_residue)]
    return format_pdb(rec.title, ac, rcode_atom, resnum, rec.chain, rec.first_atom_index,
                            d["x"][a0:a1], d["y"][a0:a1], d["z"][a0:a1], bf, res_name_override=names)



def extract_plddt(rec: "fczfile.FczRecord", digits: int) -> str:
    """Foldcomp::extract type 0 (src/foldcomp.cpp:1262-1325), float32 arithmetic and C truncation"""
    digits = min(max(int(digits), 1), 4)
    tf = fczfile.temp_factors(rec)
    mn, cf = struct.unpack_from("<ff", rec.raw, rec.o_tmp)
    maxval = np.float32(np.float32(cf) * np.float32(255.0)) + np.float32(mn)
    zero_one = bool(maxval <= np.float32(1.0)) and digits <= 2
    f32 = np.float32
    if zero_one:
        cl = np.clip(tf, f32(0), f32(1))
        d1 = (cl * f32(10)).astype(np.int32) % 10
        d2 = (cl * f32(100)).astype(np.int32) % 10
    else:
        cl = np.clip(tf, f32(0), f32(100))
        d1 = (cl / f32(10)).astype(np.int32)       # (char)(clamped / 10.0f): 100 -> 10 -> ':'
        d2 = cl.astype(np.int32) % 10
    d3 = (cl * f32(10)).astype(np.int32) % 10
    d4 = (cl * f32(100)).astype(np.int32) % 10
    parts = []
    for i in range(len(tf)):
        s = chr(48 + int(d1[i]))
        if digits > 1:
            s += chr(48 + int(d2[i]))
        if digits >= 3:
            s += "." + chr(48 + int(d3[i]))
        if digits == 4:
            s += chr(48 + int(d4[i]))
        parts.append(s)
    return ("," if digits > 1 else "").join(parts)


