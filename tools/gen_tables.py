#!/usr/bin/env python3
"""Generate foldcomp_amd/csrc/aa_tables.h and foldcomp_amd/_aa_tables.py.

Run in the build container only (needs /root/reference). The amino-acid geometry of the
reference is *data* (ideal bond lengths/angles after Peptide Builder), held in
src/amino_acid.h:69-406 as std::map<std::string,...> initialisers. This script reads that
data and re-expresses it in the dense, integer-indexed form the kernels use:

  * atom names -> a 37-entry atom-code enumeration (0=N 1=CA 2=C 3=O 4=CB ... 36=OXT)
  * per residue code (src/utility.h:133-206 order) the canonical atom list, the `-a`
    alternative order, and for every non-backbone atom j>=3 the three predecessor *slots*
    (indices into the canonical list) plus ideal length/angle as float32 bit patterns
    (the reference stores double literals into std::map<std::string,float>, i.e.
    float(double(lit)); reproduced with numpy so no decimal double-rounding slips in).

The generated files are committed; nothing reads /root/reference at run time.
"""
import re, sys, struct
import numpy as np

REF = "/root/reference/src/amino_acid.h"
# residue-code order, src/utility.h:133-206 (ALA0 ... VAL19 ASX20 GLX21 STP22 UNK23)
RES3 = ["ALA","ARG","ASN","ASP","CYS","GLN","GLU","GLY","HIS","ILE","LEU","LYS","MET",
        "PHE","PRO","SER","THR","TRP","TYR","VAL","ASX","GLX","STP","UNK"]
RES1 = "ARNDCQEGHILKMFPSTWYVBZ*X"

def parse():
    src = open(REF).read()
    # strip // comments
    src = re.sub(r"//[^\n]*", "", src)
    aas = {}
    for m in re.finditer(r'output\.emplace\("(\w+)",\s*AminoAcid\(', src):
        name = m.group(1)
        # balanced-paren scan for the constructor args
        i = m.end(); depth = 1
        while depth:
            c = src[i]
            depth += (c == '(') - (c == ')')
            i += 1
        body = src[m.end():i-1]
        # brace groups at top level
        groups = []; d = 0; start = None
        for k, c in enumerate(body):
            if c == '{':
                if d == 0: start = k
                d += 1
            elif c == '}':
                d -= 1
                if d == 0: groups.append(body[start:k+1])
        if not groups:
            aas[name] = dict(atoms=[], side={}, alt=[]); continue
        atoms = re.findall(r'"(\w+)"', groups[0])
        side = {}
        for sm in re.finditer(r'\{\s*"(\w+)"\s*,\s*\{\s*"(\w+)"\s*,\s*"(\w+)"\s*,\s*"(\w+)"\s*\}\s*\}', groups[1]):
            side[sm.group(1)] = [sm.group(2), sm.group(3), sm.group(4)]
        alt = re.findall(r'"(\w+)"', groups[2]) if len(groups) > 2 else list(atoms)
        aas[name] = dict(atoms=atoms, side=side, alt=alt)
    for kind in ("bondLengths", "bondAngles"):
        for m in re.finditer(r'output\["(\w+)"\]\.%s\s*=\s*\{(.*?)\};' % kind, src, re.S):
            d = {}
            for e in re.finditer(r'\{\s*"(\w+)"\s*,\s*([0-9.]+)\s*\}', m.group(2)):
                d[e.group(1)] = float(e.group(2))
            aas[m.group(1)][kind] = d
    return aas

def f32bits(x):
    return int(np.float32(np.float64(x)).view(np.uint32))

def main():
    aas = parse()
    # atom-code enumeration: backbone first, then first-seen order over residue codes
    names = ["N", "CA", "C", "O", "CB"]
    for r in RES3[:20]:
        for a in aas[r]["atoms"]:
            if a not in names: names.append(a)
    assert len(names) == 36, len(names)
    names.append("OXT")
    code = {a: i for i, a in enumerate(names)}
    MAXA = 14
    natoms = []; atoms = []; alts = []; prev = []; blen = []; bang = []
    for r in RES3:
        a = aas.get(r, dict(atoms=[], side={}, alt=[]))
        if r not in RES3[:20]: a = dict(atoms=[], side={}, alt=[])
        al = a["atoms"]; n = len(al)
        assert n <= MAXA
        natoms.append(n if n else 3)          # UNK & friends: backbone only (N,CA,C), 0 torsions
        row = [code[x] for x in al] + [255] * (MAXA - n)
        if n == 0: row = [0, 1, 2] + [255] * (MAXA - 3)
        atoms.append(row)
        alt = a["alt"] if a["alt"] else al
        assert sorted(alt) == sorted(al), r
        # alt_slot[j] = canonical slot of the atom printed at position j in `-a` order
        arow = [al.index(x) for x in alt] + [255] * (MAXA - n)
        if n == 0: arow = [0, 1, 2] + [255] * (MAXA - 3)
        alts.append(arow)
        prow = []; lrow = []; grow = []
        for j in range(MAXA):
            if 3 <= j < n:
                cur = al[j]; p = a["side"][cur]
                slots = [al.index(q) for q in p]
                assert all(s < j for s in slots), (r, cur)   # predecessors are already built
                prow.append(slots)
                lrow.append(f32bits(a["bondLengths"][p[2] + "_" + cur]))
                grow.append(f32bits(a["bondAngles"][p[1] + "_" + p[2] + "_" + cur]))
            else:
                prow.append([0, 0, 0]); lrow.append(0); grow.append(0)
        prev.append(prow); blen.append(lrow); bang.append(grow)
    ntors = [max(n - 3, 0) for n in natoms]
    assert ntors[:20] == [2,8,5,5,3,6,6,1,7,5,5,6,5,8,4,3,4,11,9,4], ntors

    hdr = []
    h = hdr.append
    h("// GENERATED by tools/gen_tables.py -- do not edit. Amino-acid geometry tables in dense form.")
    h("// Data source: reference src/amino_acid.h:69-406 (ideal geometry), src/utility.h:133-206 (codes).")
    h("// The table bodies live in aa_tables.inc so that a translation unit can instantiate them twice")
    h("// (e.g. a __device__ copy and a host copy) by redefining FCZ_TABLE_QUAL / FCZ_T.")
    h("#pragma once")
    h("#include <stdint.h>")
    h("#define FCZ_N_RES_CODES 24")
    h("#define FCZ_MAX_RES_ATOMS 14")
    h("#define FCZ_N_ATOM_CODES 37")
    h("#define FCZ_ATOM_OXT 36")
    h("#define FCZ_ATOM_OTHER 255")
    h("#define FCZ_RES_PRO 14")
    h("#define FCZ_RES_UNK 23")
    h("#ifndef FCZ_TABLE_QUAL")
    h("#define FCZ_TABLE_QUAL static const")
    h("#endif")
    h("#ifndef FCZ_T")
    h("#define FCZ_T(name) fcz_##name")
    h("#endif")
    h('#include "aa_tables.inc"')
    open("foldcomp_amd/csrc/aa_tables.h", "w").write("\n".join(hdr) + "\n")

    out = []
    w = out.append
    w("// GENERATED by tools/gen_tables.py -- do not edit. Included by aa_tables.h (no include guard on purpose).")
    w('FCZ_TABLE_QUAL char FCZ_T(res1)[FCZ_N_RES_CODES + 1] = "%s";' % RES1)
    w("FCZ_TABLE_QUAL char FCZ_T(res3)[FCZ_N_RES_CODES][4] = {%s};" % ",".join('"%s"' % r for r in RES3))
    w("FCZ_TABLE_QUAL char FCZ_T(atom_name)[FCZ_N_ATOM_CODES][4] = {%s};" % ",".join('"%s"' % a for a in names))
    w("// atoms per residue code (UNK/ASX/GLX/STP: backbone only)")
    w("FCZ_TABLE_QUAL uint8_t FCZ_T(res_natoms)[FCZ_N_RES_CODES] = {%s};" % ",".join(map(str, natoms)))
    def tab2(name, typ, rows, fmt):
        w("FCZ_TABLE_QUAL %s FCZ_T(%s)[FCZ_N_RES_CODES][FCZ_MAX_RES_ATOMS] = {" % (typ, name))
        for r, row in zip(RES3, rows):
            w("  {%s}, // %s" % (",".join(fmt(v) for v in row), r))
        w("};")
    w("// canonical atom order: atom code of slot j")
    tab2("res_atom", "uint8_t", atoms, str)
    w("// `-a` output order: canonical slot printed at position j")
    tab2("res_alt_slot", "uint8_t", alts, str)
    w("// predecessor slots (p0,p1,p2) of slot j>=3, packed p0 | p1<<4 | p2<<8")
    tab2("res_prev", "uint16_t", [[p[0] | p[1] << 4 | p[2] << 8 for p in row] for row in prev], lambda v: "0x%03x" % v)
    w("// ideal bond length p2-j / bond angle p1-p2-j of slot j>=3, float32 bit patterns")
    tab2("res_blen_bits", "uint32_t", blen, lambda v: "0x%08xu" % v)
    tab2("res_bang_bits", "uint32_t", bang, lambda v: "0x%08xu" % v)
    open("foldcomp_amd/csrc/aa_tables.inc", "w").write("\n".join(out) + "\n")

    py = []
    py.append('"""GENERATED by tools/gen_tables.py -- do not edit."""')
    py.append("RES3 = %r" % RES3)
    py.append("RES1 = %r" % RES1)
    py.append("ATOM_NAMES = %r" % names)
    py.append("RES_NATOMS = %r" % natoms)
    py.append("RES_ATOMS = %r" % [row[:n] for row, n in zip(atoms, natoms)])
    py.append("RES_ALT_SLOT = %r" % [row[:n] for row, n in zip(alts, natoms)])
    open("foldcomp_amd/_aa_tables.py", "w").write("\n".join(py) + "\n")
    print("atom codes:", names)

if __name__ == "__main__":
    main()
