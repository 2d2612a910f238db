"""Host-side structure ingest: PDB text -> atom table -> chain fragments -> SoA chain batch.

Mirrors the host code that sits *above* the codec hot path in the reference:
  * the fixed-column ATOM parser of the Python module (foldcomp/foldcomp.cxx:253-278),
  * removeAlternativePosition (src/atom_coordinate.cpp:362-370),
  * identifyChains / identifyDiscontinousResInd fragmenting (src/atom_coordinate.cpp:469-530,
    used by the CLI at src/main.cpp:467-480),
  * splitAtomByResidue (src/atom_coordinate.cpp:304-328) -> the residue->atom offsets of the batch.
The output is the structure-of-arrays `fcz_chain_batch` of include/fcz_hip.h.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import re

import numpy as np

from ._aa_tables import ATOM_NAMES, RES3

ATOM_CODE = {n: i for i, n in enumerate(ATOM_NAMES)}
ATOM_CODE_OTHER = 255
ATOM_CODE_OXT = ATOM_CODE["OXT"]
# names Foldcomp::compress accepts (AAS.at(name), src/sidechain.cpp:177): the 20 standard + literal UNK
RES_CODE = {n: i for i, n in enumerate(RES3) if i < 20 or n == "UNK"}


class StructureError(ValueError):
    pass


@dataclass
class AtomTable:
    """AoS-free atom records of one input file (the reference's std::vector<AtomCoordinate>)."""
    atom: List[str]
    residue: List[str]
    chain: List[str]
    atom_index: np.ndarray
    res_index: np.ndarray
    xyz: np.ndarray          # (n,3) float32
    bfac: np.ndarray         # (n,) float32

    def __len__(self):
        return len(self.atom)

    def take(self, sl: slice) -> "AtomTable":
        return AtomTable(self.atom[sl], self.residue[sl], self.chain[sl], self.atom_index[sl],
                         self.res_index[sl], self.xyz[sl], self.bfac[sl])

    def keep(self, mask: np.ndarray) -> "AtomTable":
        idx = np.nonzero(mask)[0]
        return AtomTable([self.atom[i] for i in idx], [self.residue[i] for i in idx],
                         [self.chain[i] for i in idx], self.atom_index[idx], self.res_index[idx],
                         self.xyz[idx], self.bfac[idx])


_C_SPACE = "[ \\t\\n\\v\\f\\r]*"
_STOI = re.compile(_C_SPACE + r"([+-]?[0-9]+)")
_STOF = re.compile(_C_SPACE + r"([+-]?(?:infinity|inf|nan(?:\([0-9A-Za-z_]*\))?|0[xX](?:[0-9a-fA-F]+\.?[0-9a-fA-F]*|\.[0-9a-fA-F]+)(?:[pP][+-]?[0-9]+)?"
                               r"|(?:[0-9]+\.?[0-9]*|\.[0-9]+)(?:[eE][+-]?[0-9]+)?))", re.I)


class MultipleChainsError(StructureError):
    """the binding's flag 2: a second chain name among the ATOM lines of a string given to compress"""


def _field(line: str, pos: int, n: int) -> str:
    """std::string::substr(pos, n): the characters that are there; a position beyond the end throws (std::out_of_range)"""
    if pos > len(line):
        raise StructureError("ATOM line too short")
    return line[pos:pos + n]


def _stoi(field: str) -> int:
    """std::stoi: blanks, a sign, digits -- the longest such prefix; none, or a value beyond int, throws"""
    m = _STOI.match(field)
    if not m or not -2 ** 31 <= int(m.group(1)) < 2 ** 31:
        raise StructureError("ATOM line: a field that is no integer")
    return int(m.group(1))


def _stof(field: str) -> float:
    """std::stof (strtof): the longest prefix that is a number -- decimal or hexadecimal, with an exponent, inf / nan --; none, or a
    value beyond float's range (ERANGE), throws. "0-9999.0" is 0, "1.00 5" is 1, "12.5A" is 12.5"""
    m = _STOF.match(field)
    if not m:
        raise StructureError("ATOM line: a field that is no number")
    t = m.group(1)
    low = t.lower().lstrip("+-")
    if low.startswith("0x"):
        v = float.fromhex(t)
    elif low.startswith("nan"):
        v = float("nan")
    else:
        v = float(t)
    with np.errstate(over="ignore"):
        f = float(np.float32(v)) if v == v and abs(v) != float("inf") else v
    if (abs(f) == float("inf") and not low.startswith("inf")) or (v != 0.0 and v == v and abs(f) < 1.1754943508222875e-38):
        raise StructureError("ATOM line: a number beyond float's range")
    return v


def parse_pdb(text: str, *, hetatm: bool = False, single_chain: bool = False) -> AtomTable:
    """The Python binding's reader of a PDB STRING (foldcomp/foldcomp.cxx:252-278, :636-652), field by field: lines end at '\\n'
    only (std::getline), a line that starts with ATOM gives an atom from fixed columns -- std::stoi / std::stof of the substrings:
    the longest numeric prefix of each field counts, blanks in front of it are skipped --, the occupancy is parsed as well. Where
    the reference's call THROWS (a line that ends before column 61, a field without a number, a value out of range) its extension
    ends the interpreter (the exception crosses a C frame); here that is a StructureError (foldcomp.error).

    single_chain=True reproduces `compress`: a second chain id raises StructureError("multiple chains") (flag 2 at
    foldcomp.cxx:264-266); get_data takes every ATOM line.
    """
    atom, residue, chain = [], [], []
    ai, ri, xyz, bf = [], [], [], []
    first_chain = None
    lines = text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    for line in lines:
        if line.startswith("ATOM") or (hetatm and line.startswith("HETATM")):
            ch = _field(line, 21, 1)
            if first_chain is None:
                first_chain = ch
            if single_chain and ch != first_chain:
                raise MultipleChainsError("multiple chains")
            a_name, r_name = _field(line, 12, 4).strip(" \t"), _field(line, 17, 3).strip(" \t")
            serial, resi = _stoi(_field(line, 6, 5)), _stoi(_field(line, 22, 4))
            x, y, z = _stof(_field(line, 30, 8)), _stof(_field(line, 38, 8)), _stof(_field(line, 46, 8))
            _stof(_field(line, 54, 6))                                      # the occupancy: read, not used
            b = _stof(_field(line, 60, 6))
            atom.append(a_name); residue.append(r_name); chain.append(ch)
            ai.append(serial); ri.append(resi); xyz.append((x, y, z)); bf.append(b)
    return AtomTable(atom, residue, chain, np.asarray(ai, np.int64).astype(np.int32), np.asarray(ri, np.int64).astype(np.int32),
                     np.asarray(xyz, np.float64).astype(np.float32).reshape(-1, 3),
                     np.asarray(bf, np.float64).astype(np.float32))


# ---- the command line's reader: gemmi's read_pdb as StructureReader uses it ---------------------------------------------------
# `foldcomp compress` does not read PDB text with the binding's fixed-column parser above but with gemmi 0.5.1
# (src/structure_reader.cpp:31-61 -> lib/gemmi/pdb.hpp:262-365). What that reader does beyond taking columns, restated here
# (and checked against the live reference on mutated files, tests/test_ingest_vs_reference.py):
#   * records are matched on their first four letters, case-insensitively (ATOM, HETA..., TITL..., HEAD..., MODE..., ENDM...,
#     END stops the reading); a line is at most 120 characters, the rest is dropped;
#   * an ATOM / HETATM line shorter than 54 characters (+ its line end) fails the whole file;
#   * numbers are the longest valid prefix of their field (fast_float::from_chars after blanks and one '+'), 0 when there is
#     none; integers likewise (hybrid-36 when the field starts with a letter); the B-factor is 20.0 when the line ends before
#     column 65, the residue number field carries the insertion code, columns 21-22 are the chain name, 73-76 the segment;
#   * atoms are grouped: inside one run of lines with the same chain name (MODEL / ENDMDL end a run) an atom whose residue
#     (number, insertion code, name, segment) was seen before joins THAT residue, wherever its line stands;
#   * the title is the HEADER id code (columns 63-66 of the last HEADER record long enough to hold it, right-trimmed), else the
#     TITLE records' columns 11.. right-trimmed and concatenated.

_NUM_PREFIX = re.compile(rb"-?(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?")
_SPACES = b" \t\n\v\f\r"


def _g_double(f: bytes) -> float:
    """gemmi read_double: fast_float::from_chars on the field after blanks and one '+'; 0.0 when nothing parses"""
    f = f.split(b"\0", 1)[0].lstrip(_SPACES)
    if f[:1] == b"+":
        f = f[1:]
    m = _NUM_PREFIX.match(f)
    if m:
        return float(m.group(0))
    low = f[:9].lower()
    neg = low[:1] == b"-"
    body = low[1:] if neg else low
    for word, val in ((b"infinity", float("inf")), (b"inf", float("inf")), (b"nan", float("nan"))):
        if body.startswith(word):
            return -val if neg else val
    return 0.0


def _g_int(f: bytes) -> int:
    """gemmi read_int (string_to_int, unchecked): blanks, a sign, the digits that follow; 0 when there are none"""
    i = 0
    while i < len(f) and f[i:i + 1] in (b" ", b"\t", b"\n", b"\v", b"\f", b"\r"):
        i += 1
    neg = f[i:i + 1] == b"-"
    if f[i:i + 1] in (b"-", b"+"):
        i += 1
    n = 0
    while i < len(f) and f[i:i + 1].isdigit():
        n = n * 10 + (f[i] - 48); i += 1
    n = -n if neg else n
    return ((n + 2 ** 31) % 2 ** 32) - 2 ** 31        # int arithmetic of the reader wraps


def _g_base36(f: bytes) -> int:
    z = f.split(b"\0", 1)[0].lstrip(_SPACES)
    m = re.match(rb"[+-]?[0-9a-zA-Z]*", z)
    txt = m.group(0).decode() if m else ""
    try:
        return int(txt, 36) if txt.strip("+-") else 0
    except ValueError:
        return 0


def _g_string(f: bytes) -> str:
    """gemmi read_string: left trim, stop at the end of the line, right trim"""
    f = f.lstrip(_SPACES)
    for stop in (b"\n", b"\r", b"\0"):
        k = f.find(stop)
        if k >= 0:
            f = f[:k]
    return f.rstrip(_SPACES).decode("latin-1")


def _g_id4(line: bytes) -> bytes:
    return bytes(c & ~0x20 & 0xff for c in (line + b"\0\0\0\0")[:4])


def parse_pdb_gemmi(data: bytes):
    """PDB text -> (AtomTable in the order StructureReader hands the atoms on, title or "") as gemmi + updateStructure make it
    (lib/gemmi/pdb.hpp:262-365, src/structure_reader.cpp:31-61). Raises StructureError where the reader fails the file."""
    if isinstance(data, str):
        data = data.encode("latin-1")
    ID = {k: _g_id4(k) for k in (b"ATOM", b"HETA", b"HEAD", b"TITL", b"MODE", b"ENDM", b"ANIS", b"data")}
    models: list = []          # [name, chains]; chain = [name, residues (list of [rid, atoms]), resmap]; atom = [..., u11 set]
    model = chain = resi = None
    entry_id = None; title = ""
    pos, n = 0, len(data)
    while pos < n:
        nl = data.find(b"\n", pos)
        end = n if nl < 0 else nl + 1
        line = data[pos:end][:120]                      # at most 120 characters; what follows on the line is dropped
        pos = end
        k = line.find(b"\0")
        if k >= 0:
            line = line[:k]                              # the reader works on C strings
            if not line:
                break                                    # gets() returns an empty string: the reading stops
        ln = len(line)
        buf = line + b"\0" * 8                          # reads past the end of the line see the terminator
        rid4 = _g_id4(line)
        if rid4 == ID[b"ATOM"] or rid4 == ID[b"HETA"]:
            if ln < 55:
                raise StructureError("The line is too short to be correct")
            chain_name = _g_string(buf[20:22])
            icode = buf[26:27] if buf[26:27] not in (b"\r", b"\n") else b"\0"
            if buf[22] < 65:
                f = buf[22:26]; i = 0
                seq = 0
                while i < 4:
                    if f[i:i + 1] not in (b" ", b"\t", b"\n", b"\v", b"\f", b"\r"):
                        seq = _g_int(f[i:]); break
                    i += 1
                else:
                    seq = None                           # a blank field leaves the number unset (INT_MIN in gemmi)
            else:
                seq = _g_base36(buf[22:26]) - 466560 + 10000
            resname = _g_string(buf[17:20])
            segment = _g_string(buf[72:76]) if ln > 72 else ""
            rid = (seq, icode, resname, segment)
            if chain is None or chain_name != chain[0]:
                if model is None:
                    name = str(len(models) + 1)
                    if any(m[0] == name for m in models):
                        raise StructureError("ATOM/HETATM between models")
                    model = [name, []]; models.append(model)
                chain = [chain_name, [], {}]; model[1].append(chain); resi = None
            if resi is None or resi[0] != rid:
                j = chain[2].get(rid)
                if j is None:
                    chain[2][rid] = len(chain[1]); resi = [rid, []]; chain[1].append(resi)
                else:
                    resi = chain[1][j]
            serial = _g_int(buf[6:11]) if buf[6] < 65 else _g_base36(buf[6:11]) - 16796160 + 100000
            b_iso = np.float32(_g_double(buf[60:66])) if ln > 64 else np.float32(20.0)
            if ln > 78:
                # read_charge (lib/gemmi/pdb.hpp:85-98): a digit in columns 79-80 needs a sign (or nothing) beside it
                digit, sign = buf[78], buf[79]
                if not (digit == 0x20 and sign == 0x20):
                    if 0x30 <= sign <= 0x39:
                        digit, sign = sign, digit
                    if 0x30 <= digit <= 0x39 and sign not in b"+-\0 \t\n\v\f\r":
                        raise StructureError("Wrong format for charge")
            resi[1].append([_g_string(buf[12:16]), serial, _g_double(buf[30:38]), _g_double(buf[38:46]), _g_double(buf[46:54]), b_iso, False])
        elif rid4 == ID[b"ANIS"]:
            # the reader attaches the record to the last atom read and fails the file when there is none or it has one already
            if model is None or chain is None or resi is None or not resi[1]:
                raise StructureError("ANISOU record not directly after ATOM/HETATM.")
            if resi[1][-1][6]:
                raise StructureError("Duplicated ANISOU record or not directly after ATOM/HETATM.")
            resi[1][-1][6] = np.float32(_g_int(buf[28:35])) * np.float32(1e-4) != 0
        elif rid4 == ID[b"HEAD"]:
            if ln > 66:
                e = buf[62:66].split(b"\0", 1)[0].decode("latin-1").rstrip(" \r\n\t")
                if e:
                    entry_id = e
        elif rid4 == ID[b"TITL"]:
            if ln > 10:
                title += line[10:ln - 1].decode("latin-1").rstrip(" \r\n\t")
        elif rid4 == _g_id4(b"CRYS"):
            # UnitCell::set -> calculate_properties (lib/gemmi/unitcell.hpp:155-165, 249-260): a cell whose gamma is given and
            # whose alpha or beta is exactly zero fails the file ("Impossible angle")
            if ln > 54 and _g_double(buf[47:54]) != 0.0 and (_g_double(buf[33:40]) == 0.0 or _g_double(buf[40:47]) == 0.0):
                raise StructureError("Impossible angle - N*180deg.")
        elif rid4 == ID[b"MODE"]:
            if model is not None and chain is not None:
                raise StructureError("MODEL without ENDMDL?")
            name = str(_g_int(buf[10:14]))
            model = next((m for m in models if m[0] == name), None)
            if model is None:
                model = [name, []]; models.append(model)
            if model[1]:
                raise StructureError("duplicate MODEL number: " + name)
            chain = None
        elif rid4 == ID[b"ENDM"]:
            model = None; chain = None
        elif rid4[:3] == b"END" and (rid4[3] & ~0xf) == 0:
            break
        elif rid4 == ID[b"data"] and buf[4:5] == b"_" and model is None:
            raise StructureError("Incorrect file format (perhaps it is cif not pdb?)")
        elif rid4 == _g_id4(b'{"da') and bytes(c & ~0x20 for c in buf[4:7]) == bytes(c & ~0x20 for c in b"ta_") and model is None:
            raise StructureError("Incorrect file format (perhaps it is mmJSON not pdb?)")
    atom, residue, chains, ai, ri, xyz, bf = [], [], [], [], [], [], []
    for mdl in models:
        for ch in mdl[1]:
            for rid, atoms in ch[1]:
                for (an, serial, x, y, z, b, _aniso) in atoms:
                    atom.append(an); residue.append(rid[2]); chains.append(ch[0])
                    ai.append(serial); ri.append(rid[0] if rid[0] is not None else -2 ** 31)
                    xyz.append((x, y, z)); bf.append(b)
    t = AtomTable(atom, residue, chains, np.asarray(ai, np.int64).astype(np.int32), np.asarray(ri, np.int64).astype(np.int32),
                  np.asarray(xyz, np.float64).astype(np.float32).reshape(-1, 3), np.asarray(bf, np.float32))
    return t, (entry_id if entry_id else title)


# ---- mmCIF as the reference's reader takes it (gemmi 0.5.1: cif.hpp grammar, cifdoc.hpp tables, mmcif.hpp atoms) --------------
_CIF_ORD = frozenset(b"!%&()*+,-./0123456789:<=>?@ABCDEFGHIJKLMNOPQRSTUVWXYZ\\^`abcdefghijklmnopqrstuvwxyz{|}~")   # char_table == 1
_CIF_WS = frozenset(b" \t\n\r")
_CIF_KEYWORDS = (b"data_", b"loop_", b"global_", b"save_", b"stop_")
_CIF_NUM = re.compile(rb"-?(?:[0-9]+\.?[0-9]*|\.[0-9]+)(?:[eE][+-]?[0-9]+)?")


class _CifScanner:
    """lib/gemmi/cif.hpp:37-148 restated: whitespace and comments, reserved words, the five kinds of value"""

    def __init__(self, data: bytes):
        self.d, self.n, self.i = data, len(data), 0

    def ws(self) -> bool:
        """whitespace = plus<ws_char | comment>; -> whether anything was consumed"""
        d, n, i0 = self.d, self.n, self.i
        i = i0
        while i < n:
            c = d[i]
            if c in _CIF_WS:
                i += 1
            elif c == 0x23:                                # '#': a comment runs to the end of the line
                j = d.find(b"\n", i)
                i = n if j < 0 else j + 1
            else:
                break
        self.i = i
        return i > i0

    def ws_or_eof(self) -> bool:
        return self.ws() or self.i >= self.n

    def keyword_at(self, i=None):
        i = self.i if i is None else i
        head = self.d[i:i + 7].lower()
        for k in _CIF_KEYWORDS:
            if head.startswith(k):
                return k
        return None

    def nonblank_run(self, i):
        d, n = self.d, self.n
        j = i
        while j < n and 0x21 <= d[j] <= 0x7e:
            j += 1
        return j

    def tag(self):
        if self.i < self.n and self.d[self.i] == 0x5f:
            j = self.nonblank_run(self.i + 1)
            if j > self.i + 1:
                t = self.d[self.i:j]; self.i = j
                return t
        return None

    def value(self):
        """-> the raw value (quotes / semicolons included) or None; raises on an unterminated string or text field"""
        d, n, i = self.d, self.n, self.i
        if i >= n:
            return None
        c = d[i]
        j = i
        while j < n and d[j] in _CIF_ORD:
            j += 1
        if j > i and j < n and d[j] in _CIF_WS:             # simunq
            self.i = j
            return d[i:j]
        if c in (0x27, 0x22):                              # quoted: closes at the quote that blank, '#' or the end follows
            j = i + 1
            while True:
                if j >= n or d[j] == 0x0a:
                    raise StructureError("unterminated string")
                if d[j] == c and (j + 1 >= n or d[j + 1] in b" \n\r\t#"):
                    self.i = j + 1
                    return d[i:j + 1]
                j += 1
        if c == 0x3b and (i == 0 or d[i - 1] == 0x0a):      # text field: from ';' in the first column to the next one
            j = d.find(b"\n;", i)
            if j < 0:
                raise StructureError("unterminated text field")
            self.i = j + 2
            return d[i:j + 2]
        if self.keyword_at(i) is not None or c in (0x5f, 0x24, 0x23):
            return None
        j = self.nonblank_run(i)
        if j > i:
            self.i = j
            return d[i:j]
        return None


def _cif_items(sc: _CifScanner, in_frame: bool):
    """star<dataitem | loop | frame> -> list of ("pair", tag, value) | ("loop", tags, values) | ("frame", name, items)"""
    items = []
    while True:
        t = sc.tag()
        if t is not None:
            if not sc.ws():
                raise StructureError("parse error")
            v = sc.value()
            if v is None or not sc.ws_or_eof():
                raise StructureError(t.decode("latin-1") + " has no value")        # (missing_value, or junk glued to the value)
            items.append(("pair", t, v))
            continue
        k = sc.keyword_at()
        if k == b"loop_":
            sc.i += 5
            if not sc.ws():
                raise StructureError("parse error")
            tags = []
            while True:
                t = sc.tag()
                if t is None:
                    break
                if not sc.ws():
                    raise StructureError("parse error")
                tags.append(t)
            if not tags:
                raise StructureError("parse error")
            values = []
            while True:
                at = sc.i
                v = sc.value()
                if v is None:
                    break
                if not sc.ws_or_eof():
                    sc.i = at                              # plus<...> ends before a value nothing separates from what follows
                    break
                values.append(v)
            if not values and not (sc.i >= sc.n or sc.keyword_at() is not None):
                raise StructureError("parse error")
            if sc.keyword_at() == b"stop_":
                at = sc.i; sc.i += 5
                if not sc.ws_or_eof():
                    sc.i = at
            if len(values) % len(tags) != 0:
                raise StructureError("Wrong number of values in the loop")
            items.append(("loop", tags, values))
            continue
        if k == b"save_" and not in_frame:
            j = sc.nonblank_run(sc.i + 5)
            if j == sc.i + 5:
                return items                               # a bare save_ outside a frame: nothing here takes it
            name = sc.d[sc.i + 5:j]; sc.i = j
            if not sc.ws():
                raise StructureError("parse error")
            inner = _cif_items(sc, True)
            if sc.keyword_at() != b"save_":
                raise StructureError("parse error")
            sc.i += 5
            if not sc.ws_or_eof():
                raise StructureError("parse error")
            items.append(("frame", name, inner))
            continue
        return items


def _cif_document(data: bytes):
    """cif::read_memory: the grammar, then check_for_missing_values / check_for_duplicates (cifdoc.hpp:971-1020)"""
    sc = _CifScanner(data)
    sc.ws()
    blocks = []
    if sc.i >= sc.n:
        return blocks
    while True:
        k = sc.keyword_at()
        if k == b"data_":
            j = sc.nonblank_run(sc.i + 5)
            name = sc.d[sc.i + 5:j] or b"#"; sc.i = j
        elif k == b"global_":
            name = b""; sc.i += 7
        else:
            break
        if not sc.ws_or_eof():
            raise StructureError("parse error")
        blocks.append((name, _cif_items(sc, False)))
    if not blocks:
        raise StructureError("expected block header (data_)")
    if sc.i < sc.n:
        raise StructureError("parse error")
    seen = set()
    for name, items in blocks:
        low = name.lower()
        if low in seen and name:
            raise StructureError("duplicate block name")
        seen.add(low)
    for name, items in blocks:
        tags, frames = set(), set()
        for it in items:
            for t in ([it[1]] if it[0] == "pair" else it[1] if it[0] == "loop" else []):
                if t.lower() in tags:
                    raise StructureError("duplicate tag " + t.decode("latin-1"))
                tags.add(t.lower())
            if it[0] == "frame":
                if it[1].lower() in frames:
                    raise StructureError("duplicate save_" + it[1].decode("latin-1"))
                frames.add(it[1].lower())
    return blocks


def _cif_null(v: bytes) -> bool:
    return v in (b"?", b".")


def _cif_string(v: bytes) -> bytes:
    """cif::as_string (cifdoc.hpp:82-92)"""
    if not v or _cif_null(v):
        return b""
    if v[0] in (0x22, 0x27):
        return v[1:-1]
    if v[0] == 0x3b and len(v) > 2 and v[-2] == 0x0a:
        return v[1:-3] if v[-3] == 0x0d else v[1:-2]
    return v


def _cif_number(v: bytes, default: float) -> float:
    """cif::as_number (numb.hpp:19-41): the whole value must be a number (an uncertainty in brackets may follow)"""
    s = v[1:] if v[:1] == b"+" else v
    m = _CIF_NUM.match(s)
    if not m:
        return default
    rest = s[m.end():]
    if rest[:1] == b"(":
        k = 1
        while k < len(rest) and 0x30 <= rest[k] <= 0x39:
            k += 1
        if rest[k:k + 1] == b")":
            rest = rest[k + 1:]
    return float(m.group()) if not rest else default


def _cif_int_checked(v: bytes) -> int:
    """string_to_int(str, true) (atox.hpp:72-98); what it throws is not a std::runtime_error: the reference does not survive it"""
    m = re.fullmatch(rb"[ \t\n\v\f\r]*([+-]?)([0-9]+)[ \t\n\v\f\r]*", v)
    if not m:
        raise StructureError("not an integer: " + v.decode("latin-1"))
    return _wrap_i32(int(m.group(1) + m.group(2)))


def _wrap_i32(n: int) -> int:
    return ((n + 2 ** 31) % 2 ** 32) - 2 ** 31


def parse_cif_gemmi(data: bytes):
    """mmCIF text -> (AtomTable in the order StructureReader hands the atoms on, title or "") as gemmi's make_structure
    (lib/gemmi/mmcif.hpp:560-680, 788-797) + updateStructure (src/structure_reader.cpp:31-61) make it. Raises StructureError
    where the reader fails the file."""
    if isinstance(data, str):
        data = data.encode("latin-1")
    k = data.find(b"\0")
    blocks = _cif_document(data if k < 0 else data)       # (a NUL is not a blank: the grammar fails on it by itself)
    if not blocks:
        raise StructureError("empty file")

    def find_values(items, tag):
        """Block::find_values: the first loop that has the tag (any case) or the first pair that is the tag (this case)"""
        low = tag.lower()
        for it in items:
            if it[0] == "loop":
                for c, t in enumerate(it[1]):
                    if t.lower() == low:
                        return it, c
            elif it[0] == "pair" and it[1] == tag:
                return it, 0
        return None, 0

    # make_structure_from_doc(doc, possible_chemcomp = true): monomer-library and CCD files take another route in gemmi
    # (chemcomp_xyz.hpp:106-120); no protein chain comes out of those
    if (len(blocks) == 2 and blocks[0][0] == b"comp_list") or (len(blocks) == 3 and blocks[0][0] == b"" and blocks[1][0] == b"comp_list") or \
            (len(blocks) == 1 and find_values(blocks[0][1], b"_atom_site.id")[0] is None and find_values(blocks[0][1], b"_chem_comp_atom.atom_id")[0] is not None):
        raise StructureError("a chemical-component dictionary, not a structure")
    for name, items in blocks[1:]:
        if find_values(items, b"_atom_site.id")[0] is not None:
            raise StructureError("2+ blocks are ok if only the first one has coordinates")
    items = blocks[0][1]

    def info(tag):
        it, c = find_values(items, tag)
        if it is None:
            return ""
        vals = [it[2]] if it[0] == "pair" else it[2][c::len(it[1])]
        return "; ".join(_cif_string(v).decode("latin-1") for v in vals if not _cif_null(v))

    # set_cell_from_mmcif (lib/gemmi/mmcif_impl.hpp:17-29): the six _cell. values as ONE row; UnitCell::set fails on a zero angle
    cell_tags = [b"_cell.length_a", b"_cell.length_b", b"_cell.length_c", b"_cell.angle_alpha", b"_cell.angle_beta", b"_cell.angle_gamma"]
    it0, _ = find_values(items, cell_tags[0])
    cell = None
    if it0 is not None and it0[0] == "loop":
        lows = [t.lower() for t in it0[1]]
        if all(t.lower() in lows for t in cell_tags):
            if len(it0[2]) // len(it0[1]) != 1:
                raise StructureError("Expected one value, found " + str(len(it0[2]) // len(it0[1])))
            cell = [it0[2][lows.index(t.lower())] for t in cell_tags]
    else:
        pairs = {it[1]: it[2] for it in items if it[0] == "pair"}
        if all(t in pairs for t in cell_tags):
            cell = [pairs[t] for t in cell_tags]
    if cell is not None and not any(_cif_null(v) for v in cell[:3]):
        al, be, ga = (_cif_number(v, float("nan")) for v in cell[3:])
        if ga != 0.0 and (al == 0.0 or be == 0.0):
            raise StructureError("Impossible angle - N*180deg.")
    title = info(b"_entry.id") or info(b"_struct.title")
    want = [b"id", b"?group_PDB", b"type_symbol", b"?label_atom_id", b"label_alt_id", b"?label_comp_id", b"label_asym_id", b"?label_entity_id",
            b"?label_seq_id", b"?pdbx_PDB_ins_code", b"Cartn_x", b"Cartn_y", b"Cartn_z", b"occupancy", b"B_iso_or_equiv", b"?pdbx_formal_charge",
            b"auth_seq_id", b"?auth_comp_id", b"?auth_asym_id", b"?auth_atom_id", b"?pdbx_PDB_model_num", b"?calc_flag", b"?pdbx_tls_group_id"]
    (kId, kGroup, kSymbol, kLabelAtom, kAlt, kLabelComp, kLabelAsym, kLabelEntity, kLabelSeq, kIns, kX, kY, kZ, kOcc, kB, kCharge, kAuthSeq,
     kAuthComp, kAuthAsym, kAuthAtom, kModel, kCalc, kTls) = range(23)
    loop, _ = find_values(items, b"_atom_site.id")
    pos = []
    rows = []
    if loop is not None and loop[0] == "loop":
        lows = [t.lower() for t in loop[1]]
        for w in want:
            full = b"_atom_site." + w.lstrip(b"?")
            c = lows.index(full.lower()) if full.lower() in lows else -1
            if c < 0 and not w.startswith(b"?"):
                pos = []; break
            pos.append(c)
        if pos:
            width = len(loop[1])
            rows = [loop[2][r * width:(r + 1) * width] for r in range(len(loop[2]) // width)]
    else:
        pairs = {it[1]: it[2] for it in items if it[0] == "pair"}
        one = []
        for w in want:
            full = b"_atom_site." + w.lstrip(b"?")
            if full in pairs:
                pos.append(len(one)); one.append(pairs[full])
            elif w.startswith(b"?"):
                pos.append(-1)
            else:
                pos = []; break
        if pos:
            rows = [one]
    models: list = []            # [name, chains]; chain = [name, residues]; residue = [(seq, icode, name), atoms]
    if rows:
        kAsym = kAuthAsym if pos[kAuthAsym] >= 0 else kLabelAsym
        kComp = kAuthComp if pos[kAuthComp] >= 0 else kLabelComp
        kAtom = kAuthAtom if pos[kAuthAtom] >= 0 else kLabelAtom
        if pos[kComp] < 0:
            raise StructureError("Neither _atom_site.label_comp_id nor auth_comp_id found")
        if pos[kAtom] < 0:
            raise StructureError("Neither _atom_site.label_atom_id nor auth_atom_id found")

        def model_named(nm):
            for m in models:
                if m[0] == nm:
                    return m
            models.append([nm, []])
            return models[-1]
        model = model_named(_cif_string(rows[0][pos[kModel]]) if pos[kModel] >= 0 else b"1")
        chain = resi = None
        for row in rows:
            if pos[kModel] >= 0 and row[pos[kModel]] != model[0]:
                model = model_named(_cif_string(row[pos[kModel]])); chain = None
            asym = _cif_string(row[pos[kAsym]])
            if chain is None or asym != chain[0]:
                chain = [asym, []]; model[1].append(chain); resi = None
            seqs = _cif_string(row[pos[kAuthSeq]])
            icode = " "
            if pos[kIns] >= 0:
                v = row[pos[kIns]]
                if _cif_null(v):
                    icode = " "
                elif len(v) < 2:
                    icode = chr(v[0])
                else:
                    sv = _cif_string(v)
                    if len(sv) >= 2:
                        raise StructureError("Not a single character")
                    icode = chr(sv[0]) if sv else "\0"
            num = -999                                     # SeqId::OptionalNum::None
            if seqs:
                if seqs[-1] >= 0x41:
                    if icode == " ":
                        icode = chr(seqs[-1])
                    elif icode != chr(seqs[-1]):
                        raise StructureError("Inconsistent insertion code in " + seqs.decode("latin-1"))
                    num = _cif_int_checked(seqs[:-1])
                else:
                    num = _cif_int_checked(seqs)           # (as_int(seqid, None): a null value cannot get here, it is "" already)
            rid = (num, icode, _cif_string(row[pos[kComp]]))
            if resi is None or not (resi[0][0] == rid[0] and (ord(resi[0][1]) | 0x20) == (ord(rid[1]) | 0x20) and resi[0][2] == rid[2]):
                resi = next((r for r in chain[1] if r[0][0] == rid[0] and (ord(r[0][1]) | 0x20) == (ord(rid[1]) | 0x20) and r[0][2] == rid[2]), None)
                if resi is None:
                    resi = [rid, []]; chain[1].append(resi)
                if not resi[1]:
                    if pos[kLabelSeq] >= 0 and not _cif_null(row[pos[kLabelSeq]]):
                        _cif_int_checked(row[pos[kLabelSeq]])
            alt = row[pos[kAlt]]
            if not _cif_null(alt) and len(alt) >= 2 and len(_cif_string(alt)) >= 2:
                raise StructureError("Not a single character")
            if pos[kCharge] >= 0 and not _cif_null(row[pos[kCharge]]):
                _cif_int_checked(row[pos[kCharge]])
            m = re.match(rb"[ \t\n\v\f\r]*([+-]?)([0-9]*)", row[pos[kId]])
            serial = _wrap_i32(int(m.group(1) + m.group(2))) if m.group(2) else 0
            resi[1].append((_cif_string(row[pos[kAtom]]), serial, _cif_number(row[pos[kX]], float("nan")), _cif_number(row[pos[kY]], float("nan")),
                            _cif_number(row[pos[kZ]], float("nan")), np.float32(_cif_number(row[pos[kB]], 50.0))))
    atom, residue, chains, ai, ri, xyz, bf = [], [], [], [], [], [], []
    for mdl in models:
        for ch in mdl[1]:
            for rid, atoms in ch[1]:
                for (an, serial, x, y, z, b) in atoms:
                    atom.append(an.decode("latin-1")); residue.append(rid[2].decode("latin-1")); chains.append(ch[0].decode("latin-1"))
                    ai.append(serial); ri.append(rid[0]); xyz.append((x, y, z)); bf.append(b)
    t = AtomTable(atom, residue, chains, np.asarray(ai, np.int64).astype(np.int32), np.asarray(ri, np.int64).astype(np.int32),
                  np.asarray(xyz, np.float64).astype(np.float32).reshape(-1, 3), np.asarray(bf, np.float32))
    return t, title


def coor_format_from_content(data: bytes) -> str:
    """gemmi::coor_format_from_content (lib/gemmi/mmread.hpp:31-47): what StructureReader::loadFromBuffer -- the reader of every
    `compress` input, src/main.cpp:457 -- goes by; the file's extension only says whether it is gzipped"""
    i, end = 0, len(data) - 8
    while i < end:
        c = data[i]
        if c in b" \t\n\v\f\r":
            i += 1
        elif c == 0x23:
            while i < end and data[i] != 0x0a:
                i += 1
        elif c == 0x7b:
            return "mmjson"
        elif bytes(x & ~0x20 for x in data[i:i + 4]) == bytes(x & ~0x20 for x in b"data") and data[i + 4] == 0x5f:
            return "mmcif"
        else:
            return "pdb"
    return "unknown"


def parse_structure_gemmi(data: bytes):
    """the bytes of a structure file (already inflated) -> (AtomTable, title or "") as gemmi::read_structure_from_char_array +
    StructureReader::updateStructure make them; StructureError where the reader fails the file (mmJSON and chemical-component
    dictionaries, which gemmi would read, are failed here: no protein chain comes in them)"""
    fmt = coor_format_from_content(data)
    if fmt == "pdb":
        return parse_pdb_gemmi(data)
    if fmt == "mmcif":
        return parse_cif_gemmi(data)
    raise StructureError("wrong format of coordinate file" if fmt == "unknown" else "mmJSON input is not supported")


def remove_alternative_position(t: AtomTable) -> AtomTable:
    """Drop an atom whose name equals the previous (kept) atom's name (atom_coordinate.cpp:362-370)."""
    keep = np.ones(len(t), bool)
    prev = None
    for i, a in enumerate(t.atom):
        if prev is not None and a == prev:
            keep[i] = False
        else:
            prev = a
    return t if keep.all() else t.keep(keep)


def identify_chains(t: AtomTable) -> List[slice]:
    """identifyChains (src/atom_coordinate.cpp:469-497): split where the chain id changes; the new
    fragment must start at an atom named N, otherwise atoms are skipped up to the next N."""
    out = []
    n = len(t)
    start = 0
    i = 1
    while i < n:
        if t.chain[i] != t.chain[i - 1]:
            if t.atom[i] == "N":
                out.append(slice(start, i))
                start = i
            else:
                j = next((j for j in range(i, n) if t.atom[j] == "N"), None)
                if j is None:
                    break  # (the reference would spin here; nothing compressible follows)
                out.append(slice(start, i))
                start = j
                i = start
        i += 1
    out.append(slice(start, n))
    return out


def identify_discontinuous(t: AtomTable, sl: slice) -> List[slice]:
    """identifyDiscontinousResInd (src/atom_coordinate.cpp:506-530): look only at N atoms and split
    where consecutive residue numbers differ by more than 1; a fragment starts at its first N."""
    n_idx = [i for i in range(sl.start, sl.stop) if t.atom[i] == "N"]
    if not n_idx:
        return []
    out = []
    start = n_idx[0]
    for a, b in zip(n_idx[:-1], n_idx[1:]):
        if int(t.res_index[b]) - int(t.res_index[a]) > 1:
            out.append(slice(start, b))
            start = b
    out.append(slice(start, sl.stop))
    return out


@dataclass
class Chain:
    """One gap-free single-chain fragment = one FCZ record (one `Foldcomp` object in the reference)."""
    title: str
    atoms: AtomTable


def split_residues(t: AtomTable) -> np.ndarray:
    """Residue boundaries as atom offsets [n_res+1] (splitAtomByResidue, atom_coordinate.cpp:304-328:
    a new residue starts where residue_index changes; the last atom always joins the open residue)."""
    n = len(t)
    if n == 0:
        return np.zeros(1, np.uint32)
    ri = t.res_index
    change = np.nonzero(ri[1:] != ri[:-1])[0] + 1
    change = change[change != n - 1]  # quirk: the final atom never opens a residue of its own
    return np.concatenate(([0], change, [n])).astype(np.uint32)


@dataclass
class ChainBatch:
    """SoA batch of chains == `fcz_chain_batch` (include/fcz_hip.h)."""
    res_off: np.ndarray
    atom_off: np.ndarray
    x: np.ndarray
    y: np.ndarray
    z: np.ndarray
    atom_code: np.ndarray
    res_code: np.ndarray
    bfac_ca: np.ndarray
    first_res_index: np.ndarray
    first_atom_index: np.ndarray
    chain_id: np.ndarray      # uint8 (chars)
    titles: np.ndarray        # uint8
    title_off: np.ndarray
    anchor_threshold: int = 25
    _keep: list = field(default_factory=list, repr=False)

    @property
    def n_chains(self):
        return len(self.res_off) - 1

    @property
    def n_residues(self):
        return int(self.res_off[-1])

    @property
    def n_atoms(self):
        return int(self.atom_off[-1])


def build_batch(chains: Sequence[Chain], anchor_threshold: int = 25) -> ChainBatch:
    res_off = [0]
    atom_off_parts = []
    xs, codes, rcodes, bfs = [], [], [], []
    fri, fai, cid = [], [], []
    titles = bytearray()
    title_off = [0]
    abase = 0
    for ch in chains:
        t = ch.atoms
        if len(t) == 0:
            raise StructureError("empty chain")
        ro = split_residues(t)
        nres = len(ro) - 1
        if nres > 65535 or (anchor_threshold > 0 and nres // anchor_threshold + 2 > 255):
            # uint16 nResidue / uint8 nAnchor of the FCZ header (src/foldcomp.h:120-125) would wrap
            raise StructureError(f"chain of {nres} residues does not fit the FCZ header (65535 residues, 255 anchors)")
        ac = np.fromiter((ATOM_CODE.get(a, ATOM_CODE_OTHER) for a in t.atom), np.uint8, len(t))
        rc = np.empty(nres, np.uint8)
        bf = np.zeros(nres, np.float32)
        for r in range(nres):
            name = t.residue[ro[r]]
            code = RES_CODE.get(name)
            if code is None:
                # the reference throws std::out_of_range here (src/sidechain.cpp:177)
                raise StructureError(f"residue name {name!r} is not supported by the codec")
            rc[r] = code
            seg = ac[ro[r]:ro[r + 1]]
            # N, CA, C must be present (backbone = filterBackbone order), ABI precondition
            pos = [np.nonzero(seg == k)[0] for k in (0, 1, 2)]
            if any(len(p) == 0 for p in pos) or not (pos[0][0] < pos[1][0] < pos[2][0]):
                raise StructureError("residue without N, CA, C backbone atoms in order")
            # The reference works on the FLAT list of every atom named N, CA or C (filterBackbone, src/atom_coordinate.cpp:135-143;
            # nResidue = their number / 3, src/foldcomp.cpp:462) and on every CA's B-factor (:543-547); this codec on the first N,
            # CA, C of each residue. The two agree exactly when a residue has one of each -- a second one (a line that strayed into
            # the residue) shifts every later residue of the reference's record: such a chain is refused, not compressed differently
            if any(len(p) != 1 for p in pos):
                raise StructureError("residue with a second N, CA or C atom")
            bf[r] = t.bfac[ro[r] + pos[1][0]]
        # header.lastResidue is the residue name of the chain's LAST ATOM (src/foldcomp.cpp:469), the residue codes are those of
        # each residue's first atom (getResidueNameVector, src/atom_coordinate.cpp:330-345): one value in the batch serves both
        # only when they agree (splitAtomByResidue, :304-328, always puts the last atom into the last residue, whatever it says)
        if t.residue[len(t) - 1] != t.residue[ro[nres - 1]]:
            raise StructureError("the chain's last atom carries another residue name than its residue")
        atom_off_parts.append(ro[:-1] + abase)
        abase += len(t)
        res_off.append(res_off[-1] + nres)
        xs.append(t.xyz); codes.append(ac); rcodes.append(rc); bfs.append(bf)
        fri.append(int(t.res_index[0])); fai.append(int(t.atom_index[0]))
        cid.append(ord(t.chain[0][0]) if t.chain[0] else ord(" "))
        tb = ch.title.encode("latin-1", "replace")
        titles += tb
        title_off.append(len(titles))
    xyz = np.concatenate(xs) if xs else np.zeros((0, 3), np.float32)
    atom_off = np.concatenate(atom_off_parts + [np.asarray([abase])]).astype(np.uint32)
    return ChainBatch(
        res_off=np.asarray(res_off, np.uint32), atom_off=atom_off,
        x=np.ascontiguousarray(xyz[:, 0]), y=np.ascontiguousarray(xyz[:, 1]), z=np.ascontiguousarray(xyz[:, 2]),
        atom_code=np.concatenate(codes), res_code=np.concatenate(rcodes), bfac_ca=np.concatenate(bfs),
        first_res_index=np.asarray(fri, np.int32), first_atom_index=np.asarray(fai, np.int32),
        chain_id=np.asarray(cid, np.uint8), titles=np.frombuffer(bytes(titles) or b"\0", np.uint8).copy()[:len(titles)] if len(titles) else np.zeros(0, np.uint8),
        title_off=np.asarray(title_off, np.uint32), anchor_threshold=anchor_threshold)


# ---- ctypes view of the batch ---------------------------------------------------------------
class CChainBatch(ctypes.Structure):
    _fields_ = [
        ("n_chains", ctypes.c_uint32), ("n_residues", ctypes.c_uint32), ("n_atoms", ctypes.c_uint32),
        ("anchor_threshold", ctypes.c_int32),
        ("res_off", ctypes.c_void_p), ("atom_off", ctypes.c_void_p),
        ("x", ctypes.c_void_p), ("y", ctypes.c_void_p), ("z", ctypes.c_void_p),
        ("atom_code", ctypes.c_void_p), ("res_code", ctypes.c_void_p), ("bfac_ca", ctypes.c_void_p),
        ("first_res_index", ctypes.c_void_p), ("first_atom_index", ctypes.c_void_p),
        ("chain_id", ctypes.c_void_p), ("titles", ctypes.c_void_p), ("title_off", ctypes.c_void_p),
    ]


class CIngestResult(ctypes.Structure):
    """fcz_ingest_result (include/fcz_hip.h): the device-resident batch of an ingest call + per-chain / per-file arrays"""
    _fields_ = [("batch", CChainBatch), ("chain_file", ctypes.c_void_p), ("chain_meta", ctypes.c_void_p),
                ("file_status", ctypes.c_void_p), ("refused", ctypes.c_void_p), ("n_files", ctypes.c_uint32), ("n_refused", ctypes.c_uint32),
                ("chain_name4", ctypes.c_void_p)]


class CAtomsOut(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("y", ctypes.c_void_p), ("z", ctypes.c_void_p),
                ("bfac_res", ctypes.c_void_p), ("res_code", ctypes.c_void_p), ("atom_code", ctypes.c_void_p)]


class CEntryInfo(ctypes.Structure):
    _fields_ = [("n_residues", ctypes.c_uint32), ("n_atoms_out", ctypes.c_uint32),
                ("n_atoms_header", ctypes.c_uint32), ("first_res_index", ctypes.c_int32),
                ("first_atom_index", ctypes.c_int32), ("n_anchors", ctypes.c_uint32),
                ("n_sidechain_torsions", ctypes.c_uint32), ("title_off", ctypes.c_uint32),
                ("title_len", ctypes.c_uint32), ("chain_id", ctypes.c_char),
                ("first_residue", ctypes.c_char), ("last_residue", ctypes.c_char),
                ("has_oxt", ctypes.c_uint8), ("status", ctypes.c_int32)]


def _ptr(a: Optional[np.ndarray]):
    if a is None:
        return None
    return a.ctypes.data if a.size else None


def batch_as_c(b: ChainBatch) -> CChainBatch:
    """Host-pointer view (numpy arrays must stay alive while the struct is in use)."""
    s = CChainBatch()
    s.n_chains, s.n_residues, s.n_atoms = b.n_chains, b.n_residues, b.n_atoms
    s.anchor_threshold = int(b.anchor_threshold)
    for name in ("res_off", "atom_off", "x", "y", "z", "atom_code", "res_code", "bfac_ca",
                 "first_res_index", "first_atom_index", "chain_id", "titles", "title_off"):
        setattr(s, name, _ptr(getattr(b, name)))
    return s
