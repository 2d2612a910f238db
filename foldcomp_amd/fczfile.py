"""Host-side view of one FCZ record: header fields, title, de-quantised angle lists, `extract` strings.

Pure byte/integer work plus a handful of float32 multiply-adds on values that are already in the record
(mirrors Foldcomp::read header parsing src/foldcomp.cpp:904-1036, Foldcomp::extract :1260-1336 and the
list fields of the Python get_data dict foldcomp/foldcomp.cxx:495-600). Coordinates always come from the
GPU decompressor.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass

import numpy as np

from ._aa_tables import RES1

MAGIC = b"FCMP"


class FczFormatError(ValueError):
    pass


@dataclass
class FczRecord:
    raw: bytes
    n_residues: int
    n_atoms: int
    first_res_index: int
    first_atom_index: int
    n_anchors: int
    chain: str
    n_sidechain: int
    first_residue: str
    last_residue: str
    title: str
    mins: np.ndarray      # float32[6]  phi psi omega n_ca_c ca_c_n c_n_ca
    cont_fs: np.ndarray   # float32[6]
    o_words: int
    o_sc: int
    o_tmp: int

    @property
    def words(self) -> np.ndarray:
        return np.frombuffer(self.raw, np.uint8, 8 * self.n_residues, self.o_words).reshape(-1, 8)

    @property
    def res_codes(self) -> np.ndarray:
        return self.words[:, 0] >> 3

    @property
    def has_oxt(self) -> bool:
        return self.raw[self.o_words - 13] != 0


def parse(raw: bytes) -> FczRecord:
    if len(raw) < 76 or raw[:4] != MAGIC:
        raise FczFormatError("not an FCZ record")
    n, na, ir, ia = struct.unpack_from("<HHHH", raw, 4)
    n_anchor = raw[12]
    chain = chr(raw[13])
    nsc, = struct.unpack_from("<I", raw, 16)
    fr, lr = chr(raw[20]), chr(raw[21])
    tl, = struct.unpack_from("<I", raw, 24)
    mins = np.frombuffer(raw, np.float32, 6, 28).copy()
    cfs = np.frombuffer(raw, np.float32, 6, 52).copy()
    o_title = 76 + 4 * n_anchor
    o_words = o_title + tl + 36 * n_anchor + 13
    o_sc = o_words + 8 * n
    o_tmp = o_sc + nsc
    if len(raw) < o_tmp + 8 + n:
        raise FczFormatError("truncated FCZ record")
    title = raw[o_title:o_title + tl].decode("latin-1")
    return FczRecord(raw, n, na, ir, ia, n_anchor, chain, nsc, fr, lr, title, mins, cfs, o_words, o_sc, o_tmp)


def record_size(raw) -> int:
    """byte length of the FCZ record that starts at raw[0] according to its own header (Foldcomp::getSize,
    src/foldcomp.cpp:1190-1214), -1 when raw does not start with an FCZ header"""
    if len(raw) < 76 or bytes(raw[:4]) != MAGIC:
        return -1
    n, = struct.unpack_from("<H", raw, 4)
    nsc, = struct.unpack_from("<I", raw, 16)
    tl, = struct.unpack_from("<I", raw, 24)
    return 76 + 4 * raw[12] + tl + 36 * raw[12] + 13 + 8 * n + nsc + 8 + n


def unpack_fields(rec: FczRecord):
    """-> dict of uint32 arrays (convertBytesToBackboneChain, src/foldcomp.cpp:60-77)"""
    w = rec.words.astype(np.uint32)
    return dict(res=w[:, 0] >> 3, omega=((w[:, 0] & 7) << 8) | w[:, 1], psi=(w[:, 2] << 4) | (w[:, 3] >> 4),
                phi=((w[:, 3] & 15) << 8) | w[:, 4], ca_c_n=w[:, 5], c_n_ca=w[:, 6], n_ca_c=w[:, 7])


def _deq(q, mn, cf):
    return (q.astype(np.float32) * np.float32(cf)) + np.float32(mn)


def angle_lists(rec: FczRecord):
    """phi/psi/omega/torsion_angles/bond_angles as Foldcomp::decompress leaves them (src/foldcomp.cpp:784-804)"""
    f = unpack_fields(rec)
    phi = _deq(f["phi"], rec.mins[0], rec.cont_fs[0]); psi = _deq(f["psi"], rec.mins[1], rec.cont_fs[1])
    omg = _deq(f["omega"], rec.mins[2], rec.cont_fs[2])
    nca = _deq(f["n_ca_c"], rec.mins[3], rec.cont_fs[3]); can = _deq(f["ca_c_n"], rec.mins[4], rec.cont_fs[4])
    cna = _deq(f["c_n_ca"], rec.mins[5], rec.cont_fs[5])
    n = rec.n_residues
    tors = np.stack([psi[:n - 1], omg[:n - 1], phi[:n - 1]], 1).reshape(-1)
    bonds = np.stack([can, cna, nca], 1).reshape(-1)
    return dict(phi=phi, psi=psi, omega=omg, torsion_angles=tors, bond_angles=bonds)


def temp_factors(rec: FczRecord) -> np.ndarray:
    mn, cf = struct.unpack_from("<ff", rec.raw, rec.o_tmp)
    q = np.frombuffer(rec.raw, np.uint8, rec.n_residues, rec.o_tmp + 8)
    return _deq(q, mn, cf)


def sequence(rec: FczRecord) -> str:
    """one-letter codes from the 5-bit field (extract type 1, src/foldcomp.cpp:1326-1333)"""
    return "".join(RES1[c] if c < 24 else "X" for c in rec.res_codes)


def fasta_like(title: str, data: str) -> str:
    return f">{title}\n{data}\n"          # writeFASTALike, src/foldcomp.cpp:1223-1231


def tsv_line(title: str, n_res: int, data: str) -> str:
    return f"{title}\t{n_res}\t{data}\n"  # writeTSV, src/foldcomp.cpp:1233-1237
