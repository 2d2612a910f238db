"""Python surface of the reference's `foldcomp` module on top of the MI355X codec.

Same names, argument meaning and error behaviour as foldcomp/foldcomp.cxx:
    compress(name, pdb_content, *, anchor_residue_threshold=25) -> bytes          (:295-328)
    decompress(fcz_bytes) -> (name, pdb_text)                                      (:222-239)
    get_data(bytes_or_str) -> dict(phi, psi, omega, torsion_angles, bond_angles,
                                   residues, b_factors, coordinates)               (:673-695)
    open(path, *, ids=None, decompress=True, err_on_missing=False) -> FoldcompDatabase (:333-435)
plus batch forms (`compress_many`, `decompress_many`) because one chain per call cannot fill a GPU.
All geometry runs on the GPU through libfcz_hip.so; only text parsing/formatting happens here.
"""
from __future__ import annotations

import os
import sys
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import fczfile
from ._aa_tables import RES1, RES3
from . import _lib
from .codec import Codec
from .database import DatabaseReader
from .structure import (Chain, MultipleChainsError, StructureError, build_batch, parse_pdb, remove_alternative_position)

DEFAULT_ANCHOR_THRESHOLD = 25


class error(Exception):
    """foldcomp.error of the reference module"""


FoldcompError = error
_codec: Optional[Codec] = None


def default_codec() -> Codec:
    global _codec
    if _codec is None or _codec.ctx is None:
        _codec = Codec(int(os.environ.get("FOLDCOMP_AMD_DEVICE", "0")))
    return _codec


def set_codec(c: Optional[Codec]):
    global _codec
    _codec = c


# ---- compress ------------------------------------------------------------------------------------
def _chain_from_pdb(name: str, pdb_content: str) -> Chain:
    try:
        t = parse_pdb(pdb_content, single_chain=True)
    except MultipleChainsError:
        raise error("Multiple chains found. Please provide a single chain using 'foldcomp.split_pdb_by_chain'")
    except StructureError as e:
        # (std::stoi / std::stof / substr throw inside the reference's extension and end the interpreter: an exception here)
        raise error(f"Error parsing the PDB string: {e}")
    if len(t) == 0:
        raise error("No ATOM lines found")
    return Chain(name, remove_alternative_position(t))


def compress_many(items: Sequence[Tuple[str, str]], *, anchor_residue_threshold: int = DEFAULT_ANCHOR_THRESHOLD,
                  codec: Optional[Codec] = None) -> List[bytes]:
    """[(name, pdb_content), ...] -> [fcz bytes, ...] in one GPU batch"""
    if not isinstance(anchor_residue_threshold, int):
        raise TypeError("anchor_residue_threshold must be an integer")
    chains = [_chain_from_pdb(n, p) for n, p in items]
    try:
        batch = build_batch(chains, anchor_residue_threshold)
    except StructureError as e:
        raise error(f"Error compressing: {e}")
    c = codec or default_codec()
    blob, off, st = c.compress_batch(batch, strict=False)
    if (st != 0).any():
        # (a refused chain: residue names the reference cannot process, fewer than two residues, a NaN / infinite coordinate or
        # B-factor -- FCZ_E_NONFINITE, include/fcz_hip.h)
        bad = int(st[st != 0][0])
        raise error("Error compressing: " + _lib.load().fcz_status_string(bad).decode())
    return [blob[off[i]:off[i + 1]].tobytes() for i in range(len(chains))]


def compress(name: str, pdb_content: str, *, anchor_residue_threshold: int = DEFAULT_ANCHOR_THRESHOLD) -> bytes:
    if not isinstance(name, str) or not isinstance(pdb_content, str):
        raise TypeError("compress(name: str, pdb_content: str)")
    return compress_many([(name, pdb_content)], anchor_residue_threshold=anchor_residue_threshold)[0]


# ---- decompress ----------------------------------------------------------------------------------
def decompress_many(entries: Sequence[bytes], *, alt_order: bool = False, codec: Optional[Codec] = None,
                    skip_bad: bool = False) -> List[Optional[Tuple[str, str]]]:
    """[fcz, ...] -> [(name, pdb_text), ...] in one GPU batch"""
    c = codec or default_codec()
    off = np.zeros(len(entries) + 1, np.uint64)
    off[1:] = np.cumsum([len(e) for e in entries])
    blob = np.frombuffer(b"".join(entries), np.uint8) if entries else np.zeros(0, np.uint8)
    texts, status = c.decompress_pdb(blob, off, alt_order=alt_order)   # reconstruction and PDB text both on the device
    out = []
    for i, e in enumerate(entries):
        if status[i] != 0:
            if skip_bad:
                out.append(None); continue
            raise error("Error decompressing.")
        rec = fczfile.parse(e)
        out.append((rec.title, texts[i].decode("latin-1")))
    return out


def decompress(fcz: bytes) -> Tuple[str, str]:
    if not isinstance(fcz, (bytes, bytearray, memoryview)):
        raise TypeError(f"a bytes-like object is required, not '{type(fcz).__name__}'")      # ("y#", foldcomp.cxx:204)
    return decompress_many([bytes(fcz)])[0]


# ---- get_data ------------------------------------------------------------------------------------
def get_data(input) -> dict:  # noqa: A002
    if isinstance(input, str):
        raw = input.encode("latin-1")
    elif isinstance(input, (bytes, bytearray, memoryview)):
        raw = bytes(input)
    else:
        raise TypeError(f"a bytes-like object is required, not '{type(input).__name__}'")
    if len(raw) == 0:
        raise ValueError("Input is empty")
    if len(raw) >= 4 and raw[:4] == b"FCMP":
        try:
            rec = fczfile.parse(raw)
        except fczfile.FczFormatError:
            raise ValueError("Could not read FCZ file")
        c = default_codec()
        off = np.asarray([0, len(raw)], np.uint64)
        d = c.decompress_batch(np.frombuffer(raw, np.uint8), off)
        if d["info"][0].status != 0:
            raise ValueError("Could not decompress FCZ file")
        a = fczfile.angle_lists(rec)
        coords = list(zip(d["x"].tolist(), d["y"].tolist(), d["z"].tolist()))
        return dict(phi=a["phi"].tolist(), psi=a["psi"].tolist(), omega=a["omega"].tolist(),
                    torsion_angles=a["torsion_angles"].tolist(), bond_angles=a["bond_angles"].tolist(),
                    residues=fczfile.sequence(rec), b_factors=fczfile.temp_factors(rec).tolist(), coordinates=coords)
    if len(raw) >= 4:
        # PDB text: note the reference skips removeAlternativePosition here (foldcomp.cxx:633-662)
        t = parse_pdb(raw.decode("latin-1"))
        if len(t) == 0:
            raise ValueError("No ATOM lines found in PDB file")
        batch = build_batch([Chain("", t)], DEFAULT_ANCHOR_THRESHOLD)
        ang = default_codec().compress_angles(batch)
        n = batch.n_residues
        phi, psi, omg, nca, can, cna = (ang[q][:n - 1] for q in range(6))
        tors = np.stack([psi, omg, phi], 1).reshape(-1)
        # getBondAngles order: angle at atom 1 (first N-CA-C), then (ca_c_n, c_n_ca, n_ca_c) per window
        bonds = np.concatenate([[ang[3][n - 1]], np.stack([can, cna, nca], 1).reshape(-1)]).astype(np.float32)
        return dict(phi=phi.tolist(), psi=psi.tolist(), omega=omg.tolist(), torsion_angles=tors.tolist(),
                    bond_angles=bonds.tolist(), residues="".join(RES1[c] for c in batch.res_code),
                    b_factors=batch.bfac_ca.tolist(), coordinates=list(map(tuple, t.xyz.tolist())))
    raise ValueError("Input is not a FCZ file or PDB file")


# ---- open ----------------------------------------------------------------------------------------
class FoldcompDatabase:
    """Sequence over a Foldcomp database (foldcomp.cxx:44-185): len(), db[i], iteration, context manager."""

    def __init__(self, path, ids=None, decompress=True, err_on_missing=False):
        try:
            self._reader = DatabaseReader(os.fspath(path), use_lookup=bool(ids))
        except ValueError as e:                  # an index line the reader cannot take (database.py): the module's own error, not a stray ValueError
            raise error(str(e)) from None
        self._decompress = decompress
        self._ids = None
        if ids:
            self._ids = []
            for name in ids:
                i = self._reader.id_of_name(name)
                if i < 0:
                    msg = f"Skipping entry {name} which is not in the database."
                    if err_on_missing:
                        self._reader.close()
                        raise KeyError(msg)
                    print(msg, file=sys.stderr)
                    continue
                self._ids.append(i)

    def __len__(self):
        return len(self._ids) if self._ids is not None else len(self._reader)

    def _entry(self, index: int) -> bytes:
        if index < 0 or index >= len(self):
            raise IndexError("index out of range")
        i = self._ids[index] if self._ids is not None else index
        # The reference drops the last byte of every entry (foldcomp.cxx:66,73), which is right for MMseqs2-made databases
        # (entries end in a NUL) and one byte short for the ones `foldcomp compress --db` writes (no terminator,
        # src/main.cpp:516). Here only a real terminator is dropped: the byte after the record's header-derived length, or
        # the NUL that closes a non-FCZ (text) entry.
        data = self._reader.data(i)
        size = fczfile.record_size(data)
        if size >= 0:
            return data[:size] if len(data) > size else data
        return data[:-1] if data.endswith(b"\0") else data

    # The reference's module is per entry by construction (foldcomp.cxx:44-90: one Foldcomp::read + decompress + PDB text per
    # __getitem__, :197-220). A GPU call per entry would spend its time in launches and copies, so sequential access -- the loop a
    # program written against the reference runs, `for name, pdb in db:` or db[0], db[1], ... -- is served from a read-ahead
    # WINDOW: READAHEAD entries (FOLDCOMP_READAHEAD, default 1024: ~240 MB of text at 350 residues) decoded and formatted by ONE
    # fcz_decompress_pdb call. Same values, same exception at the entry that does not decode; random access decodes the one entry.
    READAHEAD = max(1, int(os.environ.get("FOLDCOMP_READAHEAD", "1024")))

    def _window(self, start: int):
        ents = [self._entry(i) for i in range(start, min(start + self.READAHEAD, len(self)))]
        self._win_start, self._win = start, decompress_many(ents, skip_bad=True)

    def __getitem__(self, index):
        index = int(index)
        if index < 0:
            index += len(self)                        # the sequence protocol of the reference's type (sq_item behind PySequence_GetItem): db[-1] is the last entry
        if not self._decompress:
            return self._entry(index)
        if index < 0 or index >= len(self):
            raise IndexError("index out of range")
        win = getattr(self, "_win", None)
        if win is not None and self._win_start <= index < self._win_start + len(win):
            r = win[index - self._win_start]
        elif index == getattr(self, "_next", 0):
            self._window(index)                       # the access after the last one (or the first): the caller is walking the database
            r = self._win[0]
        else:
            r = decompress_many([self._entry(index)], skip_bad=True)[0]
        self._next = index + 1
        if r is None:
            raise error("Error decompressing: ")
        return r

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def decompress_all(self, batch: int = 4096):
        """GPU-sized iteration: yields (name, pdb) for every entry, decoding `batch` entries per launch"""
        for s in range(0, len(self), batch):
            ents = [self._entry(i) for i in range(s, min(s + batch, len(self)))]
            for r in decompress_many(ents):
                yield r

    def close(self):
        self._win = None
        if self._reader is not None:
            self._reader.close()
            self._reader = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def open(path, *, ids=None, decompress=True, err_on_missing=False) -> FoldcompDatabase:  # noqa: A001
    if ids is not None and not isinstance(ids, list):
        raise TypeError("user_ids must be a list.")
    if not isinstance(decompress, bool):
        raise TypeError("decompress must be a boolean")
    if not isinstance(err_on_missing, bool):
        raise TypeError("err_on_missing must be a boolean")
    return FoldcompDatabase(path, ids=ids, decompress=decompress, err_on_missing=err_on_missing)


def split_pdb_by_chain(pdb_str: str):
    """Split a PDB string into a list of PDB strings, one per run of ATOM lines with the same chain id
    (foldcomp/util.py:1-18)."""
    pdb_list, chain, cur = [], None, ""
    for line in pdb_str.splitlines():
        if line.startswith("ATOM"):
            if chain is None:
                chain = line[21]
            elif line[21] != chain:
                pdb_list.append(cur); cur = ""; chain = line[21]
            cur += line + "\n"
    pdb_list.append(cur)
    return pdb_list
