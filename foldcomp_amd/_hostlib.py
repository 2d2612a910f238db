"""The C++ host's structure readers as a library (host/libfcz_host.so, built by `make -C host`): the same rules as
structure.parse_pdb_gemmi / parse_cif_gemmi (the two implementations are held equal on mutated files,
tests/test_ingest_vs_reference.py), forty times the speed. Host code only: no device work happens here.
FCZ_PY_READERS=1 makes the command line use the Python readers instead."""
from __future__ import annotations

import ctypes
import os

import numpy as np

from .structure import AtomTable, StructureError

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None
_tried = False


class _CAtoms(ctypes.Structure):
    _fields_ = [("n", ctypes.c_uint64), ("atom", ctypes.c_void_p), ("residue", ctypes.c_void_p), ("chain", ctypes.c_void_p),
                ("atom_bytes", ctypes.c_uint64), ("residue_bytes", ctypes.c_uint64), ("chain_bytes", ctypes.c_uint64),
                ("atom_index", ctypes.c_void_p), ("res_index", ctypes.c_void_p),
                ("x", ctypes.c_void_p), ("y", ctypes.c_void_p), ("z", ctypes.c_void_p), ("bfac", ctypes.c_void_p),
                ("title", ctypes.c_void_p), ("title_len", ctypes.c_uint64), ("error", ctypes.c_char * 256)]


def load():
    """the library, or None when it has not been built (or FCZ_PY_READERS is set)"""
    global _lib, _tried
    if _tried:
        return _lib
    _tried = True
    if os.environ.get("FCZ_PY_READERS"):
        return None
    path = os.environ.get("FCZ_HOST_LIB") or os.path.join(_ROOT, "host", "libfcz_host.so")
    if not os.path.exists(path):
        return None
    try:
        lib = ctypes.CDLL(path)
    except OSError:
        return None
    lib.fcz_host_read_structure.restype = ctypes.c_int
    lib.fcz_host_read_structure.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(_CAtoms)]
    lib.fcz_host_free.restype = None
    lib.fcz_host_free.argtypes = [ctypes.POINTER(_CAtoms)]
    _lib = lib
    return lib


def read_structure(data: bytes, gz: bool = False):
    """-> (AtomTable, title or "") as structure.parse_structure_gemmi gives them; StructureError where the reader fails the file"""
    lib = load()
    if lib is None:
        raise RuntimeError("host/libfcz_host.so is not available")
    out = _CAtoms()
    rc = lib.fcz_host_read_structure(data, len(data), int(gz), ctypes.byref(out))
    if rc != 0:
        raise StructureError(out.error.decode("latin-1"))
    try:
        n = int(out.n)

        def names(ptr, nbytes):
            raw = ctypes.string_at(ptr, int(nbytes)) if nbytes else b""
            return [s.decode("latin-1") for s in raw.split(b"\0")[:n]]

        def arr(ptr, dt):
            return np.frombuffer(ctypes.string_at(ptr, 4 * n), dt).copy() if n else np.zeros(0, dt)
        xyz = np.stack([arr(out.x, np.float32), arr(out.y, np.float32), arr(out.z, np.float32)], 1) if n else np.zeros((0, 3), np.float32)
        t = AtomTable(names(out.atom, out.atom_bytes), names(out.residue, out.residue_bytes), names(out.chain, out.chain_bytes),
                      arr(out.atom_index, np.int32), arr(out.res_index, np.int32), xyz, arr(out.bfac, np.float32))
        title = ctypes.string_at(out.title, int(out.title_len)).decode("latin-1") if out.title_len else ""
    finally:
        lib.fcz_host_free(ctypes.byref(out))
    return t, title
