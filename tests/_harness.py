"""Test harness: ctypes access to the checkers (oracle/ C restatement, oracle/_ref real reference).

TEST INFRASTRUCTURE ONLY -- nothing here is imported by the product package.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import sys
from typing import List, Optional

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from foldcomp_amd.structure import (AtomTable, CAtomsOut, CChainBatch, CEntryInfo, Chain,  # noqa: E402
                                    ChainBatch, batch_as_c, build_batch)

ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libfcz_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libfoldcomp_ref.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE_TEST_DIR = "/root/reference/test"

_oracle = None
_ref = None


def build_oracle():
    src = os.path.join(ORACLE_DIR, "fcz_oracle.c")
    if (not os.path.exists(ORACLE_SO)) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libfcz_oracle.so"], stdout=subprocess.DEVNULL)


def load_oracle():
    global _oracle
    if _oracle is None:
        build_oracle()
        lib = ctypes.CDLL(ORACLE_SO)
        lib.fcz_oracle_compress_sizes.argtypes = [ctypes.POINTER(CChainBatch), ctypes.c_void_p]
        lib.fcz_oracle_compress_batch.argtypes = [ctypes.POINTER(CChainBatch), ctypes.c_void_p, ctypes.c_void_p,
                                                  ctypes.c_void_p, ctypes.c_int]
        lib.fcz_oracle_decompress_sizes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.fcz_oracle_decompress_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                                    ctypes.POINTER(CAtomsOut), ctypes.c_int]
        lib.fcz_oracle_check.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
        lib.fcz_oracle_sinf.restype = ctypes.c_float
        lib.fcz_oracle_sinf.argtypes = [ctypes.c_float]
        lib.fcz_oracle_cosf.restype = ctypes.c_float
        lib.fcz_oracle_cosf.argtypes = [ctypes.c_float]
        lib.fcz_oracle_use_restated_trig.argtypes = [ctypes.c_int]
        lib.fcz_oracle_angles_chain.argtypes = [ctypes.c_uint32] + [ctypes.c_void_p] * 13
        _oracle = lib
    return _oracle


def have_ref() -> bool:
    return os.path.exists(REF_SO)


def load_ref():
    global _ref
    if _ref is None:
        lib = ctypes.CDLL(REF_SO)
        lib.ref_compress.restype = ctypes.c_long
        lib.ref_compress.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 9 + [ctypes.c_char_p, ctypes.c_int, ctypes.c_int,
                                                                              ctypes.c_void_p, ctypes.c_long]
        lib.ref_decompress.restype = ctypes.c_int
        lib.ref_decompress.argtypes = [ctypes.c_char_p, ctypes.c_long, ctypes.c_int] + [ctypes.c_void_p] * 9 + \
                                      [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        lib.ref_decompress_pdb.restype = ctypes.c_long
        lib.ref_decompress_pdb.argtypes = [ctypes.c_char_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p, ctypes.c_long]
        lib.ref_extract.restype = ctypes.c_long
        lib.ref_extract.argtypes = [ctypes.c_char_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_long]
        lib.ref_angles.restype = ctypes.c_int
        lib.ref_angles.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 16 + [ctypes.c_int, ctypes.c_void_p]
        _ref = lib
    return _ref


def _names(strs: List[str], width: int) -> np.ndarray:
    a = np.full((len(strs), width), ord(" "), np.uint8)
    for i, s in enumerate(strs):
        b = s.encode()[:width]
        a[i, :len(b)] = np.frombuffer(b, np.uint8)
    return a


def _table_args(t: AtomTable):
    an = _names(t.atom, 4); rn = _names(t.residue, 3)
    ch = np.asarray([ord(c[0]) if c else 32 for c in t.chain], np.uint8)
    x = np.ascontiguousarray(t.xyz[:, 0]); y = np.ascontiguousarray(t.xyz[:, 1]); z = np.ascontiguousarray(t.xyz[:, 2])
    ai = np.ascontiguousarray(t.atom_index, np.int32); ri = np.ascontiguousarray(t.res_index, np.int32)
    bf = np.ascontiguousarray(t.bfac, np.float32)
    keep = (an, rn, ch, ai, ri, x, y, z, bf)
    return keep, [a.ctypes.data for a in keep]


def ref_compress(t: AtomTable, title: str, anchor_threshold: int = 25) -> bytes:
    lib = load_ref()
    keep, ptrs = _table_args(t)
    cap = 200 + len(title) + 64 * len(t)
    out = np.zeros(cap, np.uint8)
    tb = title.encode("latin-1")
    n = lib.ref_compress(len(t), *ptrs, tb, len(tb), anchor_threshold, out.ctypes.data, cap)
    if n < 0:
        raise RuntimeError(f"ref_compress failed: {n}")
    return out[:n].tobytes()


def ref_angles(t: AtomTable):
    lib = load_ref()
    keep, ptrs = _table_args(t)
    n = len(t)
    arrs = [np.zeros(n, np.float32) for _ in range(6)]
    sc = np.zeros(16 * n, np.float32)
    nsc = ctypes.c_int(0)
    m = lib.ref_angles(len(t), *ptrs, *[a.ctypes.data for a in arrs], sc.ctypes.data, len(sc), ctypes.byref(nsc))
    if m < 0:
        raise RuntimeError("ref_angles failed")
    names = ["phi", "psi", "omega", "n_ca_c", "ca_c_n", "c_n_ca"]
    d = {k: a[:m] for k, a in zip(names, arrs)}
    d["sc"] = sc[:nsc.value]
    return d


def ref_decompress(fcz: bytes, alt_order: bool = False):
    lib = load_ref()
    cap = 20 * (len(fcz) // 8 + 16)
    x = np.zeros(cap, np.float32); y = np.zeros(cap, np.float32); z = np.zeros(cap, np.float32)
    bf = np.zeros(cap, np.float32)
    an = np.zeros((cap, 4), np.uint8); rn = np.zeros((cap, 3), np.uint8)
    ai = np.zeros(cap, np.int32); ri = np.zeros(cap, np.int32); ch = np.zeros(cap, np.uint8)
    title = np.zeros(4096, np.uint8); tl = ctypes.c_int(0)
    n = lib.ref_decompress(fcz, len(fcz), int(alt_order), x.ctypes.data, y.ctypes.data, z.ctypes.data, bf.ctypes.data,
                           an.ctypes.data, rn.ctypes.data, ai.ctypes.data, ri.ctypes.data, ch.ctypes.data, cap,
                           title.ctypes.data, 4096, ctypes.byref(tl))
    if n < 0:
        raise RuntimeError(f"ref_decompress failed: {n}")
    return dict(x=x[:n], y=y[:n], z=z[:n], bfac=bf[:n],
                atom=[bytes(r).decode().strip() for r in an[:n]], residue=[bytes(r).decode().strip() for r in rn[:n]],
                atom_index=ai[:n], res_index=ri[:n], chain=[chr(c) for c in ch[:n]],
                title=bytes(title[:tl.value]).decode("latin-1"))


def ref_decompress_pdb(fcz: bytes, alt_order: bool = False) -> str:
    lib = load_ref()
    cap = 100 * (len(fcz) + 64) + 4096
    out = np.zeros(cap, np.uint8)
    n = lib.ref_decompress_pdb(fcz, len(fcz), int(alt_order), out.ctypes.data, cap)
    if n < 0:
        raise RuntimeError("ref_decompress_pdb failed")
    return out[:n].tobytes().decode("latin-1")


def ref_extract(fcz: bytes, kind: int, digits: int) -> str:
    lib = load_ref()
    cap = 16 * len(fcz) + 256
    out = np.zeros(cap, np.uint8)
    n = lib.ref_extract(fcz, len(fcz), kind, digits, out.ctypes.data, cap)
    if n < 0:
        raise RuntimeError("ref_extract failed")
    return out[:n].tobytes().decode("latin-1")


# ---- oracle batch wrappers -------------------------------------------------------------------
def ref_load_structure(data: bytes, name: str):
    """the reference's own reader (gemmi PDB / mmCIF, gz by name) + removeAlternativePosition + the fragmenting of
    src/main.cpp:457-474 -> (AtomTable, title, [(first, end, chain ordinal, fragment ordinal)], n_chains)"""
    lib = load_ref()
    lib.ref_load_structure.restype = ctypes.c_long
    lib.ref_load_structure.argtypes = [ctypes.c_char_p, ctypes.c_long, ctypes.c_char_p, ctypes.c_long] + [ctypes.c_void_p] * 9 + \
                                      [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    cap = max(16, len(data) // 8 + 4096) if not name.endswith(".gz") else 1 << 20
    an = np.zeros((cap, 4), np.uint8); rn = np.zeros((cap, 3), np.uint8); ch = np.zeros(cap, np.uint8)
    ai = np.zeros(cap, np.int32); ri = np.zeros(cap, np.int32)
    x = np.zeros(cap, np.float32); y = np.zeros(cap, np.float32); z = np.zeros(cap, np.float32); bf = np.zeros(cap, np.float32)
    title = ctypes.create_string_buffer(4096); tl = ctypes.c_int(); nf = ctypes.c_int(); nc = ctypes.c_int()
    frag = np.zeros((4096, 4), np.int32)
    n = lib.ref_load_structure(data, len(data), name.encode(), cap, an.ctypes.data, rn.ctypes.data, ch.ctypes.data, ai.ctypes.data,
                               ri.ctypes.data, x.ctypes.data, y.ctypes.data, z.ctypes.data, bf.ctypes.data, title, 4096,
                               ctypes.byref(tl), frag.ctypes.data, 4096, ctypes.byref(nf), ctypes.byref(nc))
    if n < 0:
        raise RuntimeError(f"ref_load_structure({name}) = {n}")
    def strs(a):
        return [bytes(r).rstrip(b"\0 ").decode() for r in a[:n]]
    t = AtomTable(strs(an), strs(rn), [chr(c) for c in ch[:n]], ai[:n].copy(), ri[:n].copy(),
                  np.stack([x[:n], y[:n], z[:n]], 1).copy(), bf[:n].copy())
    return t, title.raw[:tl.value].decode("latin-1"), [tuple(int(v) for v in f) for f in frag[:nf.value]], nc.value


def ref_db_write(path: str, entries, keys, names):
    """the reference's database writer (make_writer / writer_append / free_writer) on the given entries"""
    lib = load_ref()
    lib.ref_db_write.restype = ctypes.c_int
    lib.ref_db_write.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p]
    blob = b"".join(entries)
    off = np.zeros(len(entries) + 1, np.uint64); off[1:] = np.cumsum([len(e) for e in entries])
    k = np.asarray(keys, np.uint32)
    nm = b"".join(n.encode() + b"\0" for n in names)
    rc = lib.ref_db_write(path.encode(), (path + ".index").encode(), len(entries), blob, off.ctypes.data, k.ctypes.data, nm)
    assert rc == 0, rc


def ref_db_read(path: str):
    """every entry of a database through the reference's reader, in the reader's order: [(key, offset, length, name, data)]"""
    lib = load_ref()
    lib.ref_db_read.restype = ctypes.c_long
    lib.ref_db_read.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_long]
    key = ctypes.c_uint(); ln = ctypes.c_longlong(); of = ctypes.c_longlong(); name = ctypes.create_string_buffer(1024)
    n = lib.ref_db_read(path.encode(), (path + ".index").encode(), -1, ctypes.byref(key), ctypes.byref(ln), ctypes.byref(of), name, 1024, None, 0)
    out = []
    for i in range(n):
        lib.ref_db_read(path.encode(), (path + ".index").encode(), i, ctypes.byref(key), ctypes.byref(ln), ctypes.byref(of), name, 1024, None, 0)
        buf = ctypes.create_string_buffer(max(1, ln.value))
        lib.ref_db_read(path.encode(), (path + ".index").encode(), i, ctypes.byref(key), ctypes.byref(ln), ctypes.byref(of), name, 1024, buf, ln.value)
        out.append((key.value, of.value, ln.value, name.value.decode(), buf.raw[:ln.value]))
    return out


def ref_db_lookup(path: str, name: str) -> int:
    lib = load_ref()
    lib.ref_db_lookup.restype = ctypes.c_long
    lib.ref_db_lookup.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
    return int(lib.ref_db_lookup(path.encode(), (path + ".index").encode(), name.encode()))


def oracle_compress(b: ChainBatch, n_threads: int = 1):
    """-> (blob uint8, out_off uint64[C+1], status int32[C])"""
    lib = load_oracle()
    cb = batch_as_c(b)
    off = np.zeros(b.n_chains + 1, np.uint64)
    lib.fcz_oracle_compress_sizes(ctypes.byref(cb), off.ctypes.data)
    out = np.zeros(int(off[-1]), np.uint8)
    st = np.zeros(b.n_chains, np.int32)
    lib.fcz_oracle_compress_batch(ctypes.byref(cb), off.ctypes.data, out.ctypes.data, st.ctypes.data, n_threads)
    return out, off, st


def oracle_decompress(blob: np.ndarray, off: np.ndarray, alt_order: bool = False, n_threads: int = 1,
                      restated_trig: bool = False):
    lib = load_oracle()
    n = len(off) - 1
    blob = np.ascontiguousarray(blob, np.uint8); off = np.ascontiguousarray(off, np.uint64)
    info = (CEntryInfo * n)()
    res_off = np.zeros(n + 1, np.uint32); atom_off = np.zeros(n + 1, np.uint32)
    lib.fcz_oracle_decompress_sizes(blob.ctypes.data, off.ctypes.data, n, ctypes.addressof(info), res_off.ctypes.data,
                                    atom_off.ctypes.data)
    M, R = int(atom_off[-1]), int(res_off[-1])
    x = np.zeros(M, np.float32); y = np.zeros(M, np.float32); z = np.zeros(M, np.float32)
    bf = np.zeros(R, np.float32); rc = np.zeros(R, np.uint8); ac = np.zeros(M, np.uint8)
    out = CAtomsOut(x.ctypes.data or None, y.ctypes.data or None, z.ctypes.data or None, bf.ctypes.data or None,
                    rc.ctypes.data or None, ac.ctypes.data or None)
    lib.fcz_oracle_use_restated_trig(int(restated_trig))
    lib.fcz_oracle_decompress_batch(blob.ctypes.data, off.ctypes.data, n, res_off.ctypes.data, atom_off.ctypes.data,
                                    int(alt_order), ctypes.byref(out), n_threads)
    lib.fcz_oracle_use_restated_trig(0)
    return dict(x=x, y=y, z=z, bfac_res=bf, res_code=rc, atom_code=ac, res_off=res_off, atom_off=atom_off, info=info)


def mask_pad(fcz: bytes) -> bytes:
    """zero the 4 uninitialised CompressedFileHeader padding bytes"""
    b = bytearray(fcz)
    for o in (14, 15, 22, 23):
        if o < len(b):
            b[o] = 0
    return bytes(b)


# ---- the live reference in a child process ---------------------------------------------------
def _ref_child(conn):
    import faulthandler
    faulthandler.disable()          # a crash of the reference in this child is an answer ("crash"), not a report to print
    while True:
        try:
            what, args = conn.recv()
        except EOFError:
            return
        try:
            if what == "load":
                conn.send(("ok",) + tuple(ref_load_structure(*args)))
            elif what == "compress":
                conn.send(("ok", mask_pad(ref_compress(*args))))
            else:
                conn.send(("ok", globals()[what](*args)))
        except RuntimeError:
            conn.send(("fail",))


class RefWorker:
    """the live reference in a child process: on some mutated inputs the reference itself crashes or never returns (identifyChains
    spins when a chain id changes at a non-N atom with no N after it); such inputs have no reference answer. A dead child is
    noticed at once (its sentinel), a spinning one after `timeout` seconds"""

    def __init__(self, timeout=10.0):
        import multiprocessing as mp
        self.mp = mp.get_context("fork")
        self.timeout = timeout
        self.proc = self.conn = None

    def _start(self):
        self.conn, child = self.mp.Pipe()
        self.proc = self.mp.Process(target=_ref_child, args=(child,), daemon=True)
        self.proc.start()
        child.close()

    def run(self, what, *args):
        from multiprocessing.connection import wait
        if self.proc is None:
            self._start()
        self.conn.send((what, args))
        ready = wait([self.conn, self.proc.sentinel], self.timeout)
        if self.conn in ready:
            try:
                return self.conn.recv()
            except EOFError:
                pass
        self.close()
        return ("crash",)

    def load(self, data, name):
        return self.run("load", data, name)

    def compress(self, t, title, thr=25):
        return self.run("compress", t, title, thr)

    def close(self):
        if self.proc is not None:
            self.proc.kill(); self.proc.join(); self.conn.close()
        self.proc = self.conn = None

    def call(self, fn_name, *args):
        """any ref_* function of this module by name -> ("ok", result) | ("fail",) | ("crash",)"""
        return self.run(fn_name, *args)
