"""Batch codec object over the C-ABI: the host-side mirror of the reference's `Foldcomp` class
(src/foldcomp.h:267-402) for many chains at once.

    Foldcomp::compress + writeStream   ->  Codec.compress_batch(ChainBatch)   -> FCZ blob + offsets
    Foldcomp::read + decompress        ->  Codec.decompress_batch(blob, off)  -> SoA atoms

Host numpy arrays in, host numpy arrays out (copies ride the ctx stream). The device-resident entry
points used by bench.py take raw device pointers (e.g. torch tensors' .data_ptr()).
"""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np

from . import _lib
from .structure import CAtomsOut, CChainBatch, CEntryInfo, ChainBatch, batch_as_c


class Codec:
    def __init__(self, device: int = 0):
        self.lib = _lib.load()
        h = ctypes.c_void_p()
        _lib.check(self.lib.fcz_ctx_create(int(device), ctypes.byref(h)), "fcz_ctx_create")
        self.ctx = h
        self.device = device

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.fcz_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def stream(self) -> int:
        return int(self.lib.fcz_ctx_stream(self.ctx) or 0)

    def set_numerics(self, fast: bool):
        """decompress numerics: False = bit-identical to the reference (default), True = plain float arithmetic (FCZ_NUMERICS_FAST)"""
        _lib.check(self.lib.fcz_ctx_set_numerics(self.ctx, 1 if fast else 0), "fcz_ctx_set_numerics")

    def synchronize(self):
        _lib.check(self.lib.fcz_ctx_synchronize(self.ctx), "fcz_ctx_synchronize")

    # ---- compress -------------------------------------------------------------------------------
    def compress_sizes(self, b: ChainBatch) -> np.ndarray:
        cb = batch_as_c(b)
        off = np.zeros(b.n_chains + 1, np.uint64)
        _lib.check(self.lib.fcz_compress_sizes(ctypes.byref(cb), off.ctypes.data), "fcz_compress_sizes")
        return off

    def compress_batch(self, b: ChainBatch, strict: bool = True):
        """-> (blob uint8[...], off uint64[C+1], status int32[C])"""
        off = self.compress_sizes(b)
        cb = batch_as_c(b)
        out = np.zeros(int(off[-1]), np.uint8)
        st = np.zeros(b.n_chains, np.int32)
        rc = self.lib.fcz_compress_batch(self.ctx, ctypes.byref(cb), off.ctypes.data, out.ctypes.data, st.ctypes.data)
        # rc = the worst per-chain status, or a failure of the call itself (-1 is both: a refused chain, or bad arguments --
        # the latter leaves every per-chain status at 0)
        if rc != 0 and (strict or rc in (-2, -3, -8) or (rc == -1 and not (st == -1).any())):
            _lib.check(rc, "fcz_compress_batch")
        return out, off, st

    def compress_angles(self, b: ChainBatch) -> np.ndarray:
        """-> float32 [6, R]: phi, psi, omega, n_ca_c, ca_c_n, c_n_ca before quantisation"""
        cb = batch_as_c(b)
        out = np.zeros((6, b.n_residues), np.float32)
        _lib.check(self.lib.fcz_compress_angles(self.ctx, ctypes.byref(cb), out.ctypes.data), "fcz_compress_angles")
        return out

    # ---- decompress -----------------------------------------------------------------------------
    def decompress_sizes(self, blob: np.ndarray, off: np.ndarray):
        n = len(off) - 1
        blob = np.ascontiguousarray(blob, np.uint8)
        off = np.ascontiguousarray(off, np.uint64)
        info = (CEntryInfo * max(n, 1))()
        res_off = np.zeros(n + 1, np.uint32)
        atom_off = np.zeros(n + 1, np.uint32)
        _lib.check(self.lib.fcz_decompress_sizes(blob.ctypes.data, off.ctypes.data, n, ctypes.addressof(info),
                                                 res_off.ctypes.data, atom_off.ctypes.data), "fcz_decompress_sizes")
        return info, res_off, atom_off

    def decompress_batch(self, blob: np.ndarray, off: np.ndarray, alt_order: bool = False):
        blob = np.ascontiguousarray(blob, np.uint8)
        off = np.ascontiguousarray(off, np.uint64)
        n = len(off) - 1
        info, res_off, atom_off = self.decompress_sizes(blob, off)
        M, R = int(atom_off[-1]), int(res_off[-1])
        x = np.zeros(M, np.float32); y = np.zeros(M, np.float32); z = np.zeros(M, np.float32)
        bf = np.zeros(R, np.float32); rc = np.zeros(R, np.uint8); ac = np.zeros(M, np.uint8)
        out = CAtomsOut(x.ctypes.data, y.ctypes.data, z.ctypes.data, bf.ctypes.data, rc.ctypes.data, ac.ctypes.data)
        if M:
            _lib.check(self.lib.fcz_decompress_batch(self.ctx, blob.ctypes.data, off.ctypes.data, n, res_off.ctypes.data,
                                                     atom_off.ctypes.data, int(alt_order), ctypes.byref(out)),
                       "fcz_decompress_batch")
        return dict(x=x, y=y, z=z, bfac_res=bf, res_code=rc, atom_code=ac, res_off=res_off, atom_off=atom_off, info=info)

    def decompress_pdb(self, blob: np.ndarray, off: np.ndarray, alt_order: bool = False, nul_terminated: bool = False):
        """FCZ entries -> (list of PDB texts as bytes, per-entry status); decoding and text formatting both on the GPU.
        nul_terminated: every text that decodes ends in the NUL a database record carries (FCZ_PDB_NUL_TERMINATED)"""
        blob = np.ascontiguousarray(blob, np.uint8)
        off = np.ascontiguousarray(off, np.uint64)
        n = len(off) - 1
        text_off = np.zeros(n + 1, np.uint64)
        status = np.zeros(max(n, 1), np.int32)
        _lib.check(self.lib.fcz_decompress_pdb_begin(self.ctx, blob.ctypes.data, off.ctypes.data, n, int(bool(alt_order)) | (0x100 if nul_terminated else 0),
                                                     text_off.ctypes.data, status.ctypes.data), "fcz_decompress_pdb_begin")
        text = np.zeros(int(text_off[-1]), np.uint8)
        _lib.check(self.lib.fcz_decompress_pdb_fetch(self.ctx, text.ctypes.data if len(text) else None), "fcz_decompress_pdb_fetch")
        raw = text.tobytes()
        return [raw[int(text_off[i]):int(text_off[i + 1])] for i in range(n)], status[:n]

    def extract(self, blob: np.ndarray, off: np.ndarray, mode: int = 0, digits: int = 2):
        """FCZ entries -> list of data strings (bytes): pLDDT digits (mode 0) or the amino-acid sequence (mode 1); entries
        that cannot be read give b''"""
        blob = np.ascontiguousarray(blob, np.uint8)
        off = np.ascontiguousarray(off, np.uint64)
        n = len(off) - 1
        data_off = np.zeros(n + 1, np.uint64)
        _lib.check(self.lib.fcz_extract_sizes(blob.ctypes.data, off.ctypes.data, n, int(mode), int(digits), data_off.ctypes.data), "fcz_extract_sizes")
        data = np.zeros(int(data_off[-1]), np.uint8)
        _lib.check(self.lib.fcz_extract(self.ctx, blob.ctypes.data, off.ctypes.data, n, int(mode), int(digits), data_off.ctypes.data,
                                        data.ctypes.data if len(data) else None), "fcz_extract")
        raw = data.tobytes()
        return [raw[int(data_off[i]):int(data_off[i + 1])] for i in range(n)]

    # ---- structure ingest on the device -----------------------------------------------------------
    @staticmethod
    def _pack_files(texts, names):
        """bytes of the files back to back + offsets, base names + offsets, stem lengths (getFileParts: split at the last dot)"""
        file_off = np.zeros(len(texts) + 1, np.uint64)
        file_off[1:] = np.cumsum([len(t) for t in texts])
        text = np.frombuffer(b"".join(texts) or b"\0", np.uint8)
        nb = [n.encode() for n in names]
        name_off = np.zeros(len(nb) + 1, np.uint32)
        name_off[1:] = np.cumsum([len(n) for n in nb])
        name_blob = np.frombuffer(b"".join(nb) or b"\0", np.uint8)
        # byte counts of the ENCODED names (name_blob is UTF-8: a character index would cut a non-ASCII stem short)
        stem_len = np.asarray([n.rfind(b".") if b"." in n else len(n) for n in nb], np.uint32)
        return text, file_off, name_blob, name_off, stem_len

    def ingest_pdb(self, texts, names, anchor_threshold: int = 25, skip_discontinuous: bool = False):
        """PDB texts (bytes) + base names -> (ChainBatch, chain_file, chain_meta, file_status, refused[n, 2]) with every step on
        the GPU (fcz_ingest_pdb_*): what the reference's driver makes of the files before Foldcomp::compress"""
        text, file_off, name_blob, name_off, stem_len = self._pack_files(texts, names)
        counts = np.zeros(5, np.uint32)
        _lib.check(self.lib.fcz_ingest_pdb_begin(self.ctx, text.ctypes.data, file_off.ctypes.data, len(texts), name_blob.ctypes.data,
                                                 name_off.ctypes.data, stem_len.ctypes.data, int(anchor_threshold),
                                                 1 if skip_discontinuous else 0, counts.ctypes.data), "fcz_ingest_pdb_begin")
        C, R, M, TB, NR = (int(v) for v in counts)
        b = ChainBatch(res_off=np.zeros(C + 1, np.uint32), atom_off=np.zeros(R + 1, np.uint32), x=np.zeros(M, np.float32),
                       y=np.zeros(M, np.float32), z=np.zeros(M, np.float32), atom_code=np.zeros(M, np.uint8),
                       res_code=np.zeros(R, np.uint8), bfac_ca=np.zeros(R, np.float32), first_res_index=np.zeros(C, np.int32),
                       first_atom_index=np.zeros(C, np.int32), chain_id=np.zeros(C, np.uint8), titles=np.zeros(max(TB, 1), np.uint8),
                       title_off=np.zeros(C + 1, np.uint32), anchor_threshold=int(anchor_threshold))
        cb = batch_as_c(b)
        chain_file = np.zeros(C, np.uint32); chain_meta = np.zeros(C, np.uint32)
        file_status = np.zeros(len(texts), np.int32); refused = np.zeros((NR, 2), np.uint32)
        _lib.check(self.lib.fcz_ingest_pdb_fetch(self.ctx, ctypes.byref(cb), chain_file.ctypes.data, chain_meta.ctypes.data,
                                                 file_status.ctypes.data, refused.ctypes.data), "fcz_ingest_pdb_fetch")
        b.titles = b.titles[:TB]
        return b, chain_file, chain_meta, file_status, refused

    def compress_pdb(self, texts, names, anchor_threshold: int = 25, skip_discontinuous: bool = False):
        """PDB texts -> FCZ records, parse and codec both on the GPU: dict(blob, off, status, chain_file, chain_meta, file_status,
        refused)"""
        text, file_off, name_blob, name_off, stem_len = self._pack_files(texts, names)
        counts = np.zeros(5, np.uint32); nbytes = ctypes.c_uint64(0)
        _lib.check(self.lib.fcz_compress_pdb_begin(self.ctx, text.ctypes.data, file_off.ctypes.data, len(texts), name_blob.ctypes.data,
                                                   name_off.ctypes.data, stem_len.ctypes.data, int(anchor_threshold),
                                                   1 if skip_discontinuous else 0, counts.ctypes.data, ctypes.byref(nbytes)),
                   "fcz_compress_pdb_begin")
        C, NR = int(counts[0]), int(counts[4])
        off = np.zeros(C + 1, np.uint64); st = np.zeros(C, np.int32); blob = np.zeros(max(int(nbytes.value), 1), np.uint8)
        chain_file = np.zeros(C, np.uint32); chain_meta = np.zeros(C, np.uint32)
        file_status = np.zeros(len(texts), np.int32); refused = np.zeros((NR, 2), np.uint32)
        _lib.check(self.lib.fcz_compress_pdb_fetch(self.ctx, off.ctypes.data, st.ctypes.data, chain_file.ctypes.data, chain_meta.ctypes.data,
                                                   file_status.ctypes.data, refused.ctypes.data, blob.ctypes.data), "fcz_compress_pdb_fetch")
        return dict(blob=blob[:int(nbytes.value)], off=off, status=st, chain_file=chain_file, chain_meta=chain_meta,
                    file_status=file_status, refused=refused, counts=counts)

    def chain_names(self, n_chains: int):
        """names of the chains of the batch the last ingest / compress_pdb / compress_gz call left in the ctx: list of str (mmCIF chain
        names have up to four characters; chain_meta's low byte is only the first)"""
        nm = np.zeros(max(int(n_chains), 1), np.uint32)
        _lib.check(self.lib.fcz_ingest_chain_names_fetch(self.ctx, nm.ctypes.data), "fcz_ingest_chain_names_fetch")
        return [int(v).to_bytes(4, "little").rstrip(b"\0").decode("latin-1") for v in nm[:int(n_chains)]]

    # ---- gzip members on the device ----------------------------------------------------------------
    INFLATE_STATUS = {0: "ok", 1: "header", 2: "block", 3: "code", 4: "size", 5: "input", 6: "check"}

    def inflate(self, members, kind=None):
        """gzip members (bytes each) -> (list of texts, status[n]): fcz_inflate_sizes + fcz_inflate. A member with a non-zero status
        was NOT inflated on the device (the caller's zlib decides about it); its text is returned as blanks of the ISIZE it claims."""
        n = len(members)
        off = np.zeros(n + 1, np.uint64)
        off[1:] = np.cumsum([len(m) for m in members])
        raw = np.frombuffer(b"".join(members) or b"\0", np.uint8)
        kd = None if kind is None else np.ascontiguousarray(kind, np.uint8)
        toff = np.zeros(n + 1, np.uint64)
        _lib.check(self.lib.fcz_inflate_sizes(raw.ctypes.data, off.ctypes.data, n, None if kd is None else kd.ctypes.data, toff.ctypes.data),
                   "fcz_inflate_sizes")
        text = np.zeros(max(int(toff[n]), 1), np.uint8); st = np.zeros(n, np.int32)
        _lib.check(self.lib.fcz_inflate(self.ctx, raw.ctypes.data, off.ctypes.data, n, None if kd is None else kd.ctypes.data, toff.ctypes.data,
                                        text.ctypes.data, st.ctypes.data), "fcz_inflate")
        tb = text.tobytes()
        return [tb[int(toff[i]):int(toff[i + 1])] for i in range(n)], st

    def compress_gz(self, files, names, is_gz=None, anchor_threshold: int = 25, skip_discontinuous: bool = False):
        """Structure files as they lie on disk (gzip members where is_gz, by default where the name ends in .gz) -> FCZ records:
        inflate, parse and codec on the GPU. Same dict as compress_pdb; file_status 5 = the member is left to the caller's zlib."""
        data, file_off, name_blob, name_off, stem_len = self._pack_files(files, names)
        gz = np.asarray([n.endswith(".gz") for n in names] if is_gz is None else is_gz, np.uint8)
        counts = np.zeros(5, np.uint32); nbytes = ctypes.c_uint64(0)
        _lib.check(self.lib.fcz_compress_gz_begin(self.ctx, data.ctypes.data, file_off.ctypes.data, len(files), gz.ctypes.data, name_blob.ctypes.data,
                                                  name_off.ctypes.data, stem_len.ctypes.data, int(anchor_threshold),
                                                  1 if skip_discontinuous else 0, counts.ctypes.data, ctypes.byref(nbytes)),
                   "fcz_compress_gz_begin")
        C, NR = int(counts[0]), int(counts[4])
        off = np.zeros(C + 1, np.uint64); st = np.zeros(C, np.int32); blob = np.zeros(max(int(nbytes.value), 1), np.uint8)
        chain_file = np.zeros(C, np.uint32); chain_meta = np.zeros(C, np.uint32)
        file_status = np.zeros(len(files), np.int32); refused = np.zeros((NR, 2), np.uint32)
        _lib.check(self.lib.fcz_compress_pdb_fetch(self.ctx, off.ctypes.data, st.ctypes.data, chain_file.ctypes.data, chain_meta.ctypes.data,
                                                   file_status.ctypes.data, refused.ctypes.data, blob.ctypes.data), "fcz_compress_pdb_fetch")
        return dict(blob=blob[:int(nbytes.value)], off=off, status=st, chain_file=chain_file, chain_meta=chain_meta,
                    file_status=file_status, refused=refused, counts=counts)

    # ---- timing ---------------------------------------------------------------------------------
    def enable_timing(self, on: bool = True):
        self.lib.fcz_ctx_enable_timing(self.ctx, int(on))

    def reset_timing(self):
        self.lib.fcz_ctx_reset_timing(self.ctx)

    def kernel_time(self, name: str):
        ms = ctypes.c_double(0); n = ctypes.c_uint64(0)
        self.lib.fcz_ctx_kernel_time(self.ctx, name.encode(), ctypes.byref(ms), ctypes.byref(n))
        return ms.value, n.value

    # ---- numerics self-test hook ------------------------------------------------------------------
    def selftest_math(self, mode: int, start_bits: int, stride: int, count: int) -> np.ndarray:
        out = np.zeros(count, np.float32)
        _lib.check(self.lib.fcz_selftest_math(self.ctx, mode, start_bits, stride, count, out.ctypes.data), "fcz_selftest_math")
        return out
