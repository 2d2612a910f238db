"""ctypes binding of libfcz_hip.so (the C-ABI of include/fcz_hip.h).

The library is built in-tree (foldcomp_amd/libfcz_hip.so) by `__graft_entry__.build()` or
`make -C foldcomp_amd/csrc`. There is no Python/NumPy/CPU implementation of the codec in this package:
if the shared library is missing, or no HIP device is present, the codec entry points raise.
"""
from __future__ import annotations

import ctypes
import os

from .structure import CAtomsOut, CChainBatch, CEntryInfo, CIngestResult

_HERE = os.path.dirname(os.path.abspath(__file__))
# FCZ_HIP_LIB selects another build of the same library (A/B timing of kernel variants); it is still a HIP build
LIB_PATH = os.environ.get("FCZ_HIP_LIB") or os.path.join(_HERE, "libfcz_hip.so")

FCZ_OK = 0
STATUS = {0: "FCZ_OK", -1: "FCZ_E_INVALID_ARG", -2: "FCZ_E_NO_DEVICE", -3: "FCZ_E_HIP", -4: "FCZ_E_BAD_MAGIC",
          -5: "FCZ_E_TRUNCATED", -6: "FCZ_E_RESIDUE", -7: "FCZ_E_TOO_SHORT", -8: "FCZ_E_NOMEM", -9: "FCZ_E_NONFINITE"}

_lib = None


class FczLibraryError(RuntimeError):
    pass


def load():
    """Load libfcz_hip.so (once). Raises FczLibraryError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FczLibraryError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C foldcomp_amd/csrc). foldcomp_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    vp, u32, i32, u64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_uint64
    PB, PO = ctypes.POINTER(CChainBatch), ctypes.POINTER(CAtomsOut)
    sig = {
        "fcz_ctx_create": (i32, [i32, ctypes.POINTER(vp)]),
        "fcz_ctx_destroy": (None, [vp]),
        "fcz_ctx_stream": (vp, [vp]),
        "fcz_ctx_synchronize": (i32, [vp]),
        "fcz_device_count": (i32, []),
        "fcz_pinned_alloc": (vp, [ctypes.c_size_t]),
        "fcz_pinned_free": (None, [vp]),
        "fcz_ctx_set_numerics": (i32, [vp, i32]),
        "fcz_ctx_get_numerics": (i32, [vp]),
        "fcz_status_string": (ctypes.c_char_p, [i32]),
        "fcz_atom_code_name": (ctypes.c_char_p, [i32]),
        "fcz_atom_code_from_name": (i32, [ctypes.c_char_p]),
        "fcz_res_code_from_name": (i32, [ctypes.c_char_p]),
        "fcz_res_code_name": (ctypes.c_char_p, [i32]),
        "fcz_res_code_natoms": (i32, [i32]),
        "fcz_res_code_atom": (i32, [i32, i32, i32]),
        "fcz_compress_sizes": (i32, [PB, vp]),
        "fcz_compress_batch": (i32, [vp, PB, vp, vp, vp]),
        "fcz_compress_angles": (i32, [vp, PB, vp]),
        "fcz_compress_sizes_dev": (i32, [vp, PB, vp]),
        "fcz_compress_batch_dev": (i32, [vp, PB, vp, vp, vp]),
        "fcz_decompress_sizes": (i32, [vp, vp, u32, vp, vp, vp]),
        "fcz_decompress_batch": (i32, [vp, vp, vp, u32, vp, vp, i32, PO]),
        "fcz_decompress_sizes_dev": (i32, [vp, vp, vp, u32, vp, vp, ctypes.POINTER(u32), ctypes.POINTER(u32)]),
        "fcz_decompress_batch_dev": (i32, [vp, vp, vp, u32, vp, vp, i32, PO]),
        "fcz_pdb_sizes_dev": (i32, [vp, vp, vp, u32, vp, vp, PO, vp]),
        "fcz_pdb_format_dev": (i32, [vp, vp, vp, u32, vp, vp, PO, i32, vp, vp]),
        "fcz_decompress_pdb_begin": (i32, [vp, vp, vp, u32, i32, vp, vp]),
        "fcz_decompress_pdb_fetch": (i32, [vp, vp]),
        "fcz_decompress_pdb_sizes": (i32, [vp, vp, vp, u32, i32, vp, vp]),
        "fcz_extract_sizes": (i32, [vp, vp, u32, i32, i32, vp]),
        "fcz_extract": (i32, [vp, vp, vp, u32, i32, i32, vp, vp]),
        "fcz_extract_sizes_dev": (i32, [vp, vp, vp, u32, i32, i32, vp]),
        "fcz_extract_dev": (i32, [vp, vp, vp, u32, i32, i32, vp, vp]),
        "fcz_ingest_pdb_dev": (i32, [vp, vp, vp, u32, u64, vp, vp, vp, i32, i32, vp]),
        "fcz_ingest_pdb_begin": (i32, [vp, vp, vp, u32, vp, vp, vp, i32, i32, vp]),
        "fcz_ingest_pdb_fetch": (i32, [vp, PB, vp, vp, vp, vp]),
        "fcz_ingest_chain_names_fetch": (i32, [vp, vp]),
        "fcz_compress_pdb_begin": (i32, [vp, vp, vp, u32, vp, vp, vp, i32, i32, vp, ctypes.POINTER(u64)]),
        "fcz_compress_pdb_fetch": (i32, [vp, vp, vp, vp, vp, vp, vp, vp]),
        "fcz_inflate_sizes": (i32, [vp, vp, u32, vp, vp]),
        "fcz_inflate_dev": (i32, [vp, vp, vp, u32, vp, vp, vp, vp]),
        "fcz_inflate": (i32, [vp, vp, vp, u32, vp, vp, vp, vp]),
        "fcz_ingest_gz_begin": (i32, [vp, vp, vp, u32, vp, vp, vp, vp, i32, i32, vp]),
        "fcz_compress_gz_begin": (i32, [vp, vp, vp, u32, vp, vp, vp, vp, i32, i32, vp, ctypes.POINTER(u64)]),
        "fcz_check": (i32, [vp, u64]),
        "fcz_ctx_enable_timing": (i32, [vp, i32]),
        "fcz_ctx_kernel_time": (i32, [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(u64)]),
        "fcz_ctx_reset_timing": (None, [vp]),
        "fcz_selftest_math": (i32, [vp, i32, u32, u32, u32, vp]),
        "fcz_selftest_copy": (i32, [vp, u64, i32, ctypes.POINTER(ctypes.c_double)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


EXPORTS = ["fcz_ctx_create", "fcz_ctx_destroy", "fcz_ctx_stream", "fcz_ctx_synchronize", "fcz_status_string",
           "fcz_atom_code_name", "fcz_atom_code_from_name", "fcz_res_code_from_name", "fcz_res_code_name",
           "fcz_res_code_natoms", "fcz_res_code_atom", "fcz_compress_sizes", "fcz_compress_batch",
           "fcz_compress_angles", "fcz_compress_sizes_dev", "fcz_compress_batch_dev", "fcz_decompress_sizes", "fcz_decompress_batch",
           "fcz_decompress_sizes_dev", "fcz_decompress_batch_dev", "fcz_pdb_sizes_dev", "fcz_pdb_format_dev",
           "fcz_decompress_pdb_begin", "fcz_decompress_pdb_fetch", "fcz_decompress_pdb_sizes", "fcz_extract_sizes", "fcz_extract",
           "fcz_extract_sizes_dev", "fcz_extract_dev", "fcz_ingest_pdb_dev", "fcz_ingest_pdb_begin", "fcz_ingest_pdb_fetch", "fcz_ingest_chain_names_fetch",
           "fcz_compress_pdb_begin", "fcz_compress_pdb_fetch", "fcz_inflate_sizes", "fcz_inflate_dev", "fcz_inflate",
           "fcz_ingest_gz_begin", "fcz_compress_gz_begin", "fcz_check", "fcz_ctx_enable_timing",
           "fcz_ctx_kernel_time", "fcz_ctx_reset_timing", "fcz_selftest_math", "fcz_selftest_copy"]


def status_name(code: int) -> str:
    return STATUS.get(int(code), f"status {code}")


def check(code: int, what: str):
    if code != FCZ_OK:
        msg = load().fcz_status_string(int(code)).decode()
        raise FczLibraryError(f"{what}: {status_name(code)} ({msg})")
