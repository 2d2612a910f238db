"""CPU: the C-ABI shared library loads, exports every function include/fcz_hip.h declares, its pure-host
entry points (sizes, header parse, check, code tables) work, and compute entry points fail loudly without
a GPU instead of falling back to anything."""
import ctypes
import os
import re

import numpy as np
import pytest

import _harness as H
from _cases import compress_cases, db_cases, entries_blob, golden_batch
from foldcomp_amd import _lib
from foldcomp_amd.structure import CEntryInfo, batch_as_c

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "fcz_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fcz_[a-z0-9_]+)\s*\(", txt)))


def test_library_built_and_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "libfcz_hip.so missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/fcz_hip.h but not exported"
    assert set(_lib.EXPORTS) <= set(names)


def test_host_library_exports_what_its_header_declares():
    """include/fcz_host.h (the C++ host's structure readers as a library) against host/libfcz_host.so"""
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "fcz_host.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(fcz_host_[a-z0-9_]+)\s*\(", txt)))
    assert names == ["fcz_host_free", "fcz_host_read_structure"]
    path = os.path.join(ROOT, "host", "libfcz_host.so")
    assert os.path.exists(path), "host/libfcz_host.so missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(path)
    for n in names:
        assert hasattr(lib, n), n


def test_code_tables():
    lib = _lib.load()
    assert lib.fcz_atom_code_name(0) == b"N" and lib.fcz_atom_code_name(36) == b"OXT"
    assert lib.fcz_atom_code_from_name(b"CA") == 1 and lib.fcz_atom_code_from_name(b"H1") == 255
    assert lib.fcz_res_code_from_name(b"TRP") == 17 and lib.fcz_res_code_from_name(b"MSE") == -1
    assert lib.fcz_res_code_from_name(b"UNK") == 23 and lib.fcz_res_code_from_name(b"ASX") == -1
    assert lib.fcz_res_code_natoms(17) == 14 and lib.fcz_res_code_natoms(7) == 4 and lib.fcz_res_code_natoms(23) == 3
    # ALA canonical N CA C O CB, `-a` order N CA C CB O (reference src/amino_acid.h:71-74)
    assert [lib.fcz_res_code_atom(0, j, 0) for j in range(5)] == [0, 1, 2, 3, 4]
    assert [lib.fcz_res_code_atom(0, j, 1) for j in range(5)] == [0, 1, 2, 4, 3]


def test_host_sizes_match_reference_record_sizes(golden):
    z, index = golden
    lib = _lib.load()
    for name in compress_cases(index):
        b = golden_batch(z, name)
        cb = batch_as_c(b)
        off = np.zeros(2, np.uint64)
        assert lib.fcz_compress_sizes(ctypes.byref(cb), off.ctypes.data) == 0
        assert int(off[1]) == len(z[f"{name}/fcz"]), name


def test_host_entry_parse_and_check(golden):
    z, index = golden
    lib = _lib.load()
    names = compress_cases(index) + db_cases(index)
    blob, off = entries_blob([z[f"{n}/fcz"].tobytes() for n in names])
    n = len(names)
    info = (CEntryInfo * n)()
    ro = np.zeros(n + 1, np.uint32); ao = np.zeros(n + 1, np.uint32)
    assert lib.fcz_decompress_sizes(blob.ctypes.data, off.ctypes.data, n, ctypes.addressof(info), ro.ctypes.data, ao.ctypes.data) == 0
    for i, nm in enumerate(names):
        assert info[i].status == 0, nm
        assert info[i].n_atoms_out == len(z[f"{nm}/xyz0"]), nm
        e = z[f"{nm}/fcz"].tobytes()
        assert lib.fcz_check(e, len(e)) == H.load_oracle().fcz_oracle_check(e, len(e))
    # same verdicts as the oracle's parser on damaged entries
    e = z["pdb:test_af/fcz"].tobytes()
    for bad, want in ((b"XXXX" + e[4:], -4), (e[:100], -5), (e[:40], -5)):
        bb, oo = entries_blob([bad])
        inf = (CEntryInfo * 1)()
        lib.fcz_decompress_sizes(bb.ctypes.data, oo.ctypes.data, 1, ctypes.addressof(inf), ro.ctypes.data, ao.ctypes.data)
        assert inf[0].status == want
        assert ro[1] == 0 and ao[1] == 0


def test_no_cpu_fallback_without_gpu():
    """on a box without a HIP device ctx creation must fail; nothing computes on the CPU"""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("GPU present")
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.fcz_ctx_create(0, ctypes.byref(h)) == -2
    from foldcomp_amd.codec import Codec
    with pytest.raises(_lib.FczLibraryError):
        Codec(0)


def test_engine_and_status_strings_fail_loudly_without_gpu(tmp_path):
    """the C++ engine in every mode a sharded run starts it in -- also the placed decompress (sizes pass first) -- refuses to run
    without a HIP device instead of computing anything on the CPU, and prints no counts a driver could mistake for a sizes pass;
    the status added in round 5 has its text"""
    import subprocess
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    lib = _lib.load()
    lib.fcz_status_string.restype = ctypes.c_char_p
    assert b"finite" in lib.fcz_status_string(-9) and _lib.STATUS[-9] == "FCZ_E_NONFINITE"
    if has_gpu:
        pytest.skip("GPU present")
    host = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "host", "foldcomp-hip")
    if not os.path.exists(host):
        pytest.skip("host/foldcomp-hip not built")
    for ext in ("", ".index", ".lookup"):
        (tmp_path / ("db" + ext)).write_bytes(b"")
    (tmp_path / "db.dbtype").write_bytes((12).to_bytes(4, "little"))
    for extra in ([], ["--place"]):
        r = subprocess.run([host, "decompress", "-d", "-y", "--shard", "0/2", *extra, str(tmp_path / "db"), str(tmp_path / "out")],
                           input="0 0 0\n", capture_output=True, text=True, timeout=60)
        assert r.returncode != 0 and "no HIP device" in r.stderr and '"phase"' not in r.stdout, (extra, r.stdout, r.stderr)
        assert not os.path.exists(tmp_path / "out")


def test_host_extract_sizes_match_reference_strings(golden):
    """fcz_extract_sizes is pure host code: the data sizes must equal the lengths of the reference's extract strings"""
    z, index = golden
    lib = _lib.load()
    names = [n for n in index if f"{n}/plddt2" in z.files and f"{n}/fcz" in z.files]
    entries = [z[f"{n}/fcz"].tobytes() for n in names]
    off = np.zeros(len(entries) + 1, np.uint64)
    off[1:] = np.cumsum([len(e) for e in entries])
    blob = np.frombuffer(b"".join(entries), np.uint8)
    for mode, digits, key in [(0, 1, "plddt1"), (0, 2, "plddt2"), (0, 3, "plddt3"), (0, 4, "plddt4"), (1, 0, "fasta")]:
        data_off = np.zeros(len(entries) + 1, np.uint64)
        assert lib.fcz_extract_sizes(blob.ctypes.data, off.ctypes.data, len(entries), mode, digits, data_off.ctypes.data) == 0
        got = np.diff(data_off).astype(np.int64)
        exp = np.asarray([len(z[f"{n}/{key}"].tobytes()) for n in names], np.int64)
        assert np.array_equal(got, exp), (key, got[:5], exp[:5])
    # unreadable entries: zero bytes; bad arguments: status, not a crash
    bad = [b"NOPE" + entries[0][4:], entries[0][:80]]
    off2 = np.asarray([0, len(bad[0]), len(bad[0]) + len(bad[1])], np.uint64)
    d2 = np.zeros(3, np.uint64)
    assert lib.fcz_extract_sizes(np.frombuffer(b"".join(bad), np.uint8).ctypes.data, off2.ctypes.data, 2, 0, 2, d2.ctypes.data) == 0
    assert d2[2] == 0
    assert lib.fcz_extract_sizes(blob.ctypes.data, off.ctypes.data, 1, 0, 7, d2.ctypes.data) == -1   # FCZ_E_INVALID_ARG


def test_inflate_sizes_on_the_host():
    """fcz_inflate_sizes (pure host work): text offsets from the members' ISIZE as the reference's reader sizes its buffer
    (gemmi estimate_uncompressed_size, lib/gemmi/gz.hpp:25-45); plain entries by their length; what cannot be a member's text
    size (member shorter than a header + trailer, an ISIZE beyond DEFLATE's 1032 : 1) is sized 0 -- and left to zlib later"""
    import gzip, struct, zlib
    lib = _lib.load()
    texts = [b"ATOM      1  N   MET A   1\n" * k for k in (1, 40, 3000)]
    members = [gzip.compress(t, 6) for t in texts]
    lying = members[1][:-4] + struct.pack("<I", 1 << 30)            # a trailer that claims a gigabyte for a hundred bytes
    entries = [members[0], texts[1], members[2], b"\x1f\x8b\x08", lying, members[1]]
    kind = np.array([1, 0, 1, 1, 1, 1], np.uint8)
    off = np.zeros(len(entries) + 1, np.uint64); off[1:] = np.cumsum([len(e) for e in entries])
    raw = np.frombuffer(b"".join(entries), np.uint8)
    toff = np.zeros(len(entries) + 1, np.uint64)
    assert lib.fcz_inflate_sizes(raw.ctypes.data, off.ctypes.data, len(entries), kind.ctypes.data, toff.ctypes.data) == 0
    assert list(np.diff(toff.astype(np.int64))) == [len(texts[0]), len(texts[1]), len(texts[2]), 0, 0, len(texts[1])]
    # every entry a member when no kinds are given
    toff2 = np.zeros(3, np.uint64); off2 = np.array([0, len(members[0]), len(members[0]) + len(members[2])], np.uint64)
    raw2 = np.frombuffer(members[0] + members[2], np.uint8)
    assert lib.fcz_inflate_sizes(raw2.ctypes.data, off2.ctypes.data, 2, None, toff2.ctypes.data) == 0
    assert list(toff2) == [0, len(texts[0]), len(texts[0]) + len(texts[2])]
    assert zlib.decompress(members[2], 31) == texts[2]
    # the compute entry points refuse to run without a context (no CPU fallback behind them)
    st = np.zeros(2, np.int32)
    assert lib.fcz_inflate(None, raw2.ctypes.data, off2.ctypes.data, 2, None, toff2.ctypes.data, raw2.ctypes.data, st.ctypes.data) == -1
