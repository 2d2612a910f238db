"""Foldcomp / MMseqs2-style database container (reference src/database_reader.cpp, src/database_writer.cpp):
`<db>` concatenated entries, `<db>.index` lines `key\\toffset\\tlength`, `<db>.lookup` lines `key\\tname\\t0`,
`<db>.dbtype` = int32 12. Host I/O only."""
from __future__ import annotations

import mmap
import os
import struct
from typing import List, Optional

import numpy as np


def _strtoul(w: bytes) -> int:
    """strtoul / strtoull(word, NULL, 10): optional sign, then the digits the word starts with (0 when there are none)"""
    i, neg = 0, False
    if w[:1] in (b"+", b"-"):
        neg = w[:1] == b"-"; i = 1
    j = i
    while j < len(w) and 0x30 <= w[j] <= 0x39:
        j += 1
    v = int(w[i:j]) if j > i else 0
    return (-v) & 0xffffffffffffffff if neg else v


class DatabaseReader:
    def __init__(self, path: str, use_lookup: bool = True):
        self.path = path
        self._f = open(path, "rb")
        size = os.path.getsize(path)
        self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ) if size else b""
        keys, offs, lens = [], [], []
        # read_index (src/database_reader.cpp:283-311): an entry per '\n' of the file (count_lines: a last line without one is not
        # an entry), words separated by blanks and tabs (getWordsOfLine), numbers by strtoul / strtoull (the digits a word starts
        # with); a line of more than three words fails the read. A line of fewer than three is undefined there: refused here
        with open(path + ".index", "rb") as fh:
            raw = fh.read()
        for line in raw.split(b"\n")[:-1]:
            p = [w for w in line.replace(b"\t", b" ").split(b" ") if w]
            if len(p) > 3:
                raise ValueError(f"{path}.index: a line of more than three columns")
            if len(p) < 3:
                raise ValueError(f"{path}.index: a line of fewer than three columns")
            keys.append(_strtoul(p[0]) & 0xffffffff); offs.append(_strtoul(p[1])); lens.append(_strtoul(p[2]))
        order = np.argsort(np.asarray(keys, np.int64), kind="stable")     # the reader re-sorts by key
        self.keys = np.asarray(keys, np.int64)[order]
        self.offsets = np.asarray(offs, np.int64)[order]
        self.lengths = np.asarray(lens, np.int64)[order]
        self.name_to_key = {}
        self.key_to_name = {}
        if use_lookup and os.path.exists(path + ".lookup"):
            # read_lookup (:345-361): key = the first word, name = the second (words end at a blank or a tab). Deviations, on purpose:
            # the reference fails the WHOLE lookup on a line of fewer than three words and takes the name as the bytes up to the third
            # word minus one (so a name followed by several blanks keeps all but the last); here a two-word line is accepted and the
            # name ends at its first blank -- equal on every file free_writer writes ("%d\t%s\t0\n") and on the mutated ones
            # tests/test_ingest_vs_reference.py puts to the live reference
            with open(path + ".lookup", "rb") as fh:
                for line in fh.read().split(b"\n"):
                    p = [w for w in line.replace(b"\t", b" ").split(b" ") if w]
                    if len(p) >= 2:
                        nm = p[1].decode("latin-1"); k = _strtoul(p[0]) & 0xffffffff
                        # (a key or a name that comes twice: the later line wins -- the reference's stable_sort with its "<="
                        # comparators, :313-321, leaves equal elements in reverse order, and its look-ups take the first)
                        self.name_to_key[nm] = k
                        self.key_to_name[k] = nm

    def __len__(self):
        return len(self.keys)

    def id_of_key(self, key: int) -> int:
        i = int(np.searchsorted(self.keys, key))
        return i if i < len(self.keys) and self.keys[i] == key else -1

    def id_of_name(self, name: str) -> int:
        k = self.name_to_key.get(name)
        return -1 if k is None else self.id_of_key(k)

    def name(self, i: int) -> str:
        return self.key_to_name.get(int(self.keys[i]), str(int(self.keys[i])))

    def data(self, i: int, strip_nul: bool = False) -> bytes:
        o, l = int(self.offsets[i]), int(self.lengths[i])
        if strip_nul:                       # foldcomp.cxx:66,73: max(length,1)-1
            l = max(l, 1) - 1
        return bytes(self._mm[o:o + l])

    def close(self):
        if self._mm:
            self._mm.close()
        self._mm = b""
        self._f.close()


class DatabaseWriter:
    """make_writer / writer_append / free_writer (src/database_writer.cpp:36-96)"""

    def __init__(self, path: str):
        self.path = path
        self._data = open(path, "wb")
        self._entries = []
        with open(path + ".dbtype", "wb") as f:
            f.write(struct.pack("<i", 12))

    def append(self, data: bytes, key: int, name: str):
        off = self._data.tell()
        self._data.write(data)
        self._entries.append((int(key), off, len(data), name))

    def append_blob(self, blob: bytes, lengths, keys, names, base_offset: Optional[int] = None):
        """bulk form used by the sharded writer: one write, many index rows"""
        off = self._data.tell() if base_offset is None else base_offset
        self._data.write(blob)
        for l, k, n in zip(lengths, keys, names):
            self._entries.append((int(k), off, int(l), n)); off += int(l)

    def close(self):
        ent = sorted(self._entries, key=lambda e: e[0])     # stable sort by key
        with open(self.path + ".index", "w") as fi, open(self.path + ".lookup", "w") as fl:
            for k, o, l, n in ent:
                fi.write("%d\t%d\t%d\n" % (k, o, l))
                fl.write("%d\t%s\t0\n" % (k, n))
        self._data.close()
