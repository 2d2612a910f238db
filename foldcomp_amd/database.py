"""Foldcomp / MMseqs2-style database container (reference src/database_reader.cpp, src/database_writer.cpp):
`<db>` concatenated entries, `<db>.index` lines `key\\toffset\\tlength`, `<db>.lookup` lines `key\\tname\\t0`,
`<db>.dbtype` = int32 12. Host I/O only."""
from __future__ import annotations

import mmap
import os
import struct
from typing import List, Optional

import numpy as np


class DatabaseReader:
    def __init__(self, path: str, use_lookup: bool = True):
        self.path = path
        self._f = open(path, "rb")
        size = os.path.getsize(path)
        self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ) if size else b""
        keys, offs, lens = [], [], []
        with open(path + ".index") as fh:
            for line in fh:
                p = line.split()
                if len(p) < 3:
                    continue
                keys.append(int(p[0])); offs.append(int(p[1])); lens.append(int(p[2]))
        order = np.argsort(np.asarray(keys, np.int64), kind="stable")     # the reader re-sorts by key
        self.keys = np.asarray(keys, np.int64)[order]
        self.offsets = np.asarray(offs, np.int64)[order]
        self.lengths = np.asarray(lens, np.int64)[order]
        self.name_to_key = {}
        self.key_to_name = {}
        if use_lookup and os.path.exists(path + ".lookup"):
            with open(path + ".lookup") as fh:
                for line in fh:
                    p = line.rstrip("\n").split("\t")
                    if len(p) >= 2:
                        self.name_to_key.setdefault(p[1], int(p[0]))
                        self.key_to_name[int(p[0])] = p[1]

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
