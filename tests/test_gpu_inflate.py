"""Inflate on the device (fcz_inflate*, fcz_compress_gz_*, foldcomp_amd/csrc/fcz_inflate.h) -- SURVEY.md section 8 row f3, the
`.pdb.gz` / `.cif.gz` inputs.

The reference inflates with zlib (gemmi::MaybeGzipped lib/gemmi/gz.hpp:105-133, uncompressBuffer src/structure_reader.cpp:156-203), so
the oracle of this stage is zlib itself (Python's zlib module = the same library):
  * every member the device reports as inflated holds exactly zlib's bytes;
  * every stream zlib rejects (or does not finish) is refused by the device -- mutated streams: truncations, bit flips anywhere, bad
    CRC / ISIZE, distances before the start, broken code-length sets;
  * what the device hands back although zlib reads it is only what fcz_hip.h lists (header CRC, long headers, several members /
    trailing bytes, an ISIZE that is not the text size);
  * files -> FCZ records through the inflate stage == the same texts through fcz_compress_pdb_*, byte for byte; a refused member
    comes back as FCZ_INGEST_HOST_GZIP with nothing of it in the batch."""
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DBG = {0: "ok", 1: "header", 2: "block", 3: "code", 4: "size", 5: "input", 6: "check"}


def _why(st):
    st = int(st)
    return f"{DBG.get(st & 0xff, st & 0xff)}@{st >> 8}" if st >> 8 else DBG.get(st, str(st))


@pytest.fixture(scope="module")
def fixtures():
    z = np.load(os.path.join(ROOT, "tests", "golden", "reference_ingest.npz"))
    out = {k[5:]: z[k].tobytes() for k in z.files if k.startswith("file:")}
    out["test.cif"] = gzip.decompress(out["test.cif.gz"])
    return out


def gz_member(data: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=31, memlevel=8, name=None, comment=None, extra=None, hcrc=False):
    """a gzip member made by zlib, with an optional header of our own (FNAME / FCOMMENT / FEXTRA / FHCRC)"""
    if name is None and comment is None and extra is None and not hcrc:
        c = zlib.compressobj(level, zlib.DEFLATED, wbits, memlevel, strategy)
        return c.compress(data) + c.flush()
    c = zlib.compressobj(level, zlib.DEFLATED, -15, memlevel, strategy)
    body = c.compress(data) + c.flush()
    flg = (4 if extra is not None else 0) | (8 if name is not None else 0) | (16 if comment is not None else 0) | (2 if hcrc else 0)
    h = bytes([0x1f, 0x8b, 8, flg, 0, 0, 0, 0, 0, 3])
    if extra is not None:
        h += struct.pack("<H", len(extra)) + extra
    if name is not None:
        h += name + b"\0"
    if comment is not None:
        h += comment + b"\0"
    if hcrc:
        h += struct.pack("<H", zlib.crc32(h) & 0xffff)
    return h + body + struct.pack("<II", zlib.crc32(data), len(data) & 0xffffffff)


def zlib_says(member: bytes):
    """what the hosts' gunzip (inflateInit2(16 + MAX_WBITS), inflate() until Z_STREAM_END) makes of the bytes: the text, or None"""
    d = zlib.decompressobj(31)
    try:
        out = d.decompress(member)
    except zlib.error:
        return None
    return out if d.eof else None


def _corpus(fixtures):
    rng = np.random.default_rng(20261001)
    texts = [fixtures[k] for k in ("test.pdb", "test_af.pdb", "multichain.pdb", "test.cif")]
    cases = [("test.cif.gz as shipped", fixtures["test.cif.gz"], fixtures["test.cif"])]
    for nm, t in zip(("test.pdb", "test_af.pdb", "multichain.pdb", "test.cif"), texts):
        for lvl in (0, 1, 6, 9):
            cases.append((f"{nm} level {lvl}", gz_member(t, lvl), t))
        cases.append((f"{nm} fixed", gz_member(t, 6, zlib.Z_FIXED), t))
        cases.append((f"{nm} huffman only", gz_member(t, 6, zlib.Z_HUFFMAN_ONLY), t))
        cases.append((f"{nm} rle", gz_member(t, 6, zlib.Z_RLE), t))
        cases.append((f"{nm} filtered memlevel 1", gz_member(t, 4, zlib.Z_FILTERED, memlevel=1), t))
        cases.append((f"{nm} window 9", gz_member(t, 6, wbits=16 + 9), t))
        cases.append((f"{nm} gzip module", gzip.compress(t, 6), t))
    t = texts[1]
    cases.append(("name", gz_member(t, name=b"AF-A0A0B7P221-F1-model_v4.pdb"), t))
    cases.append(("name + comment + extra", gz_member(t, name=b"x.pdb", comment=b"made for a test", extra=b"AB\x04\x00abcd"), t))
    cases.append(("long name (240)", gz_member(t, name=b"n" * 230), t))
    for n in list(range(0, 70)) + [255, 256, 257, 4000, 4095, 4096, 4097, 8191, 8192, 8193, 12288, 16383, 16384, 16385, 32767, 32768, 32769, 65535, 65536, 70000]:
        d = (texts[0] * 2)[:n]
        cases.append((f"prefix {n}", gz_member(d, 6), d))
    for n in (1, 100, 5000, 70000, 200000):
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        cases.append((f"random {n}", gz_member(d, 6), d))                    # incompressible: stored blocks
        cases.append((f"random {n} level 0", gz_member(d, 0), d))
    cases.append(("zeros 1M", gz_member(bytes(1 << 20), 9), bytes(1 << 20)))  # length-258 matches at distance 1
    for period in (2, 3, 7, 63, 64, 65, 100, 257, 258, 259, 16383, 16384, 16385, 20000, 32768):
        unit = rng.integers(32, 127, period, dtype=np.uint8).tobytes()
        d = (unit * (200000 // period + 2))[:150000]
        cases.append((f"period {period}", gz_member(d, 9), d))
    far = rng.integers(0, 256, 30000, dtype=np.uint8).tobytes()              # matches at the far end of the 32 KB window
    d = far + bytes(2700) + far[:20000] + far[5000:9000] * 3
    cases.append(("far matches", gz_member(d, 9), d))
    few = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 300000))       # four symbols: one- and two-bit codes
    cases.append(("four letters", gz_member(few, 6), few))
    cases.append(("four letters huffman", gz_member(few, 6, zlib.Z_HUFFMAN_ONLY), few))
    one = b"A" * 5000
    cases.append(("one letter huffman", gz_member(one, 6, zlib.Z_HUFFMAN_ONLY), one))
    c = zlib.compressobj(6, zlib.DEFLATED, 31)                               # sync flushes: empty stored blocks between the others
    parts = [c.compress(texts[0][i:i + 30000]) + c.flush(zlib.Z_SYNC_FLUSH) for i in range(0, len(texts[0]), 30000)]
    cases.append(("sync flushes", b"".join(parts) + c.flush(), texts[0]))
    c = zlib.compressobj(6, zlib.DEFLATED, 31)
    parts = [c.compress(texts[3][i:i + 50000]) + c.flush(zlib.Z_FULL_FLUSH) for i in range(0, len(texts[3]), 50000)]
    cases.append(("full flushes", b"".join(parts) + c.flush(), texts[3]))
    return cases


def test_members_of_every_kind_inflate_to_zlibs_bytes(codec, fixtures):
    cases = _corpus(fixtures)
    for _, m, t in cases:
        assert zlib_says(m) == t
    texts, st = codec.inflate([m for _, m, _ in cases])
    bad = [(nm, _why(s)) for (nm, _, _), s in zip(cases, st) if s != 0]
    assert not bad, f"{len(bad)} of {len(cases)} members refused: {bad[:12]}"
    for (nm, _, t), got in zip(cases, texts):
        if got != t:
            k = next((i for i in range(min(len(got), len(t))) if got[i] != t[i]), min(len(got), len(t)))
            raise AssertionError(f"{nm}: text differs from zlib's at byte {k} of {len(t)} (got {len(got)})")


def test_plain_entries_are_copied(codec, fixtures):
    rng = np.random.default_rng(5)
    plain = [fixtures["test_af.pdb"], b"", b"x", fixtures["test.pdb"][:4097], rng.integers(0, 256, 100001, dtype=np.uint8).tobytes()]
    members, kind = [], []
    for i, p in enumerate(plain):
        members += [p, gz_member(fixtures["test_af.pdb"][: 1000 + 37 * i], 6)]
        kind += [0, 1]
    texts, st = codec.inflate(members, kind)
    assert (st == 0).all(), [_why(s) for s in st]
    for i, p in enumerate(plain):
        assert texts[2 * i] == p
        assert texts[2 * i + 1] == fixtures["test_af.pdb"][: 1000 + 37 * i]


def _mutations(fixtures, n_random=1100):
    rng = np.random.default_rng(77)
    base = [gz_member(fixtures["test_af.pdb"], 6), gz_member(fixtures["test_af.pdb"], 1), gz_member(fixtures["test_af.pdb"][:3000], 9),
            gz_member(fixtures["test_af.pdb"], 6, zlib.Z_FIXED), gz_member(fixtures["test_af.pdb"][:5000], 0), gz_member(fixtures["test.cif"][:40000], 6),
            gz_member(fixtures["test_af.pdb"], name=b"model.pdb")]
    out = []
    for b in base:
        t = zlib_says(b)
        out.append(("trailing byte", b + b"\0"))
        out.append(("two members", b + b))
        out.append(("bad crc", b[:-8] + struct.pack("<I", (zlib.crc32(t) ^ 1) & 0xffffffff) + b[-4:]))
        out.append(("bad isize +1", b[:-4] + struct.pack("<I", len(t) + 1)))
        out.append(("bad isize -1", b[:-4] + struct.pack("<I", len(t) - 1)))
        out.append(("isize 0", b[:-4] + struct.pack("<I", 0)))
        out.append(("no trailer", b[:-8]))
        out.append(("half trailer", b[:-3]))
        out.append(("header only", b[:10]))
        out.append(("reserved flag", b[:3] + bytes([b[3] | 0x20]) + b[4:]))
        out.append(("method 7", b[:2] + b"\x07" + b[3:]))
        out.append(("zlib wrapper", zlib.compress(t)))
        out.append(("raw deflate", b[10:-8]))
        out.append(("header crc", gz_member(t, hcrc=True)))
        out.append(("header crc wrong", (lambda m: m[:10] + bytes([m[10] ^ 1]) + m[11:])(gz_member(t, hcrc=True))))
        for cut in rng.integers(11, len(b) - 1, 12):
            out.append(("truncated", b[:int(cut)]))
    for i in range(n_random):
        b = bytearray(base[i % len(base)])
        kind = i % 5
        if kind == 0:                                   # one bit anywhere
            p = int(rng.integers(0, len(b))); b[p] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:                                 # one bit in the first block's header / code lengths
            p = int(rng.integers(10, min(len(b), 120))); b[p] ^= 1 << int(rng.integers(0, 8))
        elif kind == 2:                                 # a byte replaced
            p = int(rng.integers(0, len(b))); b[p] = int(rng.integers(0, 256))
        elif kind == 3:                                 # a few bytes cut out of the middle
            p = int(rng.integers(10, len(b) - 10)); del b[p:p + int(rng.integers(1, 5))]
        else:                                           # two bits
            for _ in range(2):
                p = int(rng.integers(10, len(b))); b[p] ^= 1 << int(rng.integers(0, 8))
        out.append((f"random {kind}", bytes(b)))
    return out


def test_mutated_streams_equal_zlib_or_are_refused(codec, fixtures):
    muts = _mutations(fixtures)
    assert len(muts) >= 1000
    expect = [zlib_says(m) for _, m in muts]
    # (an ISIZE a mutation blew up sizes the text buffer: keep the batch's total in bounds by inflating those on their own)
    texts, st = [None] * len(muts), np.zeros(len(muts), np.int32)
    for lo in range(0, len(muts), 64):
        tt, ss = codec.inflate([m for _, m in muts[lo:lo + 64]])
        texts[lo:lo + 64] = tt; st[lo:lo + 64] = ss
    accepted_by_both = refused_by_both = handed_back = 0
    handed = {}
    for (nm, m), want, got, s in zip(muts, expect, texts, st):
        if s == 0:
            assert want is not None, f"{nm}: the device inflated a stream zlib rejects"
            assert got == want, f"{nm}: text differs from zlib's"
            accepted_by_both += 1
        elif want is None:
            refused_by_both += 1
        else:
            handed_back += 1
            handed[nm] = handed.get(nm, 0) + 1
    # what zlib reads and the device hands back: only the documented classes (trailing bytes / second member / header CRC); a bit
    # flip that lands in the header's MTIME / XFL / OS bytes changes nothing and must still inflate
    assert set(handed) <= {"trailing byte", "two members", "header crc", "random 0", "random 2", "random 4", "random 1", "random 3"}, handed
    assert handed.get("trailing byte", 0) == 7 and handed.get("two members", 0) == 7 and handed.get("header crc", 0) == 7
    assert sum(v for k, v in handed.items() if k.startswith("random")) <= 12, handed
    assert refused_by_both >= 900 and accepted_by_both >= 1
    print(f"mutated streams: {accepted_by_both} inflated like zlib, {refused_by_both} refused like zlib, {handed_back} handed back to zlib: {handed}")


def test_crafted_streams_zlib_rejects(codec):
    """DEFLATE streams written bit by bit: each one is refused by zlib for a reason of its own, and by the device"""
    def bits(*fields):
        v = n = 0
        for val, width in fields:
            v |= val << n; n += width
        return v.to_bytes((n + 7) // 8, "little")

    def wrap(body, text=b""):
        return bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3]) + body + struct.pack("<II", zlib.crc32(text), len(text))
    rev = lambda code, n: int(format(code, f"0{n}b")[::-1], 2)   # Huffman codes enter the stream top bit first
    crafted = {
        "block type 3": bits((1, 1), (3, 2)),
        "stored length mismatch": bits((1, 1), (0, 2), (0, 5)) + struct.pack("<HH", 5, 5) + b"hello",
        "fixed: distance before the start": bits((1, 1), (1, 2), (rev(0x30 + 65, 8), 8), (rev(1, 7), 7), (rev(1, 5), 5), (0, 7)),    # 'A', length 3 distance 2
        "fixed: literal/length 286": bits((1, 1), (1, 2), (rev(0xC0 + 6, 8), 8), (0, 7)),
        "fixed: distance code 30": bits((1, 1), (1, 2), (rev(0x30 + 65, 8), 8), (rev(1, 7), 7), (rev(30, 5), 5), (0, 7)),
        "dynamic: too many length symbols": bits((1, 1), (2, 2), (30, 5), (0, 5), (0, 4)) + bytes(20),
        "dynamic: too many distance symbols": bits((1, 1), (2, 2), (0, 5), (31, 5), (0, 4)) + bytes(20),
        "dynamic: code lengths over-subscribed": bits((1, 1), (2, 2), (0, 5), (0, 5), (15, 4), *[(1, 3)] * 19) + bytes(40),
        "dynamic: code lengths incomplete": bits((1, 1), (2, 2), (0, 5), (0, 5), (0, 4), (1, 3), (0, 3), (0, 3), (0, 3)) + bytes(40),
        "dynamic: repeat without a previous length": bits((1, 1), (2, 2), (0, 5), (0, 5), (0, 4), (1, 3), (1, 3), (0, 3), (0, 3), (0, 1), (0, 2)) + bytes(40),
    }
    names = list(crafted)
    members = [wrap(crafted[k]) for k in names]
    for nm, m in zip(names, members):
        assert zlib_says(m) is None, nm
    _, st = codec.inflate(members)
    assert (st != 0).all(), [(nm, _why(s)) for nm, s in zip(names, st)]
    # and the same framing with a valid body is read: 'A' + end of block in the fixed code
    ok = wrap(bits((1, 1), (1, 2), (rev(0x30 + 65, 8), 8), (0, 7)), b"A")
    assert zlib_says(ok) == b"A"
    texts, st = codec.inflate([ok])
    assert st[0] == 0 and texts[0] == b"A", _why(st[0])


def test_files_to_records_through_the_inflate_stage(codec, fixtures):
    names = ["test.pdb", "test_af.pdb", "multichain.pdb", "test.cif"]
    texts = [fixtures[n] for n in names]
    want = codec.compress_pdb(texts, names)
    files = [gz_member(texts[0], 6), texts[1], gz_member(texts[2], 1), fixtures["test.cif.gz"]]
    gnames = ["test.pdb.gz", "test_af.pdb", "multichain.pdb.gz", "test.cif.gz"]
    # (the record names come from the stems: strip the .gz the way the hosts do before they pass the names)
    got = codec.compress_gz(files, names, is_gz=[n.endswith(".gz") for n in gnames])
    assert (got["file_status"] == want["file_status"]).all(), (got["file_status"], want["file_status"])
    assert np.array_equal(got["off"], want["off"]) and got["blob"].tobytes() == want["blob"].tobytes()
    assert np.array_equal(got["chain_file"], want["chain_file"]) and np.array_equal(got["chain_meta"], want["chain_meta"])
    assert int(got["counts"][0]) >= 6
    # a member that does not inflate: the file comes back, nothing of it is in the batch, the others are untouched
    broken = bytearray(files[0]); broken[len(broken) // 2] ^= 0x10
    got2 = codec.compress_gz([bytes(broken), texts[1], files[2], files[3][:-5]], names, is_gz=[1, 0, 1, 1])
    assert list(got2["file_status"]) == [5, int(want["file_status"][1]), int(want["file_status"][2]), 5]
    keep = [c for c in range(len(want["chain_file"])) if want["chain_file"][c] in (1, 2)]
    assert list(got2["chain_file"]) == [int(want["chain_file"][c]) for c in keep]
    for k, c in enumerate(keep):
        a = want["blob"][int(want["off"][c]):int(want["off"][c + 1])].tobytes()
        b = got2["blob"][int(got2["off"][k]):int(got2["off"][k + 1])].tobytes()
        assert a == b
