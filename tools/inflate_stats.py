#!/usr/bin/env python3
"""Symbol statistics of a gzip member (RFC 1951 / 1952), decoded by a plain-Python inflate: what k_inflate
(foldcomp_amd/csrc/fcz_inflate.h) spends its serial steps on. Output checked against zlib.

  python tools/inflate_stats.py file.gz [...]        (a plain file is gzipped at level 6 first)
"""
import gzip
import sys
import zlib
from collections import Counter

LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEXT = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
DBASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
DEXT = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]
ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]


class Bits:
    def __init__(self, data, pos):
        self.d, self.p = data, pos * 8

    def get(self, n):
        v = 0
        for i in range(n):
            v |= ((self.d[(self.p + i) >> 3] >> ((self.p + i) & 7)) & 1) << i
        self.p += n
        return v


def build(lens):
    """canonical code -> {(length, code): symbol}"""
    count = Counter(l for l in lens if l)
    code, nxt = 0, {}
    for l in range(1, 16):
        code = (code + count.get(l - 1, 0)) << 1
        nxt[l] = code
    table = {}
    for s, l in enumerate(lens):
        if l:
            table[(l, nxt[l])] = s
            nxt[l] += 1
    return table


def decode(b, table):
    code = 0
    for l in range(1, 16):
        code = (code << 1) | b.get(1)
        if (l, code) in table:
            return table[(l, code)], l
    raise ValueError("invalid code")


def inflate_stats(gz):
    assert gz[:3] == b"\x1f\x8b\x08"
    flg, pos = gz[3], 10
    if flg & 4:
        pos += 2 + gz[pos] + (gz[pos + 1] << 8)
    if flg & 8:
        pos = gz.index(0, pos) + 1
    if flg & 16:
        pos = gz.index(0, pos) + 1
    if flg & 2:
        pos += 2
    b = Bits(gz, pos)
    out = bytearray()
    st = Counter()
    dist_hist, len_hist, codelen_hist = Counter(), Counter(), Counter()
    while True:
        last, typ = b.get(1), b.get(2)
        st["blocks_type%d" % typ] += 1
        if typ == 0:
            b.p = (b.p + 7) & ~7
            n = b.get(16)
            b.get(16)
            out += gz[b.p >> 3:(b.p >> 3) + n]
            b.p += 8 * n
        else:
            if typ == 1:
                ll = build([8] * 144 + [9] * 112 + [7] * 24 + [8] * 8)
                dd = build([5] * 30)
            else:
                hlit, hdist, hclen = b.get(5) + 257, b.get(5) + 1, b.get(4) + 4
                cl = [0] * 19
                for i in range(hclen):
                    cl[ORDER[i]] = b.get(3)
                ct = build(cl)
                lens = []
                while len(lens) < hlit + hdist:
                    s, _ = decode(b, ct)
                    st["cl_symbols"] += 1
                    if s < 16:
                        lens.append(s)
                    elif s == 16:
                        lens += [lens[-1]] * (3 + b.get(2))
                    elif s == 17:
                        lens += [0] * (3 + b.get(3))
                    else:
                        lens += [0] * (11 + b.get(7))
                ll, dd = build(lens[:hlit]), build(lens[hlit:])
                st["max_litlen_bits"] = max(st["max_litlen_bits"], max(lens[:hlit]))
                st["max_dist_bits"] = max(st["max_dist_bits"], max(lens[hlit:]))
            while True:
                s, l = decode(b, ll)
                codelen_hist[l] += 1
                if s < 256:
                    out.append(s)
                    st["literals"] += 1
                elif s == 256:
                    break
                else:
                    ln = LBASE[s - 257] + b.get(LEXT[s - 257])
                    d, dl = decode(b, dd)
                    dist = DBASE[d] + b.get(DEXT[d])
                    st["matches"] += 1
                    st["match_bytes"] += ln
                    st["overlapping"] += dist < ln
                    st["dist_code_gt8_bits"] += dl > 8
                    dist_hist[min(dist.bit_length(), 16)] += 1
                    len_hist[min(ln, 64) // 8 * 8] += 1
                    for _ in range(ln):
                        out.append(out[-dist])
        if last:
            break
    assert bytes(out) == zlib.decompress(gz, 31), "this decoder and zlib disagree"
    st["out_bytes"], st["in_bytes"] = len(out), len(gz)
    return st, dist_hist, len_hist, codelen_hist


if __name__ == "__main__":
    for path in sys.argv[1:]:
        raw = open(path, "rb").read()
        gz = raw if raw[:2] == b"\x1f\x8b" else gzip.compress(raw, 6)
        st, dh, lh, ch = inflate_stats(gz)
        sym = st["literals"] + st["matches"]
        print(path, dict(st))
        print("  bytes/symbol %.2f  literal share of symbols %.3f  mean match %.1f" % (st["out_bytes"] / sym, st["literals"] / sym, st["match_bytes"] / max(1, st["matches"])))
        print("  distance bits:", dict(sorted(dh.items())))
        print("  match length (by 8):", dict(sorted(lh.items())))
        print("  litlen code bits:", dict(sorted(ch.items())))
