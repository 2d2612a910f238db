"""`python -m foldcomp_amd` -- the reference CLI surface (src/main.cpp) over the batch GPU codec.

    foldcomp compress   [-t N] [-b B] [-d|-z] [-y] [-r] [-f] [--skip-discontinuous] <pdb|cif|dir|tar(.gz)|db> [<out>]
    foldcomp decompress [-t N] [-a] [-d|-z] [-y] [--check] [-l ids [-m 0|1]] <fcz|dir|tar(.gz)|db> [<out>]
    foldcomp extract    [--plddt|--fasta|--amino-acid] [-p digits] [--no-merge] [--use-title] <fcz|dir|tar|db> [<out>]
    foldcomp check      <fcz|dir|tar|db>
    foldcomp rmsd       <pdb|cif> <pdb|cif>

Same flags, defaults, output naming (src/main.cpp:356-368, 444-508, 643-654) and exit-code behaviour (always
0 once arguments parse, errors go to stderr). `-t` only sizes the host-side parse/format pool: the codec
itself runs on the GPU in batches. Text parsing/formatting stays on the host; all geometry runs in
libfcz_hip.so.
"""
from __future__ import annotations

import argparse
import gzip
import os
import sys
from typing import Iterable, Iterator, List, Optional, Tuple

import numpy as np

from . import _lib, fczfile
from .api import decompress_many, default_codec
from .database import DatabaseReader, DatabaseWriter
from .structure import (AtomTable, Chain, StructureError, build_batch, identify_chains, identify_discontinuous, parse_pdb, parse_structure_gemmi,
                        remove_alternative_position)

VERSION = "0.1.0"
BATCH_CHAINS = 16384


# ---- structure files -----------------------------------------------------------------------------------
def load_structure(name: str, data: bytes) -> Tuple[AtomTable, str]:
    """StructureReader::loadFromBuffer (src/structure_reader.cpp:74-97), the reader of every `compress` input (src/main.cpp:457):
    the NAME only says whether the bytes are gzipped; whether they are PDB or mmCIF text is read off the content
    (gemmi::coor_format_from_content), and either is read by gemmi's rules (structure.parse_pdb_gemmi / parse_cif_gemmi)"""
    base = os.path.basename(name)
    from . import _hostlib
    if _hostlib.load() is not None:                      # the C++ host's readers (the same rules, held equal by the tests)
        t, title = _hostlib.read_structure(data, gz=base.endswith(".gz"))
    else:
        if base.endswith(".gz"):
            data = gzip.decompress(data)
        t, title = parse_structure_gemmi(data)
    return t, (title if title else base)


def is_compressible(stem: str, ext: str) -> bool:
    """isCompressible (utility.cpp:129-140): pdb, cif, and either of them gzipped (the .gz case looks one extension further in)"""
    return ext in ("pdb", "cif") or (ext == "gz" and file_parts(stem)[1] in ("pdb", "cif"))


def file_parts(base: str) -> Tuple[str, str]:
    """getFileParts (utility.cpp:118-126): split at the LAST '.' ("test.cif.gz" -> ("test.cif", "gz"))"""
    i = base.rfind(".")
    return (base, "") if i < 0 else (base[:i], base[i + 1:])


# ---- tar archives as the reference's microtar reads and writes them ----------------------------------------
def iter_tar(path: str) -> Iterator[Tuple[str, bytes]]:
    """TarProcessor (src/input_processor.h:109-198) over lib/microtar: 512-byte headers; a header whose checksum field starts with
    a NUL ends the archive; the member's name is the NAME FIELD ONLY, cut to 99 characters (no ustar prefix, no pax records); a GNU
    long-name record ('L', 'K') carries the next member's name; types '0', '7' and NUL are files, every other type is skipped.
    A gzipped archive is told by its NAME (.gz / .tgz). The same rules as host/foldcomp_hip.cpp (TarStream)."""
    f = gzip.open(path, "rb") if path.endswith((".gz", ".tgz")) else open(path, "rb")
    last, long_name = "", None

    def octal(b: bytes) -> int:
        if b[:1] and b[0] & 0x80:
            return int.from_bytes(bytes([b[0] & 0x7F]) + b[1:], "big")
        t = b.lstrip(b" \t\n\v\f\r"); n = 0
        while n < len(t) and 0x30 <= t[n] <= 0x37:
            n += 1
        return int(t[:n], 8) if n else 0

    with f:
        while True:
            h = f.read(512)
            if len(h) < 512:
                print(f"[Error] tar truncated after entry {last}", file=sys.stderr); return
            if h[148] == 0:
                return
            if 256 + sum(h[:148]) + sum(h[156:]) != octal(h[148:156]):
                print(f"[Error] {os.path.basename(path)}: bad tar header checksum after entry {last}", file=sys.stderr); return
            size, typ = octal(h[124:136]), h[156:157]
            padded = size + (512 - size % 512) % 512
            if typ in (b"L", b"K"):
                d = f.read(padded)
                if len(d) < padded:
                    print(f"[Error] cannot read entry {last}", file=sys.stderr); return
                long_name = d[:size].split(b"\0")[0].decode("latin-1"); continue
            name = long_name if long_name is not None else h[:99].split(b"\0")[0].decode("latin-1")
            long_name, last = None, name
            d = f.read(padded)
            if len(d) < padded:
                print(f"[Error] cannot read entry {name}", file=sys.stderr); return
            if typ in (b"0", b"7", b"\0"):
                yield name, d[:size]


def tar_header(name: str, size: int) -> bytes:
    """mtar_write_file_header (lib/microtar): zeros; name; mode 644, owner 0, size, mtime 0 printed with %o; type '0'; the checksum as
    "%06o", a NUL and a blank -- no ustar magic, no times"""
    h = bytearray(512)
    nb = name.encode("latin-1")[:99]
    h[:len(nb)] = nb
    h[100:103] = b"644"; h[108:109] = b"0"
    sz = b"%o" % size
    h[124:124 + len(sz)] = sz
    h[136:137] = b"0"; h[156:157] = b"0"
    cs = b"%06o" % (256 + sum(h[:148]) + sum(h[156:]))
    h[148:148 + len(cs)] = cs
    h[155:156] = b" "
    return bytes(h)


# ---- entry sources -------------------------------------------------------------------------------------
def iter_entries(inp: str, recursive: bool, id_list: Optional[str], id_mode: int) -> Iterator[Tuple[str, bytes]]:
    if os.path.exists(inp + ".dbtype"):
        r = DatabaseReader(inp)
        ids = [(i, None) for i in range(len(r))]
        if id_list:
            # an entry of an id list goes by the list's own line -- with --id-mode 0 that is its KEY (src/input_processor.h:262-278)
            want = [l.strip() for l in open(id_list) if l.strip()]
            ids = [(r.id_of_key(int(w)) if id_mode == 0 else r.id_of_name(w), w) for w in want]
            for i, w in ids:
                if i < 0:
                    print(f"[Warning] {w} not found in database.", file=sys.stderr)
            ids = [(i, w) for i, w in ids if i >= 0]
        for i, w in ids:
            # the stored bytes, MMseqs NUL terminator included: the codec takes the record length from the header (a record may
            # itself end in zero bytes, so nothing is stripped here)
            yield (r.name(i) if w is None else w), r.data(i)
        r.close()
    elif inp.endswith((".tar", ".tar.gz", ".tgz")) and not os.path.isdir(inp):
        yield from iter_tar(inp)
    elif os.path.isdir(inp):
        for root, dirs, files in os.walk(inp):
            dirs.sort()
            for f in sorted(files):
                p = os.path.join(root, f)
                yield p, open(p, "rb").read()
            if not recursive:
                break
    else:
        yield inp, open(inp, "rb").read()


class Sink:
    """file / directory / tar / database outputs (src/main.cpp:510-530, 656-687)"""

    def __init__(self, output: str, kind: str, overwrite: bool):
        self.kind, self.output, self.overwrite = kind, output, overwrite
        self.key = 0
        if kind == "db":
            self.db = DatabaseWriter(output)
        elif kind == "tar":
            self.tar = open(output, "wb")
        elif kind == "dir":
            os.makedirs(output, exist_ok=True)

    def put(self, name: str, data: bytes, db_name: Optional[str] = None, nul: bool = False):
        if self.kind == "db":
            self.db.append(data + (b"\0" if nul else b""), self.key, db_name or name); self.key += 1
        elif self.kind == "tar":
            self.tar.write(tar_header(os.path.basename(name), len(data)) + data + b"\0" * ((512 - len(data) % 512) % 512))
        else:
            path = name if self.kind == "file" else os.path.join(self.output, os.path.basename(name))
            if os.path.exists(path) and not self.overwrite:
                print(f"[Error] Output file already exists: {os.path.basename(path)}", file=sys.stderr)
                return
            with open(path, "wb") as f:
                f.write(data)

    def close(self):
        if self.kind == "db":
            self.db.close()
        elif self.kind == "tar":
            self.tar.write(b"\0" * 1024)     # mtar_write_finalize: two NUL records
            self.tar.close()


# ---- modes ---------------------------------------------------------------------------------------------
def host_fragments(name: str, data: bytes, a, kind: str, single: bool, output) -> List[Tuple[str, str, Chain]]:
    """one structure file -> its fragments as (output file name, database name, chain): src/main.cpp:455-508 on the host
    (parse, removeAlternativePosition, identifyChains, identifyDiscontinousResInd, names and title)"""
    out: List[Tuple[str, str, Chain]] = []
    base = os.path.basename(name)
    stem, ext = file_parts(base)
    if kind in ("tar", "db"):
        out_file = stem
    elif single:
        out_file, ext = file_parts(output)      # a single-file run names and suffixes by the OUTPUT path (src/main.cpp:449-451)
    else:
        out_file = stem
    t, title = load_structure(name, data)
    if len(t) == 0:
        print(f"[Error] No atoms found in the input file: {base}", file=sys.stderr); return out
    if title == base:
        title = out_file            # src/main.cpp:465
    t = remove_alternative_position(t)
    chains = identify_chains(t)
    for cs in chains:
        frags = identify_discontinuous(t, cs)
        if a.skip_discontinuous and len(frags) > 1:
            print(f"Skipping discontinuous chain: {base}", file=sys.stderr); continue
        for j, sl in enumerate(frags):
            fname = out_file + (t.chain[cs.start] if len(chains) > 1 else "")
            if len(frags) > 1:
                fname += f"_{j}"
            if kind != "db":
                fname += ".fcz" if is_compressible(out_file, ext) else "." + ext     # (src/main.cpp:498-502: the dot also without an extension)
            out.append((fname, out_file, Chain(title, t.take(sl))))
    return out


def run_compress(a, inputs, output, kind, single):
    sink = Sink(output, kind, a.overwrite)
    pending: List[list] = []   # [file name, db name, chain, input ordinal, record or None]

    def flush():
        if not pending:
            return
        good = list(pending)
        try:
            batch = build_batch([p[2] for p in good], a.brk)
        except StructureError:
            # isolate the offending chains one by one
            good = []
            for p in pending:
                try:
                    build_batch([p[2]], a.brk); good.append(p)
                except StructureError as ee:
                    print(f"[Error] compressing {p[0]}: {ee}", file=sys.stderr)
            batch = build_batch([p[2] for p in good], a.brk) if good else None
        if batch is not None:
            blob, off, st = default_codec().compress_batch(batch, strict=False)
            for i, p in enumerate(good):
                if st[i] != 0:
                    print(f"[Error] compressing {p[0]}: {_lib.load().fcz_status_string(int(st[i])).decode()}", file=sys.stderr); continue
                p[4] = blob[off[i]:off[i + 1]].tobytes()
        if kind in ("db", "tar"):
            for fname, dbname, _, _, rec in pending:
                if rec is not None:
                    sink.put(fname, rec, db_name=dbname)
        else:
            # files of a directory, in the order the reference's lambda meets the fragments of an input (src/main.cpp:466-531): it
            # RETURNS at the first output name that already exists -- from an earlier run, or an earlier fragment of the same input
            # (chain A, chain B, chain A again) -- unless -y, where the later fragment replaces the earlier one. A fragment refused
            # here (the reference writes its shifted record) keeps its place in that order: under a name stands what the reference
            # writes under it, or nothing (host/foldcomp_hip.cpp write_fragments_in_order)
            cur, seen, stopped = None, set(), False
            for fname, dbname, _, fidx, rec in pending:
                if fidx != cur:
                    cur, seen, stopped = fidx, set(), False
                if stopped:
                    continue
                path = fname if kind == "file" else os.path.join(output, os.path.basename(fname))
                if not a.overwrite:
                    if fname in seen or os.path.exists(path):
                        print(f"[Error] Output file already exists: {os.path.basename(path)}", file=sys.stderr); stopped = True; continue
                    seen.add(fname)
                    if rec is not None:
                        open(path, "wb").write(rec)
                else:
                    if rec is not None:
                        open(path, "wb").write(rec)
                    elif fname in seen and os.path.exists(path):
                        os.remove(path)
                    seen.add(fname)
        pending.clear()

    fidx = 0
    for inp in inputs:
        for name, data in iter_entries(inp, a.recursive, None, 1):
            fidx += 1
            try:
                pending.extend([fn, dbn, ch, fidx, None] for fn, dbn, ch in host_fragments(name, data, a, kind, single, output))
            except Exception as e:  # noqa: BLE001 - parse errors are reported and skipped like the reference
                print(f"[Error] {os.path.basename(name)}: {e}", file=sys.stderr); continue
            if len(pending) >= BATCH_CHAINS:
                flush()
    flush()
    sink.close()


def run_decompress(a, inputs, output, kind, single):
    sink = Sink(output, kind, a.overwrite)
    names, ents = [], []

    def flush():
        if not ents:
            return
        res = decompress_many(ents, alt_order=a.alt, skip_bad=True)
        for nm, r in zip(names, res):
            if r is None:
                print(f"[Error] decompressing {nm}", file=sys.stderr); continue
            stem, ext = file_parts(os.path.basename(nm))
            fname = output if single and kind == "file" else stem + ".pdb"      # src/main.cpp:646-653
            sink.put(fname, r[1].encode("latin-1"), db_name=stem, nul=True)
        names.clear(); ents.clear()

    for inp in inputs:
        for name, data in iter_entries(inp, a.recursive, a.id_list, a.id_mode):
            if a.check:
                from . import _lib
                arr = np.frombuffer(data, np.uint8)
                rc = _lib.load().fcz_check(arr.ctypes.data, len(arr))
                if rc != 0:
                    # the reference prints printValidityError's line with the record's TITLE (src/main.cpp:630-635)
                    what = {1: "Number of backbone angles does not match header", 2: "Number of sidechain angles does not match header",
                            3: "Number of temperature factors does not match header", 4: "All backbone angles are empty",
                            5: "All sidechain angles are empty", 6: "All temperature factors are empty"}
                    if rc in what and len(data) >= 76:
                        o_title = 76 + 4 * data[12]; tl = int.from_bytes(data[24:28], "little")
                        print(f"[Error] {what[rc]}: {bytes(data[o_title:o_title + tl]).decode('latin-1') if o_title + tl <= len(data) else ''}", file=sys.stderr)
                    else:
                        print(f"[Error] invalid FCZ entry skipped: {name}", file=sys.stderr)
                    continue
            names.append(name); ents.append(data)
            if len(ents) >= BATCH_CHAINS:
                flush()
    flush()
    sink.close()


def run_extract(a, inputs, output, kind, single):
    """pLDDT / sequence strings come from the device (k_extract); the FASTA-like / TSV wrapping is host text"""
    merged = [] if (a.merge and not single and kind not in ("db", "tar")) else None     # src/main.cpp:738-741
    sink = Sink(output.rstrip("/"), kind, True) if kind in ("db", "tar") else None
    out_dir_made = False
    pend: List[Tuple[str, bytes, fczfile.FczRecord]] = []

    def emit(name, rec, s):
        nonlocal out_dir_made
        title = rec.title if a.use_title else name      # the entry's name as the run met it -- a directory's file with its path (src/main.cpp:780-781)
        if a.ext_mode == 0:
            text = fczfile.fasta_like(title, s) if a.plddt_digits == 1 else fczfile.tsv_line(title, rec.n_residues, s)
        else:
            text = fczfile.fasta_like(title, s)
        if sink is not None:
            # a tar member <stem>.<suffix>, or a database record under <stem> with the MMseqs terminator (src/main.cpp:790-844)
            stem = file_parts(os.path.basename(name))[0]
            sink.put(stem + "." + a.suffix, text.encode("latin-1"), db_name=stem, nul=True)
        elif single:
            open(output, "w").write(text)
        elif merged is not None:
            merged.append(text)
        else:
            if not out_dir_made:
                os.makedirs(output, exist_ok=True); out_dir_made = True
            open(os.path.join(output, file_parts(os.path.basename(name))[0] + "." + a.suffix), "w").write(text)

    def flush():
        if not pend:
            return
        off = np.zeros(len(pend) + 1, np.uint64)
        off[1:] = np.cumsum([len(d) for _, d, _ in pend])
        blob = np.frombuffer(b"".join(d for _, d, _ in pend), np.uint8)
        digits = min(max(int(a.plddt_digits), 1), 4)
        outs = default_codec().extract(blob, off, mode=0 if a.ext_mode == 0 else 1, digits=digits)
        for (name, _, rec), s in zip(pend, outs):
            if not s and rec.n_residues:
                print(f"[Error] reading {name}", file=sys.stderr); continue
            emit(name, rec, s.decode("latin-1"))
        pend.clear()

    for inp in inputs:
        for name, data in iter_entries(inp, a.recursive, a.id_list, a.id_mode):
            try:
                rec = fczfile.parse(data)
            except fczfile.FczFormatError:
                print(f"[Error] reading {name}", file=sys.stderr); continue
            pend.append((name, data, rec))
            if len(pend) >= BATCH_CHAINS:
                flush()
    flush()
    if sink is not None:
        sink.close()
    if merged is not None:
        open(output.rstrip("/") if not output.endswith("/") else output.rstrip("/") + "." + a.suffix, "w").write("".join(merged))


def run_check(a, inputs):
    from . import _lib
    lib = _lib.load()
    # the lines of printValidityError (src/foldcomp.cpp:1534-1560), word for word: "[Error] <what>: <name>"
    msgs = {1: "Number of backbone angles does not match header", 2: "Number of sidechain angles does not match header",
            3: "Number of temperature factors does not match header", 4: "All backbone angles are empty",
            5: "All sidechain angles are empty", 6: "All temperature factors are empty"}
    for inp in inputs:
        for name, data in iter_entries(inp, a.recursive, a.id_list, a.id_mode):
            arr = np.frombuffer(data, np.uint8)
            rc = lib.fcz_check(arr.ctypes.data, len(arr)) if len(arr) else -5
            if rc == 0:
                print(f"[Info] {name} is valid.")
            else:
                print(f"[Error] {msgs[rc]}: {name}" if rc in msgs else f"[Error] {name}: not a valid FCZ entry", file=sys.stderr)


def run_rmsd(a, p1, p2):
    t1, _ = load_structure(p1, open(p1, "rb").read()); t2, _ = load_structure(p2, open(p2, "rb").read())
    if len(t1) != len(t2):
        print("[Error] The number of atoms in the two structures differ.", file=sys.stderr); return
    def rms(m):   # atom_coordinate.cpp:424-434: float accumulation, no superposition
        d = (t1.xyz[m] - t2.xyz[m]).astype(np.float32)
        return float(np.sqrt(np.float32(np.sum(d * d, dtype=np.float32)) / np.float32(max(int(m.sum()), 1))))
    bbm = np.asarray([x in ("N", "CA", "C") for x in t1.atom])
    n_res = len(set(zip(t1.chain, t1.res_index.tolist())))
    print(f"{p1}\t{p2}\t{n_res}\t{len(t1)}\t{rms(bbm):g}\t{rms(np.ones(len(t1), bool)):g}")


def main(argv=None):
    ap = argparse.ArgumentParser(prog="foldcomp", add_help=True)
    ap.add_argument("-v", "--version", action="store_true")
    ap.add_argument("-t", "--threads", type=int, default=1)
    ap.add_argument("-r", "--recursive", action="store_true")
    ap.add_argument("-f", "--file", action="store_true", dest="file_input")
    ap.add_argument("-a", "--alt", action="store_true")
    ap.add_argument("-b", "--break", type=int, default=25, dest="brk")
    ap.add_argument("-z", "--tar", action="store_true")
    ap.add_argument("-d", "--db", action="store_true")
    ap.add_argument("-y", "--overwrite", action="store_true")
    ap.add_argument("-l", "--id-list", default=None)
    ap.add_argument("-m", "--id-mode", type=int, default=1)
    ap.add_argument("--skip-discontinuous", action="store_true")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--plddt", action="store_const", const=0, dest="ext_mode", default=0)
    ap.add_argument("--fasta", "--amino-acid", action="store_const", const=1, dest="ext_mode")
    ap.add_argument("-p", "--plddt-digits", type=int, default=1)
    ap.add_argument("--no-merge", action="store_false", dest="merge")
    ap.add_argument("--use-title", action="store_true")
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--use-cache", action="store_true")
    ap.add_argument("--gpus", type=int, default=0,
                    help="compress / decompress into a database (-d) sharded over N GPUs of the node, one process per GPU "
                         "(foldcomp_amd/sharded_cli.py); 1 = the same code in a 1-rank group; 0 = this process only")
    ap.add_argument("--json-stats", action="store_true")
    ap.add_argument("mode", nargs="?")
    ap.add_argument("input", nargs="?")
    ap.add_argument("output", nargs="?")
    a = ap.parse_intermixed_args(argv)
    if a.version:
        print(f"foldcomp {VERSION}"); return 0
    if a.mode not in ("compress", "decompress", "extract", "check", "rmsd") or a.input is None:
        ap.print_usage(); return 0
    inp = a.input.rstrip("/")
    if not os.path.exists(inp):
        print(f"[Error] {inp} does not exist.", file=sys.stderr); return 1
    if a.mode == "rmsd":
        run_rmsd(a, inp, a.output); return 0
    a.plddt_digits = min(max(a.plddt_digits, 1), 4)
    a.suffix = {"compress": "fcz", "decompress": "pdb", "check": ""}.get(a.mode) if a.mode != "extract" else (
        "fasta" if a.ext_mode == 1 else ("plddt" if a.plddt_digits == 1 else "plddt.tsv"))
    inputs, singles = [inp], []
    if a.file_input:
        inputs = []
        for line in open(inp).read().splitlines():
            (singles if line.endswith((".pdb", ".pdb.gz", ".cif", ".cif.gz", ".fcz")) else inputs).append(line)
        inputs += singles
    single = (not a.file_input and os.path.isfile(inp) and not inp.endswith((".tar", ".tar.gz", ".tgz"))
              and not os.path.exists(inp + ".dbtype"))
    output = a.output.rstrip("/") if a.output else None
    if output and output.endswith(".tar"):
        a.tar = True
    if output is None and a.mode != "check":
        if a.db:
            output = inp + "_db"
        elif a.tar:
            output = f"{inp}.{a.suffix}.tar"
        elif single:
            output = os.path.splitext(inp)[0] + "." + a.suffix
        else:
            output = f"{inp}_{a.suffix}/"
    kind = "db" if a.db else "tar" if a.tar else ("file" if single else "dir")
    if a.gpus >= 1 and a.mode in ("compress", "decompress"):
        if kind != "db":
            print("[Error] --gpus shards a database run: add -d", file=sys.stderr); return 1
        from . import sharded_cli
        if a.gpus > 1 and "RANK" not in os.environ:
            return sharded_cli.launch(list(argv) if argv is not None else sys.argv[1:], a.gpus)
        return sharded_cli.run(a, inputs, output.rstrip("/"))
    # the reference's announcement of what it is about to do (src/main.cpp:392-403, :564-575, :717-734, :870-875), word for word
    verb = {"compress": "Compressing", "decompress": "Decompressing", "extract": "Extracting", "check": "Checking"}[a.mode]
    shown = (output or "").rstrip("/")
    if a.mode == "check":
        print(f"Checking {inp}" if len(inputs) == 1 else f"Checking files in {inp} using {a.threads} threads")
    elif single:
        print(f"{verb} {inp} to {shown}")
    else:
        print(f"{verb} files in {inp} using {a.threads} threads")
        print(f"Output database: {shown}" if a.db else f"Output tar file: {shown}" if a.tar else
              f"Output directory: {shown}" if (a.mode != "extract" or not a.merge) else f"Output: {shown}")
    sys.stdout.flush()
    if a.mode == "compress":
        run_compress(a, inputs, output.rstrip("/") if kind != "file" else output, kind, single)
    elif a.mode == "decompress":
        run_decompress(a, inputs, output.rstrip("/") if kind != "file" else output, kind, single)
    elif a.mode == "extract":
        run_extract(a, inputs, output, kind, single)
    else:
        run_check(a, inputs)
    return 0


def _main_guarded(argv=None):
    """main() with the database reader's refusals (a malformed .index line, database.py) printed like every other input error"""
    try:
        return main(argv)
    except ValueError as e:
        print(f"[Error] {e}", file=sys.stderr)
        return 1


if __name__ == "__main__":
    sys.exit(_main_guarded())
