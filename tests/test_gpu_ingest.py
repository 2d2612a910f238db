"""Structure ingest on the device (fcz_ingest_pdb_*, foldcomp_amd/csrc/fcz_ingest.h) -- SURVEY.md section 8 row f3.

PDB text in, fcz_chain_batch out, every step on the GPU. Parity:
  * the reference's own test files (tests/golden/reference_ingest.npz: test.pdb, test_af.pdb, multichain.pdb): the batch equals
    the batch built from what the REFERENCE's reader (gemmi) + removeAlternativePosition + identifyChains +
    identifyDiscontinousResInd hand to Foldcomp::compress (src/main.cpp:455-508), bit for bit, names and titles included;
  * seeded synthetic files, alternative positions, HETATM, chain changes with and without an N, gaps, CRLF line ends, a last
    line without a line end, TITLE / HEADER records: == the Python host's parser + fragmenting on the same text;
  * what the device does not decide (fields outside the fixed-column layout) is handed back (file_status), never guessed;
  * text -> FCZ in one call (fcz_compress_pdb_*) == host parse + fcz_compress_batch, byte for byte."""
import os
import sys

import numpy as np
import pytest

from _cases import golden_batch, mutated_pdb
from foldcomp_amd.codec import Codec
from foldcomp_amd.structure import AtomTable, Chain, StructureError, build_batch, identify_chains, identify_discontinuous, parse_pdb_gemmi, remove_alternative_position
from test_host_cpp import _pdb_text

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ing():
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_ingest.npz"))


def _read_any(text):
    """the host reader as the drivers call it: PDB or mmCIF by content (StructureReader::loadFromBuffer). The C++ readers when
    built (forty times the speed; held equal to the Python restatement and to the live reference in test_ingest_vs_reference.py)"""
    from foldcomp_amd import _hostlib
    from foldcomp_amd.structure import parse_structure_gemmi
    if _hostlib.load() is not None:
        return _hostlib.read_structure(bytes(text), gz=False)
    return parse_structure_gemmi(bytes(text))


def _host_expect(texts, names, brk=25, skip_disc=False, reader=None):
    """what the host reader (gemmi's rules, checked against the live reference in test_ingest_vs_reference.py) makes of the files:
    (batch, fragment names, chain_file, refused [(file, name)], files the reader fails)"""
    chains, out_names, cfile, refused, failed = [], [], [], [], set()
    for fi, (text, base) in enumerate(zip(texts, names)):
        stem = base.rsplit(".", 1)[0] if "." in base else base
        try:
            t, title = (reader or parse_pdb_gemmi)(text)
        except StructureError:
            failed.add(fi); continue
        t = remove_alternative_position(t)
        if len(t) == 0:
            continue
        if title == "" or title == base:
            title = stem
        cs_all = identify_chains(t)
        for cs in cs_all:
            frags = identify_discontinuous(t, cs)
            for j, sl in enumerate(frags):
                nm = stem + (t.chain[cs.start] if len(cs_all) > 1 else "") + (f"_{j}" if len(frags) > 1 else "")
                nm_ref = stem + (t.chain[cs.start][:1] if len(cs_all) > 1 else "") + (f"_{j}" if len(frags) > 1 else "")   # (a refusal is reported with the chain's first character)
                ch = Chain(title, t.take(sl))
                try:
                    if skip_disc and len(frags) > 1:
                        raise StructureError("skip")
                    build_batch([ch], brk)
                except StructureError:
                    refused.append((fi, nm_ref)); continue
                chains.append(ch); out_names.append(nm); cfile.append(fi)
    return (build_batch(chains, brk) if chains else None), out_names, cfile, refused, failed


def _name_of(base, meta, chain_name=None):
    stem = base.rsplit(".", 1)[0] if "." in base else base
    nm = stem
    if meta & (1 << 16):
        nm += chain_name if chain_name else chr(meta & 0xff)      # (mmCIF chain names have up to four characters: Codec.chain_names)
    if meta & (1 << 17):
        nm += f"_{(meta >> 8) & 0xff}"
    return nm


def _same_batch(got, exp):
    assert got.n_chains == exp.n_chains and got.n_residues == exp.n_residues and got.n_atoms == exp.n_atoms
    for k in ("res_off", "atom_off", "atom_code", "res_code", "first_res_index", "first_atom_index", "chain_id", "title_off"):
        assert np.array_equal(np.asarray(getattr(got, k)).astype(np.int64), np.asarray(getattr(exp, k)).astype(np.int64)), k
    for k in ("x", "y", "z", "bfac_ca"):
        assert np.array_equal(getattr(got, k).view(np.uint32), getattr(exp, k).view(np.uint32)), k
    assert bytes(got.titles) == bytes(exp.titles)


def _check(codec, texts, names, brk=25, skip_disc=False, reader=None):
    exp, exp_names, exp_file, exp_ref, failed = _host_expect(texts, names, brk, skip_disc, reader)
    assert not failed
    b, cfile, cmeta, fstat, refused = codec.ingest_pdb(texts, names, brk, skip_disc)
    assert set(int(v) for v in fstat) <= {0, 4}, fstat        # nothing here needs the host parser (4 = a file without atoms)
    if exp is None:
        assert b.n_chains == 0
    else:
        _same_batch(b, exp)
        assert [_name_of(names[f], m) for f, m in zip(cfile, cmeta)] == exp_names
        assert list(cfile) == exp_file
    assert sorted((int(f), _name_of(names[int(f)], int(m))) for f, m in refused) == sorted(exp_ref)
    return b, cfile, cmeta, fstat, refused


@pytest.mark.parametrize("fn", ["test.pdb", "test_af.pdb", "multichain.pdb", "test.cif.gz"])
def test_device_ingest_equals_reference_reader(codec, ing, fn):
    """the device's batch == the batch built from the REFERENCE reader's atom table and fragments (reference-minted goldens)"""
    strs = lambda a: [bytes(r).rstrip(b"\0").decode() for r in a]
    k = f"ingest:{fn}"
    t = AtomTable(strs(ing[f"{k}/atom"]), strs(ing[f"{k}/residue"]), [chr(c) for c in ing[f"{k}/chain"]], ing[f"{k}/atom_index"],
                  ing[f"{k}/res_index"], ing[f"{k}/xyz"], ing[f"{k}/bfac"])
    frag = [tuple(int(v) for v in f) for f in ing[f"{k}/frag"]]
    n_chains = int(ing[f"{k}/n_chains"][0])
    ref_title = bytes(ing[f"{k}/title"]).decode("latin-1")
    stem = fn.rsplit(".", 1)[0]
    title = stem if ref_title == fn else ref_title
    names, chains = [], []
    for (a, b, ci, fj) in frag:
        n_in = sum(1 for f in frag if f[2] == ci)
        names.append(stem + (t.chain[a] if n_chains > 1 else "") + (f"_{fj}" if n_in > 1 else ""))
        chains.append(Chain(title, t.take(slice(a, b))))
    exp = build_batch(chains, 25)
    data = ing[f"file:{fn}"].tobytes()
    if fn.endswith(".gz"):                       # (the hosts' read threads inflate; the text is parsed on the device: mmCIF here)
        import gzip
        data = gzip.decompress(data)
    got, cfile, cmeta, fstat, refused = codec.ingest_pdb([data], [fn])
    assert fstat[0] == 0 and len(refused) == 0
    _same_batch(got, exp)
    assert [_name_of(fn, int(m)) for m in cmeta] == names


def test_device_ingest_many_files_and_the_awkward_ones(codec, golden):
    z, _ = golden
    base = {n: _pdb_text(z, n) for n in ("pdb:test_af", "pdb:test", "syn:len350", "syn:len26", "syn:len129", "pdb:multichainA")}
    b0, b1 = _pdb_text(z, "pdb:multichainB_0"), _pdb_text(z, "pdb:multichainB_1")
    a = base["pdb:multichainA"]
    texts, names = [], []

    def add(name, text):
        names.append(name); texts.append(text.encode("latin-1") if isinstance(text, str) else text)
    for i, (n, t) in enumerate(base.items()):
        add(f"f{i}_{n.split(':')[1]}.pdb", t)
    # alternative positions (every 7th ATOM line doubled, once tripled), HETATM records, a REMARK, three fragments, -b 25
    la, dup = a.splitlines(), []
    for i, l in enumerate(la):
        dup.append(l)
        if l.startswith("ATOM") and i % 7 == 3:
            dup.append(l[:30] + "   9.999   9.999   9.999" + l[54:])
            if i % 21 == 3:
                dup.append(l[:30] + "  -0.001  -0.000 -99.999" + l[54:])
    het = "HETATM 9001  O   HOH A 900      11.000  12.000  13.000  1.00 30.00           O  \n"
    add("multi.pdb", "REMARK test\n" + "\n".join(dup) + "\n" + het + b0 + b1 + "END\n")
    # CRLF line ends; a last line without a line end; empty lines; a TITLE over two records; a HEADER id that wins over TITLE
    t_af = base["pdb:test_af"]
    add("crlf.pdb", t_af.replace("\n", "\r\n"))
    add("noeol.pdb", t_af.rstrip("\n").rsplit("\nTER", 1)[0])
    add("blank.pdb", "\n\n" + t_af.replace("\nATOM", "\n\nATOM", 5) + "\n\n\n")
    add("titled.pdb", "TITLE     A PROTEIN OF SOME KIND                                       \nTITLE    2 CONTINUED HERE\n" + t_af)
    add("header.pdb", "HEADER    HYDROLASE                               01-JAN-00   1ABC              \nTITLE     IGNORED\n" + t_af)
    add("title_is_name.pdb", "TITLE     title_is_name.pdb\n" + t_af)
    # chain id changes at a non-N atom: the atoms up to the next N belong to nobody; and a chain change with no N after it
    lines = [l for l in base["syn:len26"].splitlines() if l.startswith("ATOM")]
    k = next(i for i, l in enumerate(lines) if i > 40 and l[12:16].strip() == "CA")
    mixed = lines[:k] + [l[:21] + "B" + l[22:] for l in lines[k:]]
    add("chain_at_ca.pdb", "\n".join(mixed) + "\n")
    tail = lines[:-3] + [l[:21] + "C" + l[22:] for l in lines[-3:] if l[12:16].strip() != "N"]
    add("chain_no_n.pdb", "\n".join(tail) + "\n")
    # gaps in the residue numbering; a residue the codec does not know; a residue without its CA; a file without atoms
    lg = [l for l in base["syn:len129"].splitlines() if l.startswith("ATOM")]
    gap = [l[:22] + "%4d" % (int(l[22:26]) + (5 if int(l[22:26]) > 40 else 0) + (7 if int(l[22:26]) > 90 else 0)) + l[26:] for l in lg]
    add("gaps.pdb", "\n".join(gap) + "\n")
    ala = next(l[22:26] for l in t_af.splitlines() if l.startswith("ATOM") and l[17:20] == "ALA")
    add("mse.pdb", "\n".join(l[:17] + "MSE" + l[20:] if l.startswith("ATOM") and l[22:26] == ala else l for l in t_af.splitlines()) + "\n")
    # insertion codes (52, 52A, 52B): the reader keeps file order
    ins = [l[:22] + ("%4d%s" % (52, " AB"[int(l[22:26]) - 52]) if 52 <= int(l[22:26]) <= 54 else "%4d " % (int(l[22:26]) - (2 if int(l[22:26]) > 54 else 0))) + l[27:] for l in lg]
    add("icode.pdb", "\n".join(ins) + "\n")
    add("no_ca.pdb", "\n".join(l for i, l in enumerate(t_af.splitlines()) if not (l.startswith("ATOM") and l[12:16].strip() == "CA" and int(l[22:26]) == 7)) + "\n")
    add("empty.pdb", "HEADER    nothing to see\nEND\n")
    add("zero.pdb", "")
    _check(codec, texts, names)
    _check(codec, texts, names, brk=10)
    _check(codec, texts, names, skip_disc=True)
    # the same files one by one (a batch of one file has its own offsets)
    for t, n in list(zip(texts, names))[6:12]:
        _check(codec, [t], [n])


def test_device_ingest_hands_back_what_it_does_not_decide(codec, golden):
    z, _ = golden
    t_af = _pdb_text(z, "pdb:test_af")
    lines = t_af.splitlines()
    i = next(k for k, l in enumerate(lines) if l.startswith("ATOM"))
    sci = lines[:i] + [lines[i][:30] + " 1.0e+01" + lines[i][38:]] + lines[i + 1:]              # a number outside the fixed layout
    short = lines[:i] + [lines[i][:50]] + lines[i + 1:]                                          # an ATOM record cut short
    hyb = lines[:i] + [lines[i][:6] + "A0000" + lines[i][11:]] + lines[i + 1:]                    # hybrid-36 serial
    long_title = "".join("TITLE   %2d %s\n" % (k + 1, "X" * 60) for k in range(12)) + t_af       # 12 x 60 characters of title
    texts = [("\n".join(v) + "\n").encode() for v in (sci, short, hyb)] + [long_title.encode(), t_af.encode()]
    names = ["sci.pdb", "short.pdb", "hyb.pdb", "long_title.pdb", "good.pdb"]
    b, cfile, cmeta, fstat, refused = codec.ingest_pdb(texts, names)
    assert list(fstat) == [1, 1, 1, 2, 0]
    assert b.n_chains == 1 and list(cfile) == [4]
    exp, *_ = _host_expect([texts[4]], [names[4]])
    _same_batch(b, exp)


def test_text_to_fcz_in_one_call(codec, golden):
    """fcz_compress_pdb_*: the records equal those of host parse + fcz_compress_batch on the same files; and of the goldens"""
    z, _ = golden
    cases = ["pdb:test_af", "pdb:test", "syn:len350", "syn:len26", "syn:len129"]
    texts = [_pdb_text(z, n).encode() for n in cases] * 40
    names = [f"f{i:03d}.pdb" for i in range(len(texts))]
    r = codec.compress_pdb(texts, names)
    assert (r["status"] == 0).all() and (r["file_status"] == 0).all() and len(r["refused"]) == 0
    exp, exp_names, *_ = _host_expect(texts, names)
    blob, off, st = codec.compress_batch(exp)
    assert np.array_equal(off, r["off"]) and blob.tobytes() == r["blob"].tobytes()

    def no_title(f):
        na, tl = f[12], int.from_bytes(f[24:28], "little")
        return f[:24] + f[28:76 + 4 * na] + f[76 + 4 * na + tl:]
    for i, n in enumerate(cases):
        rec = r["blob"][int(r["off"][i]):int(r["off"][i + 1])].tobytes()
        assert no_title(rec) == no_title(z[f"{n}/fcz"].tobytes()), n


def test_non_ascii_file_name_keeps_its_whole_stem(codec, golden):
    """a title that falls back to the file's stem: the stem is cut at the last dot of the ENCODED name (the name blob is UTF-8;
    a character index would drop one byte per two-byte character before it)"""
    z, _ = golden
    text = "".join(l + "\n" for l in _pdb_text(z, "syn:len26").splitlines() if not l.startswith(("TITLE", "HEADER"))).encode()
    names = ["prot\u00e9ine.v2.pdb", "\u86cb\u767d\u8d28.pdb", "plain.v3.pdb"]
    r = codec.compress_pdb([text] * 3, names)
    assert (r["status"] == 0).all() and (r["file_status"] == 0).all()
    for i, nm in enumerate(names):
        rec = r["blob"][int(r["off"][i]):int(r["off"][i + 1])].tobytes()
        na, tl = rec[12], int.from_bytes(rec[24:28], "little")
        assert rec[76 + 4 * na:76 + 4 * na + tl] == nm.rsplit(".", 1)[0].encode(), nm


def test_device_ingest_fuzz_never_parses_differently(codec, golden):
    """seeded mutations of PDB files (characters replaced anywhere, lines cut, moved, duplicated, swapped, CR, tabs and lower case,
    HETATM / ANISOU / MODEL / END / junk lines spliced in; _cases.mutated_pdb, the mutations test_ingest_vs_reference.py puts to
    the live reference): whenever the device takes a file (file_status 0) its batch, names and refusals equal the host reader's;
    everything else it hands back or reports as atom-free -- it never parses differently, and never takes a file the reader fails"""
    z, _ = golden
    bases = [_pdb_text(z, "syn:len26").splitlines(), _pdb_text(z, "pdb:multichainA").splitlines()[:300] + _pdb_text(z, "pdb:multichainB_0").splitlines()[:200]]
    rng = np.random.default_rng(int(os.environ.get("FCZ_FUZZ_SEED", "20260927")))     # (FCZ_FUZZ_SEED: the same test on other mutations)
    texts, names = [], []
    for i in range(1200):
        texts.append(mutated_pdb(bases[i % 2], rng)); names.append(f"fz{i:04d}.pdb")
    b, cfile, cmeta, fstat, refused = codec.ingest_pdb(texts, names)
    ok = [i for i in range(len(texts)) if fstat[i] == 0]
    assert len(ok) > 300, "the fuzz should leave many files in the fixed layout"
    remap = {f: k for k, f in enumerate(ok)}
    exp, exp_names, exp_file, exp_ref, failed = _host_expect([texts[i] for i in ok], [names[i] for i in ok], reader=_read_any if names[0].endswith(".cif") else None)
    assert not failed, [names[ok[k]] for k in failed]
    cnames = codec.chain_names(b.n_chains)
    got_names = [_name_of(names[f], int(m), cn) for f, m, cn in zip(cfile, cmeta, cnames)]
    assert got_names == exp_names
    assert [remap[int(f)] for f in cfile] == exp_file
    _same_batch(b, exp)
    assert sorted((remap[int(f)], _name_of(names[int(f)], int(m))) for f, m in refused) == sorted(exp_ref)
    # files reported as atom-free really have no ATOM / HETATM record the reader would keep
    for i in range(len(texts)):
        if fstat[i] == 4:
            assert len(parse_pdb_gemmi(texts[i])[0]) == 0, names[i]


def test_device_mmcif_equals_host_reader_on_plain_files(codec, golden, ing):
    """mmCIF text on the device (k_ingest_parse_cif): AFDB's own file, synthetic files with the columns in another order, CR LF
    line ends, a file without a last line end, comments and blank lines between the rows -- the batch of the host reader; text ->
    FCZ in one call equals the golden record of the reference's test.cif.gz"""
    import gzip
    from test_host_cpp import _cif_text
    z, _ = golden
    af = gzip.decompress(ing["file:test.cif.gz"].tobytes())
    syn = _cif_text(z, "pdb:test_af").encode()
    lines = syn.decode().split("\n")
    k0 = next(i for i, l in enumerate(lines) if l.startswith("ATOM"))
    texts = [af, syn, syn.replace(b"\n", b"\r\n"), syn.rstrip(b"\n"),
             "\n".join(lines[:k0 + 5] + ["# a comment between the rows", "", "   "] + lines[k0 + 5:]).encode(),
             "\n".join(["# leading comment", ""] + lines).encode(),
             _cif_text(z, "pdb:multichainA", entry_id="MULT").encode()]
    names = [f"c{i}.cif" for i in range(len(texts))]
    b, cfile, cmeta, fstat, refused = _check(codec, texts, names, reader=_read_any)
    assert list(fstat) == [0] * len(texts)
    r = codec.compress_pdb([af], ["test.cif"])
    assert (r["status"] == 0).all() and (r["file_status"] == 0).all()
    rec = r["blob"][int(r["off"][0]):int(r["off"][1])].tobytes()
    ref = ing["cif:test/fcz"].tobytes()
    assert rec == ref[:len(rec)] or _same_but_title(rec, ref)


def test_device_mmcif_numeric_field_widths(codec, golden):
    """the row readers of k_ingest_parse_cif come in two widths, chosen per step from the token bounds (every numeric field at most
    eight characters -> two dwords and a 32-bit mantissa, otherwise sixteen characters and a 64-bit one): files whose rows are all
    narrow, files with ONE wide field among narrow rows (padded zeros, fifteen digits, a nine-digit serial), and fields beyond what
    the device reads (handed back, never parsed differently) -- the taken files equal the host reader's batch bit for bit"""
    from test_host_cpp import _cif_text
    z, _ = golden
    syn = _cif_text(z, "pdb:test_af")
    lines = syn.split("\n")
    rows = [i for i, l in enumerate(lines) if l.startswith("ATOM")]

    def with_row(k, col, fn):
        out = list(lines); t = out[rows[k]].split(" "); t[col] = fn(t[col]); out[rows[k]] = " ".join(t)
        return "\n".join(out).encode()
    # columns of _cif_text's rows: 1 = id, 8 = auth_seq_id, 9..11 = Cartn_x/y/z, 13 = B
    taken = [syn.encode(),
             with_row(3, 9, lambda v: v + "000000"),                       # 12.345000000: nine decimals
             with_row(70, 10, lambda v: ("-" if v.startswith("-") else "") + "0000" + v.lstrip("-")),
             with_row(5, 11, lambda v: v.split(".")[0] + ".12345678901"),  # fifteen or fewer digits in sixteen characters
             with_row(2, 13, lambda v: v + "0000000"),
             with_row(100, 1, lambda v: "%09d" % int(v)),                   # nine digits
             with_row(0, 9, lambda v: "+" + v.lstrip("-"))]                 # a plus sign: not the reader's fast path, the host decides
    beyond = [with_row(4, 9, lambda v: v + "0" * 14),                      # seventeen characters and more
              with_row(4, 1, lambda v: "%011d" % int(v))]
    texts = taken + beyond
    names = [f"w{i}.cif" for i in range(len(texts))]
    b, cfile, cmeta, fstat, refused = codec.ingest_pdb(texts, names)
    assert list(fstat[:6]) == [0] * 6, fstat
    assert all(int(v) != 0 for v in fstat[len(taken):]), fstat
    ok = [i for i in range(len(texts)) if fstat[i] == 0]
    exp, exp_names, exp_file, exp_ref, failed = _host_expect([texts[i] for i in ok], [names[i] for i in ok], reader=_read_any)
    assert not failed
    _same_batch(b, exp)
    assert [_name_of(names[f], int(m)) for f, m in zip(cfile, cmeta)] == exp_names


def _same_but_title(a, b):
    def no_title(f):
        na, tl = f[12], int.from_bytes(f[24:28], "little")
        return f[:24] + f[28:76 + 4 * na] + f[76 + 4 * na + tl:]
    return no_title(a) == no_title(b[:len(b)])


def test_device_mmcif_fuzz_never_parses_differently(codec, golden, ing):
    """seeded mutations of mmCIF files (_cases.mutated_cif: values nulled / quoted / glued, rows moved, model / chain / alt /
    insertion columns, tags removed / doubled / re-cased, blocks, comments, text fields, reserved words, cuts, random bytes -- the
    mutations test_ingest_vs_reference.py puts to the live reference): whenever the device takes a file its batch, names and
    refusals equal the host reader's; everything else it hands back -- it never takes a file the reader fails"""
    import gzip
    from _cases import mutated_cif
    from test_host_cpp import _cif_text
    z, _ = golden
    bases = [gzip.decompress(ing["file:test.cif.gz"].tobytes()).decode("latin-1"), _cif_text(z, "syn:len26"),
             _cif_text(z, "pdb:multichainA", entry_id="M1")]
    rng = np.random.default_rng(int(os.environ.get("FCZ_FUZZ_SEED", "20260927")) + 1)
    texts, names = [], []
    for i in range(900):
        texts.append(mutated_cif(bases[i % 3], rng)); names.append(f"cz{i:04d}.cif")
    b, cfile, cmeta, fstat, refused = codec.ingest_pdb(texts, names)
    ok = [i for i in range(len(texts)) if fstat[i] == 0]
    assert len(ok) > 120, f"the fuzz should leave many files in the plain shape ({len(ok)})"
    remap = {f: k for k, f in enumerate(ok)}
    exp, exp_names, exp_file, exp_ref, failed = _host_expect([texts[i] for i in ok], [names[i] for i in ok], reader=_read_any if names[0].endswith(".cif") else None)
    assert not failed, [names[ok[k]] for k in failed]
    cnames = codec.chain_names(b.n_chains)
    got_names = [_name_of(names[f], int(m), cn) for f, m, cn in zip(cfile, cmeta, cnames)]
    assert got_names == exp_names
    assert [remap[int(f)] for f in cfile] == exp_file
    _same_batch(b, exp)
    assert sorted((remap[int(f)], _name_of(names[int(f)], int(m))) for f, m in refused) == sorted(exp_ref)
    # the PDB archive's shapes are read on the device (round 6): files with chain names of several characters, insertion codes and
    # quoted atom names are among the taken ones
    taken = [texts[i] for i in ok]
    assert any(b" BB " in t_ or b" AB1x " in t_ for t_ in taken) and any(b"\"O5'\"" in t_ for t_ in taken)


def test_device_mmcif_reads_model_ensembles(codec, ing):
    """files of several models (NMR ensembles of the PDB archive; bench.py's `models` style and longer ones): models that follow each
    other under rising plain numbers are read on the device -- the same batch, names and refusals as the host reader's, which keeps
    every model (make_structure_from_block, lib/gemmi/mmcif.hpp:560-680) --; a model that comes back, a name with a leading zero or
    a letter goes to the host reader"""
    sys.path.insert(0, ROOT)
    from bench import cif_archive_from_pdb_text
    pdbs = [ing["file:test_af.pdb"].tobytes(), ing["file:multichain.pdb"].tobytes()]
    two = cif_archive_from_pdb_text(pdbs[0], "E2", "models")
    def renumber(text, how):
        """the rows of the two-model file under other model names: how(k, model) -> name of row k"""
        out, k = [], 0
        for l in text.decode("latin-1").split("\n"):
            if l.startswith(("ATOM ", "HETATM ")):
                p = l.split(" ")
                p[-1] = how(k, p[-1]); k += 1
                l = " ".join(p)
            out.append(l)
        return "\n".join(out).encode("latin-1")
    n_rows = sum(1 for l in two.decode("latin-1").split("\n") if l.startswith("ATOM "))
    def ensemble(n_models):
        """the first model's rows n_models times, under the model numbers 1 .. n_models"""
        head, rows, tail = [], [], []
        for l in two.decode("latin-1").split("\n"):
            if l.startswith(("ATOM ", "HETATM ")):
                if l.endswith(" 1"):
                    rows.append(l)
            else:
                (tail if rows else head).append(l)
        body = [" ".join(r.split(" ")[:-1] + [str(m)]) for m in range(1, n_models + 1) for r in rows]
        return "\n".join(head + body + tail).encode("latin-1")
    texts = [two,
             renumber(two, lambda k, m: {"1": "9", "2": "10"}[m]),                       # 9 -> 10: one character more
             ensemble(5),                                                                 # five models
             renumber(two, lambda k, m: {"1": "2", "2": "1"}[m]),                        # 2 -> 1: taken by the reader's order
             renumber(two, lambda k, m: ("1", "2", "1")[k * 3 // n_rows]),               # model 1 comes back
             renumber(two, lambda k, m: {"1": "01", "2": "02"}[m]),
             renumber(two, lambda k, m: {"1": "A", "2": "B"}[m]),
             cif_archive_from_pdb_text(pdbs[1], "E3", "models")]
    names = [f"ens{i}.cif" for i in range(len(texts))]
    b, cfile, cmeta, fstat, refused = codec.ingest_pdb(texts, names)
    assert [int(v) for v in fstat[:3]] == [0, 0, 0], list(fstat)                                 # read on the device
    # (the last file -- two chains rendered under one chain name, numbers starting over -- is whatever the order rule makes of it: compared below if taken)
    assert all(int(v) != 0 for v in fstat[3:7]), list(fstat)                                     # handed back
    ok = [i for i in range(len(texts)) if fstat[i] == 0]
    remap = {f: k for k, f in enumerate(ok)}
    exp, exp_names, exp_file, exp_ref, failed = _host_expect([texts[i] for i in ok], [names[i] for i in ok], reader=_read_any)
    assert not failed
    cnames = codec.chain_names(b.n_chains)
    assert [_name_of(names[f], int(m), cn) for f, m, cn in zip(cfile, cmeta, cnames)] == exp_names
    assert [remap[int(f)] for f in cfile] == exp_file
    _same_batch(b, exp)
    assert sorted((remap[int(f)], _name_of(names[int(f)], int(m))) for f, m in refused) == sorted(exp_ref)
    # (what the host reader makes of the handed-back ones is its own affair: test_ingest_vs_reference.py holds it to the live reference)


def test_device_pdb_reads_model_ensembles(codec, ing):
    """PDB files of several models (MODEL n / atoms / ENDMDL, the NMR entries of the archive): groups under rising plain numbers are
    read on the device in file order with a new chain at every group -- the same batch, names and refusals as the host reader's
    (gemmi's read_pdb keeps the models in file order, lib/gemmi/pdb.hpp:262-365) --; models out of order or named twice, atoms
    outside a model, a MODEL record inside an open model go to the host reader, which fails some of them as the reference does"""
    def atoms_of(key):
        return [l for l in ing[key].tobytes().decode("latin-1").split("\n") if l.startswith(("ATOM", "HETATM", "TER"))]
    one, multi = atoms_of("file:test_af.pdb"), atoms_of("file:multichain.pdb")
    def pdb(*parts):
        out = ["HEADER    ENSEMBLE                                01-JAN-00   1ENS"]
        for p_ in parts:
            out += p_ if isinstance(p_, list) else [p_]
        return ("\n".join(out + ["END"]) + "\n").encode("latin-1")
    M = lambda n: "MODEL     %4d" % n
    E_ = "ENDMDL"
    texts = [pdb(M(1), one, E_, M(2), one, E_),                                       # 0  two models
             pdb(M(9), one, E_, M(10), one, E_),                                      # 1
             pdb(*[x for m in range(1, 6) for x in (M(m), one, E_)]),                 # 2  five
             pdb(M(1), multi, E_, M(2), multi, E_, M(7), multi, E_),                  # 3  chains A and B in every model
             pdb(M(1), one[:40], E_, M(2), one, E_),                                  # 4  the first model ends inside a residue
             pdb(M(2), one, E_, M(1), one, E_),                                       # 5  falling numbers
             pdb(M(1), one, E_, M(1), one, E_),                                       # 6  a number twice: the reader fails the file
             pdb(one, E_, M(1), one, E_),                                             # 7  atoms before the first MODEL
             pdb(M(1), one, M(2), one, E_),                                           # 8  MODEL inside an open model
             pdb(M(1), one, E_, one),                                                 # 9  atoms behind an ENDMDL
             pdb("MODEL 1", one, E_, "MODEL 2", one, E_),                             # 10 numbers outside their columns
             pdb(M(1), one, E_, "MODEL       2A", one, E_)]                           # 11 not a plain number
    names = [f"ens{i}.pdb" for i in range(len(texts))]
    b, cfile, cmeta, fstat, refused = codec.ingest_pdb(texts, names)
    assert [int(v) for v in fstat[:5]] == [0] * 5, list(fstat)                       # read on the device
    assert all(int(v) != 0 for v in fstat[5:]), list(fstat)                          # handed back
    ok = [i for i in range(len(texts)) if fstat[i] == 0]
    remap = {f: k for k, f in enumerate(ok)}
    exp, exp_names, exp_file, exp_ref, failed = _host_expect([texts[i] for i in ok], [names[i] for i in ok])
    assert not failed
    cnames = codec.chain_names(b.n_chains)
    assert [_name_of(names[f], int(m), cn) for f, m, cn in zip(cfile, cmeta, cnames)] == exp_names
    assert [remap[int(f)] for f in cfile] == exp_file
    _same_batch(b, exp)
    assert sorted((remap[int(f)], _name_of(names[int(f)], int(m))) for f, m in refused) == sorted(exp_ref)


def test_device_pdb_reads_anisou_records_behind_their_atoms(codec, ing):
    """ANISOU records where every file of the archive has them -- directly behind the ATOM / HETATM line of their atom -- change
    nothing the codec sees (the reader attaches the record to the atom it has just read, lib/gemmi/pdb.hpp:262-365): read on the
    device, same batch as the host reader's. A record anywhere else (before the first atom, behind a TER or another ANISOU, which
    the reader may fail the file for) goes to the host reader"""
    lines = [l for l in ing["file:test_af.pdb"].tobytes().decode("latin-1").split("\n") if l.startswith(("ATOM", "HETATM", "TER"))]
    def anis(l, u11=2406):
        return "ANISOU" + l[6:28] + "%7d%7d%7d%7d%7d%7d" % (u11, 1892, 1614, 198, 519, -328) + l[70:80].ljust(10)
    every = [x for l in lines for x in ([l, anis(l)] if l.startswith(("ATOM", "HETATM")) else [l])]
    some = [x for i, l in enumerate(lines) for x in ([l, anis(l, 0)] if l.startswith("ATOM") and i % 7 == 3 else [l])]
    first_atom = next(i for i, l in enumerate(lines) if l.startswith("ATOM"))
    ter = next((i for i, l in enumerate(lines) if l.startswith("TER")), len(lines) - 1)
    def pdb(ls, tail=()):
        return ("\n".join(["HEADER    ANISOTROPIC                             01-JAN-00   1ANI"] + list(ls) + ["END"] + list(tail)) + "\n").encode("latin-1")
    texts = [pdb(every),                                                              # 0  the archive's shape (lines and steps end anywhere)
             pdb(some),                                                               # 1
             pdb(every, tail=[anis(lines[first_atom])]),                              # 2  behind END: not read
             pdb([anis(lines[first_atom])] + lines),                                  # 3  before any atom: the reader fails the file
             pdb(lines[:first_atom + 1] + [anis(lines[first_atom])] * 2 + lines[first_atom + 1:]),      # 4  twice: fails
             pdb(lines[:ter + 1] + [anis(lines[first_atom])] + lines[ter + 1:]),      # 5  behind a TER
             pdb(lines[:5] + ["REMARK   1"] + [anis(lines[4])] + lines[5:])]          # 6  a record between the atom and its ANISOU
    names = [f"ani{i}.pdb" for i in range(len(texts))]
    b, cfile, cmeta, fstat, refused = codec.ingest_pdb(texts, names)
    assert [int(v) for v in fstat[:3]] == [0] * 3, list(fstat)                       # read on the device
    assert all(int(v) != 0 for v in fstat[3:]), list(fstat)                          # handed back
    exp, exp_names, exp_file, exp_ref, failed = _host_expect(texts[:3], names[:3])
    assert not failed and not exp_ref and not len(refused)
    _same_batch(b, exp)
    assert [_name_of(names[f], int(m)) for f, m in zip(cfile, cmeta)] == exp_names and list(cfile) == exp_file
    # the host reader on two of the others: it fails the records that find no fresh atom
    for i in (3, 4):
        with pytest.raises(StructureError):
            parse_pdb_gemmi(texts[i])


def test_device_readers_keep_the_sign_of_a_zero(codec, golden, ing):
    """"-0.000" is a coordinate real files hold; the reference's reader (fast_float / strtod) keeps the sign and the codec carries it
    through anchors into the decoded atoms. PDB columns and mmCIF fields with -0.000 / -0.0 / -0 / 0.000 and a B-factor of -0.00:
    the device's batch equals the host reader's bit for bit (_same_batch compares bit patterns)"""
    from test_host_cpp import _cif_text
    z, _ = golden
    pdb_lines = ing["file:test_af.pdb"].tobytes().decode("latin-1").split("\n")
    out = []
    k = 0
    for l in pdb_lines:
        if l.startswith("ATOM") and len(l) >= 66:
            k += 1
            if k % 9 == 1: l = l[:30] + "  -0.000" + l[38:]
            if k % 9 == 4: l = l[:38] + "  -0.000" + l[46:54] + l[54:]
            if k % 9 == 7: l = l[:46] + "   0.000" + l[54:]
            if k % 13 == 2: l = l[:60] + " -0.00" + l[66:]
        out.append(l)
    pdb = "\n".join(out).encode("latin-1")
    cif_lines = _cif_text(z, "pdb:test_af").split("\n")
    k0 = next(i for i, l in enumerate(cif_lines) if l.startswith("ATOM"))
    head = [l for l in cif_lines[:k0] if l.startswith("_atom_site.")]
    cx, cb = head.index("_atom_site.Cartn_x"), head.index("_atom_site.B_iso_or_equiv")
    for i in range(k0, len(cif_lines)):
        if not cif_lines[i].startswith("ATOM"):
            continue
        t = cif_lines[i].split()
        j = i - k0
        if j % 7 == 0: t[cx] = "-0.000"
        if j % 7 == 2: t[cx + 1] = "-0.0"
        if j % 7 == 4: t[cx + 2] = "-0"
        if j % 7 == 5: t[cx] = "0.000"
        if j % 11 == 3: t[cb] = "-0.00"
        cif_lines[i] = " ".join(t)
    cif = "\n".join(cif_lines).encode()
    b, cfile, cmeta, fstat, refused = _check(codec, [pdb, cif], ["z.pdb", "z.cif"], reader=_read_any)
    assert list(fstat) == [0, 0]
    sign = lambda v: int((v.view(np.uint32) == 0x80000000).sum())
    assert sign(b.x) > 20 and sign(b.y) > 10 and sign(b.z) > 5              # (the zeros with a sign are there)
