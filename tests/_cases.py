"""Shared helpers: golden cases -> ChainBatch, seeded synthetic batches."""
import numpy as np

from foldcomp_amd.structure import ChainBatch

_KEYS = ("res_off", "atom_off", "x", "y", "z", "atom_code", "res_code", "bfac_ca", "first_res_index", "first_atom_index",
         "chain_id", "titles", "title_off")


def golden_batch(z, name) -> ChainBatch:
    kw = {k: np.ascontiguousarray(z[f"{name}/in/{k}"]) for k in _KEYS}
    return ChainBatch(anchor_threshold=int(z[f"{name}/in/anchor_threshold"][0]), **kw)


def compress_cases(index):
    return [n for n in index if n.startswith("pdb:") or n.startswith("syn:")]


def db_cases(index):
    return [n for n in index if n.startswith("db:")]


def concat_batches(batches):
    """several ChainBatch -> one (same anchor threshold)"""
    res_off = [0]; atom_off = []; abase = 0; toff = [0]
    cat = {k: [] for k in ("x", "y", "z", "atom_code", "res_code", "bfac_ca", "first_res_index", "first_atom_index", "chain_id", "titles")}
    for b in batches:
        res_off += list(res_off[-1] + b.res_off[1:].astype(np.int64))
        atom_off.append(b.atom_off[:-1].astype(np.int64) + abase); abase += int(b.atom_off[-1])
        toff += list(toff[-1] + b.title_off[1:].astype(np.int64))
        for k in cat: cat[k].append(getattr(b, k))
    atom_off.append(np.asarray([abase]))
    return ChainBatch(res_off=np.asarray(res_off, np.uint32), atom_off=np.concatenate(atom_off).astype(np.uint32),
                      title_off=np.asarray(toff, np.uint32), anchor_threshold=batches[0].anchor_threshold,
                      **{k: np.concatenate(v) for k, v in cat.items()})


def entries_blob(entries):
    """list of bytes -> (blob, off)"""
    off = np.zeros(len(entries) + 1, np.uint64)
    off[1:] = np.cumsum([len(e) for e in entries])
    blob = np.frombuffer(b"".join(entries), np.uint8).copy() if entries else np.zeros(0, np.uint8)
    return blob, off


# records a mutated PDB file may get spliced in (what gemmi's reader reacts to beyond ATOM lines)
_FOREIGN = ["ANISOU    2  CA  MET A   1     2406   1892   1614    198    519   -328       C  ",
            "ANISOU    3  C   MET A   1        0   1892   1614    198    519   -328       C  ", "MODEL        1", "MODEL        2", "ENDMDL", "data_x",
            "REMARK 465 junk", "HETATM 9001  O   HOH A 900      11.000  12.000  13.000  1.00 30.00           O  ", "TITLE     SOMETHING", "TITLE    2 MORE OF IT  ",
            "HEADER    HYDROLASE                               01-JAN-00   1ABC              ", "junk", "", "END", "TER",
            "atom      1  N   MET A   1      27.340  24.430   2.614  1.00  9.67           N",
            "CRYST1   52.000   58.600   61.900  90.00   0.00  90.00 P 21 21 21    8", "CRYST1   52.000   58.600   61.900  90.00  90.00   0.00 P 1",
            "CRYST1    1.000    1.000    1.000   0.50  90.00  90.00 P 1", "CRYST1    1.000    1.000    1.000  90.00 120.00  90.00 P 1", '{"data_x": 1}',
            "SSBOND   1 CYS A    6    CYS A  127                          1555   1555  2.03", "SSBOND   x CYS Z  999    CYS", "LINK         O   GLY A  49                CA    CA A 501     1555   1555  2.65",
            "LINK  junk", "CISPEP   1 SER A   58    PRO A   59          0         0.30", "CISPEP junk junk", "HELIX    1   1 ALA A    2  GLY A   10  1                                   9",
            "HELIX  x", "SHEET    1   A 2 THR A   4  VAL A   8  0", "SHEET zz", "SEQRES   1 A   26  MET ALA GLY", "SEQRES junk", "DBREF  1ABC A    1    26  UNP    P12345   X_HUMAN          1     26",
            "MTRIX1   1  1.000000  0.000000  0.000000        0.00000    1", "MTRIX2   1  x", "SCALE1      0.019231  0.000000  0.000000        0.00000", "SCALE2 junk",
            "ORIGX1      1.000000  0.000000  0.000000        0.00000", "REMARK 350   BIOMT1   1  1.000000  0.000000  0.000000        0.00000", "REMARK 350 APPLY THE FOLLOWING TO CHAINS: A, B",
            "REMARK   2 RESOLUTION.    1.74 ANGSTROMS.", "REMARK   3   R VALUE            (WORKING SET) : 0.18", "CONECT  413  412  414", "COMPND    MOL_ID: 1;", "EXPDTA    X-RAY DIFFRACTION",
            "NUMMDL    2", "MASTER      351    0    0    4    8    0    0    6 1215    1    0   11", "HETNAM     HOH WATER", "MODRES 1ABC MSE A    1  MET  SELENOMETHIONINE", "TER     216      PRO A  26",
            "SIGATM    1  N   MET A   1       0.010   0.010   0.010  0.00  0.00           N", "JRNL        AUTH   A.B.C", "KEYWDS    X",
            "MODEL        3", "MODEL       10", "ENDMDL", "MODEL 4"]


def mutated_pdb(base_lines, rng, max_edits=5):
    """one seeded mutation of a PDB file: characters replaced anywhere, lines cut, duplicated elsewhere, swapped, removed, given a
    CR, chain ids and residue numbers changed from some line on, foreign records spliced in -- what tests of the readers feed them"""
    alphabet = list("0123456789 .-+ANCOHETMabcxyz\t*")
    lines = list(base_lines)
    for _ in range(int(rng.integers(1, max_edits + 1))):
        kind = int(rng.integers(0, 10)); j = int(rng.integers(0, len(lines))); l = lines[j]
        if kind == 0 and l:
            k = int(rng.integers(0, len(l))); lines[j] = l[:k] + alphabet[int(rng.integers(0, len(alphabet)))] + l[k + 1:]
        elif kind == 1:
            lines[j] = l[:int(rng.integers(0, len(l) + 1))]
        elif kind == 2:
            lines.insert(int(rng.integers(0, len(lines))), l)
        elif kind == 3 and j + 1 < len(lines):
            lines[j], lines[j + 1] = lines[j + 1], lines[j]
        elif kind == 4:
            ch = "BCD"[int(rng.integers(0, 3))]
            lines[j:] = [x[:21] + ch + x[22:] if x.startswith("ATOM") and len(x) > 22 else x for x in lines[j:]]
        elif kind == 5:
            lines[j:] = [x[:22] + "%4d" % (int(x[22:26]) + 3) + x[26:] if x.startswith("ATOM") and x[22:26].strip().lstrip("-").isdigit() else x for x in lines[j:]]
        elif kind == 6:
            lines.insert(j, _FOREIGN[int(rng.integers(0, len(_FOREIGN)))])
        elif kind == 7:
            lines[j] = l + "\r"
        elif kind == 9:                                     # the charge columns (79-80): the reader fails a digit beside a non-sign
            lines[j] = l[:78].ljust(78) + ["1+", "2-", "1A", "A1", "+1", "9 ", " 9", "x5", "5", "7\t"][int(rng.integers(0, 10))]
        else:
            del lines[j]
    return ("\n".join(lines) + ("\n" if rng.integers(0, 4) else "")).encode("latin-1")


def reference_would_spin(t) -> bool:
    """identifyChains of the reference (src/atom_coordinate.cpp:469-497) never returns when a chain id changes at an atom that is
    not an N and no N follows: such inputs cannot be put to the live reference"""
    n = len(t); i = 1
    while i < n:
        if t.chain[i] != t.chain[i - 1] and t.atom[i] != "N":
            j = next((j for j in range(i, n) if t.atom[j] == "N"), None)
            if j is None:
                return True
            i = j
        i += 1
    return False


# ---- coordinates no structure has: what the angle code makes of zero-length bonds, straight lines, coinciding atoms ----------
def _first_atom(b, chain, res):
    return int(b.atom_off[int(b.res_off[chain]) + res])


def degenerate_cases():
    """[(name, mutate(batch, x, y, z))]: in-place edits of a synthetic batch's coordinates. NaN angles (a zero-length bond gives
    0 / 0 in getCosineTheta), the NaN guard of the dihedrals, min / max over arrays that hold NaN (std::min_element keeps what it
    holds when a comparison with NaN is false), overflow to infinity, every value equal (a quantiser with max == min)."""
    def ca_on_n(res):
        def f(b, x, y, z):
            for ch in range(b.n_chains):
                a = _first_atom(b, ch, res); x[a + 1], y[a + 1], z[a + 1] = x[a], y[a], z[a]
        return f

    def c_on_ca(b, x, y, z):
        for ch in range(b.n_chains):
            a = _first_atom(b, ch, 0); x[a + 2], y[a + 2], z[a + 2] = x[a + 1], y[a + 1], z[a + 1]

    def straight(b, x, y, z):
        for ch in range(b.n_chains):
            a = _first_atom(b, ch, 7)
            x[a + 1], y[a + 1], z[a + 1] = x[a] + 1.0, y[a], z[a]
            x[a + 2], y[a + 2], z[a + 2] = x[a] + 2.0, y[a], z[a]

    def residue_at_origin(b, x, y, z):
        for ch in range(b.n_chains):
            a0, a1 = _first_atom(b, ch, 3), _first_atom(b, ch, 4)
            x[a0:a1] = 0; y[a0:a1] = 0; z[a0:a1] = 0

    def far_away(b, x, y, z):
        for ch in range(b.n_chains):
            x[_first_atom(b, ch, 9) + 1] = 1e30

    def noise(b, x, y, z):
        rng = np.random.default_rng(1)
        for v in (x, y, z):
            v[:] = rng.normal(0, 10, len(v)).astype(np.float32).round(3)

    def lattice(b, x, y, z):
        rng = np.random.default_rng(2)
        for v in (x, y, z):
            v[:] = rng.integers(-3, 4, len(v)).astype(np.float32)

    def one_point(b, x, y, z):
        x[:] = 1.0; y[:] = 2.0; z[:] = 3.0

    def neg_zero(b, x, y, z):
        # "-0.000" is a coordinate PDB files hold (strtod keeps the sign): anchors, ordinary atoms, whole residues in a plane
        x[::5] = -0.0; y[::7] = -0.0; z[::11] = -0.0
        for ch in range(b.n_chains):
            n = int(b.res_off[ch + 1] - b.res_off[ch])
            for r in range(0, n, 25):
                a = _first_atom(b, ch, r); x[a] = -0.0; y[a + 1] = -0.0; z[a + 2] = -0.0

    def n1_on_c0(b, x, y, z):
        for ch in range(b.n_chains):
            a, a1 = _first_atom(b, ch, 0), _first_atom(b, ch, 1); x[a1], y[a1], z[a1] = x[a + 2], y[a + 2], z[a + 2]

    def coincidences(seed):
        def f(b, x, y, z):
            rng = np.random.default_rng(seed)
            for ch in range(b.n_chains):
                n = int(b.res_off[ch + 1] - b.res_off[ch])
                for _ in range(int(rng.integers(1, 6))):
                    r = int(rng.integers(0, n)) if rng.random() < 0.6 else (0 if rng.random() < 0.5 else n - 1)
                    a = _first_atom(b, ch, r); na = _first_atom(b, ch, r + 1) - a if r + 1 < n else 4
                    i, j = int(rng.integers(0, min(na, 5))), int(rng.integers(0, min(na, 5)))
                    x[a + i], y[a + i], z[a + i] = x[a + j], y[a + j], z[a + j]
        return f

    return [("N of residue 1 on C of residue 0", n1_on_c0)] + [(f"random coincidences {k}", coincidences(k)) for k in range(6)] + [("CA on N, residue 5", ca_on_n(5)), ("CA on N, first residue", ca_on_n(0)), ("C on CA, first residue", c_on_ca),
            ("three atoms in a line", straight), ("a residue at the origin", residue_at_origin), ("an atom 1e30 away", far_away),
            ("noise", noise), ("integer lattice", lattice), ("every atom at one point", one_point), ("coordinates of -0.0", neg_zero)]


def degenerate_batch(mutate, lens=(40, 350, 90), seed=3):
    from foldcomp_amd import synthetic
    b = synthetic.to_chain_batch(synthetic.generate(len(lens), list(lens), seed=seed))
    x, y, z = b.x.copy(), b.y.copy(), b.z.copy()
    mutate(b, x, y, z)
    b.x, b.y, b.z = x, y, z
    return b


def distorted_batch(n_chains, sigma, seed, lo=16, hi=700):
    """chains of the generator with every atom moved by N(0, sigma) in x, y and z and written at PDB precision (three decimals):
    bond lengths and angles as far from ideal as refined experimental models (sigma 0.02-0.05), poor ones (0.3) and wrecks (1.5)
    have them -- the generator alone has ideal bond lengths. Lengths log-uniform in [lo, hi]"""
    from foldcomp_amd import synthetic
    rng = np.random.default_rng(seed)
    lens = np.exp(rng.uniform(np.log(lo), np.log(hi), n_chains)).astype(np.int64)
    b = synthetic.to_chain_batch(synthetic.generate(n_chains, lens, seed=seed))
    for k in ("x", "y", "z"):
        v = getattr(b, k).astype(np.float64) + rng.normal(0.0, sigma, len(getattr(b, k)))
        setattr(b, k, (np.round(v * 1000.0) / 1000.0).astype(np.float32))
    return b


# ---- input variants for the differential fuzz -------------------------------------------------------------------------------------
def _variant_base(rng, n, lo=16, hi=600):
    from foldcomp_amd import synthetic
    lens = np.exp(rng.uniform(np.log(lo), np.log(hi), n)).astype(np.int64)
    return synthetic.to_chain_batch(synthetic.generate(n, lens, seed=int(rng.integers(1, 1 << 30))))


def _variant_xyz(b, f):
    for k in ("x", "y", "z"):
        setattr(b, k, np.ascontiguousarray(f(getattr(b, k).astype(np.float64), k), dtype=np.float32))
    return b


def _variant_atoms(b, rng, mode):
    """a batch with its side chains edited residue by residue (N, CA, C stay where they are): mode "drop" removes each side-chain
    atom with probability 0.15 (and whole side chains with 0.03), "shuffle" permutes the atoms behind the backbone, "extra" inserts
    up to three unnamed atoms (code 255: hydrogens, ligand atoms) anywhere behind the backbone"""
    x, y, z, code, aoff = [], [], [], [], [0]
    for r in range(b.n_residues):
        s, e = int(b.atom_off[r]), int(b.atom_off[r + 1])
        idx = np.arange(s, e)
        head, tail = idx[:3], idx[3:]
        if mode == "drop":
            tail = tail[rng.random(len(tail)) >= 0.15] if rng.random() >= 0.03 else tail[:0]
        elif mode == "shuffle":
            tail = rng.permutation(tail)
        idx = np.concatenate((head, tail))
        xs, ys, zs, cs = b.x[idx], b.y[idx], b.z[idx], b.atom_code[idx]
        if mode == "extra":
            for _ in range(int(rng.integers(0, 4))):
                at = int(rng.integers(3, len(xs) + 1))
                p3 = np.round((np.asarray([xs[1], ys[1], zs[1]], np.float64) + rng.normal(0, 2, 3)) * 1000.0) / 1000.0
                xs = np.insert(xs, at, np.float32(p3[0])); ys = np.insert(ys, at, np.float32(p3[1])); zs = np.insert(zs, at, np.float32(p3[2])); cs = np.insert(cs, at, np.uint8(255))
        x.append(xs); y.append(ys); z.append(zs); code.append(cs); aoff.append(aoff[-1] + len(xs))
    return ChainBatch(res_off=b.res_off, atom_off=np.asarray(aoff, np.uint32), x=np.concatenate(x).astype(np.float32), y=np.concatenate(y).astype(np.float32),
                      z=np.concatenate(z).astype(np.float32), atom_code=np.concatenate(code).astype(np.uint8), res_code=b.res_code, bfac_ca=b.bfac_ca,
                      first_res_index=b.first_res_index, first_atom_index=b.first_atom_index, chain_id=b.chain_id,
                      titles=b.titles, title_off=b.title_off, anchor_threshold=b.anchor_threshold)


def input_variants(rng, n):
    """(name, batch) pairs: chains of the generator put through what real inputs have and the generator does not -- distortions at PDB
    precision and as raw floats, translations to the edge of the PDB columns and beyond, scalings, reflections, coarse precision,
    signed zeros and denormals, odd B-factors, every short length, one residue type per batch, numbering and chain ids. Used by
    tests/test_gpu_parity_fuzz.py (GPU == restatement) and tools/dbg/parity_fuzz.py (the same at any size and seed)"""
    from foldcomp_amd import synthetic
    base = lambda rng_: _variant_base(rng_, n)
    xyz = _variant_xyz
    r3 = lambda v: np.round(v * 1000.0) / 1000.0
    yield "plain", base(rng)
    for s in (0.01, 0.05, 0.2, 0.6, 3.0, 10.0):
        yield f"noise {s} (3 decimals)", xyz(base(rng), lambda v, k: r3(v + rng.normal(0, s, len(v))))
    for s in (0.001, 0.05, 0.5):
        yield f"noise {s} (raw float)", xyz(base(rng), lambda v, k: v + rng.normal(0, s, len(v)))
    for t in (100.0, 999.0, 5000.0, 9000.0, -9999.0, 1e5, 1e7):
        yield f"moved by {t}", xyz(base(rng), lambda v, k: r3(v + t))
    yield "moved by (1000, -2000, 3000) + noise", xyz(base(rng), lambda v, k: r3(v + {"x": 1000.0, "y": -2000.0, "z": 3000.0}[k] + rng.normal(0, 0.1, len(v))))
    for sc in (0.5, 0.9, 1.1, 2.0, 1e-3, 1e3):
        yield f"scaled by {sc}", xyz(base(rng), lambda v, k: r3(v * sc) if sc >= 0.5 else v * sc)
    # nearly collinear bonds: what a reader makes of columns that ran together ("-1999.7482999.922": z = 82999.92 for some atoms,
    # 3000.03 for others) -- cosines that round beyond +-1, acos domain errors, NaN bond angles at the head of a chain
    yield "z of half the atoms + 80 000", xyz(_variant_base(rng, 6 * n, 16, 40), lambda v, k: r3(v + (np.where(rng.random(len(v)) < 0.5, 80000.0, 0.0) if k == "z" else 0.0)))
    yield "x of a tenth of the atoms + 1e6", xyz(base(rng), lambda v, k: r3(v + (np.where(rng.random(len(v)) < 0.1, 1e6, 0.0) if k == "x" else 0.0)))
    yield "atoms on a line + noise 1e-3", xyz(base(rng), lambda v, k: r3(np.arange(len(v)) * {"x": 1.5, "y": 0.0, "z": 0.0}[k] + rng.normal(0, 1e-3, len(v))))
    yield "mirrored", xyz(base(rng), lambda v, k: -v if k == "x" else v)
    yield "two decimals", xyz(base(rng), lambda v, k: np.round(v * 100.0) / 100.0)
    yield "one decimal", xyz(base(rng), lambda v, k: np.round(v * 10.0) / 10.0)
    yield "integers", xyz(base(rng), lambda v, k: np.round(v))
    yield "-0.0 sprinkled + noise", xyz(base(rng), lambda v, k: np.where(rng.random(len(v)) < 0.05, -0.0, r3(v + rng.normal(0, 0.05, len(v)))))
    yield "zeros sprinkled + noise", xyz(base(rng), lambda v, k: np.where(rng.random(len(v)) < 0.05, 0.0, r3(v + rng.normal(0, 0.05, len(v)))))
    yield "one coordinate plane (z = -0.0)", xyz(base(rng), lambda v, k: np.full_like(v, -0.0) if k == "z" else v)
    yield "one coordinate plane (y = 0)", xyz(base(rng), lambda v, k: np.zeros_like(v) if k == "y" else v)
    yield "-0.0 sprinkled", xyz(base(rng), lambda v, k: np.where(rng.random(len(v)) < 0.02, -0.0, v))
    yield "tiny values sprinkled", xyz(base(rng), lambda v, k: np.where(rng.random(len(v)) < 0.02, rng.choice([1e-38, -1e-38, 1e-45, 1e-30, -1e-20, 1e-10], len(v)), v))
    # B-factors
    for name, f in (("B constant", lambda v: np.full_like(v, 50.0)), ("B zero", lambda v: np.zeros_like(v)), ("B negative", lambda v: -v), ("B huge", lambda v: v * 1e4),
                    ("B two values", lambda v: np.where(rng.random(len(v)) < 0.5, 10.0, 90.0).astype(np.float32)), ("B with -0.0", lambda v: np.where(rng.random(len(v)) < 0.1, -0.0, v).astype(np.float32)),
                    ("B raw floats", lambda v: rng.normal(50, 30, len(v)).astype(np.float32)),
                    ("B at the ends of the float range", lambda v: rng.choice(np.asarray([3e38, -3e38, 1e38, 0.0, 1e-38, 50.0], np.float32), len(v))),
                    ("B tiny", lambda v: (v * 1e-40).astype(np.float32))):
        b = base(rng); b.bfac_ca = np.ascontiguousarray(f(b.bfac_ca.copy()), dtype=np.float32); yield name, b
    # short chains, one length each
    for k in (1, 2, 3, 4, 5, 8, 15, 16, 17, 24, 25, 26, 27, 49, 50, 51, 52, 63, 64, 65, 75, 76, 77):
        yield f"chains of {k}", synthetic.to_chain_batch(synthetic.generate(min(n, 64), [k] * min(n, 64), seed=int(rng.integers(1, 1 << 30))))
    # all of one residue type
    for rc in range(24):                                   # (20-22: ASX, GLX, STP are refused by both; 23 = UNK)
        yield f"all residues of type {rc}", xyz(synthetic.to_chain_batch(synthetic.generate(min(n, 48), [int(v) for v in rng.integers(20, 200, min(n, 48))], seed=int(rng.integers(1, 1 << 30)), res_code=rc)),
                                                 lambda v, k: r3(v + rng.normal(0, 0.05, len(v))))
    # chains long enough for the segment-parallel decode (1 024 residues and more), alone and among short ones
    m = min(n, 24)
    yield "long chains + noise + -0.0", xyz(_variant_base(rng, m, 1024, 3000), lambda v, k: np.where(rng.random(len(v)) < 0.02, -0.0, r3(v + rng.normal(0, 0.05, len(v)))))
    lens = [int(v) for v in np.where(rng.random(m) < 0.3, rng.integers(1024, 2600, m), rng.integers(2, 300, m))]
    yield "long and short chains in one batch", xyz(synthetic.to_chain_batch(synthetic.generate(m, lens, seed=int(rng.integers(1, 1 << 30)))), lambda v, k: r3(v + rng.normal(0, 0.1, len(v))))
    # side chains as deposited models have them: atoms missing, in another order, hydrogens and other unnamed atoms among them
    for mode in ("drop", "shuffle", "extra"):
        yield f"side chains: {mode}", _variant_atoms(xyz(_variant_base(rng, min(n, 48), 16, 300), lambda v, k: r3(v + rng.normal(0, 0.05, len(v)))), rng, mode)
    # first / chain numbering
    b = base(rng); b.first_res_index = rng.integers(-500, 9000, b.n_chains).astype(b.first_res_index.dtype); b.first_atom_index = rng.integers(0, 90000, b.n_chains).astype(b.first_atom_index.dtype); yield "numbering", b
    b = base(rng); b.chain_id = rng.integers(32, 127, b.n_chains).astype(np.uint8); yield "chain ids", b



def composite_pdb(rng, b, title="COMPOSITE") -> bytes:
    """a PDB file as depositions look, made of chains of a batch: two to four chains under different names, TER records, residues
    missing in the middle (the reference splits the chain into fragments), alternative locations (A kept, B dropped), insertion
    codes, numbering from below zero, waters and a ligand as HETATM behind the chains, CRLF line ends now and then"""
    import sys as _sys, os as _os
    _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "oracle"))
    import host_text
    res_of_atom = np.repeat(np.arange(b.n_residues), np.diff(b.atom_off.astype(np.int64)))
    short = [c for c in range(b.n_chains) if 4 <= int(b.res_off[c + 1] - b.res_off[c]) <= 160] or [0]
    out = ["HEADER    TEST                                    01-JAN-00   1CMP", "TITLE     " + title]
    serial = 1
    for ci, c in enumerate(rng.choice(short, size=min(len(short), int(rng.integers(2, 5))), replace=False)):
        r0, r1 = int(b.res_off[c]), int(b.res_off[c + 1])
        a0, a1 = int(b.atom_off[r0]), int(b.atom_off[r1]); sl = slice(a0, a1)
        first = int(rng.choice([1, 1, -7, 95, 990]))
        lines = host_text.format_pdb("", b.atom_code[sl], b.res_code[res_of_atom[sl]], first + res_of_atom[sl] - r0, "ABCDEFG"[ci], 1,
                                     b.x[sl], b.y[sl], b.z[sl], b.bfac_ca[res_of_atom[sl]]).split("\n")
        lines = [l for l in lines if l.startswith("ATOM")]
        n = r1 - r0
        mode = int(rng.integers(0, 5))
        if mode == 1 and n > 12:                                                # residues missing in the middle
            g0 = int(rng.integers(3, n - 6)); g1 = g0 + int(rng.integers(1, 4))
            lines = [l for l in lines if not (first + g0 <= int(l[22:26]) < first + g1)]
        elif mode == 2:                                                         # alternative locations on some side-chain atoms
            new = []
            for l in lines:
                if l[12:16].strip() not in ("N", "CA", "C", "O") and rng.random() < 0.15:
                    new.append(l[:16] + "A" + l[17:]); new.append(l[:16] + "B" + l[17:30] + "%8.3f" % (float(l[30:38]) + 0.5) + l[38:])
                else:
                    new.append(l)
            lines = new
        elif mode == 3 and n > 8:                                               # insertion codes: 7, 7A, 7B, 8 ...
            k = first + int(rng.integers(2, n - 4))
            def renum(l):
                q = int(l[22:26])
                if q == k + 1: return l[:22] + "%4d" % k + "A" + l[27:]
                if q == k + 2: return l[:22] + "%4d" % k + "B" + l[27:]
                if q > k + 2: return l[:22] + "%4d" % (q - 2) + l[26:]
                return l
            lines = [renum(l) for l in lines]
        for l in lines:
            out.append(l[:6] + "%5d" % serial + l[11:]); serial += 1
        last = lines[-1]
        out.append("TER   %5d      %s %s%s" % (serial, last[17:20], last[21], last[22:27])); serial += 1
    wch = last[21] if rng.random() < 0.8 else "W"           # (a chain of their own behind the last chain: the reference's identifyChains never returns)
    for w in range(int(rng.integers(0, 6))):
        out.append("HETATM%5d  O   HOH %s%4d    %8.3f%8.3f%8.3f  1.00 30.00           O  " % (serial, wch, 2000 + w, *rng.normal(0, 20, 3))); serial += 1
    out.append("END")
    eol = "\r\n" if rng.random() < 0.1 else "\n"
    return (eol.join(out) + eol).encode("latin-1")


# ---- FCZ records with random payloads (the header, the residue codes and the anchors stay) ----------------------------------
def payload_mutations(records, per_record=10, seed=20260927, temp_params=False):
    """random angle words (every one / a tenth / all bits set or clear / one residue / bond-angle bytes only), side-chain torsion
    bytes and B-factor bytes; temp_params: also the B-factor quantiser's two floats (what `extract` prints from)"""
    from foldcomp_amd import fczfile
    rng = np.random.default_rng(seed)
    out = []
    for e in records:
        r = fczfile.parse(e)
        for v in range(per_record):
            b = bytearray(e)
            w = np.frombuffer(e, np.uint8, 8 * r.n_residues, r.o_words).reshape(-1, 8).copy()
            rnd = rng.integers(0, 256, w.shape, dtype=np.uint8)
            how = v % 5
            if how == 0:
                sel = np.ones(len(w), bool)
            elif how == 1:
                sel = rng.random(len(w)) < 0.1
            elif how == 2:
                sel = np.ones(len(w), bool); rnd[:] = np.where(rng.random(w.shape) < 0.5, 0, 255).astype(np.uint8)
            elif how == 3:
                sel = np.zeros(len(w), bool); sel[int(rng.integers(0, len(w)))] = True
            else:
                sel = np.ones(len(w), bool); rnd[:, :5] = w[:, :5]
            rnd[:, 0] = (w[:, 0] & 0xf8) | (rnd[:, 0] & 0x07)         # the residue code stays (it decides the atom counts)
            w[sel] = rnd[sel]
            b[r.o_words:r.o_words + 8 * r.n_residues] = w.tobytes()
            if v >= per_record // 2:
                b[r.o_sc:r.o_sc + r.n_sidechain] = rng.integers(0, 256, r.n_sidechain, dtype=np.uint8).tobytes()
                o_t = r.o_sc + r.n_sidechain + 8
                b[o_t:o_t + r.n_residues] = rng.integers(0, 256, r.n_residues, dtype=np.uint8).tobytes()
            if temp_params and v % 3 == 0:
                mn = np.float32(rng.normal(0, 1) * 10.0 ** int(rng.integers(-2, 4)))
                cf = np.float32(abs(rng.normal(0, 1)) * 10.0 ** int(rng.integers(-3, 2)))
                b[r.o_tmp:r.o_tmp + 8] = mn.tobytes() + cf.tobytes()
            out.append(bytes(b))
    return out


def golden_records(golden):
    """the FCZ records of the goldens, cut to their own size (the database entries end in a NUL)"""
    from foldcomp_amd import fczfile
    z, index = golden
    recs = []
    for n in compress_cases(index) + db_cases(index):
        e = z[f"{n}/fcz"].tobytes()
        recs.append(e[:fczfile.record_size(e)])
    return recs


# ---- mutated mmCIF files --------------------------------------------------------------------------------------------------
def mutated_cif(text: str, rng, max_edits=4) -> bytes:
    """one seeded mutation of an mmCIF file: values of _atom_site rows replaced (null, quoted, lower case, letters glued to numbers,
    uncertainties, junk), rows moved / doubled / dropped / swapped, model / chain / alt / insertion-code columns changed from some
    row on, and the file around the loop disturbed (tags removed, doubled or re-cased, blocks, comments, text fields, stray reserved
    words, a cut, random bytes)"""
    lines = text.split("\n")
    cols = [l.strip() for l in lines if l.startswith("_atom_site.")]
    rows = [i for i, l in enumerate(lines) if l.startswith(("ATOM", "HETATM"))]
    col = {c.split(".", 1)[1]: k for k, c in enumerate(cols)}

    def set_tok(i, name, val):
        p = lines[i].split()
        if name in col and col[name] < len(p):
            p[col[name]] = val; lines[i] = " ".join(p)

    for _ in range(int(rng.integers(1, max_edits + 1))):
        kind = int(rng.integers(0, 16))
        rows = [i for i, l in enumerate(lines) if l.startswith(("ATOM", "HETATM", "atom", "hetatm"))]
        if not rows:
            break
        i = rows[int(rng.integers(0, len(rows)))]
        if kind == 0:                                       # one value of one row
            name = list(col)[int(rng.integers(0, len(col)))]
            p = lines[i].split()
            old = p[col[name]] if col[name] < len(p) else "x"
            val = ["?", ".", "'" + old + "'", '"' + old + '"', old.lower(), old + "A", old + "(3)", "+" + old, "-" + old, "1e2", "x y", "''", old + "'",
                   ";", "#c", "_t", "$v", "loop_", "1.5.2", "", "0x10", " ".join([old, old])][int(rng.integers(0, 22))]
            set_tok(i, name, val)
        elif kind == 1:                                     # a row moved
            l = lines.pop(i); lines.insert(rows[int(rng.integers(0, len(rows)))], l)
        elif kind == 2:
            lines.insert(i, lines[i])
        elif kind == 3:
            del lines[i]
        elif kind == 4 and i + 1 < len(lines):
            lines[i], lines[i + 1] = lines[i + 1], lines[i]
        elif kind == 5:                                     # a column changed from this row on
            name, val = [("pdbx_PDB_model_num", "2"), ("auth_asym_id", "B"), ("label_asym_id", "B"), ("label_alt_id", "A"), ("pdbx_PDB_ins_code", "A"),
                         ("auth_comp_id", "ALA"), ("label_comp_id", "GLY"), ("auth_atom_id", "CA"), ("B_iso_or_equiv", "?"), ("occupancy", "."),
                         ("pdbx_PDB_model_num", "1"), ("auth_seq_id", "7"), ("label_seq_id", "x"),
                         # (round 6, the PDB archive's shapes: chain names of several characters, lower-case insertion codes, atom names in quotes)
                         ("auth_asym_id", "BB"), ("auth_asym_id", "AB1x"), ("auth_asym_id", "ABCDE"), ("pdbx_PDB_ins_code", "b"),
                         ("auth_atom_id", "\"O5'\""), ("label_atom_id", "\"O5'\""),
                         # (ensembles: models numbered upwards are read where they lie; a model that comes back, or is not a plain number, is sorted by the reader)
                         ("pdbx_PDB_model_num", "3"), ("pdbx_PDB_model_num", "10"), ("pdbx_PDB_model_num", "02"), ("pdbx_PDB_model_num", "A"),
                         ("pdbx_PDB_model_num", "0")][int(rng.integers(0, 24))]
            stop = rows.index(i) + int(rng.integers(1, 40))
            for j in rows[rows.index(i):stop]:
                set_tok(j, name, val)
        elif kind == 6:                                     # residue numbers shifted from here on (both columns)
            for j in rows[rows.index(i):]:
                p = lines[j].split()
                for name in ("auth_seq_id", "label_seq_id"):
                    if name in col and col[name] < len(p) and p[col[name]].lstrip("-").isdigit():
                        p[col[name]] = str(int(p[col[name]]) + 3)
                lines[j] = " ".join(p)
        elif kind == 7:                                     # a tag line removed, doubled or re-cased
            tl = [k for k, l in enumerate(lines) if l.startswith("_")]
            k = tl[int(rng.integers(0, len(tl)))]
            how = int(rng.integers(0, 4))
            if how == 0:
                del lines[k]
            elif how == 1:
                lines.insert(k, lines[k])
            elif how == 2:
                lines[k] = lines[k].upper()
            else:
                lines[k] = lines[k].split()[0]
        elif kind == 8:                                     # the atom_site tags
            tl = [k for k, l in enumerate(lines) if l.startswith("_atom_site.")]
            k = tl[int(rng.integers(0, len(tl)))]
            how = int(rng.integers(0, 3))
            if how == 0:
                del lines[k]
            elif how == 1:
                lines[k] = lines[k].replace("_atom_site.", "_ATOM_SITE.")
            else:
                lines[k], lines[tl[0]] = lines[tl[0]], lines[k]
        elif kind == 9:                                     # something between the rows
            lines.insert(i, ["# a comment", "", ";text\n;", "loop_", "stop_", "data_second", "save_x", "global_", "_new.tag 1", "'quoted value'", "\t"][int(rng.integers(0, 11))])
        elif kind == 10:                                    # another block at the end
            lines += ["data_more", "_atom_site.id 1"] if rng.random() < 0.5 else ["data_more", "_x.y z"]
        elif kind == 11:                                    # cut
            t = "\n".join(lines); lines = t[:int(rng.integers(0, len(t)))].split("\n")
        elif kind == 12:                                    # a random byte
            t = bytearray("\n".join(lines).encode("latin-1")); t[int(rng.integers(0, len(t)))] = int(rng.integers(0, 256)); lines = t.decode("latin-1").split("\n")
        elif kind == 13:                                    # the loop as pairs: one atom
            tl = [k for k, l in enumerate(lines) if l.startswith("_atom_site.")]
            p = lines[rows[0]].split()
            if len(p) == len(tl):
                for k, v in zip(tl, p):
                    lines[k] = lines[k].strip() + " " + v
                lines = [l for k, l in enumerate(lines) if k not in set(rows) and not (k == tl[0] - 1 and l.strip() == "loop_")]
        elif kind == 14:                                    # CR LF
            lines = [l + "\r" for l in lines]
        else:                                               # the head of the file
            lines[0] = ["data_", "DATA_x", "global_", "# no block", "data_x y", "loop_"][int(rng.integers(0, 6))]
    return "\n".join(lines).encode("latin-1")
