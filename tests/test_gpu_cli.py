"""GPU: the CLI surface (python -m foldcomp_amd = reference src/main.cpp) end to end on files, directories
and databases, checked against reference-minted goldens."""
import os

import numpy as np
import pytest

from _cases import golden_batch
import host_text as pdbio   # oracle/host_text.py: host restatement of the reference writer
from foldcomp_amd.__main__ import file_parts, main
from foldcomp_amd.database import DatabaseReader

pytestmark = pytest.mark.gpu


def _pdb_text(z, name, chain=None):
    b = golden_batch(z, name)
    res_of_atom = np.repeat(np.arange(b.n_residues), np.diff(b.atom_off.astype(np.int64)))
    return pdbio.format_pdb("", b.atom_code, b.res_code[res_of_atom], int(b.first_res_index[0]) + res_of_atom,
                            chain or chr(b.chain_id[0]), int(b.first_atom_index[0]), b.x, b.y, b.z, b.bfac_ca[res_of_atom])


@pytest.fixture(autouse=True)
def _use_codec(codec):
    from foldcomp_amd import api
    api.set_codec(codec)
    yield
    api.set_codec(None)


def test_file_parts():
    # getFileParts splits at the LAST dot (reference src/utility.cpp:118-126; the real reference names test.cif.gz -> test.cif.fcz)
    from foldcomp_amd.__main__ import is_compressible
    assert file_parts("test.pdb") == ("test", "pdb") and file_parts("test.cif.gz") == ("test.cif", "gz")
    assert file_parts("d1asha_") == ("d1asha_", "")
    assert is_compressible("test.cif", "gz") and is_compressible("x", "pdb") and not is_compressible("x.fcz", "gz") and not is_compressible("x", "txt")


def test_single_file_roundtrip(tmp_path, golden):
    z, _ = golden
    p = tmp_path / "test_af.pdb"
    p.write_text(_pdb_text(z, "pdb:test_af"))
    assert main(["compress", str(p)]) == 0
    fcz = (tmp_path / "test_af.fcz").read_bytes()
    exp = z["pdb:test_af/fcz"].tobytes()
    # single-file title = output path stem (src/main.cpp:452-465): only the title bytes differ from the golden
    tl = int.from_bytes(fcz[24:28], "little"); tl0 = int.from_bytes(exp[24:28], "little")
    na = fcz[12]
    assert fcz[76 + 4 * na:76 + 4 * na + tl].decode() == str(tmp_path / "test_af")
    assert fcz[76 + 4 * na + tl:] == exp[76 + 4 * na + tl0:]
    assert main(["decompress", str(tmp_path / "test_af.fcz"), str(tmp_path / "back.pdb")]) == 0
    back = (tmp_path / "back.pdb").read_text()
    ref = z["pdb:test_af/pdb0"].tobytes().decode()
    assert [l for l in back.splitlines() if l.startswith(("ATOM", "TER"))] == [l for l in ref.splitlines() if l.startswith(("ATOM", "TER"))]
    assert main(["check", str(tmp_path / "test_af.fcz")]) == 0
    assert main(["extract", "--fasta", str(tmp_path / "test_af.fcz"), str(tmp_path / "seq.fasta")]) == 0
    assert (tmp_path / "seq.fasta").read_text().splitlines()[1] == z["pdb:test_af/fasta"].tobytes().decode()


def test_directory_to_db_and_back(tmp_path, golden):
    z, _ = golden
    d = tmp_path / "in"
    d.mkdir()
    names = ["pdb:test_af", "pdb:test", "syn:len350", "syn:len26"]
    for n in names:
        (d / (n.split(":")[1] + ".pdb")).write_text(_pdb_text(z, n))
    # a two-chain file with a numbering gap -> 3 records (multichainA, multichainB_0, multichainB_1)
    (d / "multichain.pdb").write_text(_pdb_text(z, "pdb:multichainA") + _pdb_text(z, "pdb:multichainB_0") + _pdb_text(z, "pdb:multichainB_1"))
    assert main(["compress", "-d", str(d), str(tmp_path / "db")]) == 0
    r = DatabaseReader(str(tmp_path / "db"))
    assert len(r) == 7
    got = {}
    for i in range(len(r)):
        got.setdefault(r.name(i), []).append(r.data(i))
    r.close()
    def no_title(f):   # everything except lenTitle and the title bytes (CLI title = file stem)
        na, tl = f[12], int.from_bytes(f[24:28], "little")
        return f[:24] + f[28:76 + 4 * na] + f[76 + 4 * na + tl:]
    for n in names:
        stem = n.split(":")[1]
        assert no_title(got[stem][0]) == no_title(z[f"{n}/fcz"].tobytes()), n
        if n.startswith("pdb:"):
            assert got[stem][0] == z[f"{n}/fcz"].tobytes(), n      # title == file stem == golden title
    assert len(got["multichain"]) == 3
    assert main(["decompress", str(tmp_path / "db"), str(tmp_path / "out")]) == 0
    for n in names:
        stem = n.split(":")[1]
        got_pdb = (tmp_path / "out" / (stem + ".pdb")).read_text().splitlines()
        ref_pdb = z[f"{n}/pdb0"].tobytes().decode("latin-1").splitlines()
        assert [l for l in got_pdb if not l.startswith("TITLE")] == [l for l in ref_pdb if not l.startswith("TITLE")], n
    assert main(["extract", "--plddt", "-p", "4", str(tmp_path / "db"), str(tmp_path / "plddt.tsv")]) == 0
    lines = (tmp_path / "plddt.tsv").read_text().splitlines()
    assert len(lines) == 7
