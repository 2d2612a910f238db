import sys, numpy as np
sys.path[:0]=['/root/repo','/root/repo/tests','/root/repo/oracle']
from test_host_cpp import _pdb_text
from foldcomp_amd.codec import Codec
z=np.load('/root/repo/tests/golden/reference_vectors.npz')
base=_pdb_text(z, "syn:len26").splitlines()
c=Codec(0)
phe=next(l for l in base if l[13:16]=="CE2" and l[17:20]=="PHE")
def run(lines, tag):
    tt=("\n".join(lines)+"\n").encode()
    b, cfile, cmeta, fstat, refused = c.ingest_pdb([tt],["a.pdb"])
    print(tag, "fstat", fstat, "chains", b.n_chains, "res", b.n_residues, "atoms", b.n_atoms, "refused", refused.tolist())
run(base, "base")
for at in (5, 20, 60, 63, 64, 65, 100, 151, 152, 153, 200):
    run(base[:at]+[phe]+base[at:], f"phe@{at} ({base[at][17:26]})")
# residue number going down by one line
for at in (30, 100):
    l=base[at]; run(base[:at]+[l[:22]+"%4d"%(int(l[22:26])-1)+l[26:]]+base[at+1:], f"down@{at}")
