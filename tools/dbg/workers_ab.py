"""GPU box: host/foldcomp-hip compress -d on the four input kinds (plain .pdb / .cif, .pdb.gz / .cif.gz; 4 096 files x 12 walks) with 2 and
3 workers per GPU, two runs each: steady residues/s."""
import gzip, json, os, subprocess, sys, tempfile
sys.path.insert(0, os.getcwd())
import torch
torch.cuda.init()
from foldcomp_amd import synthetic
from foldcomp_amd.codec import Codec
from concurrent.futures import ThreadPoolExecutor
from bench import cif_from_pdb_text
n, nd = 4096, 512
b = synthetic.to_chain_batch(synthetic.generate(nd, [350] * nd, seed=3))
with Codec(0) as c:
    blob, off, st = c.compress_batch(b)
    texts, _ = c.decompress_pdb(blob, off)
cifs = [cif_from_pdb_text(t, f"S{i:05d}") for i, t in enumerate(texts)]
tmp = tempfile.mkdtemp(prefix="wab_", dir="/tmp")
with ThreadPoolExecutor(16) as ex:
    gzp = list(ex.map(lambda t: gzip.compress(t, 6), texts)); gzc = list(ex.map(lambda t: gzip.compress(t, 6), cifs))
sets = {"pdb": (texts, ".pdb"), "cif": (cifs, ".cif"), "pdb_gz": (gzp, ".pdb.gz"), "cif_gz": (gzc, ".cif.gz")}
host = os.path.join(os.getcwd(), "host", "foldcomp-hip")
for kind, (data, ext) in sets.items():
    d = os.path.join(tmp, kind); os.mkdir(d)
    for i in range(n):
        open(os.path.join(d, f"s{i:06d}{ext}"), "wb").write(data[i % nd])
    lst = os.path.join(tmp, kind + ".txt"); open(lst, "w").write((d + "\n") * 12)
    for wpg in (2, 3, 2, 3):
        r = subprocess.run([host, "compress", "-d", "-y", "-t", "16", "--gpus", "1", "--workers-per-gpu", str(wpg), "--json-stats", "-f", lst, os.path.join(tmp, "db")], capture_output=True, text=True)
        s = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        print(kind, wpg, round(s["residues"] / (s["wall_s"] - s["ctx_ready_s"]) / 1e6, 1), "M res/s steady; wall", s["wall_s"], flush=True)
