import gzip, json, os, subprocess, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import torch
torch.cuda.init()
import numpy as np
from foldcomp_amd import synthetic
from foldcomp_amd.codec import Codec
from concurrent.futures import ThreadPoolExecutor
n = 4096
b = synthetic.to_chain_batch(synthetic.generate(512, [350] * 512, seed=3))
with Codec(0) as c:
    blob, off, st = c.compress_batch(b)
    texts, _ = c.decompress_pdb(blob, off)
tmp = tempfile.mkdtemp(prefix="jobtest_", dir="/tmp")
d = os.path.join(tmp, "gz"); os.mkdir(d)
with ThreadPoolExecutor(16) as ex:
    gz = list(ex.map(lambda t: gzip.compress(t, 6), texts))
for i in range(n):
    open(os.path.join(d, f"s{i:06d}.pdb.gz"), "wb").write(gz[i % 512])
lst = os.path.join(tmp, "l.txt"); open(lst, "w").write((d + "\n") * 24)
host = os.path.join(os.getcwd(), "host", "foldcomp-hip")
for wpg in (2, 3):
    for job in (1024, 2048, 2560, 4096, 8192):
        r = subprocess.run([host, "compress", "-d", "-y", "-t", "16", "--gpus", "1", "--workers-per-gpu", str(wpg), "--job-files", str(job), "--json-stats", "-f", lst, os.path.join(tmp, "db")], capture_output=True, text=True)
        st = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        steady = st["wall_s"] - st["ctx_ready_s"]
        print(wpg, job, round(st["residues"] / steady / 1e6, 1), "M res/s steady", st["wall_s"], st["records"], flush=True)
