#!/bin/bash
# GPU box: per-kernel time of the gzipped-directory route (host/foldcomp-hip compress -d on .pdb.gz files; KIND=cif: .cif.gz) under rocprofv3 --kernel-trace --stats
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
python3 - <<'PY'
import gzip, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
torch.cuda.init()
from foldcomp_amd import synthetic
from foldcomp_amd.codec import Codec
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
kind = os.environ.get("KIND", "pdb")
b = synthetic.to_chain_batch(synthetic.generate(512, [350] * 512, seed=3))
with Codec(0) as c:
    blob, off, st = c.compress_batch(b)
    texts, _ = c.decompress_pdb(blob, off)
d = "/tmp/gzk/gz"; os.makedirs(d, exist_ok=True)
with ThreadPoolExecutor(16) as ex:
    if kind == "cif":
        from bench import cif_from_pdb_text
        texts = [cif_from_pdb_text(t, f"S{i:05d}") for i, t in enumerate(texts)]
    gz = list(ex.map(lambda t: gzip.compress(t, 6), texts))
for i in range(4096):
    open(os.path.join(d, f"s{i:06d}.{kind}.gz"), "wb").write(gz[i % 512])
open("/tmp/gzk/l.txt", "w").write((d + "\n") * 12)
PY
rm -rf /tmp/rp_gzk
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_gzk -o k -- $REPO/host/foldcomp-hip compress -d -y -t 16 --gpus 1 --json-stats -f /tmp/gzk/l.txt /tmp/gzk/db > /tmp/gzk/out.txt 2>/tmp/gzk/err.txt
tail -1 /tmp/gzk/out.txt | cut -c1-400
f=$(find /tmp/rp_gzk -name "*kernel_stats.csv" | head -1)
cp $f $REPO/gpurun_out/r6_gz_e2e_${KIND:-pdb}_kernel_stats.csv
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-60s calls %5s avg %9.3f ms total %9.1f ms  %5s%%" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
