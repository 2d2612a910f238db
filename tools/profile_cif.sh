#!/bin/bash
# GPU box (via gpurun): the mmCIF ingest kernels alone under rocprofv3 -- kernel statistics, then FETCH_SIZE and WRITE_SIZE in passes of
# their own -- on tools/dbg/cif_ab.py's batch (AFDB-shaped files of 350 residues). usage: tools/profile_cif.sh <tag> [files]
set -u
TAG=${1:-cif}; FILES=${2:-16384}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() { local name=$1; shift; rm -rf /tmp/rpc_$name
  rocprofv3 "$@" --output-format csv -d /tmp/rpc_$name -o $name -- python $REPO/tools/dbg/cif_ab.py $FILES > $OUT/${name}_run.txt 2>&1
  python3 - /tmp/rpc_$name $OUT $name <<'PY'
import csv, glob, os, sys, collections
src, out, name = sys.argv[1:4]
for f in glob.glob(os.path.join(src, "**", "*.csv"), recursive=True):
    base = os.path.basename(f)
    if base.endswith("_counter_collection.csv"):
        per = collections.defaultdict(lambda: collections.defaultdict(dict))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "k_ingest" not in k: continue
            d = per[k][r["Counter_Name"]]; d[r["Dispatch_Id"]] = d.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
        with open(os.path.join(out, name + "_per_kernel.csv"), "w") as o:
            w = csv.writer(o); w.writerow(["kernel", "counter", "dispatches", "largest_dispatch", "mean_of_the_full_size_dispatches"])
            for k in sorted(per):
                for c, dv in sorted(per[k].items()):
                    big = max(dv.values()); full = [v for v in dv.values() if v >= 0.5 * big]
                    w.writerow([k, c, len(dv), big, sum(full) / len(full)])
    elif base.endswith("_kernel_stats.csv"):
        rows = [r for r in csv.DictReader(open(f)) if "k_ingest" in r.get("Name", "")]
        with open(os.path.join(out, name + "_kernel_stats.csv"), "w") as o:
            if rows:
                w = csv.DictWriter(o, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
PY
  rm -rf /tmp/rpc_$name; }
pass stats --kernel-trace --stats
pass fetch --kernel-include-regex "k_ingest" --pmc FETCH_SIZE
pass write --kernel-include-regex "k_ingest" --pmc WRITE_SIZE
tail -n 1 $OUT/stats_run.txt; cat $OUT/stats_kernel_stats.csv $OUT/fetch_per_kernel.csv $OUT/write_per_kernel.csv 2>/dev/null
