#!/bin/bash
# Runs on the GPU box: counter passes of k_inflate alone (tools/inflate_bench.py); per-dispatch sums -> gpurun_out/<tag>_pmc.txt
# usage: tools/inflate_pmc.sh <tag> [inflate_bench args...]
set -u
TAG=${1:-inflate}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${TAG}_pmc.txt
cd /tmp && export TMPDIR=/tmp
: > $OUT
pass() {
  rm -rf /tmp/rp_inf
  rocprofv3 --kernel-include-regex "k_inflate" --pmc "$@" --output-format csv -d /tmp/rp_inf -o p -- python $REPO/tools/inflate_bench.py --reps 1 $ARGS > /tmp/rp_inf.out 2> /tmp/rp_inf.err
  grep files /tmp/rp_inf.out | tail -1 >> $OUT
  python3 - /tmp/rp_inf >> $OUT <<'PY'
import csv, glob, os, sys, collections
per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_inflate_merge" in r["Kernel_Name"]: continue
        per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for c, dv in sorted(per.items()):
    vals = list(dv.values())
    print("  %-28s dispatches %d  last %.6g" % (c, len(vals), vals[-1]))
PY
}
ARGS="$@"
pass SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU
pass SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_IFETCH SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT
cat $OUT
