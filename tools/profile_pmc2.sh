#!/bin/bash
# extra SQ counter groups (instruction mix) for the fcz kernels; usage: tools/profile_pmc2.sh <tag> [bench args]
set -u
TAG=${1:-x}; shift || true
ARGS=${@:-"--chains 65536 --steps 2 --warmup 1 --cpu-sample 0 --no-parity"}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  rm -rf /tmp/rp_$name
  rocprofv3 --kernel-include-regex "fcz" --pmc "$@" --output-format csv -d /tmp/rp_$name -o $name -- python $REPO/bench.py $ARGS > /dev/null 2> $OUT/${name}.err
  python3 - /tmp/rp_$name $OUT $name <<'PY'
import csv, glob, os, sys, collections
src, out, name = sys.argv[1:4]
for f in glob.glob(os.path.join(src, "**", "*_counter_collection.csv"), recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"].split("(")[0]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    with open(os.path.join(out, "pmc_%s_per_kernel.csv" % name), "w") as o:
        w = csv.writer(o); w.writerow(["kernel", "dispatches", "counter", "sum", "per_dispatch"])
        for k in sorted(agg):
            for c, v in sorted(agg[k].items()):
                w.writerow([k, len(cnt[k]), c, v, v / max(len(cnt[k]), 1)])
PY
  rm -rf /tmp/rp_$name
}
run a SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES
run b SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT
run c SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run d SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
