#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes of bench.py; keeps only
# the small CSV summaries under gpurun_out/prof_<tag>/ (copy the ones to be judged into profiles/).
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
ARGS=${@:-"--chains 262144 --steps 3 --warmup 1 --cpu-sample 0 --no-parity --mixed-chains 0 --e2e-files 0 --pdb-sample 0 --host-chains 0"}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, extra rocprof flags...
  local name=$1; shift
  rm -rf /tmp/rp_$name
  rocprofv3 "$@" --output-format csv -d /tmp/rp_$name -o $name -- python $REPO/bench.py $ARGS > $OUT/${name}_bench.json 2> $OUT/${name}_bench.err
  python3 - /tmp/rp_$name $OUT <<'PY'
import csv, glob, os, sys, collections, shutil
src, out = sys.argv[1], sys.argv[2]
for f in glob.glob(os.path.join(src, "**", "*.csv"), recursive=True):
    base = os.path.basename(f)
    if base.endswith("_counter_collection.csv"):
        # per kernel and counter: the value of every dispatch. A kernel that is also launched on an (almost always empty) work
        # list -- k_compress_angles after k_compress_angles_w, the 128-residue k_sidechain launch -- would halve a plain
        # per-dispatch average: dispatches whose SQ_WAVES / first counter is below 5 % of the kernel's largest are not counted
        per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r["Kernel_Name"].split("(")[0]
                if "fcz" not in k: continue
                per[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        with open(os.path.join(out, base.replace("_counter_collection.csv", "_per_kernel.csv")), "w") as o:
            w = csv.writer(o); w.writerow(["kernel", "dispatches", "counter", "sum", "per_dispatch"])
            for k in sorted(per):
                ref = per[k].get("SQ_WAVES") or per[k][sorted(per[k])[0]]
                big = max(ref.values()) if ref else 0.0
                full = {d for d, v in ref.items() if v >= 0.05 * big} or set(ref)
                for c, dv in sorted(per[k].items()):
                    v = sum(x for d, x in dv.items() if d in full)
                    w.writerow([k, len(full), c, v, v / max(len(full), 1)])
    elif base.endswith("_kernel_trace.csv"):
        continue
    elif os.path.getsize(f) < (4 << 20):
        shutil.copy(f, out)
PY
  rm -rf /tmp/rp_$name
}
if [ -z "${PMC_ONLY:-}" ]; then
  [ -n "${FULL_ONLY:-}" ] || run stats --kernel-trace --stats
  # the same trace at the default bench size (1 M chains), the line the driver records, minus the legs that launch the
  # same kernels at OTHER sizes (mixed-length leg, host-pointer leg) or on the CPU: their launches would mix into the
  # per-kernel averages, which are to be compared with roofline.avg_launch_ms of the headline workload
  if [ -n "${FULL_STATS:-}" ]; then SAVE_ARGS=$ARGS; ARGS="--cpu-sample 0 --e2e-files 0 --mixed-chains 0 --host-chains 0"; run stats_full --kernel-trace --stats; ARGS=$SAVE_ARGS; fi
fi
[ -n "${FULL_ONLY:-}" ] && exit 0
# PMC passes: counters only (no trace domains besides kernel dispatch), one group per pass
run pmc_sq1 --kernel-include-regex "fcz" --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run pmc_sq2 --kernel-include-regex "fcz" --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU
if [ -z "${NO_MEM:-}" ]; then run pmc_fetch --kernel-include-regex "fcz" --pmc FETCH_SIZE
run pmc_write --kernel-include-regex "fcz" --pmc WRITE_SIZE
# the same traffic in exact units: the L2's requests towards the memory controllers counted in 32-byte pieces (a 64-byte request
# counts 2, a 128-byte one 4), reads and writes in separate passes. Like FETCH_SIZE / WRITE_SIZE they sit on the L2's fabric side;
# tools/hbm_busy_probe.py holds them against the memory controllers' own activity level.
run pmc_dram_rd --kernel-include-regex "fcz" --pmc TCC_EA0_RDREQ_DRAM_32B_sum
run pmc_dram_wr --kernel-include-regex "fcz" --pmc TCC_EA0_WRREQ_WRITE_DRAM_32B_sum; fi
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts in the kernels' own access patterns
if [ -z "${NO_MEM:-}" ]; then
  hipcc --offload-arch=gfx950 -O3 -w -o /tmp/pmc_calibrate $REPO/tools/pmc_calibrate.hip
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/rp_cal
    rocprofv3 --kernel-include-regex "fczcal" --pmc $ctr --output-format csv -d /tmp/rp_cal -o cal -- /tmp/pmc_calibrate > $OUT/cal_$ctr.out 2>&1
    python3 - /tmp/rp_cal $OUT/cal_$ctr.csv <<'PY'
import csv, glob, os, sys, collections
agg = collections.defaultdict(float); cnt = collections.defaultdict(set)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]; agg[(k, r["Counter_Name"])] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
with open(sys.argv[2], "w") as o:
    w = csv.writer(o); w.writerow(["kernel", "counter", "per_dispatch"])
    for (k, c), v in sorted(agg.items()): w.writerow([k, c, v / max(len(cnt[k]), 1)])
PY
  done
fi
ls -la $OUT
