#!/bin/bash
# round 5, VERDICT item 4: HBM-side traffic of k_backbone<0> with one ring slot per group (v0) and with the persistent grid (v1):
# L2 fabric-side counters (FETCH_SIZE / WRITE_SIZE / TCC_EA0_*_DRAM_32B) per launch, and the memory controllers' activity level.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r5_ring; mkdir -p $OUT
ARGS="--chains 1000000 --steps 2 --warmup 1 --cpu-sample 0 --no-parity --mixed-chains 0 --e2e-files 0 --pdb-sample 0 --host-chains 0"
cd /tmp && export TMPDIR=/tmp
for v in v0p v1p; do
  for ctr in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum; do
    rm -rf /tmp/rp
    FCZ_HIP_LIB=$REPO/build/libfcz_$v.so rocprofv3 --kernel-include-regex "k_backbone" --pmc $ctr --output-format csv -d /tmp/rp -o p -- python $REPO/bench.py $ARGS > /tmp/b.json 2> /tmp/b.err
    python3 - /tmp/rp $v $ctr >> $OUT/pmc.txt <<'PY'
import csv, glob, os, sys, collections
per = collections.defaultdict(float); n = collections.defaultdict(set)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]; per[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for (k, c), v in sorted(per.items()): print(sys.argv[2], k, c, "per_dispatch=%.0f" % (v / max(len(n[k]), 1)), "dispatches=%d" % len(n[k]))
PY
  done
  FCZ_HIP_LIB=$REPO/build/libfcz_$v.so python $REPO/tools/hbm_busy_probe.py --chains 1000000 --seconds 3 --probes k_backbone > $OUT/busy_$v.json 2> $OUT/busy_$v.err
done
cat $OUT/pmc.txt
# (same call) the two-rank sharded decompress BEFORE the write-once change: its exchange_and_splice_s against engine_s
cd $REPO && python bench.py --chains 65536 --steps 1 --warmup 0 --cpu-sample 0 --no-parity --mixed-chains 0 --pdb-sample 0 --host-chains 0 --e2e-files 8192 > $OUT/e2e_before.json 2> $OUT/e2e_before.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5_ring/e2e_before.json"))
print(json.dumps(d.get("end_to_end", {}).get("sharded"), indent=1))
PY
