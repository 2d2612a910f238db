#!/bin/bash
# round 5: where k_compress_pack spends a short chain (SQ counters, 2 M x 37-residue chains). usage: tools/r5_pack_pmc.sh <lib>
REPO=${GRAFT_REPO_ROOT:-/root/repo}; LIB=$1
OUT=$REPO/gpurun_out/r5_pack_pmc; mkdir -p $OUT
ARGS="--residues 37 --chains 2000000 --steps 2 --warmup 1 --cpu-sample 0 --no-parity --mixed-chains 0 --e2e-files 0 --pdb-sample 0 --host-chains 0"
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_IFETCH SQ_CYCLES SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT"; do
  rm -rf /tmp/rp
  FCZ_HIP_LIB=$REPO/$LIB rocprofv3 --kernel-include-regex "k_compress_pack" --pmc $grp --output-format csv -d /tmp/rp -o p -- python $REPO/bench.py $ARGS > /tmp/b.json 2> /tmp/b.err
  python3 - /tmp/rp $LIB >> $OUT/pmc.txt <<'PY'
import csv, glob, os, sys, collections
per = collections.defaultdict(float); n = collections.defaultdict(set)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]; per[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for (k, c), v in sorted(per.items()): print(sys.argv[2], k, c, "per_dispatch=%.0f" % (v / max(len(n[k]), 1)), "dispatches=%d" % len(n[k]))
PY
done
cat $OUT/pmc.txt
