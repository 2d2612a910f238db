cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA"; do
  rm -rf /tmp/rp
  rocprofv3 --kernel-include-regex "k_backbone_fast|k_sidechain" --pmc $grp --output-format csv -d /tmp/rp -o p -- python $REPO/bench.py --numerics fast --chains 131072 --steps 2 --warmup 1 --cpu-sample 0 --pdb-sample 0 --mixed-chains 0 --no-parity > /dev/null 2>&1
  python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in glob.glob("/tmp/rp/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
R = 131072 * 350
for k in agg:
    n = len(cnt[k])
    print(k, {c: round(v / n / R, 3) for c, v in agg[k].items()})
PY
done
