#!/bin/bash
# round 5: should chains of 65..128 residues go four to a wavefront too (8 rounds of 16)? usage: tools/r5_rows8_ab.sh <tag> <lib>...
TAG=$1; shift
OUT=gpurun_out/rows8_$TAG.txt; : > $OUT
run() {
  local lib=$1 label=$2; shift 2
  FCZ_HIP_LIB=$PWD/$lib python bench.py "$@" --steps 3 --warmup 1 --cpu-sample 0 --pdb-sample 0 --mixed-chains 0 --e2e-files 0 --host-chains 0 > /tmp/ab.json 2> /tmp/ab.err || { echo "$lib $label FAILED" >> $OUT; tail -3 /tmp/ab.err >> $OUT; return; }
  python - "$lib" "$label" >> $OUT <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json")); k = d["roofline"]["kernel_ms"]; p = d.get("parity") or {}
print(sys.argv[1], sys.argv[2], "Gres/s=%.3f" % (d["value"] / 1e9), "step_ms=%.2f" % d["ms_per_step"], " ".join(f"{n.replace('compress_','c_').replace('decompress_','d_')}={v:.3f}" for n, v in k.items()), "parity=%s/%s/%s" % (p.get("chains_checked"), p.get("fcz_bit_exact"), p.get("coords_bit_exact")))
PY
}
for lib in "$@"; do
  run $lib res16 --residues 16 --chains 2000000 --parity-chains 131072 --seed-base 66000000000
  run $lib res37 --residues 37 --chains 2000000 --parity-chains 131072
  run $lib res100 --residues 100 --chains 1000000 --parity-chains 131072 --seed-base 44000000000
  run $lib res129 --residues 129 --chains 1000000 --parity-chains 65536 --seed-base 55000000000
  run $lib mixed --mixed --chains 542000 --parity-chains 65536
  run $lib res350 --chains 262144 --parity-chains 8192
done
cat $OUT
