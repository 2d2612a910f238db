#!/bin/bash
# A/B timing of libfcz_hip.so variants on the GPU box: tools/ab_bench.sh <tag> <lib>... ; per-kernel ms of each into gpurun_out/ab_<tag>.txt
TAG=$1; shift
OUT=gpurun_out/ab_$TAG.txt; : > $OUT
for lib in "$@"; do
  for rep in ${REPS:-1 2}; do
    FCZ_HIP_LIB=$PWD/$lib python bench.py --chains ${CHAINS:-262144} --steps 4 --warmup 1 --cpu-sample 0 --pdb-sample 0 --mixed-chains 0 --e2e-files 0 --host-chains 0 > /tmp/ab.json 2> /tmp/ab.err || { echo "$lib FAILED"; tail -3 /tmp/ab.err; }
    python - "$lib" >> $OUT <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json")); k = d["roofline"]["kernel_ms"]
a = d.get("alt_numerics") or {}
print(sys.argv[1], "step_ms=%.2f" % d["ms_per_step"], "alt[%s]: dec=%s bb=%s sc=%s |" % (a.get("mode"), a.get("decompress_ms"), (a.get("kernel_ms") or {}).get("decompress_backbone"), (a.get("kernel_ms") or {}).get("decompress_sidechain")), " ".join(f"{n.replace('compress_','c_').replace('decompress_','d_')}={v:.3f}" for n, v in k.items()), "parity=%s/%s" % (d["parity"]["fcz_bit_exact"], d["parity"]["coords_bit_exact"]))
PY
  done
done
cat $OUT
