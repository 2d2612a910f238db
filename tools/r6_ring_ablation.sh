#!/bin/bash
# Runs on the GPU box: k_backbone with its ring traffic folded into an L2-resident window (measurement builds of libfcz_hip:
# -DFCZ_ABL_NO_TRING, -DFCZ_ABL_NO_RING; foldcomp_amd/csrc/fcz_kernels.h BB_RROW / BB_TROW) beside the product, on the headline batch
# and the mixed-length batch. Output: gpurun_out/r6_ab_ring_ablation.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r6_ab_ring_ablation.txt
: > $OUT
for v in libfcz_hip libfcz_hip_notring libfcz_hip_noring libfcz_hip; do
  for rep in 1 2; do
    FCZ_HIP_LIB=$REPO/foldcomp_amd/$v.so python $REPO/bench.py --steps 10 --no-parity --cpu-sample 0 --e2e-files 0 --pdb-sample 0 --host-chains 0 --mixed-steps 5 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']; m=d['mixed']['kernel_ms']
print('$v run $rep: 1Mx350 backbone %.2f ms  step %.2f ms | mixed 542k backbone %.2f ms  step %.2f ms | sidechain %.2f index %.2f' % (k['decompress_backbone'], d['ms_per_step'], m['decompress_backbone'], d['mixed']['ms_per_step'], k['decompress_sidechain'], k['decompress_index']))" >> $OUT
  done
done
cat $OUT
