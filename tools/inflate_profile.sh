#!/bin/bash
# Runs on the GPU box: rocprofv3 --kernel-trace --stats of k_inflate alone (tools/inflate_bench.py), then the counter passes
# (tools/inflate_pmc.sh). Summaries -> gpurun_out/<tag>_*; copy the ones to be judged into profiles/.
# usage: tools/inflate_profile.sh <tag> [inflate_bench args...]
set -u
TAG=${1:-r6_inflate}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_inf_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_inf_stats -o s -- python $REPO/tools/inflate_bench.py --reps 5 "$@" > $REPO/gpurun_out/${TAG}_bench.json 2> /tmp/rp_inf_stats.err
f=$(find /tmp/rp_inf_stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $REPO/gpurun_out/${TAG}_kernel_stats.csv
cd $REPO && tools/inflate_pmc.sh $TAG "$@" > /dev/null 2>&1
