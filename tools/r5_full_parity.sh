#!/bin/bash
# round 5: full-batch parity runs of the final code -- EVERY chain of the batch against the oracle (FCZ bytes + coordinates), and the
# first 65 536 chains against the live reference (cpu_baseline sample). Lines land in gpurun_out/r5_parity_*.json.
OFF="--steps 2 --warmup 1 --pdb-sample 0 --mixed-chains 0 --e2e-files 0 --host-chains 0"
python bench.py --parity-chains 1000000 --cpu-sample 65536 $OFF > gpurun_out/r5_parity_1m.json 2> gpurun_out/r5_parity_1m.err
python bench.py --mixed --chains 542000 --parity-chains 542000 --cpu-sample 32768 $OFF > gpurun_out/r5_parity_mixed_542k.json 2> gpurun_out/r5_parity_mixed_542k.err
: > gpurun_out/r5_parity_sweeps.jsonl
python bench.py --residues 37 --chains 2000000 --seed-base 11000000000 --parity-chains 2000000 --cpu-sample 65536 $OFF >> gpurun_out/r5_parity_sweeps.jsonl 2> gpurun_out/r5_parity_sweeps.err
python bench.py --residues 129 --chains 1000000 --seed-base 55000000000 --parity-chains 1000000 --cpu-sample 32768 $OFF >> gpurun_out/r5_parity_sweeps.jsonl 2>> gpurun_out/r5_parity_sweeps.err
python bench.py --residues 16 --chains 2000000 --seed-base 66000000000 --parity-chains 2000000 --cpu-sample 65536 $OFF >> gpurun_out/r5_parity_sweeps.jsonl 2>> gpurun_out/r5_parity_sweeps.err
python bench.py --residues 32 --chains 2000000 --seed-base 99000000000 --parity-chains 2000000 --cpu-sample 65536 $OFF >> gpurun_out/r5_parity_sweeps.jsonl 2>> gpurun_out/r5_parity_sweeps.err
python bench.py --residues 64 --chains 1000000 --seed-base 77000000000 --parity-chains 1000000 --cpu-sample 32768 $OFF >> gpurun_out/r5_parity_sweeps.jsonl 2>> gpurun_out/r5_parity_sweeps.err
python bench.py --residues 65 --chains 1000000 --seed-base 88000000000 --parity-chains 1000000 --cpu-sample 32768 $OFF >> gpurun_out/r5_parity_sweeps.jsonl 2>> gpurun_out/r5_parity_sweeps.err
python - <<'PY'
import json
for f in ("gpurun_out/r5_parity_1m.json", "gpurun_out/r5_parity_mixed_542k.json"):
    d = json.load(open(f)); print(f, d["config"]["workload"], round(d["value"] / 1e9, 3), json.dumps(d["parity"]))
for l in open("gpurun_out/r5_parity_sweeps.jsonl"):
    if l.startswith("{"):
        d = json.loads(l); print(d["config"]["workload"], round(d["value"] / 1e9, 3), d["ms_per_step"], json.dumps(d["parity"]), json.dumps(d["roofline"]["kernel_ms"]))
PY
