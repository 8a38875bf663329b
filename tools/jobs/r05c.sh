#!/usr/bin/env bash
# round 5, third pass: auto-graphed small U(1) transitions, overlapped exchange, whole GPU tier, U(1) bench block
set -u
cd "$(dirname "$0")/../.."
o=gpurun_out/r05c; mkdir -p $o
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu -x > $o/t_all.log 2>&1; echo "all rc=$?" | tee -a $o/summary.txt
tail -12 $o/t_all.log
python - > $o/u1.json 2> $o/u1.err <<'PY'
import sys, json
sys.path.insert(0, 'l2hmc-qcd_amd'); sys.path.insert(0, '.')
import torch
import bench
out = bench.secondary_u1()
out['published'] = bench.published_u1()
for k, v in out.items():
    if isinstance(v, dict):
        print(k, {a: v[a] for a in ('ms_per_trajectory', 'value', 'kernel_time_fraction_of_wall', 'hip_graph', 'train_step_s', 'eval_step_s', 'hmc_step_s') if a in v})
    else:
        print(k, v)
json.dump(out, open(sys.argv[0] if False else 'gpurun_out/r05c/u1_full.json', 'w'), indent=1, default=str)
PY
echo "u1 rc=$?" | tee -a $o/summary.txt
cat $o/u1.json; tail -5 $o/u1.err
