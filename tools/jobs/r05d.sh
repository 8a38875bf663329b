#!/usr/bin/env bash
# round 5: A/B of the sliced heads kernel (V1: both tanh in the q period; V2: s-head tanh moved, short exp), then the
# kernel tests of the new default and the bench line
set -u
cd "$(dirname "$0")/../.."
o=gpurun_out/r05d; mkdir -p $o
export TMPDIR=/tmp GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
BOTH=0 TAILN=3 bash tools/gpu_job_ab.sh tools/time_heads_sliced.py slv1 slv2 > $o/ab.txt 2>&1
cat $o/ab.txt
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "heads" > $o/t_heads.log 2>&1; echo "heads tests rc=$?" | tee -a $o/summary.txt
tail -3 $o/t_heads.log
timeout 1200 python -m pytest tests/test_dynamics_gpu.py -q -m gpu -x -k "auto_graphed" > $o/t_ag.log 2>&1; echo "auto graph test rc=$?" | tee -a $o/summary.txt
tail -3 $o/t_ag.log
python bench.py --no-u1 --no-cpu-baseline > $o/bench_l2hmc.json 2> $o/bench.err; echo "bench rc=$?" | tee -a $o/summary.txt
python - $o <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + '/bench_l2hmc.json').readline())
print(d['value'], d['ms_per_step'], d['roofline']['avg_ms'], d['roofline']['frac'])
for k, v in d['kernels'].items():
    print(k, v['avg_ms'], v['share'])
PY
