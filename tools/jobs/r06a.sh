#!/usr/bin/env bash
# Round 6, first session: the new wilson_loops tensor tests, the full GPU suite, the bench line with
# roofline.hbm_kernels / roofline.int8, and 8 ranks sharing the one GPU over gloo (sampling and training).
set -u
cd "$(dirname "$0")/../.."
tag="${1:-r06a}"
o="gpurun_out/$tag"; mkdir -p "$o"
export TMPDIR=/tmp
python -m pytest tests -q -m gpu -x 2>&1 | tail -4 | tee "$o/pytest_gpu.txt"
python bench.py > "$o/bench_l2hmc.json" 2> "$o/bench_l2hmc.err"; echo "bench rc=$?"
L2Q_BENCH_SHARE_GPU=1 L2Q_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 3 --warmup 1 --nchains 32 --no-u1 > "$o/bench_8ranks_gloo_shared_gpu.json" 2> "$o/bench_8ranks.err"; echo "8 ranks sampling rc=$?"
L2Q_BENCH_SHARE_GPU=1 L2Q_BENCH_BACKEND=gloo timeout 1200 python bench.py --gpus 8 --mode train --steps 3 --warmup 1 --nchains 32 --no-u1 > "$o/bench_train_8ranks_gloo_shared_gpu.json" 2> "$o/bench_train_8ranks.err"; echo "8 ranks train rc=$?"
tail -3 "$o/bench_8ranks.err" "$o/bench_train_8ranks.err"
python - "$o" <<'PY'
import json, sys
o = sys.argv[1]
for f in ('bench_l2hmc', 'bench_8ranks_gloo_shared_gpu', 'bench_train_8ranks_gloo_shared_gpu'):
    try:
        d = json.loads(open(f'{o}/{f}.json').readline())
        r = d.get('roofline') or {}
        print(f, d['value'], d['ms_per_step'], d.get('rccl_ranks'), (r.get('kernel') or '')[:36], r.get('frac'))
        print('   hbm_kernels', json.dumps(r.get('hbm_kernels'))[:600])
        print('   int8', json.dumps(r.get('int8'))[:400])
        print('   per_rank', json.dumps((d.get('per_rank') or {}).get('self_check'))[:500])
        print('   train', json.dumps(d.get('train'))[:500])
        print('   workload', d['config']['workload'][-200:])
    except Exception as e:
        print(f, 'failed', e)
PY
