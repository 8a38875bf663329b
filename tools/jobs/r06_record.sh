#!/usr/bin/env bash
# Round-6 record (tag = first argument): full GPU suite, bench lines, kernel tables (graph replay and train), PMC
# passes at 8^4 and at the 16^4 shard, 8 ranks sharing the one GPU over gloo (sampling and training).
set -u
cd "$(dirname "$0")/../.."
tag="${1:-r06r}"
o="gpurun_out/$tag"; mkdir -p "$o"
export TMPDIR=/tmp
M=su3_assemble_tah_kernel
python -m pytest tests -q -m gpu 2>&1 | tail -4 | tee "$o/pytest_gpu.txt"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee "$o/smoke.txt"
python bench.py > "$o/bench_l2hmc.json" 2> "$o/bench_l2hmc.err"
python bench.py --mode hmc --no-u1 > "$o/bench_hmc.json" 2>> "$o/bench_l2hmc.err"
python bench.py --mode train --no-u1 --no-cpu-baseline --no-spot-check --no-comm-probe > "$o/bench_train.json" 2> "$o/bench_train.err"
python bench.py --lattice 16 16 16 16 --beta 6.2 --steps 3 --warmup 2 --no-cpu-baseline --no-spot-check --no-u1 > "$o/bench_cfg5_shard.json" 2> "$o/bench_cfg5.err"
L2Q_BENCH_SHARE_GPU=1 L2Q_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 3 --warmup 1 --nchains 32 --no-u1 > "$o/bench_8ranks_gloo_shared_gpu.json" 2> "$o/bench_8ranks.err"
L2Q_BENCH_SHARE_GPU=1 L2Q_BENCH_BACKEND=gloo timeout 1200 python bench.py --gpus 8 --mode train --steps 3 --warmup 1 --nchains 32 --no-u1 > "$o/bench_train_8ranks_gloo_shared_gpu.json" 2> "$o/bench_train_8ranks.err"
L2Q_BENCH_SKIP_INSTRUMENTED=1 KSTATS_MARKER=$M KSTATS_LAST=5 bash tools/kstats.sh "$o/bench_l2hmc_kernel_stats.txt" python bench.py --no-cpu-baseline --no-spot-check --no-u1 --no-comm-probe > "$o/kstats_l2hmc.log" 2>&1
KSTATS_MARKER=$M KSTATS_LAST=5 bash tools/kstats.sh "$o/bench_l2hmc_eager_instrumented_kernel_stats.txt" python bench.py --no-cpu-baseline --no-spot-check --no-u1 --no-comm-probe > "$o/kstats_l2hmc_eager.log" 2>&1
KSTATS_MARKER=$M KSTATS_LAST=5 bash tools/kstats.sh "$o/bench_train_kernel_stats.txt" python bench.py --mode train --no-cpu-baseline --no-spot-check --no-u1 --no-comm-probe > "$o/kstats_train.log" 2>&1
bash tools/pmc_collect.sh "$tag" > "$o/pmc.log" 2>&1
L2Q_KPROF_LATTICE="16 16 16 16" L2Q_KPROF_NB=256 bash tools/pmc_collect.sh "${tag}_16x4" > "$o/pmc16.log" 2>&1
cp profiles/${tag}_pmc_counters.txt profiles/${tag}_16x4_pmc_counters.txt profiles/pmc_traffic.json "$o/" 2>/dev/null
python bench.py --no-cpu-baseline --no-spot-check --no-u1 --no-comm-probe > "$o/bench_l2hmc_after_pmc.json" 2>/dev/null
python bench.py --lattice 16 16 16 16 --beta 6.2 --steps 3 --warmup 2 --no-cpu-baseline --no-spot-check --no-u1 --no-comm-probe > "$o/bench_cfg5_shard_after_pmc.json" 2>/dev/null
head -16 "$o/bench_l2hmc_kernel_stats.txt"; head -8 "$o/bench_l2hmc_eager_instrumented_kernel_stats.txt"; head -12 "$o/bench_train_kernel_stats.txt"
python - "$o" <<'PY'
import json, sys
o = sys.argv[1]
for f in ('bench_l2hmc', 'bench_hmc', 'bench_train', 'bench_cfg5_shard', 'bench_8ranks_gloo_shared_gpu', 'bench_train_8ranks_gloo_shared_gpu', 'bench_l2hmc_after_pmc', 'bench_cfg5_shard_after_pmc'):
    try:
        d = json.loads(open(f'{o}/{f}.json').readline())
        r = d.get('roofline') or {}
        print(f, d['value'], d['ms_per_step'], d.get('launch_path', '')[:12], (r.get('kernel') or '')[:36], r.get('frac'), r.get('traffic'))
        for rr in d.get('rooflines', []):
            print('    ', rr['kernel'][:44], rr.get('avg_ms'), rr.get('frac'), rr.get('traffic'))
    except Exception as e:
        print(f, 'failed', e)
PY
