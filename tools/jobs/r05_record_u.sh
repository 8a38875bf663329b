#!/usr/bin/env bash
# End-of-round record after the cfg-3 session (tag r05u): full GPU suite, smoke, the driver's bench line (with the U(1)
# block), rocprofv3 table + PMC traffic of the cfg-3 kernels, the stand-alone timings of the new kernels.
set -u
cd "$(dirname "$0")/../.."
tag="${1:-r05u}"
o="gpurun_out/$tag"; mkdir -p "$o"
export TMPDIR=/tmp
python -m pytest tests -q -m gpu 2>&1 | tail -4 | tee "$o/pytest_gpu.txt"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee "$o/smoke.txt"
python bench.py > "$o/bench_l2hmc.json" 2> "$o/bench_l2hmc.err"
python bench.py --mode train --no-u1 --no-cpu-baseline --no-spot-check --no-comm-probe > "$o/bench_train.json" 2> "$o/bench_train.err"
bash tools/kstats.sh "$o/u1_cfg3_kernel_stats.txt" python tools/kprof_u1_cfg3.py > "$o/kstats_u1.log" 2>&1
python tools/time_heads_h.py 0 2 2>&1 | grep "^\[" | tee "$o/time_heads_h.txt"
python tools/time_gemm_h_input.py 0 1 2>&1 | grep "^\[" | tee "$o/time_gemm_h_input.txt"
L2Q_KPROF_DESC='U(1) 64x64, 8192 chains, fp16 layers / fp32 lattice (BASELINE cfg-3), dense [256, 256] network' L2Q_KPROF_LATTICE='64 64' L2Q_KPROF_NB=8192 L2Q_KPROF_SCRIPT=tools/kprof_u1_cfg3.py L2Q_PMC_JSON=profiles/pmc_traffic_u1_cfg3.json bash tools/pmc_collect.sh "${tag}_u1_cfg3" > "$o/pmc_u1.log" 2>&1
cp profiles/${tag}_u1_cfg3_pmc_counters.txt profiles/pmc_traffic_u1_cfg3.json "$o/" 2>/dev/null
head -14 "$o/u1_cfg3_kernel_stats.txt" | cut -c1-170
python - "$o" <<'PY'
import json, sys
o = sys.argv[1]
for f in ('bench_l2hmc', 'bench_train'):
    try:
        d = json.loads(open(f'{o}/{f}.json').readline())
        print(f, d['value'], d['ms_per_step'], d.get('launch_path', '')[:14])
        for tag, rec in (d.get('secondary_u1') or {}).items():
            if isinstance(rec, dict) and 'ms_per_trajectory' in rec:
                print('   ', tag, rec['ms_per_trajectory'], rec['value'], (rec.get('dominant_kernel') or {}).get('frac'))
    except Exception as e:
        print(f, 'failed', e)
PY
