#!/usr/bin/env bash
# force_tile 7 (plaquette sharing) vs 5 (thread per link) INSIDE the trajectory: two interleaved rounds of bench.py
cd "$(dirname "$0")/../.."
o=gpurun_out/r06ft; mkdir -p $o
for r in 1 2; do for ft in 5 7; do
  L2Q_TUNING=force_tile=$ft python bench.py --no-cpu-baseline --no-spot-check --no-u1 --no-comm-probe > $o/b_${ft}_$r.json 2>/dev/null
  python - $o/b_${ft}_$r.json $ft <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).readline())
k = d['kernels'].get('l2q_su3_force', {})
print('force_tile', sys.argv[2], d['value'], d['ms_per_step'], 'force avg_ms', k.get('avg_ms'), (d['roofline'].get('hbm_kernels') or {}).get('force', {}).get('symbol'))
PY
done; done


