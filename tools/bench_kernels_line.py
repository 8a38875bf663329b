#!/usr/bin/env python3
"""One-line digest of a bench.py run for A/B jobs: value, ms per step and the HIP-event average of every
timed entry point."""
import json
import subprocess
import sys

out = subprocess.run([sys.executable, 'bench.py', '--no-cpu-baseline', '--no-spot-check'] + sys.argv[1:],
                     capture_output=True, text=True).stdout.strip().splitlines()[-1]
d = json.loads(out)
print(f"value {d['value']:.0f} ms/step {d['ms_per_step']:.3f}  " +
      '  '.join(f"{k[4:]} {v['avg_ms']:.4f}" for k, v in d.get('kernels', {}).items() if v['avg_ms'] > 0.05))
