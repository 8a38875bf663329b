#!/usr/bin/env bash
# Per-dispatch durations of the fp64 heads kernel (rocprofv3 --kernel-trace) for a command, in launch
# order, with the gap to the previous kernel's end:  tools/ktrace_heads.sh <out.txt> <cmd ...>
out="$1"; shift
export TMPDIR=/tmp
d=$(mktemp -d /tmp/ktrace.XXXX)
(cd /tmp && timeout 900 rocprofv3 --kernel-trace -d "$d" -o r --output-format csv -- "$@" > "$d/stdout.log" 2> "$d/stderr.log")
python3 - "$d" "$out" <<'PY'
import csv, glob, sys
d, out = sys.argv[1:3]
f = glob.glob(d + '/**/r_kernel_trace.csv', recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r['Start_Timestamp']))
prev_end, prev_name = None, ''
lines = []
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if 'fused_heads' in r['Kernel_Name']:
        lines.append(f"{(e - s) / 1e3:9.1f} us  gap {((s - prev_end) / 1e3 if prev_end else 0):8.1f} us  after {prev_name[:50]}")
    prev_end, prev_name = e, r['Kernel_Name']
open(out, 'w').write('\n'.join(lines) + '\n')
print(f'{len(lines)} heads dispatches'); print('\n'.join(lines[:70]))
PY
tail -5 "$d/stdout.log"
