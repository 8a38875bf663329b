#!/usr/bin/env bash
# Kernel sequence of ONE trajectory of a command (the dispatches between the last two momentum draws,
# l2q::su3_assemble_tah_kernel), with durations:  tools/ktrace_seq.sh <out.txt> <cmd ...>
out="$1"; shift
export TMPDIR=/tmp
d=$(mktemp -d /tmp/ktrace.XXXX)
(cd /tmp && timeout 900 rocprofv3 --kernel-trace -d "$d" -o r --output-format csv -- "$@" > "$d/stdout.log" 2> "$d/stderr.log")
python3 - "$d" "$out" <<'PY'
import csv, glob, re, sys
d, out = sys.argv[1:3]
f = glob.glob(d + '/**/r_kernel_trace.csv', recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'su3_assemble_tah' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
lines = []
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    n = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')[:70]
    lines.append(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  {n}")
lines.append(f'span {(int(rows[b]["Start_Timestamp"]) - t0) / 1e6:.3f} ms, {b - a} dispatches')
open(out, 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
PY
