#!/usr/bin/env bash
# How much of a command's GPU timeline is idle between kernels?  rocprofv3 --kernel-trace, then over the
# LAST `frac` of the dispatches (the timed steps of bench.py): busy time, span, the gap histogram and the
# kernels that most often follow a long gap.   tools/ktrace_gaps.sh <out.txt> <cmd ...>
out="$1"; shift
export TMPDIR=/tmp
d=$(mktemp -d /tmp/ktrace.XXXX)
(cd /tmp && timeout 900 rocprofv3 --kernel-trace -d "$d" -o r --output-format csv -- "$@" > "$d/stdout.log" 2> "$d/stderr.log")
python3 - "$d" "$out" <<'PY'
import collections, csv, glob, sys
d, out = sys.argv[1:3]
f = glob.glob(d + '/**/r_kernel_trace.csv', recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows) // 2:]
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows)
span = int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])
gaps = []
after = collections.Counter()
for a, b in zip(rows, rows[1:]):
    g = int(b['Start_Timestamp']) - int(a['End_Timestamp'])
    gaps.append(g)
    if g > 20000:
        after[(b['Kernel_Name'][:60], a['Kernel_Name'][:40])] += g
lines = [f'dispatches {len(rows)}  span {span / 1e6:.3f} ms  busy {busy / 1e6:.3f} ms  idle {(span - busy) / 1e6:.3f} ms ({100 * (span - busy) / span:.1f} %)']
for lo, hi in ((0, 2000), (2000, 5000), (5000, 20000), (20000, 100000), (100000, 10 ** 12)):
    sel = [g for g in gaps if lo <= g < hi]
    lines.append(f'  gaps {lo / 1e3:6.0f}-{hi / 1e3:<9.0f} us: {len(sel):5d}  total {sum(sel) / 1e6:.3f} ms')
for (k, p), g in after.most_common(12):
    lines.append(f'  {g / 1e6:7.3f} ms idle before {k}  (after {p})')
open(out, 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
PY
tail -2 "$d/stdout.log" | cut -c1-200
