#!/usr/bin/env bash
# rocprofv3 --kernel-trace --stats summary of a command:  tools/kstats.sh <out.txt> <cmd ...>
# (run on the GPU box; writes the top-40 kernels by total time)
out="$1"; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
d=$(mktemp -d /tmp/kstats.XXXX)
# the command runs from /tmp (rocprofv3 wants a writable cwd): make its script path absolute
args=()
for a in "$@"; do
  if [ -f "$root/$a" ]; then args+=("$root/$a"); else args+=("$a"); fi
done
set -- "${args[@]}"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$d" -o r --output-format csv -- "$@" > "$d/stdout.log" 2> "$d/stderr.log")
python3 - "$d" "$out" "$*" <<'PY'
import csv, glob, re, sys
d, out, cmd = sys.argv[1:4]
f = glob.glob(d + '/**/r_kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
if not rows:      # say why instead of writing an empty table
    import os
    print('kstats: no r_kernel_stats.csv under', d, [os.path.relpath(p, d) for p in glob.glob(d + '/**', recursive=True)][:40])
    for log in ('stderr.log', 'stdout.log'):
        try:
            print(f'--- {log} (tail)'); print(open(os.path.join(d, log)).read()[-3000:])
        except OSError:
            pass
with open(out, 'w') as o:
    o.write(f'# rocprofv3 --kernel-trace --stats -- {cmd}  (MI355X)\n')
    o.write(f'{"kernel":100s} {"calls":>6s} {"total_ms":>10s} {"avg_us":>10s} {"%":>6s}\n')
    for r in rows[:40]:
        n = re.sub(r'\(.*', '', r['Name']).replace('void ', '')[:100]
        o.write(f'{n:100s} {int(r["Calls"]):6d} {float(r["TotalDurationNs"])/1e6:10.3f} {float(r["AverageNs"])/1e3:10.2f} {float(r["Percentage"]):6.2f}\n')
print(open(out).read())
PY
