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
import csv, glob, os, re, sys
d, out, cmd = sys.argv[1:4]
# KSTATS_MARKER=<kernel name fragment> KSTATS_LAST=<n>: summarise only the dispatches from the start of the
# n-th last marker kernel on (bench.py: one su3_assemble_tah_kernel per trajectory, so KSTATS_LAST =
# --steps selects exactly the timed region and leaves model set-up / warm-up out).
marker, last = os.environ.get('KSTATS_MARKER'), int(os.environ.get('KSTATS_LAST', '0'))
short = lambda n: re.sub(r'\(.*', '', n).replace('void ', '')[:100]
rows, note = [], ''
tr = glob.glob(d + '/**/r_kernel_trace.csv', recursive=True)
if marker and last and tr:
    disp = sorted(csv.DictReader(open(tr[0])), key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(disp) if marker in r['Kernel_Name']]
    if len(marks) >= last:
        t0 = int(disp[marks[-last]]['Start_Timestamp'])
        sel = [r for r in disp if int(r['Start_Timestamp']) >= t0]
        agg = {}
        for r in sel:
            a = agg.setdefault(short(r['Kernel_Name']), [0, 0])
            a[0] += 1
            a[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        tot = sum(a[1] for a in agg.values())
        span = int(sel[-1]['End_Timestamp']) - t0
        rows = [{'Name': k, 'Calls': a[0], 'TotalDurationNs': a[1], 'AverageNs': a[1] / a[0],
                 'Percentage': 100.0 * a[1] / tot} for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])]
        note = (f'# only the dispatches of the last {last} trajectories (from the {last}-th last `{marker}` on): '
                f'{len(sel)} dispatches, kernel time {tot / 1e6:.3f} ms in a span of {span / 1e6:.3f} ms '
                f'({len(disp) - len(sel)} earlier dispatches = model set-up and warm-up, not listed)\n')
if not rows:
    f = glob.glob(d + '/**/r_kernel_stats.csv', recursive=True)
    rows = list(csv.DictReader(open(f[0]))) if f else []
if not rows:      # say why instead of writing an empty table
    print('kstats: no r_kernel_stats.csv under', d, [os.path.relpath(p, d) for p in glob.glob(d + '/**', recursive=True)][:40])
    for log in ('stderr.log', 'stdout.log'):
        try:
            print(f'--- {log} (tail)'); print(open(os.path.join(d, log)).read()[-3000:])
        except OSError:
            pass
with open(out, 'w') as o:
    o.write(f'# rocprofv3 --kernel-trace --stats -- {cmd}  (MI355X)\n' + note)
    o.write(f'{"kernel":100s} {"calls":>6s} {"total_ms":>10s} {"avg_us":>10s} {"%":>6s}\n')
    for r in rows[:40]:
        n = short(r['Name'])
        o.write(f'{n:100s} {int(r["Calls"]):6d} {float(r["TotalDurationNs"])/1e6:10.3f} {float(r["AverageNs"])/1e3:10.2f} {float(r["Percentage"]):6.2f}\n')
print(open(out).read())
PY
