#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max.
usage: rocpd_summary.py results.db [out.txt]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r'\(.*', '', name)                 # drop the argument list
    name = name.replace('void ', '')
    return name if len(name) <= 110 else name[:107] + '...'


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = [f'{"kernel":110s} {"calls":>6s} {"total_ms":>10s} {"avg_us":>10s} {"min_us":>9s} '
             f'{"max_us":>9s} {"%":>6s} {"vgpr":>5s} {"agpr":>5s} {"sgpr":>5s} {"lds":>7s} {"scr":>5s}']
    for n, c, t, a, mn, mx, vg, ag, sg, lds, scr in rows:
        lines.append(f'{short(n):110s} {c:6d} {t/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:9.2f} '
                     f'{mx/1e3:9.2f} {100*t/tot:6.2f} {vg or 0:5d} {ag or 0:5d} {sg or 0:5d} '
                     f'{lds or 0:7d} {scr or 0:5d}')
    lines.append(f'total kernel time: {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches')
    txt = '\n'.join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(txt + '\n')
    print(txt)


if __name__ == '__main__':
    main()
