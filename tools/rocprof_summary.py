#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite .db or *_kernel_stats.csv) into a small
text file that can be committed under profiles/.   usage: rocprof_summary.py <results.db|stats.csv> <out.txt> [note]"""
import csv
import sqlite3
import sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ''
    rows = []
    if src.endswith('.db'):
        db = sqlite3.connect(src)
        for name, calls, total, avg, pct in db.execute('select name,total_calls,total_duration,average,percentage from top_kernels'):
            rows.append((name, calls, total, avg, pct))
        regs = {}
        for name, v, s, lds in db.execute('select name, max(vgpr_count), max(sgpr_count), max(lds_size) from kernels group by name'):
            regs[name] = (v, s, lds)
    else:
        regs = {}
        with open(src) as f:
            for r in csv.DictReader(f):
                rows.append((r['Name'], int(r['Calls']), float(r['TotalDurationNs']), float(r['AverageNs']), float(r['Percentage'])))
    with open(out, 'w') as f:
        f.write(f'# rocprofv3 --kernel-trace --stats summary  ({note})\n')
        f.write('# durations in microseconds\n')
        f.write(f'{"calls":>6} {"total_us":>14} {"avg_us":>12} {"pct":>7}  {"vgpr":>5} {"sgpr":>5} {"lds":>6}  kernel\n')
        for name, calls, total, avg, pct in rows:
            scale = 1e-3 if src.endswith('.csv') else 1.0
            v, s, lds = regs.get(name, ('', '', ''))
            short = name if len(name) < 150 else name[:147] + '...'
            f.write(f'{calls:>6} {total*scale:>14.1f} {avg*scale:>12.1f} {pct:>7.3f}  {str(v):>5} {str(s):>5} {str(lds):>6}  {short}\n')
    print(open(out).read()[:1500])


if __name__ == '__main__':
    main()
