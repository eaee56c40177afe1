#!/usr/bin/env python
"""Summarise a rocprofv3 run (rocpd sqlite .db, `--kernel-trace --stats`) as a small
markdown table for profiles/.  usage: rocprof_summary.py <results.db> <out.md> [title]"""
import sqlite3
import sys


def main(db_path, out_path, title='rocprofv3 --kernel-trace --stats'):
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    with open(out_path, 'w') as f:
        f.write(f'# {title}\n\nsource: `{db_path}` (durations in microseconds)\n\n')
        f.write('| kernel | calls | total us | avg us | % |\n|---|---:|---:|---:|---:|\n')
        for name, calls, total, avg, pct in rows:
            if pct < 0.01:
                continue
            short = name if len(name) < 110 else name[:107] + '...'
            f.write(f'| `{short}` | {calls} | {total:.0f} | {avg:.1f} | {pct:.2f} |\n')
    print(open(out_path).read())


if __name__ == '__main__':
    main(*sys.argv[1:])
