"""Per-kernel summary (calls, total, average, share) of a `rocprofv3 --kernel-trace --stats` run whose output is the rocpd
SQLite database (the default output format of ROCm 7.2's rocprofv3): prints the CSV that `--output-format csv` would have
written as *_kernel_stats.csv.  Usage: python profiles/kernel_stats_from_db.py RESULTS.db > profiles/rNN_x_kernel_stats.csv"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    rows = list(cur.execute('select %s, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) '
                            'from kernels group by %s order by 3 desc' % (name, name)))
    total = float(sum(r[2] for r in rows)) or 1.0
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for n, c, t, a, lo, hi in rows:
        print('"%s",%d,%d,%.1f,%.2f,%d,%d' % (n, c, t, a, 100.0 * t / total, lo, hi))


if __name__ == '__main__':
    main(sys.argv[1])
