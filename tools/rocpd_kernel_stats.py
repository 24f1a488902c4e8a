#!/usr/bin/env python
"""rocprofv3's rocpd database (`<dir>/<name>_results.db`, the default output of
`rocprofv3 --kernel-trace --stats -d <dir> -o <name> -- cmd`) -> the per-kernel statistics table
in the column layout of rocprofv3's `*_kernel_stats.csv`.  usage: rocpd_kernel_stats.py DB [OUT.csv]"""
import csv, math, sqlite3, sys

db = sqlite3.connect(sys.argv[1])
rows = {}
for name, dur in db.execute('select name, duration from kernels'):
  rows.setdefault(name, []).append(int(dur))
total = sum(sum(v) for v in rows.values())
out = csv.writer(open(sys.argv[2], 'w', newline='') if len(sys.argv) > 2 else sys.stdout,
                 quoting=csv.QUOTE_NONNUMERIC)
out.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'StdDev'])
for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
  n, s = len(v), sum(v)
  mean = s / n
  sd = math.sqrt(sum((x - mean) ** 2 for x in v) / (n - 1)) if n > 1 else 0.0
  out.writerow([name, n, s, round(mean, 6), round(100.0 * s / total, 4), min(v), max(v), round(sd, 6)])
