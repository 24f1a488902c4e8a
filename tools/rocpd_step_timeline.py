#!/usr/bin/env python
"""One step of a repeated workload as a kernel timeline from rocprofv3's rocpd database: the
launches between the last two occurrences of an anchor kernel (start offset, duration, gap to the
previous kernel's end, name).  usage: rocpd_step_timeline.py DB ANCHOR_SUBSTRING [OUT.txt]"""
import sqlite3, sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
start = 'start' if 'start' in cols else 'start_ns'
end = 'end' if 'end' in cols else 'end_ns'
ks = list(db.execute('select name, %s, %s from kernels order by %s' % (start, end, start)))
idx = [i for i, k in enumerate(ks) if sys.argv[2] in k[0]]
a, b = idx[-2], idx[-1]
out = open(sys.argv[3], 'w') if len(sys.argv) > 3 else sys.stdout
t0, prev_end, busy = ks[a][1], ks[a][1], 0
for name, s, e in ks[a:b]:
  short = name.replace('(anonymous namespace)::', '').replace('void ', '').replace('at::native::', '')
  out.write('%9.1f us  dur %8.1f  gap %6.1f  %s\n' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, short[:110]))
  prev_end = max(prev_end, e)
  busy += e - s
out.write('step: %.1f us wall, %.1f us in %d kernels\n' % ((ks[b][1] - t0) / 1e3, busy / 1e3, b - a))
