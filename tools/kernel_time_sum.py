"""Sum of kernel durations in a rocprofv3 --kernel-trace run, per step: usage DIR STEPS"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
steps = int(sys.argv[2])
tot = 0.0
rows = []
for r in csv.DictReader(open(f)):
  tot += float(r['TotalDurationNs'])
  rows.append((float(r['TotalDurationNs']), r['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:56], int(r['Calls'])))
print('kernel time per step: %.3f ms (all %d steps incl. warm-up, setup kernels included)' % (tot / steps / 1e6, steps))
for d, n, c in sorted(rows, reverse=True)[:14]:
  print('  %-58s %6.3f ms/step  %5.1f calls/step' % (n, d / steps / 1e6, c / steps))
