#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc ... --kernel-trace --output-format csv run per kernel:
average duration, effective shader clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) and MFMA-busy
fraction (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * cycles)).  usage: pmc_summary.py DIR [PREFIX]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
cc = glob.glob(d + '/*counter_collection.csv')[0]
kt = glob.glob(d + '/*kernel_trace.csv')[0]
dur = {}
for r in csv.DictReader(open(kt)):
  dur[r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
  k = r['Kernel_Name']
  k = k.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:40]
  agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
  agg[k]['_dur'].append(dur[r['Dispatch_Id']])
for k, v in agg.items():
  m = {c: sum(x) / len(x) for c, x in v.items()}
  line = '%-42s n=%3d dur=%9.1f us' % (k, len(v['_dur']), m['_dur'] / 1e3)
  if 'GRBM_GUI_ACTIVE' in m:
    cyc = m['GRBM_GUI_ACTIVE'] / 8.0
    line += ' clk=%.3f GHz' % (cyc / m['_dur'])
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in m and m['SQ_VALU_MFMA_BUSY_CYCLES'] > 0:
      line += ' mfma_busy=%.3f' % (m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cyc))
  for c in m:
    if c not in ('_dur', 'GRBM_GUI_ACTIVE', 'SQ_VALU_MFMA_BUSY_CYCLES'):
      line += ' %s=%.4g' % (c, m[c])
  print(line)
