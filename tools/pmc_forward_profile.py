#!/usr/bin/env python
"""Counter passes behind bench.py's `roofline` (run ON THE GPU BOX, e.g. through gpurun):

    python tools/pmc_forward_profile.py OUTDIR [-- extra bench.py args]

Three separate counters-only rocprofv3 runs (--pmc X --kernel-trace, nothing else — MI355X_MICROARCH.md
"rocprofv3 PMC slots": FETCH_SIZE and WRITE_SIZE do not fit one pass) of

    python bench.py --steps 7 --warmup 2 --no-cpu-baseline --no-secondary

for SQ_INSTS_VALU_MFMA_MOPS_F32, FETCH_SIZE and WRITE_SIZE, plus the tile plan of the timed batch
(LNZ_BENCH_DUMP_PLAN).  Writes OUTDIR/forward_pmc.json: per-launch averages of the fused forward
kernel (and of the other kernels of the step), with the guide's gfx950 correction applied to
FETCH_SIZE (x2: it reports half the bytes of wide coalesced reads; WRITE_SIZE is uncalibrated), and
OUTDIR/forward_tile_plan.npz.  Copy both to profiles/ to have them judged.
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
  return name.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]


def run_pass(counter, outdir, extra):
  d = os.path.join(outdir, 'pmc_' + counter)
  env = dict(os.environ, TMPDIR='/tmp')
  if counter == 'SQ_INSTS_VALU_MFMA_MOPS_F32':
    env['LNZ_BENCH_DUMP_PLAN'] = os.path.join(outdir, 'forward_tile_plan.npz')
  cmd = ['rocprofv3', '--pmc', counter, '--kernel-trace', '--output-format', 'csv', '-d', d, '--',
         sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '7', '--warmup', '2',
         '--no-cpu-baseline', '--no-secondary'] + extra
  with open(os.path.join(outdir, 'pmc_%s.log' % counter), 'w') as log:
    subprocess.run(cmd, check=True, cwd='/tmp', env=env, stdout=log, stderr=subprocess.STDOUT)
  cc = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
  kt = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
  dur = {r['Dispatch_Id']: int(r['End_Timestamp']) - int(r['Start_Timestamp'])
         for r in csv.DictReader(open(kt))}
  agg = collections.defaultdict(lambda: dict(v=[], d=[]))
  for r in csv.DictReader(open(cc)):
    if r['Counter_Name'] != counter:
      continue
    k = short(r['Kernel_Name'])
    agg[k]['v'].append(float(r['Counter_Value']))
    agg[k]['d'].append(dur[r['Dispatch_Id']])
  return {k: dict(launches=len(x['v']), per_launch=sum(x['v']) / len(x['v']),
                  avg_duration_us=sum(x['d']) / len(x['d']) / 1e3) for k, x in agg.items()}


def main():
  outdir = os.path.abspath(sys.argv[1])
  extra = sys.argv[3:] if len(sys.argv) > 2 and sys.argv[2] == '--' else []
  os.makedirs(outdir, exist_ok=True)
  res = {c: run_pass(c, outdir, extra) for c in
         ('SQ_INSTS_VALU_MFMA_MOPS_F32', 'FETCH_SIZE', 'WRITE_SIZE')}
  fwd = [k for k in res['SQ_INSTS_VALU_MFMA_MOPS_F32'] if k.startswith('lanczosnet_forward') or k.startswith('lanczosnet_strip')]
  # (the exact-fp32 forward of the step: the opt-in split-precision leg of the same bench run launches
  # its own instantiation more often, but issues 0.5 % of the fp32 matrix instructions)
  fwd = max(fwd, key=lambda k: res['SQ_INSTS_VALU_MFMA_MOPS_F32'][k]['per_launch'])
  mops = res['SQ_INSTS_VALU_MFMA_MOPS_F32'][fwd]
  fetch, write = res['FETCH_SIZE'][fwd], res['WRITE_SIZE'][fwd]
  out = {
      'command': 'bench.py --steps 7 --warmup 2 --no-cpu-baseline --no-secondary ' + ' '.join(extra),
      'kernel': fwd,
      'SQ_INSTS_VALU_MFMA_MOPS_F32_per_launch': mops['per_launch'],
      'mfma_flops_per_launch': mops['per_launch'] * 512.0,
      # one counted "MOP" = 512 flop: v_mfma_f32_16x16x4_f32 (strip / 16 x 16-tile kernels) = 4,
      # v_mfma_f32_32x32x2_f32 (the 32-row-tile kernels) = 8
      'mfma_instruction': 'v_mfma_f32_32x32x2_f32' if fwd.startswith('lanczosnet_forward_kernel') else 'v_mfma_f32_16x16x4_f32',
      'mfma_instructions_per_launch': mops['per_launch'] / (8.0 if fwd.startswith('lanczosnet_forward_kernel') else 4.0),
      'launches': mops['launches'],
      'avg_duration_us_under_counters': mops['avg_duration_us'],
      'FETCH_SIZE_KB': fetch['per_launch'], 'WRITE_SIZE_KB': write['per_launch'],
      'hbm_bytes_per_launch': int(2 * fetch['per_launch'] * 1024 + write['per_launch'] * 1024),
      'corrections': 'FETCH_SIZE x 2 (gfx950 reports 1/2 of wide coalesced reads, '
                     'MI355X_MICROARCH.md HBM section); WRITE_SIZE uncalibrated, taken as is; '
                     'separate counters-only passes',
      'other_kernels': {c: {k: v for k, v in res[c].items() if k != fwd} for c in res},
  }
  json.dump(out, open(os.path.join(outdir, 'forward_pmc.json'), 'w'), indent=1)
  print(json.dumps({k: out[k] for k in ('kernel', 'mfma_instructions_per_launch',
                                        'mfma_flops_per_launch', 'hbm_bytes_per_launch',
                                        'launches')}))


if __name__ == '__main__':
  main()
