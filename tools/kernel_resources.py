#!/usr/bin/env python
"""Register / spill table of every kernel of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage,
cross-compiles without a GPU):   python tools/kernel_resources.py lanczosnet_amd/csrc/conv_forward.hip"""
import re
import subprocess
import sys


def table(src, extra=()):
  cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-pass-failed',
         '-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', '/dev/null'] + list(extra)
  err = subprocess.run(cmd, capture_output=True, text=True).stderr
  rows, cur = [], None
  for line in err.splitlines():
    m = re.search(r'remark:\s+([\w \[\]/]+?):\s+(\S+) \[-Rpass', line)
    if not m:
      continue
    k, v = m.group(1).strip(), m.group(2)
    if k == 'Function Name':
      cur = {'name': v}
      rows.append(cur)
    elif cur is not None:
      cur[k] = v
  return rows


if __name__ == '__main__':
  for r in table(sys.argv[1], sys.argv[2:]):
    name = subprocess.run(['c++filt', r['name']], capture_output=True,
                          text=True).stdout.strip()
    name = re.sub(r'\(anonymous namespace\)::|\(lnz_forward_args\)|^void ', '', name)
    print('%-58s vgpr %3s agpr %3s sgpr %3s | spill v %3s s %3s scratch %4s B | occ %s lds %s' % (
        name[:58], r.get('VGPRs'), r.get('AGPRs'), r.get('TotalSGPRs'), r.get('VGPRs Spill'),
        r.get('SGPRs Spill'), r.get('ScratchSize [bytes/lane]'), r.get('Occupancy [waves/SIMD]'),
        r.get('LDS Size [bytes/block]')))
