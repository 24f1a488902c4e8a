#!/usr/bin/env python
"""Where a kernel's spill traffic sits: scratch accesses and SGPR-spill lane moves
(v_readlane / v_writelane) of every kernel of a .hip file, counted by the depth of the loop whose
body they are in (the assembler comments `Loop Header: Depth=N` / `in Loop: ... Depth=N` that hipcc
-S emits), next to the MFMA count of the same depth.  A spill at depth 1 of the fused forward runs
once per conv layer; one at depth >= 3 would sit in a GEMM loop.
    python tools/spill_depth.py lanczosnet_amd/csrc/conv_forward.hip [-DFLAGS ...]"""
import os
import re
import subprocess
import sys
import tempfile

src = sys.argv[1]
with tempfile.TemporaryDirectory() as d:
  out = os.path.join(d, 'k.s')
  subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
                  '-Wno-pass-failed', '-S', '--cuda-device-only', '-I' + os.path.dirname(os.path.abspath(src)),
                  src, '-o', out] + sys.argv[2:], check=True, stderr=subprocess.DEVNULL)
  lines = open(out).read().split('\n')
starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
ends = [i for i, l in enumerate(lines) if l.startswith('.Lfunc_end')]
for i in starts:
  e = min(x for x in ends if x > i)
  name = subprocess.run(['c++filt', lines[i].split(':')[0]], capture_output=True, text=True).stdout.strip()
  name = re.sub(r'\(anonymous namespace\)::|\(lnz_forward_args\)|^void ', '', name)[:64]
  cur, hist = 0, {}
  for b in lines[i:e]:
    m = re.search(r'Loop Header: Depth=(\d+)', b) or re.search(r';\s+in Loop: Header=\S+ Depth=(\d+)', b)
    if m:
      cur = int(m.group(1))
    for key, tag in (('scratch_', 'scratch'), ('v_readlane', 'lane'), ('v_writelane', 'lane'), ('v_mfma', 'mfma')):
      if key in b:
        hist.setdefault(cur, {}).setdefault(tag, 0)
        hist[cur][tag] += 1
  print(name)
  for depth in sorted(hist):
    h = hist[depth]
    print('   depth %d: %4d MFMA  %3d scratch accesses  %4d SGPR-spill lane moves'
          % (depth, h.get('mfma', 0), h.get('scratch', 0), h.get('lane', 0)))
