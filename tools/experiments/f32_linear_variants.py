"""Run tools/bench_f32_linear.py (+ a repeatability check at awkward K) once per A/B build of the
library under tools/experiments/_variants/ (build_f32_variants.sh), one subprocess each."""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHECK = r'''
import os, sys, torch
sys.path.insert(0, %r)
from lanczosnet_amd import ops
torch.manual_seed(0)
bad = 0
for (M, N, K) in ((1024, 4096, 4064), (1024, 1056, 4096), (512, 4096, 832), (128, 128, 64), (128, 128, 96), (200, 300, 4096)):
  x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') / K ** 0.5
  ref = x.double() @ w.double().t()
  outs = [ops.f32_linear(x, w, None, relu=False) for _ in range(5)]
  e = float((outs[0].double() - ref).abs().max() / ref.abs().max())
  same = all(torch.equal(outs[0], o) for o in outs[1:])
  if e > 5e-6 or not same: bad += 1
  print('check', M, N, K, 'err %%.2e' %% e, 'repeatable', same)
print('CHECK', 'FAILED' if bad else 'ok')
''' % ROOT
libs = sorted(glob.glob(os.path.join(ROOT, 'tools', 'experiments', '_variants', 'liblnz_f32_*.so')))
only = sys.argv[1:]
for lib in libs:
  name = os.path.basename(lib)[len('liblnz_f32_'):-3]
  if only and name not in only:
    continue
  env = dict(os.environ, LANCZOSNET_HIP_LIB=lib)
  print('=== variant', name, flush=True)
  for rep in range(2):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'bench_f32_linear.py')], env=env,
                       capture_output=True, text=True, timeout=600)
    print(r.stdout.strip() or r.stderr[-1500:], flush=True)
  r = subprocess.run([sys.executable, '-c', CHECK], env=env, capture_output=True, text=True, timeout=600)
  print(r.stdout.strip() or r.stderr[-1500:], flush=True)
