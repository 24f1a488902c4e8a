"""Diagnostic: where a wave of lnz_f32_linear spends its main loop.  Needs a -DLNZ_F32LIN_STAMP
build of the library (tools/experiments/build_f32_variants.sh stamp_base:"-DLNZ_F32LIN_STAMP ...");
prints, per variant, mean / max over the waves of: cycles waiting for the wave's own copies
(vmcnt(0)), cycles at the workgroup barrier, cycles of the whole main loop."""
import ctypes as C, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import ctypes as C, json, os, sys, torch
sys.path.insert(0, %r)
from lanczosnet_amd import _lib, ops
lib = _lib.load()
M, N, K = 1024, 4096, 4096
x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') / K ** 0.5
out = torch.empty(M, N, device='cuda')
st = torch.zeros(256 * 8 * 4, dtype=torch.int64, device='cuda')
for _ in range(3):
  _lib.check(lib.lnz_f32_linear(ops._ptr(x), K, ops._ptr(w), K, None, 0, M, N, K, ops._ptr(out), N, ops._ptr(st), ops._stream()))
torch.cuda.synchronize()
s = st.view(256 * 8, 4).double().cpu()
vm, bar, tot, T = s[:, 0], s[:, 1], s[:, 2], s[:, 3]
print(json.dumps({'slices': float(T[0]), 'loop_cycles_mean': float(tot.mean()), 'loop_cycles_max': float(tot.max()),
                  'vmcnt_wait_mean': float(vm.mean()), 'vmcnt_wait_max': float(vm.max()),
                  'barrier_wait_mean': float(bar.mean()), 'barrier_wait_max': float(bar.max()),
                  'barrier_frac_mean': float((bar / tot).mean()), 'vm_frac_mean': float((vm / tot).mean()),
                  'groupA_barrier_frac': float((bar / tot).view(256, 8)[:, :4].mean()),
                  'groupB_barrier_frac': float((bar / tot).view(256, 8)[:, 4:].mean()),
                  'ideal_mfma_cycles_per_simd': 128 * 64 * 2 * 32}))
''' % ROOT
for lib in sorted(glob.glob(os.path.join(ROOT, 'tools', 'experiments', '_variants', 'liblnz_f32_stamp*.so'))):
  print('=== ', os.path.basename(lib), flush=True)
  r = subprocess.run([sys.executable, '-c', CODE], env=dict(os.environ, LANCZOSNET_HIP_LIB=lib),
                     capture_output=True, text=True, timeout=300)
  print(r.stdout.strip() or r.stderr[-2000:], flush=True)
