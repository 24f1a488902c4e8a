#!/usr/bin/env python
"""lnz_f16x3_linear at the shapes of AdaLanczosNet's filter MLPs (M = batch = 1024): launch time and
the rate in fp16-product flops (3 products per multiply-add pair), next to the library form of the
same arithmetic (one fp16 GEMM of three times the depth + the split kernel)."""
import json
import sys
import numpy as np
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from lanczosnet_amd import ops  # noqa: E402


def timeit(fn, reps=20):
  fn(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps


rs = np.random.RandomState(0)
M = 1024
for N, K in ((4096, 832), (4096, 4096), (1056, 4096)):
  X = torch.from_numpy(rs.randn(M, K).astype(np.float32)).cuda()
  W = torch.from_numpy((rs.randn(N, K) / np.sqrt(K)).astype(np.float32)).cuda()
  b = torch.from_numpy(rs.randn(N).astype(np.float32)).cuda()
  xp, wp = ops.f16x3_split(X), ops.f16x3_pack_weight(W)
  outp = torch.zeros((2, 1024, (N + 63) // 64 * 64), dtype=torch.float16, device='cuda')
  outf = torch.empty((M, N), dtype=torch.float32, device='cuda')
  t_planes = timeit(lambda: ops.f16x3_linear(xp, wp, b, M, N, out_planes=outp))
  t_f32 = timeit(lambda: ops.f16x3_linear(xp, wp, b, M, N, relu=False, out_f32=outf))
  x3, w3 = ops.split_f16x3(X), ops.split_weight_f16x3(W)
  t_lib = timeit(lambda: torch.mm(x3, w3.t(), out_dtype=torch.float32))
  h = torch.mm(x3, w3.t(), out_dtype=torch.float32)
  t_split = timeit(lambda: ops.split_f16x3(h, bias=b, alpha=1 / 1024.0, relu=True))
  fl = 2.0 * M * N * ((K + 63) // 64 * 64) * 3
  print(json.dumps(dict(M=M, N=N, K=K, hand_written_ms=round(t_planes, 4), hand_written_f32_out_ms=round(t_f32, 4),
                        hand_written_PFLOPs=round(fl / t_planes / 1e12, 3), library_gemm_ms=round(t_lib, 4),
                        library_split_ms=round(t_split, 4),
                        library_PFLOPs_gemm_only=round(fl / t_lib / 1e12, 3))))
