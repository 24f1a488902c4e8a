#!/usr/bin/env python
"""One shape of lnz_f16x3_linear in a loop, for rocprofv3 --pmc runs (tools/pmc_summary.py)."""
import sys
import numpy as np
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from lanczosnet_amd import ops  # noqa: E402
M, N, K = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 4096
rs = np.random.RandomState(0)
X = torch.from_numpy(rs.randn(M, K).astype(np.float32)).cuda()
W = torch.from_numpy((rs.randn(N, K) / 64).astype(np.float32)).cuda()
b = torch.zeros(N, device='cuda')
xp, wp = ops.f16x3_split(X), ops.f16x3_pack_weight(W)
outp = torch.zeros((2, 1024, (N + 63) // 64 * 64), dtype=torch.float16, device='cuda')
for _ in range(12):
  ops.f16x3_linear(xp, wp, b, M, N, out_planes=outp)
torch.cuda.synchronize()
