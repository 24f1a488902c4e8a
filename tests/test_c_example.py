"""The drop-in boundary is a C ABI: the header compiles as C99, and a C program with no Python and
no torch in the process (examples/ritz_pairs.c) produces the reference's (D, V)."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, 'include')
LIBDIR = os.path.join(ROOT, 'lanczosnet_amd', 'csrc')


def test_header_is_plain_c99(tmp_path):
  src = tmp_path / 'hdr.c'
  src.write_text('#include "lanczosnet_hip.h"\n'
                 'int main(void) { return (int)sizeof(lnz_forward_args) > 0 ? 0 : 1; }\n')
  for cmd in (['gcc', '-std=c99', '-pedantic', '-Wall', '-Werror'], ['g++', '-std=c++11', '-x', 'c++']):
    subprocess.run(cmd + ['-I', INC, '-c', str(src), '-o', str(tmp_path / 'hdr.o')], check=True)


@pytest.mark.gpu
def test_c_program_on_the_abi_matches_the_reference_pipeline(tmp_path):
  rocm = os.environ.get('ROCM_PATH', '/opt/rocm')
  if not shutil.which('gcc') or not os.path.exists(os.path.join(rocm, 'include', 'hip', 'hip_runtime_api.h')):
    pytest.skip('needs gcc and the ROCm headers')
  exe = tmp_path / 'ritz_pairs'
  subprocess.run(['gcc', '-std=c99', '-Wall', '-D__HIP_PLATFORM_AMD__', '-I', os.path.join(rocm, 'include'),
                  '-I', INC, os.path.join(ROOT, 'examples', 'ritz_pairs.c'), '-L', LIBDIR,
                  '-llanczosnet_hip', '-L', os.path.join(rocm, 'lib'), '-lamdhip64',
                  '-Wl,-rpath,' + LIBDIR, '-Wl,-rpath,' + os.path.join(rocm, 'lib'), '-o', str(exe)],
                 check=True)
  out = subprocess.run([str(exe)], check=True, capture_output=True, text=True, timeout=120).stdout
  rows = [ln for ln in out.splitlines() if ln.startswith('n=')]
  assert len(rows) == 5
  for ln in rows:
    n = int(re.match(r'n=(\d+)', ln).group(1))
    D = np.array([float(x) for x in ln.split('D=')[1].split('|')[0].split()])
    a = np.zeros((n, n))
    for i in range(n - 1):
      a[i, i + 1] = a[i + 1, i] = 1.0
    Dr = oracle.graph_laplacian_eigs(a, 6)[0]           # utils/data_helper.py:197-223 restated
    ref = np.zeros(6)
    ref[:min(n, 6)] = Dr[:min(n, 6)]
    assert np.abs(D - ref).max() < 1e-6, (n, D, ref)
    assert abs(float(ln.split('|v0|^2=')[1]) - 1.0) < 1e-5
  assert 'rc=-3' in out   # LNZ_ENOTSUP with a message, no exception machinery
