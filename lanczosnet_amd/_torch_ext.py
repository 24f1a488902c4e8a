"""The torch extension (`lanczosnet_amd/csrc/torch_ext.cpp` -> `liblanczosnet_torch.so`, in-tree):
every kernel of the C ABI registered with the dispatcher as `torch.ops.lanczosnet.*` — the
high-level ops of the forward step, `fused_launch` for the argument-block launches and one
`raw_<name>` op per remaining entry point (`csrc/torch_ext_abi.inc`, generated from the header by
`tools/gen_torch_ext.py`).  `build()` compiles it (host code only: g++ against torch's headers, linked to
liblanczosnet_hip.so through $ORIGIN); `load()` registers it, loudly failing when it is not built —
there is no fallback for the ops that go through it."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
EXT_PATH = os.path.join(CSRC, 'liblanczosnet_torch.so')
_loaded = False


def build(force=False):
  import torch
  from torch.utils import cpp_extension as ce
  src = os.path.join(CSRC, 'torch_ext.cpp')
  inc = os.path.join(CSRC, 'torch_ext_abi.inc')   # generated: tools/gen_torch_ext.py
  hdr = os.path.join(os.path.dirname(_HERE), 'include', 'lanczosnet_hip.h')
  lib = os.path.join(CSRC, 'liblanczosnet_hip.so')
  if not force and os.path.exists(EXT_PATH) and \
      os.path.getmtime(EXT_PATH) >= max(os.path.getmtime(p) for p in (src, inc, hdr, lib)):
    return EXT_PATH
  tlib = os.path.join(os.path.dirname(torch.__file__), 'lib')
  inc = ce.include_paths('cuda') if ce.include_paths.__code__.co_argcount else ce.include_paths()
  rocm = os.environ.get('ROCM_HOME', '/opt/rocm')
  cmd = ['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-D__HIP_PLATFORM_AMD__=1', '-DUSE_ROCM=1',
         '-D_GLIBCXX_USE_CXX11_ABI=%d' % int(torch._C._GLIBCXX_USE_CXX11_ABI), '-Wno-deprecated-declarations',
         src, '-o', EXT_PATH]
  cmd += ['-I' + p for p in inc] + ['-I' + os.path.join(rocm, 'include')]
  cmd += ['-L' + tlib, '-ltorch', '-ltorch_cpu', '-lc10', '-lc10_hip', '-ltorch_hip',
          '-L' + CSRC, '-llanczosnet_hip', '-Wl,-rpath,$ORIGIN', '-Wl,-rpath,' + tlib]
  subprocess.run(cmd, check=True)
  return EXT_PATH


def load():
  """Register torch.ops.lanczosnet.* (once).  ImportError if the extension is not built."""
  global _loaded
  if _loaded:
    return
  import torch
  from . import _lib
  if not os.path.exists(EXT_PATH):
    raise ImportError(
        'lanczosnet_amd: torch extension not built: %s is missing. Build it with '
        '`python -c "import __graft_entry__ as g; g.build()"`. There is no fallback for the ops '
        'registered through it.' % EXT_PATH)
  _lib.load()   # the C ABI library first (same HIP runtime as torch's, see _lib.load)
  torch.ops.load_library(EXT_PATH)
  if torch.ops.lanczosnet.abi_version() != _lib.ABI_VERSION:
    raise ImportError('lanczosnet_amd: torch extension / C ABI version mismatch (rebuild)')
  _loaded = True
