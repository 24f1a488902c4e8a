"""oracle/_ref: the REAL reference operator, built from its own source (test infrastructure).

The only compiled code of the reference on this path is `operators/`.  Its CUDA
file cannot be built here; its CPU file (`operators/src/segment_reduction.cpp`, 57 lines) can:
g++ on that file where it lies under /root/reference, plus our pybind binding
(`oracle/segment_reduction_ref_binding.cpp`) and an empty `THC/THC.h` shim (the include the file no
longer finds in modern torch and does not use).  Nothing of the reference is copied: the output is
one shared object under `oracle/_ref/` (git-ignored, travels to the GPU box with the snapshot).

`build()` runs in the build container (`__graft_entry__.build()` calls it when /root/reference is
present); `load()` imports the prebuilt module anywhere, or returns None.
"""
import glob
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '_ref')
NAME = 'segment_reduction_ref'
REF = os.environ.get('LANCZOS_REFERENCE', '/root/reference')


def build(verbose=False):
  src = os.path.join(REF, 'operators', 'src', 'segment_reduction.cpp')
  if not os.path.exists(src):
    return None
  from torch.utils.cpp_extension import load
  os.makedirs(OUT, exist_ok=True)
  return load(name=NAME, sources=[src, os.path.join(HERE, 'segment_reduction_ref_binding.cpp')],
              extra_include_paths=[os.path.join(HERE, 'ref_shim'), os.path.dirname(src)],
              extra_cflags=['-O2', '-Wno-deprecated-declarations'], build_directory=OUT,
              with_cuda=False, verbose=verbose)


def load():
  """The prebuilt reference operator module, or None when it was never built."""
  so = glob.glob(os.path.join(OUT, NAME + '*.so'))
  if not so:
    return None
  import torch  # noqa: F401  (libtorch must be loaded first)
  spec = importlib.util.spec_from_file_location(NAME, so[0])
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


if __name__ == '__main__':
  m = build(verbose=True)
  print('built' if m is not None else 'reference source not found', file=sys.stderr)
