"""Alias of the package under the name the build brief uses (`lanczosnetwork_amd`): the code lives in
`lanczosnet_amd/`; `import lanczosnetwork_amd.model` (ops, dataset, dist, ...) resolves to the very same
modules — no second copy of anything."""
import importlib
import sys

import lanczosnet_amd as _pkg

for _name in ('ops', 'model', 'dataset', 'dist', 'train', 'synthetic', 'utils', 'operators', '_lib', '_torch_ext'):
  sys.modules[__name__ + '.' + _name] = importlib.import_module('lanczosnet_amd.' + _name)
  globals()[_name] = sys.modules[__name__ + '.' + _name]
__all__ = getattr(_pkg, '__all__', [])
