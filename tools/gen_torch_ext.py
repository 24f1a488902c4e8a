#!/usr/bin/env python
"""Generate lanczosnet_amd/csrc/torch_ext_abi.inc from include/lanczosnet_hip.h: one dispatcher op
`torch.ops.lanczosnet.raw_<name>` per C entry point `lnz_<name>` (the four launches that take a
`lnz_forward_args` block are written by hand in torch_ext.cpp).

Mapping of a C parameter to the op schema:
  lnz_stream_t                     dropped: the CURRENT HIP stream of the calling thread
  int / int64_t ; float / double   int ; float
  const int32_t* <name>_host       int[]?     host array (None -> NULL)
  const float* const*              Tensor[]   host array of device pointers
  const T*                         Tensor?    input  (None -> NULL); dtype of T checked
  T*                               Tensor(x!)? output / in-out, written in place
An entry with a stream returns its status code, which the wrapper turns into an exception carrying
lnz_last_error() (no return value); an entry without one is a host-side size query returning int.

    python tools/gen_torch_ext.py            # rewrite the .inc   (tests/test_host_cpu.py checks it is current)
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, 'include', 'lanczosnet_hip.h')
OUT = os.path.join(ROOT, 'lanczosnet_amd', 'csrc', 'torch_ext_abi.inc')
SKIP = {'lnz_abi_version', 'lnz_last_error', 'lnz_last_kernel', 'lnz_stream_create_cu_masked'}
DTYPE = {'float': 'at::kFloat', 'double': 'at::kDouble', 'int32_t': 'at::kInt', 'uint32_t': 'at::kInt',
         'int64_t': 'at::kLong', 'uint8_t': 'at::kByte', 'unsigned long long': 'at::kLong'}


def prototypes():
  h = re.sub(r'/\*.*?\*/', '', open(HDR).read(), flags=re.S)
  out = []
  for ret, name, args in re.findall(r'\b(int|int64_t)\s+(lnz_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;', h, flags=re.S):
    args = ' '.join(args.split())
    params = []
    for p in ([a.strip() for a in args.split(',')] if args not in ('', 'void') else []):
      m = re.match(r'(.*?)(\w+)$', p)
      params.append((m.group(1).strip(), m.group(2)))
    out.append((ret, name, params))
  return out


def generate():
  defs, impls, body = [], [], []
  for ret, name, params in prototypes():
    if name in SKIP or any('lnz_forward_args' in t for t, _ in params):
      continue
    op = 'raw_' + name[4:]
    has_stream = any(t == 'lnz_stream_t' for t, _ in params)
    schema, cargs, call, pre = [], [], [], []
    alias = iter('abcdefghijklmnopqrstuvwxyz')
    tensors = []
    for t, n in params:
      if t == 'lnz_stream_t':
        call.append('cur_stream()')
      elif t in ('int', 'int64_t'):
        schema.append('int %s' % n)
        cargs.append('int64_t %s' % n)
        call.append('(%s)%s' % (t, n))
      elif t in ('float', 'double'):
        schema.append('float %s' % n)
        cargs.append('double %s' % n)
        call.append('(%s)%s' % (t, n))
      elif t == 'const int32_t*' and n.endswith('_host'):
        schema.append('int[]? %s' % n)
        cargs.append('at::OptionalIntArrayRef %s' % n)
        pre.append('  std::vector<int32_t> %s_v;\n  if (%s.has_value()) %s_v.assign(%s->begin(), %s->end());'
                   % (n, n, n, n, n))
        call.append('%s.has_value() ? %s_v.data() : nullptr' % (n, n))
      elif t == 'const float* const*':
        schema.append('Tensor[] %s' % n)
        cargs.append('at::TensorList %s' % n)
        pre.append('  std::vector<const float*> %s_v;\n  for (const Tensor& t_ : %s) {\n'
                   '    need(t_, at::kFloat, "%s[i]");\n    %s_v.push_back(t_.data_ptr<float>());\n  }'
                   % (n, n, n, n))
        call.append('%s_v.data()' % n)
        tensors.append('(%s.empty() ? nullptr : &%s[0])' % (n, n))
      elif t.endswith('*'):
        const = t.startswith('const ')
        base = t[6:-1].strip() if const else t[:-1].strip()
        schema.append('Tensor%s? %s' % ('' if const else '(%s!)' % next(alias), n))
        cargs.append('const c10::optional<Tensor>& %s' % n)
        if base == 'void':
          chk = 'raw_any(%s, "%s")' % (n, n)
        elif base == 'uint16_t':
          chk = 'raw_2byte(%s, "%s")' % (n, n)
        else:
          chk = 'raw_ptr(%s, %s, "%s")' % (n, DTYPE[base], n)
        call.append('(%s%s*)%s' % ('const ' if const else '', base, chk))
        tensors.append('(%s.has_value() ? &*%s : nullptr)' % (n, n))
      else:
        raise SystemExit('gen_torch_ext: unhandled parameter type %r in %s' % (t, name))
    rets = '()' if has_stream else 'int'
    defs.append('  m.def("%s(%s) -> %s");' % (op, ', '.join(schema), rets))
    lines = ['%s %s(%s) {' % ('void' if has_stream else 'int64_t', op, ', '.join(cargs))]
    lines += pre
    if has_stream:
      lines.append('  const Tensor* first_ = first_defined({%s});' % ', '.join(tensors))
      lines.append('  c10::OptionalDeviceGuard guard_;')
      lines.append('  if (first_) guard_.reset_device(first_->device());')
      lines.append('  check(%s(%s), "%s");' % (name, ', '.join(call), name[4:]))
    else:
      lines.append('  return (int64_t)%s(%s);' % (name, ', '.join(call)))
    lines.append('}')
    body.append('\n'.join(lines))
    impls.append(('  m.impl("%s", %s);' % (op, op), has_stream and bool(tensors), op))
  txt = ['// GENERATED by tools/gen_torch_ext.py from include/lanczosnet_hip.h — do not edit.',
         '// One dispatcher op torch.ops.lanczosnet.raw_<name> per C entry point lnz_<name>.', '']
  txt += body
  txt += ['', '#define LNZ_RAW_DEFS(m) \\'] + [d + ' \\' for d in defs] + ['', '']
  dev = [i for i, on_dev, _ in impls if on_dev]
  host = [i for i, on_dev, _ in impls if not on_dev]
  txt += ['#define LNZ_RAW_IMPLS_DEVICE(m) \\'] + [d + ' \\' for d in dev] + ['', '']
  txt += ['#define LNZ_RAW_IMPLS_HOST(m) \\'] + [d + ' \\' for d in host] + ['', '']
  return '\n'.join(txt)


if __name__ == '__main__':
  txt = generate()
  if '--check' in sys.argv:
    sys.exit(0 if os.path.exists(OUT) and open(OUT).read() == txt else 1)
  open(OUT, 'w').write(txt)
  print('wrote', OUT, txt.count('m.def('), 'ops')
