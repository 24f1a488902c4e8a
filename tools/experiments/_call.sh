cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
V=tools/experiments/_variants
for lib in "" $V/liblnz_conv_forward_f16_vform.so; do LANCZOSNET_HIP_LIB=$lib timeout 300 python bench.py --gemm f16x3 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f16x3 fwd [$lib]', d['ms_per_step'], d['config']['stage_ms'], d.get('parity_rel_err'))"; done
for lib in "" $V/liblnz_conv_large_vform.so; do LANCZOSNET_HIP_LIB=$lib timeout 300 python tools/bench_config5.py --reps 2 2>&1 | tail -1 | cut -c1-400; done
for lib in "" $V/liblnz_f16x3_linear_vform.so; do LANCZOSNET_HIP_LIB=$lib timeout 300 python tools/bench_f16x3_linear.py 2>&1 | tail -2 | cut -c1-300; done
