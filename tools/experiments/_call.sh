cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
V=tools/experiments/_variants
timeout 200 python tools/experiments/forward_ab.py 2>&1 | tail -1
LANCZOSNET_HIP_LIB=$V/liblnz_conv_strip_prio1.so timeout 200 python tools/experiments/forward_ab.py 2>&1 | tail -1
timeout 200 python tools/experiments/forward_ab.py 2>&1 | tail -1
LANCZOSNET_HIP_LIB=$V/liblnz_conv_strip_prio1.so timeout 200 python tools/experiments/forward_ab.py 2>&1 | tail -1
PROBE_NAMES="prologue layer-head long-block lift edge-gemm1 gemm2 epilogue head total subtiles" LANCZOSNET_HIP_LIB=$V/liblnz_conv_strip_phases.so timeout 300 python tools/phase_probe16.py 2>&1 | tail -4
