cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c15
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_graph.py tests/test_graph_runner_dropin.py -m gpu -q 2>&1 | tail -6) > gpurun_out/c15/pytest.log 2>&1
(timeout 300 python tools/bench_ritz_wg.py 2>&1 | grep -v amdgpu | cut -c1-200) > gpurun_out/c15/ritz_wg.log 2>&1
(LANCZOSNET_HIP_LIB=tools/libprobe_ritz_wg.so timeout 300 python tools/ritz_wg_phase_probe.py 2>&1 | tail -10) > gpurun_out/c15/ritz_probe.log 2>&1
tail -4 gpurun_out/c15/pytest.log; cat gpurun_out/c15/ritz_wg.log gpurun_out/c15/ritz_probe.log
