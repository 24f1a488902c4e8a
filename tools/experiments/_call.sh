# scratch: the command file of the last gpurun call
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_graph.py tests/test_graph_runner_dropin.py -x -q 2>&1 | tail -2
LANCZOSNET_HIP_LIB=tools/experiments/_variants/liblnz_lanczos_ritz_wg_probe.so timeout 300 python tools/ritz_wg_phase_probe.py workgroup 2>&1 | grep -v amdgpu.ids | head -9
timeout 300 python tools/bench_ritz_wg.py 2>/dev/null > gpurun_out/ritz_wg.jsonl
python -c "
import sys, json
for l in open('gpurun_out/ritz_wg.jsonl'):
    if l.startswith('{'):
        d = json.loads(l); print(d['case'], d['B'], d['N'], {k: v['ms'] for k, v in d.items() if isinstance(v, dict)})
"
