#!/bin/bash
# one GPU call: Ritz-kernel parity tests, the preparation launch against its parts, phase stamps
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q -x -k "ritz or prepare or full_size or end_to_end or pipelined or collate" 2>&1 | tail -15 > gpurun_out/r05_ritz_tests.log
python tools/experiments/prep_parts.py > gpurun_out/r05_prep_parts.json 2>&1
LANCZOSNET_HIP_LIB=tools/experiments/_variants/liblnz_lanczos_ritz_phases.so python tools/ritz_phase_probe.py > gpurun_out/r05_ritz_phases.txt 2>&1
tail -5 gpurun_out/r05_ritz_tests.log; cat gpurun_out/r05_prep_parts.json; tail -32 gpurun_out/r05_ritz_phases.txt
