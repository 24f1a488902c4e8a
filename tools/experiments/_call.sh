# scratch: the command file of the last gpurun call
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "large or graph or general" 2>&1 | tail -2
rm -rf gpurun_out/graphleg; mkdir -p gpurun_out/graphleg
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/graphleg -o gl -- python $GRAFT_REPO_ROOT/tools/experiments/graph_leg.py 20 2>&1 | grep "laplacian_l4"
cd $GRAFT_REPO_ROOT
python tools/rocpd_kernel_stats.py gpurun_out/graphleg/gl_results.db gpurun_out/graphleg/stats.csv > /dev/null 2>&1
rm -f gpurun_out/graphleg/gl_results.db
