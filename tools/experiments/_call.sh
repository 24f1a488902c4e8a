cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/f2
for v in 1 0; do
rm -rf /tmp/rp_busy$v
(cd /tmp && LNZ_STRIPS=$v LNZ_FORWARD16=1 timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/rp_busy$v -- python $GRAFT_REPO_ROOT/tools/experiments/forward_ab.py > /dev/null 2>&1)
python tools/pmc_summary.py $(dirname $(find /tmp/rp_busy$v -name '*counter_collection.csv' | head -1)) 2>&1 | grep -i "lanczosnet\|spectral\|prepare" > gpurun_out/f2/busy_$v.txt
cat gpurun_out/f2/busy_$v.txt
done
rm -rf /tmp/rp_busy2
(cd /tmp && timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/rp_busy2 -- python $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu-baseline > /dev/null 2>&1)
python tools/pmc_summary.py $(dirname $(find /tmp/rp_busy2 -name '*counter_collection.csv' | head -1)) 2>&1 | grep -i "lanczosnet\|spectral\|prepare" > gpurun_out/f2/busy_bench.txt
cat gpurun_out/f2/busy_bench.txt
