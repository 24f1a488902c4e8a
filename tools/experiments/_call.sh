cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
V=tools/experiments/_variants
for t in 2 1; do
LNZ_GAINS_TILES=$t timeout 200 python tools/bench_gains.py 200 2>&1 | tail -1 | sed "s/^/tiles=$t base /"
LNZ_GAINS_TILES=$t LANCZOSNET_HIP_LIB=$V/liblnz_spectral_gains_il.so timeout 200 python tools/bench_gains.py 200 2>&1 | tail -1 | sed "s/^/tiles=$t interleave /"
done
