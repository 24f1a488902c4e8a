#!/bin/bash
# rocprofv3 --kernel-trace --stats of one command, per-kernel table to stdout (and, with a name as
# $RP_OUT, the csv under gpurun_out/):   tools/rp_stats.sh <name> <command...>
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
name=$1; shift
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/rp_$name
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/rp_$name -o $name -- "$@" > /tmp/rp_$name.log 2>&1
db=$(find /tmp/rp_$name -name '*_results.db' | head -1)
mkdir -p $ROOT/gpurun_out
[ -n "$db" ] && python $ROOT/tools/rocpd_kernel_stats.py $db $ROOT/gpurun_out/${name}_kernel_stats.csv
grep '^{' /tmp/rp_$name.log | tail -2
head -25 $ROOT/gpurun_out/${name}_kernel_stats.csv
