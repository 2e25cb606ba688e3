#!/bin/bash
# One rocprofv3 --pmc pass over a small run: the process must exit by itself with rc 0 (no "stream_stack.cpp: Check
# failed" abort at exit, no kill by the guard timeout).   bash tools/pmc_exit_check.sh <outdir-under-gpurun_out>
export TMPDIR=/tmp SAAMGE_AMD_SERIAL=1
R=$PWD; O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp
T0=$(date +%s)
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "sell_" --kernel-trace --output-format csv -d $O/exitcheck -o p -- python3 $R/tools/smoother_bench.py 64 3 quick > $O/exitcheck.log 2>&1
RC=$?
T1=$(date +%s)
echo "pmc exit check: rc=$RC after $((T1-T0)) s; 'Check failed' lines: $(grep -c 'Check failed' $O/exitcheck.log)" | tee $O/exitcheck_result.txt
