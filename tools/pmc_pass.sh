#!/bin/bash
# One rocprofv3 counter pass over tools/smoother_bench.py (smoother steps only: "quick"), setup without worker threads.
#   bash tools/pmc_pass.sh <outdir-under-gpurun_out> <tag> "<counters>" [n] [extra env assignments...]
export TMPDIR=/tmp SAAMGE_AMD_SERIAL=1
R=$PWD; O=$R/gpurun_out/$1; TAG=$2; C=$3; N=${4:-256}
shift 4 || true
for kv in "$@"; do export "$kv"; done
mkdir -p $O
cd /tmp
timeout -k 10 150 rocprofv3 --pmc $C --kernel-include-regex "sell_" --kernel-trace --output-format csv -d $O/$TAG -o p -- python3 $R/tools/smoother_bench.py $N 3 quick > $O/$TAG.log 2>&1
echo "pass $TAG rc=$?"
