#!/bin/bash
# rocprofv3 counter passes over tools/spmv_lab (GPU box):  bash tools/pmc_lab.sh <outdir-under-gpurun_out> "<counters pass 1>" "<counters pass 2>" ...
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp
i=0
for C in "$@"; do
  i=$((i+1))
  timeout -k 10 90 rocprofv3 --pmc $C --kernel-include-regex "sell_|stream_mix|lab_kernel" --kernel-trace --output-format csv -d $O/p$i -o p -- $R/tools/spmv_lab 257 > $O/p$i.log 2>&1
  echo "pass $i ($C) rc=$?"
done
cd $R
python3 tools/pmc_summary.py $O | cut -c1-1500
