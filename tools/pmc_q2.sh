#!/bin/bash
# HBM traffic of the dictionary-coded smoother kernel (Q2 elasticity 48^3, two levels): FETCH_SIZE and WRITE_SIZE in
# separate passes, each under a timeout (a guard only: the exit-time abort of rounds 1-3 was the library's own static destructors calling HIP, fixed in round 4).
#   bash tools/pmc_q2.sh <outdir-under-gpurun_out>
export TMPDIR=/tmp SAAMGE_AMD_SERIAL=1
R=$PWD; O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-include-regex "sell_gpair" --kernel-trace --output-format csv -d $O/$C -o p -- python3 $R/tools/smoother_bench.py 48 2 quick q2 > $O/$C.log 2>&1
  echo "pass $C rc=$?"
done
