#!/bin/bash
# HBM traffic of the fine-level smoother kernel IN THE LIBRARY'S OWN PROCESS (round 4: counter passes over the real path
# exit cleanly): FETCH_SIZE and WRITE_SIZE in separate passes over tools/smoother_bench.py <n> 3 quick (one set-up, then 21
# smoother applications of 10 fused steps on the fine level), summarised into gpurun_out/<out>/pmc_traffic.json.
#   bash tools/pmc_smoother_traffic.sh <outdir-under-gpurun_out> [n]
export TMPDIR=/tmp SAAMGE_AMD_SERIAL=1
R=$PWD; O=$R/gpurun_out/$1; N=${2:-256}
mkdir -p $O
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-include-regex "sell_staged|sell_tiles_kernel" --kernel-trace --output-format csv -d $O/$C -o p -- python3 $R/tools/smoother_bench.py $N 3 quick > $O/$C.log 2>&1
  echo "pass $C rc=$?"
done
cd $R
python3 tools/pmc_smoother_traffic.py $O
