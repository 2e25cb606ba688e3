#!/bin/bash
# PMC / kernel-trace passes over tools/smoother_bench.py (fine-level smoother steps in isolation).  Usage (GPU box):
#   bash tools/pmc_smoother.sh <outdir-under-gpurun_out> [n]
set -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-pmc_smoother}
N=${2:-256}
mkdir -p $O
( while sleep 45; do echo "[heartbeat] $(date +%T)"; done ) &
HB=$!
trap 'kill $HB 2>/dev/null' EXIT
cd /tmp
export SAAMGE_AMD_SERIAL=1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/tools/smoother_bench.py $N 3 pcg1 > $O/stats.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-include-regex "sell_spmv|pcg_update" --kernel-trace --output-format csv -d $O/pmc_$tag -o p -- python3 $R/tools/smoother_bench.py $N 3 pcg1 > $O/pmc_$tag.log 2>&1 || echo "pass $c failed"
  echo "pass $c done"
done
cd $R
python3 tools/pmc_summary.py $O > $O/summary.txt 2>&1
cat $O/summary.txt
