#!/bin/bash
# SQ counters of the fine-level smoother kernel (sell_staged*_kernel) in the library's own process: two passes over
# tools/smoother_bench.py <n> 3 quick, summarised by tools/pmc_summary.py.   bash tools/pmc_sq_smoother.sh <outdir-under-gpurun_out> [n]
export TMPDIR=/tmp SAAMGE_AMD_SERIAL=1
R=$PWD; O=$R/gpurun_out/$1; N=${2:-256}
mkdir -p $O
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/tools/smoother_bench.py $N 3 quick > $O/stats.log 2>&1
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-include-regex "sell_staged" --kernel-trace --output-format csv -d $O/p$i -o p -- python3 $R/tools/smoother_bench.py $N 3 quick > $O/p$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $R
python3 tools/pmc_summary.py $O sell_staged | cut -c1-1600
