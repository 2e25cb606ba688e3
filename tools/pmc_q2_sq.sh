#!/bin/bash
# SQ counters of the dictionary-coded smoother kernel and the wide-band triangular solves (Q2 elasticity 48^3, two levels):
#   bash tools/pmc_q2_sq.sh <outdir-under-gpurun_out>     then   python tools/pmc_summary.py gpurun_out/<outdir>
export TMPDIR=/tmp SAAMGE_AMD_SERIAL=1
R=$PWD; O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-include-regex "sell_gpair|ss_trsolve_win" --kernel-trace --output-format csv -d $O/p$i -o p -- python3 $R/tools/smoother_bench.py 48 2 quick q2 > $O/p$i.log 2>&1
  echo "pass $i rc=$?"
done
