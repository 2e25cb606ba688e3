#!/bin/bash
# SQ counters of the level-0 topology kernels (one bench step, setup without worker threads).
#   bash tools/pmc_topology.sh <tag>
set -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp
export SAAMGE_AMD_SERIAL=1
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $C --kernel-include-regex "ae_to_dof|row_sort|d2e_|key_|elem_ldof|d2ae_|ae_pack|mis_insert|mis_rep|topo_check" \
      --kernel-trace --output-format csv -d $O/p$i -o p -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-general --no-others --warmup 0 --steps 1 > $O/p$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $R
python3 tools/pmc_summary.py $O | cut -c1-1200 > $O/counters.txt
