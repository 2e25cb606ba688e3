#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/<tag>/ (summaries are copied to profiles/ afterwards).
#   bash tools/collect_profiles.sh r03 [part ...]      parts: stats pmc sq timing bench workloads (default: all but workloads)
# Every rocprofv3 pass runs under a timeout as a GUARD only.  Until round 3 counter passes aborted at process exit
# ("stream_stack.cpp: Check failed", then a hang in the signal handler): the library's static buffers were destroyed
# after main() and called hipEventRecord into a runtime that was already torn down.  Round 4: no static object of the
# library owns device memory through a destructor any more (csrc/eig.hip arena(), assemble.hip, hierarchy.hip).
set -o pipefail
export TMPDIR=/tmp
R=$PWD
TAG=${1:-r03}
shift || true
PARTS=${@:-stats pmc sq timing bench}
O=$R/gpurun_out/$TAG
mkdir -p $O
( while sleep 45; do echo "[heartbeat] $(date +%T)"; done ) &
HB=$!
trap 'kill $HB 2>/dev/null' EXIT
has() { [[ " $PARTS " == *" $1 "* ]]; }

if has stats; then
  cd /tmp
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-general --no-others --warmup 1 --steps 2 > $O/stats_bench.json 2> $O/stats_bench.err
  echo "stats rc=$?"
  cd $R
fi
if has pmc; then
  # HBM-side traffic and L2 behaviour of the SpMV family on the headline's fine-level operator shape (tools/spmv_lab:
  # 257^3 rows, 27-point stencil, essential boundary eliminated -- the kernels and sizes of bench.py's level 0)
  bash tools/pmc_lab.sh $TAG/pmc_lab "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
      "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
      "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" > $O/pmc_lab.txt 2>&1
  echo "pmc lab done"
fi
if has sq; then
  # SQ counters of the setup kernels (one bench step, setup without worker threads: SAAMGE_AMD_SERIAL)
  cd /tmp
  export SAAMGE_AMD_SERIAL=1
  i=0
  for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout -k 10 150 rocprofv3 --pmc $C --kernel-include-regex "chol_band_lds2|ae_rows8|ae_build|rap_numeric|rap_symbolic|ss_trsolve|gj_panel|gj_apply|coarse_elmat|chol_panel|band_trail_mfma|sbr_fused|mis_svd|ss_solve_lds|ss_rr" \
        --kernel-trace --output-format csv -d $O/sq/p$i -o p -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-general --no-others --warmup 0 --steps 1 > $O/sq_p$i.log 2>&1
    echo "sq pass $i rc=$?"
  done
  unset SAAMGE_AMD_SERIAL
  cd $R
  python3 tools/pmc_summary.py $O/sq | cut -c1-1200 > $O/sq_counters.txt
fi
if has timing; then
  SAAMGE_AMD_TIMING=1 python3 bench.py --no-cpu-baseline --no-roofline --no-general --no-others --warmup 1 --steps 1 2> $O/phase_timing.err > /dev/null
  grep TIMING $O/phase_timing.err | tail -31 > $O/phase_timing_256.txt
  echo "timing done"
fi
if has bench; then
  python3 bench.py > $O/bench_poisson256.json 2> $O/bench_poisson256.err
  echo "default bench done rc=$?"
fi
if has workloads; then
  python3 bench.py --workload poisson128 --no-cpu-baseline --warmup 1 --steps 5 > $O/bench_poisson128.json 2> $O/bench_poisson128.err
  python3 bench.py --workload aniso128 --no-cpu-baseline --warmup 1 --steps 5 > $O/bench_aniso128.json 2> $O/bench_aniso128.err
  echo "128 workloads done"
  python3 bench.py --workload elasticity_q2 --no-cpu-baseline --warmup 1 --steps 3 > $O/bench_elasticity_q2.json 2> $O/bench_elasticity_q2.err
  echo "elasticity done rc=$?"
  cd /tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_aniso -o an -- python3 $R/bench.py --workload aniso128 --no-cpu-baseline --no-roofline --warmup 1 --steps 2 > /dev/null 2> $O/stats_aniso.err
  cd $R
  echo "workload stats done"
fi
ls $O
