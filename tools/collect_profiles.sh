#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/r02/ (copied to profiles/ afterwards):
# rocprofv3 kernel stats of the default bench, the two PMC traffic passes, phase timing, bench lines of every workload.
set -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r02}
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r02 -- python3 $R/bench.py --no-cpu-baseline --no-roofline --warmup 1 --steps 2 > $O/stats_bench.json 2> $O/stats_bench.err
echo "stats done" 
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --no-cpu-baseline --no-roofline --warmup 0 --steps 1 > /dev/null 2> $O/pmc_fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --no-cpu-baseline --no-roofline --warmup 0 --steps 1 > /dev/null 2> $O/pmc_write.err
echo "write done"
cd $R
python3 tools/pmc_traffic.py $(ls $O/pmc_fetch/*counter_collection.csv | head -1) $(ls $O/pmc_write/*counter_collection.csv | head -1) $O/pmc_traffic.json > $O/pmc_traffic.txt
SAAMGE_AMD_TIMING=1 python3 bench.py --no-cpu-baseline --no-roofline --warmup 1 --steps 1 2> $O/phase_timing.err > /dev/null
grep TIMING $O/phase_timing.err | tail -31 > $O/phase_timing_256.txt
echo "timing done"
python3 bench.py > $O/bench_poisson256.json 2> $O/bench_poisson256.err
echo "default bench done"
python3 bench.py --workload poisson128 --no-cpu-baseline > $O/bench_poisson128.json 2> $O/bench_poisson128.err
python3 bench.py --workload aniso128 --no-cpu-baseline > $O/bench_aniso128.json 2> $O/bench_aniso128.err
python3 bench.py --workload elasticity_q2 --no-cpu-baseline --warmup 0 > $O/bench_elasticity_q2.json 2> $O/bench_elasticity_q2.err
echo "workloads done"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_aniso -o an -- python3 $R/bench.py --workload aniso128 --no-cpu-baseline --no-roofline --warmup 1 --steps 2 > /dev/null 2> $O/stats_aniso.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_el -o el -- python3 $R/bench.py --workload elasticity_q2 --no-cpu-baseline --no-roofline --warmup 0 --steps 1 > /dev/null 2> $O/stats_el.err
cd $R
echo "workload stats done"
ls $O
