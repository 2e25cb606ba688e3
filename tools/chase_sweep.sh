#!/bin/bash
# usage: tools/chase_sweep.sh  (on the GPU box) -- times the band chase for the slot layouts
for cfg in 4,1024 2,512 2,1024 1,256 1,512; do
  echo "== SAAMGE_AMD_CHASE=$cfg"
  SAAMGE_AMD_CHASE=$cfg timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "eigens" 2>&1 | tail -1
  SAAMGE_AMD_CHASE=$cfg timeout -k 10 200 python bench.py --size 128 --levels 2 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "band_chase" 
done
