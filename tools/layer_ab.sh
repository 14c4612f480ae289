#!/bin/bash
# GPU box: per-layer conv timing tables (bench roofline leg with VINCE_PROFILE_DUMP) for each env setting given.
mkdir -p gpurun_out
i=0
for envs in "$@"; do
  i=$((i+1))
  env $envs VINCE_PROFILE_DUMP=gpurun_out/lt_$i.csv timeout 400 python bench.py --steps 10 --warmup 3 --cpu-steps 0 > gpurun_out/lt_$i.json 2> gpurun_out/lt_$i.err
  timeout 20 python tools/bench_brief.py gpurun_out/lt_$i.json "[$envs]"
  timeout 20 python tools/layer_table.py gpurun_out/lt_$i.csv 3 > gpurun_out/lt_$i.txt 2>&1
  rm -f gpurun_out/lt_$i.csv
done
