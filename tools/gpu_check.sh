#!/bin/bash
# GPU box: full GPU test-suite, then bench lines for the environment settings given as arguments ("A=1 B=2" strings).
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log | head -2
i=0
for envs in "$@"; do
  i=$((i+1))
  env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-extras > gpurun_out/ab_$i.json 2> gpurun_out/ab_$i.err
  timeout 20 python tools/bench_brief.py gpurun_out/ab_$i.json "[$envs]"
done
