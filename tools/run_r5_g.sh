mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
for i in 1 2; do
 for e in "VINCE_KNOBS=gram_max_k=128" "VINCE_KNOBS=gram_max_k=256"; do
  env $e timeout 300 python bench.py --steps 30 --warmup 5 --no-extras > gpurun_out/ab.json 2> gpurun_out/ab.err
  python tools/bench_brief.py gpurun_out/ab.json "[$e]" | cut -c1-70
 done
done
(cd ab_base && timeout 300 python bench.py --steps 30 --warmup 5 --no-extras > ../gpurun_out/ab0.json 2> ../gpurun_out/ab0.err)
python tools/bench_brief.py gpurun_out/ab0.json "[round-4 tree]" | cut -c1-60
