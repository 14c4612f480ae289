mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py tests/test_full_size_gpu.py -x -q -m gpu -k "deferred or solver or infonce or g5 or three_steps or g2 or self_sim or materialised" > gpurun_out/pytest_d.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_d.log
tail -3 gpurun_out/pytest_d.log
timeout 300 python tools/group_bound.py 256 10 2>&1 | tail -2
timeout 300 python tools/step_phases.py 20 bf16 2>&1 | tail -1
for i in 1 2; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-extras > gpurun_out/ab.json 2> gpurun_out/ab.err
  python tools/bench_brief.py gpurun_out/ab.json "[new]" | cut -c1-60
  (cd ab_base && timeout 300 python bench.py --steps 30 --warmup 5 --no-extras > ../gpurun_out/ab0.json 2> ../gpurun_out/ab0.err)
  python tools/bench_brief.py gpurun_out/ab0.json "[round-4 tree]" | cut -c1-60
done
