mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "deferred or solver or algebra or g5 or three_steps or g9 or bn3" > gpurun_out/pytest_c.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_c.log
tail -3 gpurun_out/pytest_c.log
timeout 300 python tools/step_phases.py 20 bf16 2>&1 | tail -1
for i in 1 2; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-extras > gpurun_out/ab.json 2> gpurun_out/ab.err
  python tools/bench_brief.py gpurun_out/ab.json "[new]" | cut -c1-60
done
