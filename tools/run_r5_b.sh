mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_x3_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "deferred or solver or x3h_operands or stream_copy or g13 or g5 or three_steps" > gpurun_out/pytest_b.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_b.log
tail -4 gpurun_out/pytest_b.log
for e in "VINCE_DEFER_STEM=1" "VINCE_DEFER_STEM=0" "VINCE_DEFER_STEM=1" "VINCE_DEFER_STEM=0"; do
  env $e timeout 300 python bench.py --steps 30 --warmup 5 --no-extras > gpurun_out/ab.json 2> gpurun_out/ab.err
  python tools/bench_brief.py gpurun_out/ab.json "[$e]" | cut -c1-60
done
timeout 300 python tools/group_bound.py 256 10 2>&1 | tail -1
