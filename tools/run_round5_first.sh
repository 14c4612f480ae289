mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
bash tools/fwd_kstats.sh "VINCE_KNOBS=gram_max_k=128" "VINCE_KNOBS=gram_max_k=256" 2>&1 | tail -6
timeout 600 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; echo bench rc=$?
python tools/bench_brief.py gpurun_out/bench_full.json "[full]"
