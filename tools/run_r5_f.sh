mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gram or folded or join" > gpurun_out/pytest_f.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_f.log
tail -3 gpurun_out/pytest_f.log
bash tools/fwd_kstats.sh "VINCE_KNOBS=gram_max_k=128" "VINCE_KNOBS=gram_max_k=256" 2>&1 | tail -4
grep -E "finalize|Li2ELb0ELi2EEE" gpurun_out/r2/fwd_kstats_2.txt | cut -c1-150
