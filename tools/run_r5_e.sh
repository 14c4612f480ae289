mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "deferred or gram or infonce" > gpurun_out/pytest_e.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_e.log
tail -3 gpurun_out/pytest_e.log
bash tools/fwd_kstats.sh "VINCE_KNOBS=gram_max_k=128" "VINCE_KNOBS=gram_max_k=256" 2>&1 | tail -4
grep -E "finalize|wgrad_reduce|wgrad_tr|Li2ELb0ELi2EEE|bn_apply_kernel" gpurun_out/r2/fwd_kstats_2.txt | cut -c1-150
grep -E "finalize" gpurun_out/r2/fwd_kstats_1.txt | cut -c1-150
