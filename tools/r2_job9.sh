mkdir -p gpurun_out/r2
python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_full_size_gpu.py -q -s -k "gram or expand_join or g3_trunk or g5_first or solver_own_step or bf16 or g9" 2>&1 | grep -E "^E  +assert|^E  +Assertion|^E  +Mismatch|^E  +Max|^E .*err|FAILED|passed|failed|Error|G9 " | head -40
for v in 1 0; do
  VINCE_GRAM_TRAIN=$v python bench.py --steps 10 --warmup 3 --fp32-steps 0 --cpu-steps 0 --profile-steps 1 > gpurun_out/r2/bench_gt$v.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r2/bench_gt$v.json").read().strip().splitlines()[-1])
print("GRAM_TRAIN=$v", d["value"], d["ms_per_step"], d["fwd_infonce"], d["config"]["final_loss"])
PY
done
