mkdir -p gpurun_out/r2
python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -k "gram or g3_trunk or g5_first or solver_own_step" 2>&1 | grep -E "^E  +assert|^E  +Assertion|^E  +Mismatch|^E  +Max|^E .*err|FAILED|passed|failed|Error" | head -40
for v in 1 0; do
  VINCE_GRAM_JOIN=$v python bench.py --steps 10 --warmup 3 --fp32-steps 0 --cpu-steps 0 --profile-steps 1 > gpurun_out/r2/bench_gram$v.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r2/bench_gram$v.json").read().strip().splitlines()[-1])
print("GRAM_JOIN=$v", d["value"], d["ms_per_step"], d["fwd_infonce"])
PY
done
