mkdir -p gpurun_out/r2
for v in "VINCE_BN_NT=0" "VINCE_BN_NT=1" "VINCE_BN_NT=2" "VINCE_BN_NT=3" "VINCE_BN_NT=3 VINCE_BN_NT_MIN=0" "VINCE_BN_NT=0"; do
  env $v python bench.py --steps 15 --warmup 3 --fp32-steps 0 --cpu-steps 0 --profile-steps 1 > gpurun_out/r2/bench_nt.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r2/bench_nt.json").read().strip().splitlines()[-1])
print("$v", d["value"], d["ms_per_step"], d["fwd_infonce"]["ms"])
PY
done
