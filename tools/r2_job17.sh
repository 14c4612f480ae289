#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv" 2>&1 | tail -2
for e in "VINCE_ROT=0 VINCE_KC8_MIN_K=0" "VINCE_ROT=9 VINCE_KC8_MIN_K=2048" "VINCE_ROT=0 VINCE_KC8_MIN_K=0" "VINCE_ROT=9 VINCE_KC8_MIN_K=2048"; do
  echo "$e: $(env $e timeout 300 python tools/fwd_profile.py 30 2>/dev/null | grep 'forward ms')"
  env $e timeout 600 python bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   step ms', d['ms_per_step'], 'frames/s', d['value'])"
done
