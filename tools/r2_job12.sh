#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/rot.txt
for k in 0 1 2 4 8 15 0; do
  VINCE_ROT=$k timeout 300 python tools/conv_micro4.py "ROT=$k" 2>/dev/null >> gpurun_out/rot.txt
done
VINCE_ROT=15 timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv" > gpurun_out/rot_tests.txt 2>&1
tail -3 gpurun_out/rot_tests.txt
cat gpurun_out/rot.txt
for k in 0 15; do
echo "ROT=$k: $(VINCE_ROT=$k timeout 300 python tools/fwd_profile.py 20 2>/dev/null | grep 'forward ms')"
done
