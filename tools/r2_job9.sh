#!/bin/bash
mkdir -p gpurun_out
./tools/micro/feed_micro > gpurun_out/feed2.txt 2>&1
for k in 0 1 2 4 8 15; do
  VINCE_KC8=$k timeout 300 python tools/conv_micro4.py "KC8=$k" >> gpurun_out/kc8.txt 2>&1
done
VINCE_KC8=15 timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv" > gpurun_out/kc8_tests.txt 2>&1
tail -3 gpurun_out/kc8_tests.txt
cat gpurun_out/kc8.txt
cat gpurun_out/feed2.txt
