#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/persist.txt; rm -f $O
for pe in 0 1; do for r in 0 13; do
  VINCE_PERSIST=$pe VINCE_ROT=$r timeout 300 python tools/conv_micro4.py "PERSIST=$pe ROT=$r" 2>/dev/null >> $O
done; done
cat $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv" > gpurun_out/persist_tests.txt 2>&1
tail -3 gpurun_out/persist_tests.txt
for pe in 0 1; do for r in 0 13; do
echo "PERSIST=$pe ROT=$r: $(VINCE_PERSIST=$pe VINCE_ROT=$r timeout 300 python tools/fwd_profile.py 20 2>/dev/null | grep 'forward ms')"
done; done
