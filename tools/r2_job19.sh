#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/stagger.txt; rm -f $O
for st in 0 1 2 3 5 0; do
  VINCE_CONV_STAGGER=$st timeout 300 python tools/conv_micro4.py "STAGGER=$st" 2>/dev/null >> $O
done
cat $O
for st in 0 1 2 3; do
echo "STAGGER=$st: $(VINCE_CONV_STAGGER=$st timeout 300 python tools/fwd_profile.py 20 2>/dev/null | grep 'forward ms')"
done
