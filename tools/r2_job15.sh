#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/sweep.txt; rm -f $O
for r in 0 13; do
echo "ROT=$r" >> $O
VINCE_ROT=$r python tools/conv_sweep.py 16 256 256 3 64,128,192,256,320,384,512,768,1024 2>/dev/null >> $O
VINCE_ROT=$r python tools/conv_sweep.py 14 256 256 3 256 2>/dev/null >> $O
VINCE_ROT=$r python tools/conv_sweep.py 8 512 512 3 128,256,512,1024 2>/dev/null >> $O
VINCE_ROT=$r python tools/conv_sweep.py 16 256 1024 1 128,256,512 2>/dev/null >> $O
done
cat $O
echo "base ROT=13: $(VINCE_ROT=13 timeout 300 python tools/fwd_profile.py 20 2>/dev/null | grep 'forward ms')"
echo "base ROT=0: $(VINCE_ROT=0 timeout 300 python tools/fwd_profile.py 20 2>/dev/null | grep 'forward ms')"
