#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/abl3.txt
for a in 64 144 128; do
  VINCE_CONV_ABLATE=$a timeout 300 python tools/conv_micro4.py "ABLATE=$a" 2>/dev/null >> gpurun_out/abl3.txt
done
P=$PWD/vince_amd/lib/libvince_hip_prio.so
for r in 0 15; do
  VINCE_ROT=$r timeout 300 python tools/conv_micro4.py "base ROT=$r" 2>/dev/null >> gpurun_out/abl3.txt
  VINCE_HIP_LIB=$P VINCE_ROT=$r timeout 300 python tools/conv_micro4.py "prio ROT=$r" 2>/dev/null >> gpurun_out/abl3.txt
done
cat gpurun_out/abl3.txt
for r in 0 15; do
echo "base ROT=$r: $(VINCE_ROT=$r timeout 300 python tools/fwd_profile.py 20 2>/dev/null | grep 'forward ms')"
echo "prio ROT=$r: $(VINCE_HIP_LIB=$P VINCE_ROT=$r timeout 300 python tools/fwd_profile.py 20 2>/dev/null | grep 'forward ms')"
done
