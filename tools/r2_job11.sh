#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/abl.txt
for a in 0 1 2 3 4; do
  VINCE_CONV_ABLATE=$a timeout 300 python tools/conv_micro4.py "ABLATE=$a" 2>/dev/null >> gpurun_out/abl.txt
  VINCE_CONV_ABLATE=$a NOSTATS=1 timeout 300 python tools/conv_micro4.py "ABLATE=$a nostats" 2>/dev/null >> gpurun_out/abl.txt
done
cat gpurun_out/abl.txt
