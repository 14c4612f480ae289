mkdir -p gpurun_out/r2
for v in "" "VINCE_CONV_ABLATE=8" "NOSTATS=1" "VINCE_BIG_MIN_K=256" "VINCE_BIG_MIN_K=512" "VINCE_BIG_MIN_K=256 VINCE_BIG_MIN_TILES=128" "VINCE_BIG_MIN_K=128"; do
  env $v python tools/conv_micro4.py "[$v]" 2>&1 | grep -v amdgpu
done > gpurun_out/r2/conv_knobs2.txt
cat gpurun_out/r2/conv_knobs2.txt
