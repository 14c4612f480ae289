mkdir -p gpurun_out/r2
for v in "" "VINCE_CT64_MAX_K=64" "VINCE_CT64_MAX_K=256" "VINCE_CT64_MAX_K=512" "NOSTATS=1" "VINCE_DLDS_CFG=4" "VINCE_NARROW256_MIN_TILES=0"; do
  env $v python tools/conv_micro4.py "[$v]" 2>&1 | grep -v amdgpu
done > gpurun_out/r2/conv_knobs.txt
cat gpurun_out/r2/conv_knobs.txt
