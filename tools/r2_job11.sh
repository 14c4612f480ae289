for v in "" "VINCE_S3_MIN_K=128" "VINCE_S3_MIN_K=256" "VINCE_S3_MIN_K=512"; do
  env $v python tools/conv_micro4.py "[$v]" 2>&1 | grep -v amdgpu
done
