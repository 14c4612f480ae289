python tools/conv_micro4.py "[A]" 2>&1 | grep -v amdgpu
VINCE_HIP_LIB=$PWD/vince_amd/lib/libvince_hip_b.so python tools/conv_micro4.py "[B interleave]" 2>&1 | grep -v amdgpu
VINCE_HIP_LIB=$PWD/vince_amd/lib/libvince_hip_b.so python -m pytest tests/test_ops_gpu.py -q -k "conv" 2>&1 | tail -2
