#!/bin/bash
# GPU box: vince_conv_igemm on 13 representative layers with parts switched off (measurement build: python -m vince_amd.build --measure).
# VINCE_CONV_ABLATE bits: 1 no DMA, 2 no MFMA, 4 no barrier, 8 no statistics atomics, 16 no main loop, 32 no output stores, 128 no epilogue
cd $GRAFT_REPO_ROOT
export VINCE_HIP_LIB=$PWD/vince_amd/lib/libvince_hip_measure.so
for A in 0 128 8 32 40 16 144; do VINCE_KNOBS="m8=0" VINCE_CONV_ABLATE=$A timeout 120 python tools/conv_micro4.py "ABLATE=$A" 2>&1 | tail -1; done
for A in 0 32 2; do VINCE_M8_ABLATE=$A timeout 120 python tools/conv_micro4.py "M8 ABLATE=$A" 2>&1 | tail -1; done
