cd $GRAFT_REPO_ROOT
export VINCE_HIP_LIB=$PWD/vince_amd/lib/libvince_hip_measure.so
for A in 0 1 2 4 8 16 6 3 7 32 34 38 39; do VINCE_M8_ABLATE=$A timeout 120 python tools/conv_micro4.py "M8 ABLATE=$A" 2>&1 | tail -1 | awk -F'|' '{print $1 "|" $5 "|" $6}'; done
