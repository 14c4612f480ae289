#!/bin/bash
# Ablations of conv_m8 (measurement build) on one layer: tools/m8_ablate.sh hw ci co k  -> gpurun_out/m8_ablate.txt
# then SQ counters of the full kernel.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export VINCE_HIP_LIB=$R/vince_amd/lib/libvince_hip_measure.so VINCE_KNOBS=m8_min_k=1,m8_min_tiles=1
OUT=$R/gpurun_out/m8_ablate.txt
: > $OUT
FIRST="$1"
for shape in "$@"; do
  set -- $shape
  for a in ${M8_ABLS:-0 64 1 16 2 4 3 7 32}; do
    echo "== shape $shape ablate $a" >> $OUT
    (cd $R && VINCE_M8_ABLATE=$a timeout 120 python tools/conv_sweep.py $1 $2 $3 $4 256 2>&1 | grep "hw " >> $OUT)
  done
done
cat $OUT
# SQ counters of the unablated kernel on the first shape
set -- $FIRST
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"; do
  O=/tmp/m8pmc; rm -rf $O; mkdir -p $O
  (cd $R && VINCE_HIP_LIB=${PMC_LIB:-$R/vince_amd/lib/libvince_hip.so} timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $O -o p -- python tools/conv_one.py $1 $2 $3 $4 10 > $O/log.txt 2>&1)
  DB=$(find $O -name '*.db' | head -1)
  if [ -z "$DB" ]; then tail -20 $O/log.txt >> $R/gpurun_out/m8_pmc.txt; fi
  echo "=== [$grp]" >> $R/gpurun_out/m8_pmc.txt
  (cd $R && timeout 60 python tools/pmc_all.py $DB conv_m8 >> $R/gpurun_out/m8_pmc.txt 2>&1)
done
cat $R/gpurun_out/m8_pmc.txt
