#!/bin/bash
# Ablations of the weight-gradient kernel (measurement build), timed alone, then SQ counters of the unablated kernel on layer3's 3x3.
#   tools/wgrad_ablate.sh  -> gpurun_out/wgrad_ablate.txt, gpurun_out/wgrad_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/wgrad_ablate.txt
: > $OUT
for a in ${WG_ABLS:-0 1 2 4 8 16 12 24 28 29}; do
  (cd $R && VINCE_HIP_LIB=$R/vince_amd/lib/libvince_hip_measure.so VINCE_WGRAD_ABLATE=$a ${WG_KNOBS:+VINCE_KNOBS=$WG_KNOBS} timeout 200 python tools/wgrad_micro.py "ablate=$a" 2>&1 | tail -1 >> $OUT)
done
cat $OUT
P=$R/gpurun_out/wgrad_pmc.txt; : > $P
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"; do
  O=/tmp/wgpmc; rm -rf $O; mkdir -p $O
  (cd $R && WGRAD_SHAPES=${WG_PMC_SHAPES:-0} WGRAD_REPS=5 ${WG_KNOBS:+VINCE_KNOBS=$WG_KNOBS} timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $O -o p -- python tools/wgrad_micro.py pmc > $O/log.txt 2>&1)
  DB=$(find $O -name '*.db' | head -1)
  if [ -z "$DB" ]; then tail -20 $O/log.txt >> $P; fi
  echo "=== [$grp]" >> $P
  (cd $R && timeout 60 python tools/pmc_all.py $DB conv_wgrad >> $P 2>&1)
done
cat $P
