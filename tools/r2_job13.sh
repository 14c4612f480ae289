#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/pmc_l3.txt; rm -f $OUT
run() {  # label, env, counters
  O=gpurun_out/pmcx; rm -rf $O; mkdir -p $O
  env $2 timeout 300 rocprofv3 --kernel-trace --pmc $3 -d $O -o p -- python tools/conv_one.py 14 256 256 3 10 > $O/log.txt 2>&1
  DB=$(find $O -name '*.db' | head -1)
  echo "=== $1 [$3]" >> $OUT
  python tools/pmc_all.py $DB conv_igemm >> $OUT 2>&1
  rm -rf $O
}
for e in "VINCE_ROT=0" "VINCE_ROT=1"; do
run "$e" "$e" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES"
run "$e" "$e" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"
run "$e" "$e" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INST_LEVEL_LDS"
done
cat $OUT
