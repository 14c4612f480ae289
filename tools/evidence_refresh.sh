#!/bin/bash
# GPU box: the part of tools/final_profile.sh that depends on the bf16 step's kernels -- after a late kernel change: bench line + per-layer
# roofline, kernel stats (overlapped and serialised), the PMC passes (re-stamping profiles/pmc_conv_igemm.json), the same-box A/B, and
# the bench line once more with the fresh stamp.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace -d $O/kt -o kt -- python bench.py --no-extras > $O/kt.log 2>&1
DB=$(find $O/kt -name '*.db' | head -1); timeout 60 python tools/rocpd_stats.py $DB 45 > $O/kernel_stats.txt 2>&1
timeout 60 python tools/rocpd_overlap.py $DB > $O/step_overlap.txt 2>&1; rm -rf $O/kt
VINCE_OVERLAP_KEY=0 VINCE_KNOBS=wgrad_stream=0,ds_stream=0 timeout 600 rocprofv3 --kernel-trace -d $O/kts -o kt -- python bench.py --no-extras > $O/kts.log 2>&1
DB=$(find $O/kts -name '*.db' | head -1); timeout 60 python tools/rocpd_stats.py $DB 45 > $O/kernel_stats_serialised.txt 2>&1; rm -rf $O/kts
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $C -d $O/pmc_$C -o pmc -- python bench.py --steps 2 --warmup 1 --no-extras > $O/pmc_$C.log 2>&1
done
F=$(find $O/pmc_FETCH_SIZE -name '*.db' | head -1); W=$(find $O/pmc_WRITE_SIZE -name '*.db' | head -1)
timeout 60 python tools/rocpd_pmc.py $F 14 > $O/pmc_fetch_size.txt 2>&1
timeout 60 python tools/rocpd_pmc.py $W 14 > $O/pmc_write_size.txt 2>&1
timeout 60 python tools/pmc_summary.py $F $W $O/pmc_conv_igemm.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), FETCH_SIZE x2 (gfx950 128-B requests tallied at 64 B); bench.py --steps 2 --warmup 1 --no-extras" > $O/pmc_summary.txt 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES -d $O/pmc_mfma -o pmc -- python bench.py --steps 2 --warmup 1 --no-extras > $O/pmc_mfma.log 2>&1
M=$(find $O/pmc_mfma -name '*.db' | head -1); timeout 60 python tools/pmc_mfma.py $M > $O/pmc_mfma_util.txt 2>&1
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_mfma
head -2 $O/pmc_summary.txt
timeout 300 python tools/step_phases.py 20 bf16 2>/dev/null | tail -1 > $O/step_phases.txt
timeout 1500 python tools/ab.py 4 40 "BASE" "VINCE_DEFER_STEM=1" "VINCE_KNOBS=xjoin_next=0" "VINCE_KNOBS=bn_nt=0" "VINCE_KNOBS=xjoin_next=0,bn_nt=0" > $O/ab.txt 2>&1
cat $O/ab.txt
cp $O/pmc_conv_igemm.json profiles/pmc_conv_igemm.json
VINCE_PROFILE_DUMP=$O/layers.csv timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 60 python tools/layer_roofline.py $O/layers.csv 3 > $O/layer_roofline.txt 2>&1
timeout 20 python tools/bench_brief.py $O/bench.json bench_final | cut -c1-200
