#!/bin/bash
# GPU box: the round's evidence set.  bench.py full line, rocprofv3 kernel stats of the same command, HBM traffic PMC passes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final; rm -rf $O; mkdir -p $O
VINCE_PROFILE_DUMP=$O/layers.csv timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 20 python tools/bench_brief.py $O/bench.json bench
# per-layer roofline table of every conv launch of the instrumented steps (bench.py --profile-steps, default 3)
timeout 60 python tools/layer_roofline.py $O/layers.csv 3 > $O/layer_roofline.txt 2>&1
# the no-grad forward alone (the forward + InfoNCE leg's trunk)
bash tools/fwd_kstats.sh VINCE_KNOBS=gram_join=1 VINCE_KNOBS=gram_join=0 > $O/fwd_ms.txt 2>&1; cp gpurun_out/r2/fwd_kstats_1.txt $O/fwd_kernel_stats.txt; cp gpurun_out/r2/fwd_kstats_2.txt $O/fwd_kernel_stats_separate_passes.txt
timeout 600 rocprofv3 --kernel-trace -d $O/kt -o kt -- python bench.py --no-extras > $O/kt.log 2>&1
DB=$(find $O/kt -name '*.db' | head -1); timeout 60 python tools/rocpd_stats.py $DB 45 > $O/kernel_stats.txt 2>&1
# which stream is busy when (per queue, kernels in flight, per-millisecond bins) of one overlapped step of the same trace
timeout 60 python tools/rocpd_overlap.py $DB > $O/step_overlap.txt 2>&1; rm -rf $O/kt
# the same with every stream serialised (one kernel at a time, like the instrumented steps behind bench.py's `roofline`): the
# per-kernel average durations of THIS file are the ones that agree with roofline.avg_us
VINCE_OVERLAP_KEY=0 VINCE_KNOBS=wgrad_stream=0,ds_stream=0 timeout 600 rocprofv3 --kernel-trace -d $O/kts -o kt -- python bench.py --no-extras > $O/kts.log 2>&1
DB=$(find $O/kts -name '*.db' | head -1); timeout 60 python tools/rocpd_stats.py $DB 45 > $O/kernel_stats_serialised.txt 2>&1; rm -rf $O/kts
# the weight-gradient kernels layer by layer, timed alone: conv_wgrad_tr against the kernel it replaced
(timeout 200 python tools/wgrad_micro.py conv_wgrad_tr; VINCE_KNOBS=wgrad_tr=0 timeout 200 python tools/wgrad_micro.py conv_wgrad_dlds) 2>&1 | grep "14x14" | sed 's/ | /\n    /g' > $O/wgrad_micro.txt
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
cat $O/pmc_summary.txt | head -12
# GPU input stage: kernel stats of 512 MoCo-v2 views + the step fed from raw uint8 frames
bash tools/aug_profile.sh > /dev/null 2>&1; cp gpurun_out/aug_kstats.txt $O/input_stage_kernel_stats.txt
timeout 300 python bench.py --no-extras --input u8aug > $O/bench_u8aug.json 2>/dev/null
timeout 20 python tools/bench_brief.py $O/bench_u8aug.json u8aug
# ---- the split-half modes: x3 (strict parity) and x3f (x3 forward + bf16 twin backward), each profiled like the product
bash tools/x3_profile.sh final_x3 x3 > /dev/null 2>&1; cp gpurun_out/final_x3/kernel_stats_serialised.txt $O/x3_kernel_stats_serialised.txt; cp gpurun_out/final_x3/bench.json $O/x3_bench.json
cp gpurun_out/final_x3/layer_roofline.txt $O/x3_layer_roofline.txt; cp gpurun_out/final_x3/pmc_summary.txt $O/x3_pmc_summary.txt
bash tools/x3_profile.sh final_x3f x3f > /dev/null 2>&1; cp gpurun_out/final_x3f/kernel_stats_serialised.txt $O/x3f_kernel_stats_serialised.txt; cp gpurun_out/final_x3f/bench.json $O/x3f_bench.json
cp gpurun_out/final_x3f/layer_roofline.txt $O/x3f_layer_roofline.txt; cp gpurun_out/final_x3f/pmc_summary.txt $O/x3f_pmc_summary.txt
timeout 900 python tools/x3f_check.py g9 g12 g14 2>&1 | cut -c1-420 > $O/x3f_gradients_vs_reference.txt
X3_MODES=x3,x1,bf16 X3_OPS=dgrad,wgrad timeout 300 python tools/x3_micro.py x1b 2>&1 | grep -v amdgpu > $O/x1b_micro.txt
./tools/micro/mfma_micro > $O/mfma_micro.txt 2>&1
timeout 300 python tools/x3_micro.py final > $O/x3_micro.txt 2>&1
bash tools/fwd_pmc.sh $O/fwd_pmc > /dev/null 2>&1; cp $O/fwd_pmc/fwd_pmc_summary.txt $O/fwd_pmc_summary.txt
timeout 300 python tools/ceilings.py > $O/ceilings.txt 2>&1
timeout 120 python tools/strip_dgrad_micro.py > $O/strip_dgrad_micro.txt 2>&1
timeout 300 python tools/xjoin_micro.py > $O/xjoin_micro.txt 2>&1
timeout 200 python tools/dgrad_epi_micro.py > $O/dgrad_epi_micro.txt 2>&1
# the full-size parity fixtures, with what each dtype measured against the reference printed (G9, G12: fp32 / x3 / bf16)
timeout 1800 python -m pytest tests/test_full_size_gpu.py -q -s -k "g9 or g12 or g14 or serialised" 2>&1 | grep -E "G9|G12|G14|serialised vs|passed|failed" | cut -c1-600 > $O/full_size_parity.txt
# ---- round 5: where the step's time is without a profiler, the grouped-launch bound, the same-box A/B of the round's switches
timeout 300 python tools/step_phases.py 20 bf16 2>/dev/null | tail -1 > $O/step_phases.txt
timeout 300 python tools/step_phases.py 10 x3 2>/dev/null | tail -1 >> $O/step_phases.txt
timeout 300 python tools/step_phases.py 10 x3f 2>/dev/null | tail -1 >> $O/step_phases.txt
timeout 400 python tools/group_bound.py 256 10 2>/dev/null | tail -2 > $O/group_bound.txt
timeout 900 python tools/ab.py 3 40 "VINCE_DEFER_STEM=1" "VINCE_DEFER_STEM=0" "VINCE_KNOBS=xjoin_next=0" > $O/ab.txt 2>&1
AB_ARGS="--dtype x3f" timeout 900 python tools/ab.py 2 20 "VINCE_KNOBS=" "VINCE_KNOBS=gram_shadow=0" "VINCE_KNOBS=x3f_alg=0" "VINCE_KNOBS=alg_split=0" "VINCE_X3F_HYBRID=0" >> $O/ab.txt 2>&1
# the BatchNorm-backward algebra's input gradient at engine scale: masked pixel sums vs fp64, single bf16 matrices against hi + lo parts
(timeout 200 python tools/alg_op_probe.py 200704 64 0.02; timeout 200 python tools/alg_op_probe.py 200704 128 0.1) 2>&1 | grep -v amdgpu > $O/alg_op_probe.txt
export VINCE_GIT_HEAD=${VINCE_GIT_HEAD:-unknown}
# the bench line once more, now that profiles/pmc_conv_igemm.json of THIS build exists (traffic_stale false)
cp $O/pmc_conv_igemm.json profiles/pmc_conv_igemm.json
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 20 python tools/bench_brief.py $O/bench.json bench_final
