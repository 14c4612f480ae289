#!/bin/bash
# GPU box: the x3 training step (fp32 tensors, split-half convolutions) profiled like the bf16 one -- kernel stats with the streams
# serialised (per-kernel averages = each kernel alone), the bench line with its per-layer roofline (against 2.5 PF / 3), HBM bytes per
# step and per kernel family by PMC (FETCH_SIZE x 2 + WRITE_SIZE, separate passes).  Usage: bash tools/x3_profile.sh [tag] [x3|x3f]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-x3}
DT=${2:-x3}
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
VINCE_OVERLAP_KEY=0 VINCE_KNOBS=wgrad_stream=0,ds_stream=0 timeout 600 rocprofv3 --kernel-trace -d $O/kts -o kt -- python bench.py --dtype $DT --steps 3 --warmup 2 --no-extras > $O/kts.log 2>&1
DB=$(find $O/kts -name '*.db' | head -1); timeout 60 python tools/rocpd_stats.py $DB 45 > $O/kernel_stats_serialised.txt 2>&1; rm -rf $O/kts
VINCE_PROFILE_DUMP=$O/layers.csv timeout 400 python bench.py --dtype $DT --steps 10 --warmup 3 --fp32-steps 0 --config-steps 0 --cpu-steps 0 > $O/bench.json 2> $O/bench.err
timeout 60 python tools/layer_roofline.py $O/layers.csv 3 5.28 x3 > $O/layer_roofline.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $C -d $O/pmc_$C -o pmc -- python bench.py --dtype $DT --steps 2 --warmup 1 --no-extras > $O/pmc_$C.log 2>&1
done
F=$(find $O/pmc_FETCH_SIZE -name '*.db' | head -1); W=$(find $O/pmc_WRITE_SIZE -name '*.db' | head -1)
timeout 60 python tools/pmc_summary.py $F $W $O/pmc_x3.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), FETCH_SIZE x2; bench.py --dtype $DT --steps 2 --warmup 1 --no-extras" > $O/pmc_summary.txt 2>&1
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
head -3 $O/pmc_summary.txt; tail -1 $O/layer_roofline.txt
tail -c 600 $O/bench.json
