#!/bin/bash
# GPU box: kernel stats of the x3 training step (fp32 tensors, split-half convolutions), streams serialised so that the per-kernel
# averages are each kernel alone.  Usage: bash tools/x3_profile.sh [tag]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-x3}
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
VINCE_OVERLAP_KEY=0 VINCE_KNOBS=wgrad_stream=0,ds_stream=0 timeout 600 rocprofv3 --kernel-trace -d $O/kts -o kt -- python bench.py --dtype x3 --steps 3 --warmup 2 --no-extras > $O/kts.log 2>&1
DB=$(find $O/kts -name '*.db' | head -1); timeout 60 python tools/rocpd_stats.py $DB 45 > $O/kernel_stats_serialised.txt 2>&1; rm -rf $O/kts
timeout 300 python bench.py --dtype x3 --steps 10 --warmup 3 --no-extras > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
