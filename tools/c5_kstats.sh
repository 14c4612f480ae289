#!/bin/bash
# GPU box: rocprofv3 kernel stats of BASELINE config 5's per-GPU step (4 frames per clip, inter + self comparison, jigsaw side)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c5; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace -d $O/kt -o kt -- python bench.py --mode vince --no-extras --steps 10 --warmup 3 > $O/kt.log 2>&1
DB=$(find $O/kt -name '*.db' | head -1); timeout 60 python tools/rocpd_stats.py $DB 60 > $O/kernel_stats.txt 2>&1; rm -rf $O/kt
timeout 300 python bench.py --mode vince --no-extras --steps 20 --warmup 5 2>/dev/null | tail -c 400
