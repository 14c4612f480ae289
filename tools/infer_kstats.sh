#!/bin/bash
# GPU box: rocprofv3 kernel stats of eval-mode extract_features (folded BatchNorms), ResNet-50 B=256 bf16
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/infer; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace -d $O/kt -o kt -- python tools/infer_ab.py child > $O/kt.log 2>&1
DB=$(find $O/kt -name '*.db' | head -1); timeout 60 python tools/rocpd_stats.py $DB 40 > $O/kernel_stats.txt 2>&1; rm -rf $O/kt
tail -1 $O/kt.log
