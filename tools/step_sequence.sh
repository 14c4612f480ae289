#!/bin/bash
# GPU box: kernel trace of the training step (bench.py --no-extras) -> per-queue kernel sequence of one step + overlap summary
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/seq}; rm -rf $O; mkdir -p $O
shift
timeout 600 rocprofv3 --kernel-trace -d $O/kt -o kt -- python bench.py --steps 6 --warmup 3 --no-extras "$@" > $O/kt.log 2>&1
DB=$(find $O/kt -name '*.db' | head -1)
timeout 60 python tools/rocpd_sequence.py $DB > $O/sequence.txt 2>&1
timeout 60 python tools/rocpd_overlap.py $DB > $O/overlap.txt 2>&1
timeout 60 python tools/rocpd_stats.py $DB 40 > $O/kernel_stats.txt 2>&1
rm -rf $O/kt
head -12 $O/overlap.txt
