#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/fseq; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace -d $O -o kt -- python tools/fwd_profile.py 6 > $O/log.txt 2>&1
DB=$(find $O -name '*.db' | head -1)
python tools/rocpd_fwd_sequence.py $DB > gpurun_out/fwd_sequence.txt 2>&1
rm -rf $O
head -5 gpurun_out/fwd_sequence.txt
