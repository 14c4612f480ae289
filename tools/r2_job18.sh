#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/cp; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $O -o kt -- python tools/fwd_profile.py 4 > $O/log.txt 2>&1
DB=$(find $O -name '*.db' | head -1)
python tools/rocpd_copies.py $DB > gpurun_out/copies.txt 2>&1
rm -rf $O
cat gpurun_out/copies.txt | head -60
