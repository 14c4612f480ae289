#!/bin/bash
# GPU box: rocprofv3 kernel stats of the no-grad forward alone for each env setting given -> gpurun_out/r2/fwd_kstats_<i>.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
i=0
for envs in "$@"; do
  i=$((i+1))
  O=gpurun_out/kst_$i; rm -rf $O; mkdir -p $O
  # wall time WITHOUT the profiler (under rocprofv3 the ~230 launches of a forward become host-bound), then the kernel trace
  echo "$envs: $(env $envs timeout 300 python tools/fwd_profile.py 20 2>/dev/null | grep 'forward ms') (no profiler)"
  env $envs timeout 600 rocprofv3 --kernel-trace -d $O -o kt -- python tools/fwd_profile.py 10 > $O/log.txt 2>&1
  echo "$envs: $(grep 'forward ms' $O/log.txt) (under rocprofv3 --kernel-trace)"
  DB=$(find $O -name '*.db' | head -1)
  timeout 60 python tools/rocpd_stats.py $DB 30 > gpurun_out/r2/fwd_kstats_$i.txt 2>&1
  rm -rf $O
done
