#!/bin/bash
# GPU box: rocprofv3 kernel stats of a short bench run for each env setting given -> gpurun_out/kstats_<i>.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for envs in "$@"; do
  i=$((i+1))
  O=gpurun_out/kst_$i; rm -rf $O; mkdir -p $O
  env $envs timeout 600 rocprofv3 --kernel-trace -d $O -o kt -- python bench.py --steps 10 --warmup 3 --no-extras > $O/log.txt 2>&1
  DB=$(find $O -name '*.db' | head -1)
  timeout 60 python tools/rocpd_stats.py $DB 60 > gpurun_out/kstats_$i.txt 2>&1
  timeout 60 python tools/rocpd_sequence.py $DB > gpurun_out/kseq_$i.txt 2>&1
  rm -rf $O
done
