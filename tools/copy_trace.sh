#!/bin/bash
# GPU box: which memory copies does a training step issue (rocprofv3 --memory-copy-trace), by size and direction
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/copytrace; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --memory-copy-trace --hip-runtime-trace -d $O/t -o t --output-format csv -- python bench.py --steps 4 --warmup 2 --no-extras > $O/log.txt 2>&1
ls -R $O/t | head -20
python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/copytrace/t/**/*memory_copy_trace.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(f, len(rows), rows[0].keys() if rows else None)
    c = collections.Counter()
    for r in rows:
        c[(r.get('Direction'), r.get('Size') or r.get('Bytes'))] += 1
    for k, v in c.most_common(25):
        print(v, k)
for f in glob.glob('gpurun_out/copytrace/t/**/*hip_api_trace.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    c = collections.Counter(r.get('Function') for r in rows)
    print(f, len(rows))
    for k, v in c.most_common(25):
        print(v, k)
PY
