cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/seq; mkdir -p gpurun_out/seq
rocprofv3 --kernel-trace -d gpurun_out/seq -o seq -- python bench.py --steps 6 --warmup 3 --no-extras > gpurun_out/seq/bench.log 2>&1
DB=$(find gpurun_out/seq -name '*.db' | head -1)
python tools/rocpd_sequence.py $DB > gpurun_out/seq/sequence.txt 2>&1
python tools/rocpd_overlap.py $DB > gpurun_out/seq/overlap.txt 2>&1
python tools/rocpd_stats.py $DB 45 > gpurun_out/seq/stats.txt 2>&1
rm -f $DB
tail -3 gpurun_out/seq/bench.log; cat gpurun_out/seq/overlap.txt
