set -x
mkdir -p gpurun_out/r2
python tools/ceilings.py > gpurun_out/r2/ceilings.txt 2>&1
python tools/conv_micro4.py base > gpurun_out/r2/conv_micro_base.txt 2>&1
python tools/bn_micro.py base > gpurun_out/r2/bn_micro_base.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r2/bench_base.json 2> gpurun_out/r2/bench_base.err
tail -c 600 gpurun_out/r2/ceilings.txt
