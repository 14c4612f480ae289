cd $GRAFT_REPO_ROOT
for k in "gram_wgs64=512,gram_wgs128=512" "gram_wgs64=1024,gram_wgs128=768" "gram_wgs64=2048,gram_wgs128=1024" "gram_wgs64=768,gram_wgs128=640"; do
  echo "== $k"; bash tools/fwd_kstats.sh VINCE_KNOBS=$k 2>&1 | tail -2; grep "bn_apply_gram\|wgrad_reduce" gpurun_out/r2/fwd_kstats_1.txt | cut -c1-60,95-140
done
