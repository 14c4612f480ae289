cd $GRAFT_REPO_ROOT
for shape in "14 256 1024 1" "7 512 2048 1" "28 128 512 1" "56 64 256 1"; do
  for k in "x=0" "big_min_k=64,big_min_tiles=1" "kc8_min_k=64" "s3_min_k=64" "m8_min_k=1,m8_min_tiles=1"; do
    echo -n "$k : "; VINCE_KNOBS=$k timeout 120 python tools/conv_sweep.py $shape 256 2>&1 | grep "hw "
  done
done
