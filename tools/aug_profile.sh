#!/bin/bash
# GPU box: rocprofv3 kernel stats of the GPU input stage alone (512 MoCo-v2 views of 224x224 from 256x320 uint8 frames)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/augprof; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace -d $O -o kt -- python -c "
import torch, numpy as np
from vince_amd.utils import transforms as T
tf = T.MoCoV2ImagenetTransform(224, seed=0)
pool = torch.randint(0, 256, (256, 256, 320, 3), dtype=torch.uint8, device='cuda')
p = tf.draw(512, (256, 320), src_index=np.tile(np.arange(256), 2))
for _ in range(5):
    v = tf.apply(pool, p); v.float_tensor(torch.bfloat16)
torch.cuda.synchronize()
" > $O/log.txt 2>&1
DB=$(find $O -name '*.db' | head -1)
timeout 60 python tools/rocpd_stats.py $DB 14 > gpurun_out/aug_kstats.txt 2>&1
rm -rf $O
cat gpurun_out/aug_kstats.txt
