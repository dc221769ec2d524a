"""rocprofv3 --pmc target: the screened retrieval (csrc/scores_screen.hip) on two database panels of configs[2]'s shard --
10 000 queries x 16 384 rows x 49 152, top-20, cosine: every launch of the screening GEMM has the shape it has in the full
shard (10 000 x 8 192 x 24 576 per launch, two K chunks per panel), next to the merge / compaction / re-scoring kernels.
usage: python tools/pmc_target_screen.py [reps=2]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from anyloc_amd import retrieval  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda", 0)
db = bench.synthetic_db(16384, 32, 1536, dev, seed=100)
qu = bench.synthetic_db(10000, 32, 1536, dev, seed=500)
index = retrieval.FlatIndex(db, "cosine", planes=True)
for _ in range(reps):
    d, i = index.search(qu, 20)
torch.cuda.synchronize()
print("ok", tuple(i.shape))
