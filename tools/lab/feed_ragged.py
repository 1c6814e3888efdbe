"""bench.py's `secondary.feed_ragged` alone (BucketedBatchFeeder over ragged TVR batches): python tools/lab/feed_ragged.py [steps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    torch.cuda.set_device(0)
    import hero_amd
    hero_amd.set_compute_dtype(torch.bfloat16)
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    print(json.dumps(bench.feed_ragged_run("cuda:0", 0, steps=steps, warmup=6, n_buckets=nb)))
