"""A few forwards of the bench workload (for rocprofv3 --kernel-trace): python tools/fwd_steps.py [steps] [boundary]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd import build_batch_octree, load_config, model_factory  # noqa: E402
from hotformerloc_amd import synthetic as syn  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
boundary = len(sys.argv) > 2 and sys.argv[2] == 'boundary'
params, depth = load_config('wild-places')
model = model_factory(params)
syn.fill_synthetic_weights(model, 'init')
model = model.cuda().eval()
octree = build_batch_octree(syn.make_clouds(2, 32, 4096, params.coordinates), depth, 2, 'cuda')
with torch.inference_mode():
    for _ in range(3):
        model({'octree': octree})
    torch.cuda.synchronize()
    for _ in range(steps):
        if boundary:
            octree.drop_forward_caches()
        model({'octree': octree})
    torch.cuda.synchronize()
