"""Probe: oracle (CPU port of the reference forward) clouds/s vs torch thread count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hotformerloc_amd import load_config, synthetic as syn
from oracle import hotformer_ref
from oracle.testing import oracle_octree, synthetic_state_dict
params, depth = load_config('wild-places')
sd = synthetic_state_dict(params, 'init')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
octree = oracle_octree(syn.make_clouds(2, B, 4096, params.coordinates), depth)
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    hotformer_ref.forward(sd, params, octree)
    t0 = time.perf_counter(); hotformer_ref.forward(sd, params, octree); dt = time.perf_counter() - t0
    print('threads', th, 'B', B, '%.2f s' % dt, '%.3f clouds/s' % (B / dt), flush=True)
