import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hotformerloc_amd import build_batch_octree, load_config, synthetic as syn
params, depth = load_config('wild-places')
clouds = syn.make_clouds(2, 32, 4096, params.coordinates)
for rep in range(3):
    octree = build_batch_octree(clouds, depth, 2, 'cuda')
    octree.construct_all_neigh(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for d in (6, 5):
        src, slot, edges = octree.sparse_taps(d)
    torch.cuda.synchronize()
    print('sparse_taps d6+d5: %.2f ms' % ((time.perf_counter() - t0) * 1e3), [e[-1] for e in (octree.sparse_taps(6)[2], octree.sparse_taps(5)[2])],
          'dense pairs', 27 * int(octree.nnum_nempty[6]), 27 * int(octree.nnum_nempty[5]))
