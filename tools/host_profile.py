"""cProfile of the host side of one forward (issue only, no sync): python tools/host_profile.py [early 0/1]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from hotformerloc_amd import build_batch_octree, load_config, model_factory, synthetic as syn  # noqa: E402

params, depth = load_config('wild-places')
model = model_factory(params)
syn.fill_synthetic_weights(model, 'init')
model = model.cuda().eval()
octree = build_batch_octree(syn.make_clouds(2, 32, 4096, params.coordinates), depth, 2, 'cuda')
batch = {'octree': octree}
with torch.inference_mode():
    for _ in range(8):
        model(batch)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        model(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('host issue %.2f ms/step, wall %.2f ms/step' % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        model(batch)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats('cumulative').print_stats(45)
