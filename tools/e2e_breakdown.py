"""Where does a FRESH batch spend its time before the forward?  (each stage timed with a device synchronisation)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd import build_batch_octree, load_config, model_factory, synthetic as syn
from hotformerloc_amd.plan import WindowPlan
params, depth = load_config('wild-places')
model = model_factory(params); syn.fill_synthetic_weights(model, 'init'); model = model.cuda().eval()
clouds = syn.make_clouds(2, 32, 4096, params.coordinates)
dev_clouds = [torch.from_numpy(c).cuda() for c in clouds]
def T(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return r, (time.perf_counter() - t0) * 1e3
with torch.inference_mode():
    for it in range(4):
        o, t_build = T(lambda: build_batch_octree(dev_clouds, depth, 2, 'cuda', construct_neigh=False))
        _, t_neigh = T(lambda: o.construct_all_neigh())
        _, t_fwd1 = T(lambda: model({'octree': o}))
        _, t_fwd2 = T(lambda: model({'octree': o}))
        print('iter %d: build+merge %.2f ms, neighbours+tap lists %.2f ms, first forward (plan, tiles, caches) %.2f ms, second forward %.2f ms'
              % (it, t_build, t_neigh, t_fwd1, t_fwd2))
