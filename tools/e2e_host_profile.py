import os, sys, time, torch, cProfile, pstats
sys.path.insert(0, '/root/repo')
from hotformerloc_amd import build_batch_octree, load_config, model_factory, synthetic as syn
params, depth = load_config('wild-places')
model = model_factory(params); syn.fill_synthetic_weights(model, 'init'); model = model.cuda().eval()
clouds = syn.make_clouds(2, 32, 4096, params.coordinates)
dev_clouds = [torch.from_numpy(c).cuda() for c in clouds]
def fresh():
    o = build_batch_octree(dev_clouds, depth, 2, 'cuda', construct_neigh=True)
    return model({'octree': o})
with torch.inference_mode():
    for _ in range(3): fresh()
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10): fresh()
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr); st.sort_stats('cumulative').print_stats(45)
