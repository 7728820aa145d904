"""cProfile of the HOST side of one resident forward (issue only; sorted by self time)."""
import os, sys, torch, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd import build_batch_octree, load_config, model_factory, synthetic as syn
params, depth = load_config('wild-places')
model = model_factory(params); syn.fill_synthetic_weights(model, 'init'); model = model.cuda().eval()
octree = build_batch_octree(syn.make_clouds(2, 32, 4096, params.coordinates), depth, 2, 'cuda', construct_neigh=True)
batch = {'octree': octree}
with torch.inference_mode():
    for _ in range(5): model(batch)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10): model(batch)
    pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(28)
