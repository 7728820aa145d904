"""Which torch.nn.functional.linear / mm / matmul / addmm / bmm calls the TRAINING path still makes (i.e. what does not go through
the hand-written GEMMs), by caller and shape, for one forward + backward of the CS-Wild-Places config."""
import collections, sys, traceback, torch
sys.path.insert(0, '/root/repo' if __import__('os').path.exists('/root/repo/bench.py') else '.')
import torch.nn.functional as F
from hotformerloc_amd import build_batch_octree, load_config, model_factory, synthetic as syn
cnt = collections.Counter()
def wrap(name, fn):
    def w(*a, **k):
        st = traceback.extract_stack(limit=6)
        who = ' < '.join('%s:%d' % (f.name, f.lineno) for f in reversed(st[:-1]) if 'hotformerloc_amd' in f.filename)[:110]
        shp = tuple(tuple(x.shape) for x in a[:2] if hasattr(x, 'shape'))
        cnt[(name, who, shp)] += 1
        return fn(*a, **k)
    return w
F.linear = wrap('F.linear', F.linear)
torch.mm = wrap('mm', torch.mm); torch.matmul = wrap('matmul', torch.matmul); torch.addmm = wrap('addmm', torch.addmm); torch.bmm = wrap('bmm', torch.bmm)
params, depth = load_config('cs-wild-places')
model = model_factory(params); syn.fill_synthetic_weights(model, 'init'); model = model.cuda().train()
clouds = syn.make_clouds(2, 8, 4096, params.coordinates)
octree = build_batch_octree(clouds, depth, 2, 'cuda')
y = model({'octree': octree})['global']; y.sum().backward(); torch.cuda.synchronize()
cnt.clear()
model.zero_grad(set_to_none=True)
y = model({'octree': octree})['global']; y.sum().backward(); torch.cuda.synchronize()
for k, v in cnt.most_common(25): print(v, k)
print('total python-level calls', sum(cnt.values()))
