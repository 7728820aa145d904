"""Which torch (non-library) ops does one inference forward still launch, and from where?  (torch.profiler, stacks)"""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd import build_batch_octree, load_config, model_factory, synthetic as syn
params, depth = load_config('wild-places')
model = model_factory(params); syn.fill_synthetic_weights(model, 'init'); model = model.cuda().eval()
octree = build_batch_octree(syn.make_clouds(2, 32, 4096, params.coordinates), depth, 2, 'cuda', construct_neigh=True)
batch = {'octree': octree}
with torch.inference_mode():
    for _ in range(3): model(batch)
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA],
                                with_stack=True) as prof:
        model(batch); torch.cuda.synchronize()
rows = []
for ev in prof.key_averages(group_by_stack_n=8):
    t = getattr(ev, 'self_device_time_total', 0) or 0
    if t <= 0 or not ev.key.startswith('aten::'):
        continue
    frames = [f for f in (ev.stack or []) if 'hotformerloc_amd' in f]
    where = ' <- '.join(f.split('hotformerloc_amd/')[-1].split(' ')[0] for f in frames[:3]) if frames else '?'
    rows.append((t, ev.count, ev.key, where))
for t, n, name, where in sorted(rows, reverse=True)[:45]:
    print('%-26s x%3d %8.1f us  %s' % (name, n, t, where))
