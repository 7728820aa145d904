import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29591')
mode = sys.argv[1]
if mode != 'none':
    dist.init_process_group('nccl', rank=0, world_size=1)
    t = torch.ones(4, device='cuda'); dist.all_reduce(t); torch.cuda.synchronize()
from hotformerloc_amd import build_batch_octree, load_config, model_factory, synthetic as syn
params, depth = load_config('wild-places')
model = model_factory(params); syn.fill_synthetic_weights(model, 'init'); model = model.cuda().eval()
octree = build_batch_octree(syn.make_clouds(2, 32, 4096, params.coordinates), depth, 2, 'cuda')
batch = {'octree': octree}
with torch.inference_mode():
    for _ in range(8): model(batch)
    torch.cuda.synchronize()
    n = 20; t0 = time.perf_counter()
    for _ in range(n): model(batch)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(mode, 'host issue %.2f ms/step, wall %.2f ms/step' % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3), 'affinity', len(os.sched_getaffinity(0)))
if mode != 'none': dist.destroy_process_group()
