"""How long does the host take to ISSUE one forward (no synchronisation) vs how long the GPU takes to run it?"""
import faulthandler, functools, os, sys, time
faulthandler.enable()
print = functools.partial(print, flush=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hotformerloc_amd import build_batch_octree, load_config, model_factory, synthetic as syn
params, depth = load_config('wild-places')
model = model_factory(params); syn.fill_synthetic_weights(model, 'init'); model = model.cuda().eval()
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
print('host issue %.2f ms/step, wall %.2f ms/step, OMP_NUM_THREADS=%s, torch threads %d'
      % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, os.environ.get('OMP_NUM_THREADS'), torch.get_num_threads()))

# GPU-only time of the same forward: capture it once into a HIP graph and replay (no host issue work at all)
try:
    static_out = None
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.inference_mode():
        for _ in range(2):
            model(batch)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.inference_mode(), torch.cuda.graph(g):
        static_out = model(batch)['global']
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    with torch.inference_mode():
        ref = model(batch)['global']
    print('graph replay %.2f ms/step (GPU-only), max |replayed - eager| = %.2e'
          % ((t1 - t0) / n * 1e3, (static_out - ref).abs().max().item()))
except Exception as e:                                   # noqa: BLE001
    print('graph capture failed:', repr(e)[:500])
