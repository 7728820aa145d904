"""Capture one forward of the default workload NATIVELY (hipStreamBeginCapture / hipStreamEndCapture around the model call, relaxed
mode, no torch.cuda.graphs: that path crashes in capture_end on this stack, profiles/r05_g_hip_graph_probe.log) and replay it:
the GPU-only time of a forward against the eager step.  Logs the outcome either way."""
import ctypes, faulthandler, functools, os, sys, time
faulthandler.enable()
print = functools.partial(print, flush=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hotformerloc_amd import build_batch_octree, load_config, model_factory, synthetic as syn

hip = ctypes.CDLL('libamdhip64.so')
for fn in ('hipStreamBeginCapture', 'hipStreamEndCapture', 'hipGraphInstantiate', 'hipGraphLaunch', 'hipGraphGetNodes'):
    getattr(hip, fn).restype = ctypes.c_int
params, depth = load_config('wild-places')
model = model_factory(params); syn.fill_synthetic_weights(model, 'init'); model = model.cuda().eval()
octree = build_batch_octree(syn.make_clouds(2, 32, 4096, params.coordinates), depth, 2, 'cuda')
batch = {'octree': octree}
n = 20
s = torch.cuda.Stream()
with torch.inference_mode(), torch.cuda.stream(s):
    for _ in range(8):
        ref = model(batch)['global']
    s.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        model(batch)
    s.synchronize()
    t1 = time.perf_counter()
print('eager (window plan cached on the octree): %.2f ms/step' % ((t1 - t0) / n * 1e3))

handle = ctypes.c_void_p(s.cuda_stream)
rc = hip.hipStreamBeginCapture(handle, 2)                       # hipStreamCaptureModeRelaxed
print('hipStreamBeginCapture ->', rc)
if rc != 0:
    sys.exit(0)
err = None
try:
    with torch.inference_mode(), torch.cuda.stream(s):
        out = model(batch)['global']
except Exception as e:                                           # noqa: BLE001
    err = repr(e)[:600]
graph = ctypes.c_void_p()
rc = hip.hipStreamEndCapture(handle, ctypes.byref(graph))
print('hipStreamEndCapture ->', rc, '| exception inside the captured forward:', err)
if rc != 0 or not graph.value:
    sys.exit(0)
nn = ctypes.c_size_t(0)
hip.hipGraphGetNodes(graph, None, ctypes.byref(nn))
print('captured graph: %d nodes' % nn.value)
gexec = ctypes.c_void_p()
rc = hip.hipGraphInstantiate(ctypes.byref(gexec), graph, None, None, ctypes.c_size_t(0))
print('hipGraphInstantiate ->', rc)
if rc != 0:
    sys.exit(0)
for _ in range(3):
    rc = hip.hipGraphLaunch(gexec, handle)
s.synchronize()
print('hipGraphLaunch ->', rc)
t0 = time.perf_counter()
for _ in range(n):
    hip.hipGraphLaunch(gexec, handle)
s.synchronize()
t1 = time.perf_counter()
print('graph replay: %.2f ms/step (no host issue work); max |replayed - eager descriptors| = %.2e'
      % ((t1 - t0) / n * 1e3, (out - ref).abs().max().item()))
