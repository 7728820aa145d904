#!/usr/bin/env python3
"""bench.py -- clouds/s of the HOTFormerLoc encoder forward on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched as
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, one rank per
GPU over RCCL -- or plainly as `python bench.py --gpus N`: with no RANK in the environment the script starts its own
ranks that way (a child process; this parent never touches the GPU) and relays rank 0's line and the exit code.  W untimed warm-up steps, then EXACTLY K timed steps bracketed by a barrier +
`torch.cuda.synchronize()`; max over ranks; rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1]): batch of 32 synthetic 4096-point clouds per GPU,
Wild-Places cfg (octree depth 7, cylindrical, K=48), forward only, eval mode, random-init
style closed-form weights.  The timed step is what the reference's `model(batch)` is
(`models/hotformerloc.py:33-59`): the batch octree is on the device with its 27-neighbour tables
built (`misc/torch_utils.py:47-51` happens before `model(batch)`), and EVERYTHING the reference
derives from it inside the forward is derived inside the timed step here too -- the window / relay-token
plan (`OctreeT.__init__` + `build_t`, `models/hotformerloc_backbone.py:712-716`), the live-tap lists and
row-tile tables of the octree convolutions (ocnn's `octree2col` per call): `Octree.drop_forward_caches()`
runs at the top of every step.  For N > 1 the step ends with the RCCL all-gather of the (B_local,256)
descriptors.  Weak scaling: each rank encodes its own contiguous slice of the global batch.

What one default run times (same W-warm-up / K-step / barrier protocol for every leg, rank 0, N = 1):
  value          boundary-faithful step, Linear layers as 3-term split-bf16 products with fp32 accumulation on the
                 hand-written MFMA GEMM (`--gemm x3`), window attention on fp16 (hi, lo) pairs
  resident_plan  the same step with the plan / tap tables kept on the octree between steps (what a caller that
                 re-submits one octree sees; round 2's headline)
  matched_precision   the boundary-faithful step at the REFERENCE'S ARITHMETIC on hand-written kernels: every
                 transformer-block Linear on hfl_linear_x6 (fp32-grade products: three bf16 planes per operand, six plane
                 products, f32 accumulation; csrc/gemm_x6.hip), f32 LayerNorm / softmax / GELU, f32-MFMA window attention;
                 with its own roofline (hfl_linear_x6), its parity against the reference golden, and, as context, the same
                 step on the fp32 library GEMM (`fp32_library_gemm`).  Repeated under config.matched_precision
  e2e            a FRESH octree per step from device-resident points: device build + neighbour tables + forward
  train_cs       BASELINE config 3 (CS-Wild-Places cfg, B = 64, forward + backward), in a child process; activations kept
                 (checkpoint policy 'auto': they fit) -- train_cs_checkpointed = the same with the reference's
                 per-block recomputation (bitwise the same gradients)
  oxford         BASELINE config 5's per-rank workload (Oxford cfg, B = 64, octree depth 9), in a child process
  unpinned_host  the headline step with the CPU affinity left alone, in a child process (the headline itself runs on its rank's
                 eighth of the host's logical CPUs: --pin-cores); every leg's `host_issue` = host time to queue the K steps vs
                 their wall time
and then, outside any timed value: the `roofline` legs (HIP events per launch: `roofline` = the step's dominant kernel, the
fused MLP launch; `roofline_window` the fp16 window kernel, `roofline_fused` for the
one-kernel LayerNorm -> qkv -> attention launch of the OctFormer stage, `roofline_ws` for the one with relay tokens of the
finest pyramid level, `roofline_fp32` for the f32-MFMA window kernel of the matched-precision leg) and the `cpu_baseline`
(the CPU oracle, a port of the reference forward, BASELINE.md section 3 protocol) whose descriptors are also the
`parity` reference of the timed workload's GPU descriptors (BASELINE metric: "descriptor L2 vs ref").
"""

import argparse
import json
import os
import statistics
import subprocess
import sys
import time

# hipBLASLt schedule (hotformerloc_amd/__init__.py): data-parallel for the forward path; the training step's weight
# gradients contract over ~10^5 rows into small (N, K) outputs and need the stream-K split (145 vs 206 ms/step)
os.environ.setdefault('TENSILE_STREAMK_DATA_PARALLEL', '0' if '--train' in sys.argv else '1')
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32 matrix peak (v_mfma_f32_16x16x4_f32)
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16 / bf16 matrix peak (v_mfma_f32_16x16x32_f16)
ATTN = 'hfl_window_attention_fwd'


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', default='wild-places')
    ap.add_argument('--batch', type=int, default=None, help='clouds per GPU (default 32; 64 for cs-wild-places --train)')
    ap.add_argument('--points', type=int, default=4096)
    ap.add_argument('--no-drop-path', action='store_true',
                    help='--train: switch the config\'s per-cloud stochastic depth (drop_path = 0.5) off')
    ap.add_argument('--keep-drop-path', action='store_true', help='(default now; kept for old command lines)')
    ap.add_argument('--points-max', type=int, default=None,
                    help='variable density: per-cloud point count ~ U{points..points_max}, forest / unit-ball mix '
                         '(default for cs-wild-places: 32768, BASELINE config 3)')
    ap.add_argument('--gemm', default='x3', choices=['x3', 'bf16x3', 'fp32', 'x6'],
                    help="Linear layers of the headline `value`: 'x3' = hand-written split-bf16 MFMA GEMM with fused "
                         "bias/GELU/residual epilogues (default); 'bf16x3' = the same three-term split as one hipBLASLt bf16 "
                         "GEMM over K-concatenated operands; 'fp32' = hipBLASLt fp32 GEMMs; 'x6' = matched precision: hand-written "
                         "fp32-grade GEMM on three bf16 planes per operand (csrc/gemm_x6.hip) + fp32-MFMA attention")
    ap.add_argument('--resident-plan', action='store_true',
                    help='headline step keeps the window plan / tap tables cached on the octree (round-2 behaviour)')
    ap.add_argument('--no-streams', action='store_true', help='pyramid depths on one stream')
    ap.add_argument('--serial-streams', action='store_true',
                    help="the step's own launch schedule on ONE stream (what the roofline leg times: a kernel trace of this "
                         'run shows every kernel alone; profiles/*_serial_*)')
    ap.add_argument('--attn-variant', type=int, default=0, help='A/B: window attention kernel variant (0 = default)')
    ap.add_argument('--train', action='store_true', help='time forward+backward (BASELINE config 3) instead of forward')
    ap.add_argument('--multistaged', action='store_true',
                    help='with --train: the full multi-staged step (stage 1 no-grad encode, TruncatedSmoothAP on the '
                         'all-gathered descriptors, stage 3 forward+backward, gradient all-reduce, AdamW step)')
    ap.add_argument('--no-train-x3', action='store_true', help='A/B with --train: Linear layers as fp32 torch GEMMs')
    ap.add_argument('--x3-nt', type=int, default=None, help='A/B: non-temporal store bits of the hand-written GEMM (0..3)')
    ap.add_argument('--no-collective', action='store_true', help='A/B: skip the descriptor all-gather (N > 1 diagnostics)')
    ap.add_argument('--no-extras', action='store_true', help='only the headline timed region (no other legs)')
    ap.add_argument('--no-train-leg', action='store_true', help='skip the config-3 child process')
    ap.add_argument('--no-oxford-leg', action='store_true', help="skip the config-5 per-rank workload's child process")
    ap.add_argument('--no-pinned-leg', action='store_true', help='skip the unpinned-host child process')
    ap.add_argument('--dry-launch', action='store_true',
                    help='--gpus N without a launcher: print the child command this process would start, and exit')
    ap.add_argument('--pin-cores', type=int, default=-1,
                    help='restrict this process to N logical CPUs before anything else runs: -1 (default) = 1/8 of the host\'s, '
                         'the slice of this rank (LOCAL_RANK) -- the share one of 8 ranks on a node has, which is also how a '
                         'deployment pins one process per GPU; 0 = leave the affinity alone (the `unpinned_host` leg)')
    ap.add_argument('--master-port', type=int, default=None, help='self-launch: rendezvous port (default: a free one)')
    ap.add_argument('--train-leg-steps', type=int, default=5)
    ap.add_argument('--train-leg-warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-threads', type=int, default=16,
                    help='torch threads of the CPU baseline (8-16 is the optimum measured on the 2x64-core host; more threads are slower)')
    ap.add_argument('--cpu-budget-s', type=float, default=240.0,
                    help='stop adding timed CPU runs once this much wall time is spent (at least 1 run per batch size)')
    return ap.parse_args()


def cpu_baseline(params, depth, args):
    """Oracle forward (port of the reference, torch CPU fp32), BASELINE.md section 3: octree prebuilt with neighbour
    tables (the model boundary), eval / inference mode, 2 warm-ups + 5 timed runs, median, at B=1 (BASELINE config 1)
    and at the GPU workload's batch (B=32, the first clouds of the same workload).  Bounded by --cpu-budget-s.
    Returns (record, descriptors of the B = args.batch run as numpy) -- the latter is the parity reference."""
    import torch
    from hotformerloc_amd import synthetic as syn
    from oracle import hotformer_ref
    from oracle.testing import oracle_octree, synthetic_state_dict
    cores = min(args.cpu_threads, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    sd = synthetic_state_dict(params, 'init')
    t_start = time.perf_counter()
    out = {}
    want = None
    for b in (1, args.batch):
        clouds = syn.make_clouds(2, b, args.points, params.coordinates)
        t0 = time.perf_counter()
        octree = oracle_octree(clouds, depth)                                # boundary: prebuilt, timed separately
        t_build = time.perf_counter() - t0
        runs = []
        for i in range(2 + 5):
            if i >= 3 and time.perf_counter() - t_start > args.cpu_budget_s * (0.1 if b == 1 else 1.0):
                break
            t0 = time.perf_counter()
            with torch.inference_mode():
                y = hotformer_ref.forward(sd, params, octree)
            dt = time.perf_counter() - t0
            log('cpu baseline: B=%d run %d %.2f s%s' % (b, i, dt, ' (warm-up)' if i < 2 else ''))
            if i >= 2:
                runs.append(dt)
        if b == args.batch:
            want = y.numpy().copy()
        med = statistics.median(runs)
        out[b] = {'clouds_per_s': round(b / med, 4), 'median_s': round(med, 3), 'timed_runs': len(runs),
                  'warmups': 2, 'octree_build_s': round(t_build, 3)}
    big = out[args.batch]
    return {'value': big['clouds_per_s'], 'unit': 'clouds/s', 'cores': cores, 'kind': 'port',
            'sample': 'oracle forward (CPU port of the reference), %s cfg, first %d clouds x %d pts of the GPU workload as '
                      'one batch, octree + neighbours prebuilt, 2 warm-ups + %d timed runs, median %.2f s; torch %d '
                      'threads (host has %d logical CPUs)'
                      % (args.config, args.batch, args.points, big['timed_runs'], big['median_s'],
                         torch.get_num_threads(), os.cpu_count() or 1),
            'b1': out[1], 'b%d' % args.batch: big}, want


def parity_record(got, want):
    """Relative L2 distance per cloud between GPU descriptors and the oracle's (north_star: <= 1e-3)."""
    import numpy as np
    rel = np.linalg.norm(got - want, axis=1) / np.linalg.norm(want, axis=1)
    return {'max_rel_l2': float('%.3e' % rel.max()), 'mean_rel_l2': float('%.3e' % rel.mean()),
            'clouds': int(got.shape[0]), 'bar': 1e-3, 'ok': bool(np.isfinite(got).all() and rel.max() <= 1e-3)}


def train_leg(args, env_extra=None):
    """BASELINE config 3 (CS-Wild-Places cfg, B = 64, 4096..32768 points per cloud, forward + backward) timed by this same
    script in a CHILD process (its own hipBLASLt schedule, its own allocator), same warm-up / step / barrier protocol."""
    saved = os.environ.pop('TENSILE_STREAMK_DATA_PARALLEL', None)
    try:
        return child_leg(['--config', 'cs-wild-places', '--train', '--steps', str(args.train_leg_steps), '--warmup',
                          str(args.train_leg_warmup)],
                         'child process: python bench.py --config cs-wild-places --train (BASELINE config 3)',
                         keys=('peak_memory_GiB', 'checkpointing'), env_extra=env_extra)
    finally:
        if saved is not None:
            os.environ['TENSILE_STREAMK_DATA_PARALLEL'] = saved


def child_leg(extra, what, timeout=900, keys=(), env_extra=None):
    """One more workload timed by this same script in a CHILD process (same warm-up / step / barrier protocol)."""
    cmd = [sys.executable, os.path.abspath(__file__), '--no-extras', '--no-cpu-baseline'] + extra
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.update(env_extra or {})
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    except subprocess.TimeoutExpired:
        return {'error': 'timeout'}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    if r.returncode != 0 or not lines:
        return {'error': 'rc %d' % r.returncode, 'stderr_tail': r.stderr[-400:]}
    j = json.loads(lines[-1])
    out = {'value': j['value'], 'unit': j['unit'], 'ms_per_step': j['ms_per_step'], 'steps': j['steps'],
           'warmup': j['warmup'], 'workload': j['config']['workload'], 'dtype': j['dtype'],
           'wall_s': round(time.perf_counter() - t0, 1), 'what': what}
    for k in keys:
        if k in j:
            out[k] = j[k]
    return out


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start N fresh ranks with torch.distributed.run as a CHILD
    process and relay rank 0's JSON line and the children's exit code.  This parent never touches the GPU (no HIP call, no
    torch import) and never exec()s; the under-torchrun path below is what the children run."""
    import socket
    port = args.master_port
    if port is None:
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
    argv = [a for a in sys.argv[1:] if a != '--dry-launch']
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + argv
    if args.dry_launch:
        print(json.dumps({'launch': cmd}), flush=True)
        return 0
    log('no RANK in the environment: launching %d ranks: %s' % (args.gpus, ' '.join(cmd)))
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    r = subprocess.run(cmd, env=env)                       # stdout / stderr inherited: rank 0's line goes straight out
    return r.returncode


_T0 = time.perf_counter()


def log(*a):
    print('[bench %.1fs]' % (time.perf_counter() - _T0), *a, file=sys.stderr, flush=True)


def rank_report(dist, elapsed, steps, device):
    """Per-rank ms/step (all-gathered, in rank order) and the MAX over ranks of the timed region's seconds."""
    import torch
    world = dist.get_world_size()
    tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
    allt = [torch.zeros_like(tt) for _ in range(world)]
    dist.all_gather(allt, tt)
    per_rank = [round(float(x.item()) / steps * 1e3, 3) for x in allt]
    t = tt.clone()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return per_rank, float(t.item())


def pin_host(args):
    """One process per GPU, each on its own slice of the host's logical CPUs (args.pin_cores; -1: an eighth, by LOCAL_RANK).
    The launching parent of --gpus N pins nothing (its children do).  Measured on the 256-CPU bench box: 2874-2936 clouds/s
    with the scheduler free to move the Python thread and the HIP runtime's helpers, 2938-2960 pinned (alternating runs,
    profiles/r04_ad_ab_host_pinning.log)."""
    args.host_affinity = 'not set'
    if not hasattr(os, 'sched_setaffinity'):
        return
    if args.pin_cores == 0:
        # a child leg of a pinned parent inherits its mask: back to what the parent started with
        orig = os.environ.get('HFL_BENCH_AFFINITY0')
        if orig:
            try:
                os.sched_setaffinity(0, {int(c) for c in orig.split(',')})
            except (OSError, ValueError):
                pass
        return
    if args.gpus > 1 and 'RANK' not in os.environ:
        return
    allowed = sorted(os.sched_getaffinity(0))
    os.environ.setdefault('HFL_BENCH_AFFINITY0', ','.join(str(c) for c in allowed))
    n = args.pin_cores if args.pin_cores > 0 else max(1, (os.cpu_count() or len(allowed)) // 8)
    n = min(n, len(allowed))
    slot = int(os.environ.get('LOCAL_RANK', '0')) if args.pin_cores < 0 else 0
    first = slot * n if (slot + 1) * n <= len(allowed) else 0
    os.sched_setaffinity(0, set(allowed[first:first + n]))
    os.environ['OMP_NUM_THREADS'] = str(n)
    args.host_affinity = '%d of %d logical CPUs (slice %d)' % (n, os.cpu_count() or len(allowed), slot)


def main():
    args = parse()
    pin_host(args)
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(self_launch(args))
    if args.dry_launch:
        print(json.dumps({'launch': None, 'note': 'nothing to launch: --gpus 1, or already under a launcher'}), flush=True)
        return
    import torch
    import torch.distributed as dist
    from hotformerloc_amd import build_batch_octree, load_config, model_factory, ops
    from hotformerloc_amd import synthetic as syn
    from hotformerloc_amd.model import set_gemm_mode, set_pyramid_streams
    set_gemm_mode(args.gemm)
    set_pyramid_streams('serial' if args.serial_streams else not args.no_streams)
    if args.no_train_x3:
        from hotformerloc_amd.model import set_train_x3
        set_train_x3(False)
    if args.batch is None:
        args.batch = 64 if (args.config == 'cs-wild-places' and args.train) else 32
    if args.points_max is None and args.config == 'cs-wild-places':
        args.points_max = 32768

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks' % (args.gpus, world))
    n_dev = torch.cuda.device_count()                   # counting devices does not initialise the GPU
    if n_dev < world or local_rank >= n_dev:
        raise SystemExit('bench.py rank %d: need %d GPUs on this node (one rank per GPU), found %d'
                         % (rank, world, n_dev))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    use_dist = world > 1 or 'RANK' in os.environ        # under torchrun the RCCL path runs even at N=1
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        # no device_id: binding the group to the device at init (eager communicator) cost 0.8 ms per step on the
        # host-bound issue path of this model (19.3 vs 20.1 ms/step at N=1 under the launcher)
        dist.init_process_group('nccl')

    params, depth = load_config(args.config)
    if args.train:
        if args.no_drop_path:
            params.drop_path = 0.0
    model = model_factory(params)
    syn.fill_synthetic_weights(model, 'init')
    model = model.to(dev).eval()
    log('model ready')

    # this rank's contiguous slice of the global batch (ordered)
    clouds = bench_clouds(syn, params, args, rank)
    octree = build_batch_octree(clouds, depth, 2, dev, construct_neigh=True)
    batch = {'octree': octree}
    torch.cuda.synchronize()
    log('octree ready', octree.nnum_nempty.tolist())
    if args.attn_variant:
        from hotformerloc_amd import _native
        _native.load().hfl_set_variant(b'window_attention', args.attn_variant)
    if args.x3_nt is not None:
        from hotformerloc_amd import _native
        _native.load().hfl_set_variant(b'x3_dbg', 0x100 | (args.x3_nt & 3))
    from hotformerloc_amd.distributed import all_gather_descriptors

    if args.train:
        model.train()
        proj = torch.from_numpy(syn.hash_uniform(99, args.batch * params.output_dim).reshape(
            args.batch, params.output_dim).astype('float32')).to(dev)

    if args.train and args.multistaged:
        from hotformerloc_amd.losses import TruncatedSmoothAP
        from hotformerloc_amd.training import multistaged_training_step
        n_tot = args.batch * world
        lab = torch.arange(n_tot) // 4                     # groups of 4 consecutive clouds are mutual positives
        pos_mask = ((lab[:, None] == lab[None, :]) & ~torch.eye(n_tot, dtype=torch.bool)).to(dev)
        neg_mask = (lab[:, None] != lab[None, :]).to(dev)
        loss_fn = TruncatedSmoothAP(tau1=0.01, positives_per_query=4)
        optim = torch.optim.AdamW(model.parameters(), lr=1e-5)
        from hotformerloc_amd.training import OverlappedGradReducer
        reducer = OverlappedGradReducer(model.parameters())     # gradient all-reduce bucket by bucket behind the backward

    collective = use_dist and not args.no_collective and not args.train
    state = {'fresh_plan': not args.resident_plan}

    def step():
        # the reference boundary: octree + 27-neighbour tables resident, everything else derived inside model(batch)
        if state['fresh_plan']:
            octree.drop_forward_caches()
        if args.train and args.multistaged:
            multistaged_training_step(model, [batch], pos_mask, neg_mask, loss_fn, optim, n_total=n_tot, reducer=reducer)
            return torch.ones(1, device=dev)
        if args.train:
            model.zero_grad(set_to_none=True)
            y = model(batch)['global']
            (y * proj).sum().backward()
            return y.detach()
        y = model(batch)['global']
        if collective:
            all_gather_descriptors(y, args.batch * world, force=True)      # (B_total, 256) on every rank
        return y

    dev_clouds = [torch.from_numpy(c).to(dev) for c in clouds]             # e2e leg: points resident, octree not

    def step_e2e():
        fresh = build_batch_octree(dev_clouds, depth, 2, dev, construct_neigh=True)
        return model({'octree': fresh})['global']

    last = {}

    def timed(fn, steps, warmup, tag=None):
        """W untimed warm-up steps, then exactly K timed steps bracketed by barrier + synchronize; seconds (this rank)."""
        for i in range(warmup):
            fn()
            torch.cuda.synchronize()
        if use_dist:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            y = fn()
        t_issued = time.perf_counter() - t0           # the host has queued everything; the GPU may still be running
        if use_dist:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert torch.isfinite(y).all()
        if tag is not None:
            last[tag] = y.detach().float().cpu().numpy()
            last[tag + '_issue_s'] = t_issued
        return dt

    def leg(dt):
        return {'value': round(args.batch * args.steps / dt, 2), 'unit': 'clouds/s',
                'ms_per_step': round(dt / args.steps * 1e3, 3), 'steps': args.steps, 'warmup': args.warmup}

    extras = world == 1 and not args.train and not args.no_extras
    resident_line = fp32_line = e2e_line = kern_iso = kern_all = iso_sizes = kern_iso32 = fused_roof = ws_roof = mlp_roof = None
    lib_line = x6_sizes = None
    with (torch.enable_grad() if args.train else torch.inference_mode()):
        # ---- the headline: W warm-ups, K timed steps, nothing instrumented inside the timed region
        elapsed = timed(step, args.steps, args.warmup, tag='value')
        log('timed region done: %.3f s for %d steps' % (elapsed, args.steps))
        if extras:
            state['fresh_plan'] = not state['fresh_plan']
            dt = timed(step, args.steps, args.warmup)
            state['fresh_plan'] = not state['fresh_plan']
            resident_line = leg(dt)
            resident_line['what'] = ('window plan, tap lists and tile tables %s between steps'
                                     % ('rebuilt' if args.resident_plan else 'kept on the octree'))
            log('resident-plan leg: %.3f s' % dt)
            # ---- the MATCHED-PRECISION leg: the reference's own arithmetic (fp32-grade Linear products, fp32 LayerNorm / softmax
            # / GELU, fp32-MFMA window attention) on hand-written kernels -- no library GEMM in the transformer blocks
            other = 'x6' if args.gemm != 'x6' else 'x3'
            set_gemm_mode(other)
            dt = timed(step, args.steps, args.warmup, tag='other')
            fp32_line = leg(dt)
            fp32_line['gemm'] = other
            log('%s Linear leg: %.3f s' % (other, dt))
            if other == 'x6':
                # roofline of the leg's dominant kernel (hfl_linear_x6) and of the fp32-MFMA window kernel it runs (v4), the
                # step's schedule on one stream: a launch's events then bracket the kernel alone
                set_pyramid_streams('serial' if not args.no_streams else False)
                step()
                torch.cuda.synchronize()
                with ops.KernelTimer(only=[ATTN, 'hfl_linear_x6']) as t32:
                    for _ in range(args.steps):
                        step()
                    torch.cuda.synchronize()
                kern_iso32 = t32.summary()
                x6_sizes = t32.by_size('hfl_linear_x6')
                set_pyramid_streams(not args.no_streams)
                # context: the same step with the fp32 library GEMM (hipBLASLt through F.linear) where hfl_linear_x6 runs
                set_gemm_mode('fp32')
                dt = timed(step, args.steps, args.warmup, tag='lib')
                lib_line = leg(dt)
                lib_line['gemm'] = 'fp32'
                log('fp32 library-GEMM leg: %.3f s' % dt)
            set_gemm_mode(args.gemm)
            dt = timed(step_e2e, args.steps, args.warmup)
            e2e_line = leg(dt)
            e2e_line['what'] = ('fresh octree every step from device-resident points: HIP octree build + merge + '
                                'neighbour tables + live-tap lists (one device->host read of counts each for build and '
                                'taps) + window plan + forward')
            log('e2e leg: %.3f s' % dt)
            # ---- roofline legs (not part of any value): every hand-written kernel instrumented, streams as in the
            # timed region, then once more with the pyramid depths on ONE stream.  With three streams sharing the GPU
            # a kernel's start-to-end event time includes the CUs it lent to its neighbours; serialised, the events
            # bracket the kernel alone (what rocprofv3 --kernel-trace reports, profiles/).
            with ops.KernelTimer() as t_all:
                for _ in range(args.steps):
                    step()
                torch.cuda.synchronize()
            kern_all = t_all.summary()
            if args.gemm == 'x3':
                # the default path issues its launches from native code (hfl_block_forward_x3, and ONE window-attention
                # launch for the two coarse pyramid levels of an iteration): the library records a HIP event pair around
                # every fp16 window-attention launch, on the launch stream -- the launches of the timed region itself,
                # first with the step's streams, then the same schedule on one stream ('serial': events bracket the
                # kernel alone, what rocprofv3 --kernel-trace reports)
                kern_all[ATTN], _ = native_attention_timing(step, args.steps)
                if not args.no_streams:
                    set_pyramid_streams('serial')
                    step()
                    rec, iso_sizes = native_attention_timing(step, args.steps)
                    kern_iso = {ATTN: rec}
                    fused_roof = native_fused_timing(step, args.steps)
                    ws_roof = native_ws_timing(step, args.steps)
                    # the fused MLP launches (the step's largest kernel group) alone, same one-stream schedule
                    with ops.KernelTimer(only=['hfl_ln_mlp_fused']) as t_mlp:
                        for _ in range(args.steps):
                            step()
                        torch.cuda.synchronize()
                    mlp_roof = rowtile_roofline(t_mlp.by_size('hfl_ln_mlp_fused'), t_mlp.summary().get('hfl_ln_mlp_fused'))
                    set_pyramid_streams(True)
            elif not args.no_streams:
                set_pyramid_streams(False)
                step()
                torch.cuda.synchronize()
                with ops.KernelTimer(only=[ATTN]) as t_iso:
                    for _ in range(args.steps):
                        step()
                    torch.cuda.synchronize()
                kern_iso = t_iso.summary()
                iso_sizes = t_iso.by_size(ATTN)
                set_pyramid_streams(True)

    # ---- per-rank times (N > 1 diagnostics) and the collective alone
    per_rank = None
    allgather_ms = None
    if use_dist:
        per_rank, elapsed_max = rank_report(dist, elapsed, args.steps, dev)
        if collective:
            y = torch.zeros((args.batch, params.output_dim), device=dev)
            for _ in range(3):
                all_gather_descriptors(y, args.batch * world, force=True)
            dist.barrier(device_ids=[local_rank])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                all_gather_descriptors(y, args.batch * world, force=True)
            torch.cuda.synchronize()
            allgather_ms = round((time.perf_counter() - t0) / args.steps * 1e3, 4)
    if use_dist:
        elapsed = elapsed_max

    if rank == 0:
        total_clouds = args.batch * world * args.steps
        roof = roofline_block(kern_iso or kern_all, kern_all, iso_sizes, args,
                              'hand-written fp16 (hi, lo) MFMA window kernel (v5)' if args.gemm != 'fp32'
                              else 'fp32-MFMA window kernel (v4)', MFMA_F32_PEAK_TFLOPS, bool(kern_iso))
        roof32 = roofline_block({ATTN: kern_iso32[ATTN]}, None, None, args, 'fp32-MFMA window kernel (v4), matched-precision leg',
                                MFMA_F32_PEAK_TFLOPS, True) if (kern_iso32 and ATTN in kern_iso32) else None
        roof_x6 = x6_roofline(x6_sizes, kern_iso32.get('hfl_linear_x6')) if kern_iso32 else None
        others = {}
        for name, (kn, kms, kb, kf, kmv) in (kern_all or {}).items():
            others[name] = {'launches_per_step': kn // args.steps,
                            'ms_per_step': round(kms / args.steps, 4),
                            'GBps': round(kb / (kms * 1e-3) / 1e9, 1) if kms > 0 else None}
        split_txt = ('f32 storage and accumulation; Linear products as 3-term bf16 (hi, lo) split (16 significant bits per '
                     'operand); window-attention QK^T / PV on fp16 (hi, lo) pairs (22 significant bits), softmax in f32; '
                     'fp32_linear leg = f32 GEMMs + f32-MFMA attention (the reference\'s arithmetic)')
        split_txt = split_txt.replace('fp32_linear leg = f32 GEMMs + f32-MFMA attention', 'matched_precision leg = fp32-grade '
                                      'Linear products (3 bf16 planes per operand, 6 plane products) + f32-MFMA attention')
        x6_txt = ('f32 storage and accumulation; Linear products fp32-grade: every operand the exact sum of three bf16 planes, six '
                  'plane products with f32 accumulation (error vs fp64 below the fp32 library GEMM\'s); LayerNorm / softmax / GELU '
                  'in f32; window-attention QK^T / PV on the f32 matrix cores')
        mode_txt = {'fp32': 'f32', 'bf16x3': split_txt, 'x3': split_txt, 'x6': x6_txt}
        plan_txt = ('window plan + tap tables cached on the octree' if args.resident_plan else
                    'window plan + tap tables rebuilt inside every step (reference boundary)')
        line = {
            'metric': 'point-clouds/sec (4096 pts, Wild-Places cfg)' if (args.config == 'wild-places' and not args.points_max
                                                                         and args.points == 4096)
                      else 'point-clouds/sec (%s cfg)' % args.config,
            'value': round(total_clouds / elapsed, 2),
            'unit': 'clouds/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': mode_txt[args.gemm],
            'data': 'synthetic',
            'config': {'workload': '%s cfg, batch=%d clouds/GPU x %s pts, octree depth %d, %s, '
                                   'octree+neighbours resident, %s'
                                   % (args.config, args.batch,
                                      '%d..%d (forest/ball mix)' % (args.points, args.points_max) if args.points_max
                                      else '%d' % args.points, depth,
                                      ('multi-staged training step (stage 1 + TruncatedSmoothAP + stage 3 + grad all-reduce + AdamW)'
                                       if args.multistaged else ('forward+backward, stochastic depth %s' % ('off' if args.no_drop_path else 'on (drop_path = %.2f, as the config trains)' % params.drop_path))) if args.train else 'forward-only',
                                      plan_txt),
                       'global_batch': args.batch * world, 'parallelism': 'dp%d' % world, 'gemm': args.gemm,
                       'value_leg': ('split precision (16-bit-operand products; inside the 1e-3 tolerance, narrower than the '
                                     'reference): the matched-precision figure is matched_precision.value' if args.gemm in ('x3', 'bf16x3')
                                     else 'matched precision (the reference\'s arithmetic)'),
                       'host_affinity': getattr(args, 'host_affinity', 'not set'),
                       'collective': 'rccl all_gather (B_local,256) f32' if collective and world > 1 else
                                     ('rccl all_gather at world size 1' if collective else 'none')},
            # the step's dominant kernel (26-28 % of its kernel time: profiles/r05_summary_table.md) is the fused MLP launch;
            # `roofline` describes it when the leg ran (below), `roofline_window` the fp16 window-attention kernel
            'roofline': roof,
        }
        if roof32:
            line['roofline_fp32'] = roof32
        if fused_roof:
            line['roofline_fused'] = fused_roof
        if ws_roof:
            line['roofline_ws'] = ws_roof
        if mlp_roof:
            line['roofline_window'] = line['roofline']
            line['roofline'] = mlp_roof
        if per_rank is not None:
            line['per_rank_ms_per_step'] = {'min': min(per_rank), 'max': max(per_rank), 'ranks': per_rank}
        if allgather_ms is not None:
            line['allgather_ms'] = allgather_ms
        if resident_line:
            line['boundary_plan' if args.resident_plan else 'resident_plan'] = resident_line
        if fp32_line and fp32_line['gemm'] == 'x6':
            mp = dict(fp32_line)
            mp['what'] = ('the same step at the reference\'s arithmetic: every transformer-block Linear on hfl_linear_x6 (hand-'
                          'written; three bf16 planes per operand, six plane products, f32 accumulation; bias / GELU / residual '
                          'in its epilogue), f32 LayerNorm, f32-MFMA window attention (v4), f32 relay attention; stem '
                          'convolutions and pooling head f32')
            mp['dtype'] = x6_txt
            if roof_x6:
                mp['roofline'] = roof_x6
            if lib_line:
                mp['fp32_library_gemm'] = dict(lib_line, what='context: hipBLASLt fp32 GEMMs (F.linear) in place of hfl_linear_x6, '
                                                               'everything else the same')
            line['matched_precision'] = mp
            # (the driver's record keeps `config` and `roofline` whole: the figure is repeated there)
            line['config']['matched_precision'] = {'value': mp['value'], 'unit': 'clouds/s', 'ms_per_step': mp['ms_per_step'],
                                                   'gemm': 'x6 (hfl_linear_x6, hand-written)',
                                                   'roofline_frac': None if not roof_x6 else roof_x6['frac']}
        elif fp32_line:
            line['split_linear'] = fp32_line
        if e2e_line:
            line['e2e'] = e2e_line
        cpe = cpe_l1_bound() if (args.config == 'wild-places' and not args.train) else None
        if cpe:
            others = dict(others or {})
            others['hfl_cpe_forward.bound'] = cpe
        if others:
            line['kernels'] = others
        if args.train:
            line['peak_memory_GiB'] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
            from hotformerloc_amd import model as _M
            # activation checkpointing: the reference's configs set grad_checkpoint (a second forward of every block to save
            # memory); policy 'auto' keeps the activations while more than half of the device's memory is free -- bitwise the
            # same gradients (tests/test_gpu_model.py::test_grad_checkpoint_recomputes_the_same_gradients)
            # (recomputing: peak memory tells -- 14.4 GiB with the recomputation, 40.7 GiB without on this workload)
            line['checkpointing'] = {'policy': _M._CHECKPOINT_POLICY,
                                     'keeps_activations_when': 'estimated activation bytes of a stage < %.2f x free device memory'
                                                               % _M._CHECKPOINT_FREE_FRACTION,
                                     'device_memory_GiB': round(torch.cuda.get_device_properties(0).total_memory / 2 ** 30, 1)}
        # host side of the timed region: seconds until the last launch of the K steps was queued (no synchronisation
        # inside the region), next to the region's wall time.  issue ~ wall means the host is the bound.
        line['host_issue'] = {'ms_per_step_issue': round(last['value_issue_s'] / args.steps * 1e3, 3),
                              'ms_per_step_wall': line['ms_per_step'],
                              'logical_cpus_allowed': len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else None,
                              'logical_cpus_host': os.cpu_count()}
        if extras and not args.no_train_leg and args.config == 'wild-places':
            log('config-3 training leg (child process) ...')
            line['train_cs'] = train_leg(args)
            log('train leg:', line['train_cs'])
            # the same step with the reference's checkpointing (every block's forward twice, a third of the memory)
            line['train_cs_checkpointed'] = train_leg(args, {'HFL_CHECKPOINT': 'always'})
            log('train leg (checkpointed):', line['train_cs_checkpointed'])
        if extras and not args.no_oxford_leg and args.config == 'wild-places':
            log('config-5 per-rank workload (Oxford cfg, B = 64, depth 9; child process) ...')
            line['oxford'] = child_leg(['--config', 'oxford', '--batch', '64', '--steps', str(args.steps), '--warmup',
                                        str(args.warmup)],
                                       'child process: python bench.py --config oxford --batch 64 (BASELINE config 5: the '
                                       'per-rank workload of batch 512 on 8 GPUs; reference cfg config/config_oxford.txt:21)',
                                       keys=('host_issue',))
            log('oxford leg:', line['oxford'])
        if extras and not args.no_pinned_leg and args.config == 'wild-places' and args.pin_cores != 0:
            # the headline is measured with the process on its rank's slice of the host's CPUs (pin_host); the same step with the
            # affinity left alone, for contrast
            log('unpinned-host leg (child process) ...')
            line['unpinned_host'] = child_leg(['--pin-cores', '0', '--steps', str(args.steps), '--warmup', str(args.warmup),
                                               '--no-extras', '--no-cpu-baseline'],
                                              'child process: the headline step with the CPU affinity left alone (all %d '
                                              'logical CPUs; the headline runs on %s)' % (os.cpu_count() or 0, args.host_affinity),
                                              keys=('host_issue',))
            log('unpinned leg:', line['unpinned_host'])
        if world == 1 and not args.no_cpu_baseline and not args.train:
            line['cpu_baseline'], want = cpu_baseline(params, depth, args)
            line['gpu_over_cpu'] = round(line['value'] / line['cpu_baseline']['value'], 1)
            # descriptors of the LAST timed step of each leg against the oracle's on the same clouds and weights
            par = {'reference': 'oracle/hotformer_ref.py forward (pinned to the reference model on tests/golden/model_*.npz), '
                                'same %d clouds, same closed-form weights' % args.batch,
                   args.gemm: parity_record(last['value'], want)}
            if 'other' in last:
                par[fp32_line['gemm']] = parity_record(last['other'], want)
            line['parity'] = par
        # descriptors of the timed workload against the REFERENCE's own output for this exact batch and these weights (the
        # reference model files run in the build container: oracle/gen_golden.py::WORKLOAD_CASES -> tests/golden/
        # model_wild_places_b32.npz; tests/test_gpu_model.py::test_full_size_workloads_match_reference_golden checks the same)
        gold = os.path.join(ROOT, 'tests', 'golden', 'model_wild_places_b32.npz')
        if (world == 1 and not args.train and args.config == 'wild-places' and args.batch == 32 and args.points == 4096
                and not args.points_max and os.path.exists(gold) and 'value' in last):
            import numpy as np
            ref = np.load(gold)['descriptors']
            pr = {'reference': 'tests/golden/model_wild_places_b32.npz (reference model output, oracle/gen_golden.py)',
                  args.gemm: parity_record(last['value'], ref)}
            if 'other' in last and fp32_line:
                pr[fp32_line['gemm']] = parity_record(last['other'], ref)
            if 'lib' in last:
                pr['fp32'] = parity_record(last['lib'], ref)
            line['parity_reference'] = pr
            if 'matched_precision' in line and 'x6' in pr:
                line['matched_precision']['parity_reference'] = pr['x6']
                line['config']['matched_precision']['parity_reference_max_rel_l2'] = pr['x6'].get('max_rel_l2')
        # the keys a truncated record must keep come first
        head = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'matched_precision', 'parity_reference', 'parity', 'roofline', 'cpu_baseline', 'gpu_over_cpu')
        line = {**{k: line[k] for k in head if k in line}, **{k: v for k, v in line.items() if k not in head}}
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


def bench_clouds(syn, params, args, rank):
    """Rank `rank`'s contiguous slice [rank * B, (rank + 1) * B) of the global synthetic batch."""
    if args.points_max:
        clouds = []
        for i in range(rank * args.batch, (rank + 1) * args.batch):        # forest / unit-ball mix, n ~ U{points..max}
            clouds += syn.make_clouds(3, 1, args.points, params.coordinates, kind='forest' if i % 2 == 0 else 'ball',
                                      n_points_max=args.points_max, first_index=i)
        return clouds
    return syn.make_clouds(2, args.batch, args.points, params.coordinates, first_index=rank * args.batch)


def native_attention_timing(step, steps):
    """(launches, ms, algorithmic bytes, flops, moved bytes) and the by-size groups of the fp16 window-attention launches of
    `steps` product-path steps, from the event pairs the library records around each launch (hfl_internal_attn_timing)."""
    import ctypes
    import torch
    from hotformerloc_amd import _native
    lib = _native.load()
    lib.hfl_internal_attn_timing.argtypes = [ctypes.c_int]
    lib.hfl_internal_attn_timing_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    lib.hfl_internal_attn_timing(1)
    try:
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        cap = 1 << 16
        ms, nb, fl = (ctypes.c_double * cap)(), (ctypes.c_double * cap)(), (ctypes.c_double * cap)()
        n = lib.hfl_internal_attn_timing_read(ms, nb, fl, cap)
        assert 0 < n <= cap, n
    finally:
        lib.hfl_internal_attn_timing(0)
    groups = {}
    for i in range(n):
        c, t = groups.get(int(nb[i]), (0, 0.0))
        groups[int(nb[i])] = (c + 1, t + ms[i])
    tot_b = sum(nb[i] for i in range(n))
    rec = (n, sum(ms[i] for i in range(n)), int(tot_b), int(sum(fl[i] for i in range(n))), int(tot_b))
    return rec, [(b, c, t) for b, (c, t) in sorted(groups.items(), reverse=True)]


def native_fused_timing(step, steps):
    """HIP-event timing of the hfl_attn_fused_fwd launches (LN -> qkv -> window attention in one kernel) of `steps` steps:
    None when the step issues none."""
    import ctypes
    import torch
    from hotformerloc_amd import _native
    lib = _native.load()
    lib.hfl_internal_fused_timing.argtypes = [ctypes.c_int]
    lib.hfl_internal_fused_timing_read.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int]
    lib.hfl_internal_fused_timing(1)
    try:
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        cap = 1 << 14
        ms, nb, fg, fa = ((ctypes.c_double * cap)() for _ in range(4))
        n = lib.hfl_internal_fused_timing_read(ms, nb, fg, fa, cap)
    finally:
        lib.hfl_internal_fused_timing(0)
    if n <= 0:
        return None
    n = min(n, cap)
    t = sum(ms[i] for i in range(n)) * 1e-3
    useful = sum(fg[i] + fa[i] for i in range(n))
    # issued on the matrix cores: 3 bf16 MFMA terms per GEMM product, 3.5 fp16 MFMA per useful attention flop (see roofline.mfma)
    issued = sum(3.0 * fg[i] + 3.5 * fa[i] for i in range(n))
    nbytes = sum(nb[i] for i in range(n))
    return {'kernel': 'hfl_attn_fused_fwd', 'what': 'LayerNorm -> qkv -> window attention in ONE kernel, q / k / v never in HBM '
            '(OctFormer stage: C = 128, 8 heads, K = 48, no relay tokens); outputs bitwise equal to hfl_ln_qkv_fused + the fp16 '
            'window kernel', 'bound': 'mfma', 'unit': 'TFLOP/s', 'peak': MFMA_F16_PEAK_TFLOPS,
            'achieved': round(issued / t / 1e12, 1), 'frac': round(issued / t / 1e12 / MFMA_F16_PEAK_TFLOPS, 4),
            'useful_tflops_fp32_equivalent': round(useful / t / 1e12, 2),
            'useful_over_f32_mfma_peak': round(useful / t / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
            'launches': n, 'avg_launch_us': round(t / n * 1e6, 2),
            'hbm': {'algorithmic_bytes_per_launch': int(nbytes / n), 'GBps': round(nbytes / t / 1e9, 1),
                    'frac_of_8TBps': round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4),
                    'note': 'x in + split2 out = 8 B per (row, channel) + 8 B of metadata per token; the two launches it '
                            'replaces move 24 B per (row, channel)'},
            'mfma_busy_pmc': pmc_mfma_busy('r05_fused_counters.txt') or pmc_mfma_busy('r04_fused_counters.txt'),
            'timing': 'HIP event pair around every launch, recorded by the library, serialised schedule, after the timed region'}


def native_ws_timing(step, steps):
    """HIP-event timing of the hfl_attn_ws_fwd launches (LN -> qkv -> window attention with relay tokens in one kernel: the
    finest pyramid level's blocks) of `steps` steps; None when the step issues none."""
    import ctypes
    import torch
    from hotformerloc_amd import _native
    lib = _native.load()
    lib.hfl_internal_ws_timing.argtypes = [ctypes.c_int]
    lib.hfl_internal_ws_timing_read.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int]
    lib.hfl_internal_ws_timing(1)
    try:
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        cap = 1 << 14
        ms, nb, fg, fa = ((ctypes.c_double * cap)() for _ in range(4))
        n = lib.hfl_internal_ws_timing_read(ms, nb, fg, fa, cap)
    finally:
        lib.hfl_internal_ws_timing(0)
    if n <= 0:
        return None
    n = min(n, cap)
    t = sum(ms[i] for i in range(n)) * 1e-3
    useful = sum(fg[i] + fa[i] for i in range(n))
    issued = sum(3.0 * fg[i] + 3.5 * fa[i] for i in range(n))
    nbytes = sum(nb[i] for i in range(n))
    return {'kernel': 'hfl_attn_ws_fwd', 'what': 'LayerNorm -> qkv -> window attention (48 tokens + the relay token per window) in '
            'ONE kernel with specialised GEMM / attention waves, q / k / v never in HBM (pyramid blocks of the finest level: '
            'C = 256, 16 heads); token rows bitwise equal to hfl_ln_qkv_fused + the fp16 window kernel', 'bound': 'mfma',
            'unit': 'TFLOP/s', 'peak': MFMA_F16_PEAK_TFLOPS, 'achieved': round(issued / t / 1e12, 1),
            'frac': round(issued / t / 1e12 / MFMA_F16_PEAK_TFLOPS, 4),
            'useful_tflops_fp32_equivalent': round(useful / t / 1e12, 2),
            'useful_over_f32_mfma_peak': round(useful / t / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
            'launches': n, 'avg_launch_us': round(t / n * 1e6, 2),
            'hbm': {'algorithmic_bytes_per_launch': int(nbytes / n), 'GBps': round(nbytes / t / 1e9, 1),
                    'frac_of_8TBps': round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4),
                    'note': 'x in + split2 out = 8 B per (row, channel) + metadata + the relay rows\' operands; the two '
                            'launches it replaces move 24 B per (row, channel)'},
            'mfma_busy_pmc': pmc_mfma_busy('r05_ws_counters.txt'),
            'timing': 'HIP event pair around every launch, recorded by the library, serialised schedule, after the timed region'}


def rowtile_roofline(groups, total):
    """MFMA roofline of the fused LN2 -> fc1 -> GELU -> fc2 -> residual launches of the step (hfl_ln_mlp_fused): per launch size
    (ops.KernelTimer.by_size: algorithmic bytes = 8 B per (row, channel)) and over all of them.  FLOP issued = 3 bf16 MFMA
    terms per product of the 16 M C^2 useful ones."""
    if not total or not groups:
        return None
    n, ms, nbytes, flops, _ = total
    issued = 3.0 * flops
    by = []
    for b, c, t in groups:
        # flops of a launch of b algorithmic bytes: rows x C = b / 8; C from the step's two widths is not recoverable from b
        # alone, so the per-size rows report time and rows x channels only
        by.append({'rows_x_channels': int(b // 8), 'launches': c, 'avg_launch_us': round(t / c * 1e3, 2)})
    return {'kernel': 'hfl_ln_mlp_fused', 'what': 'LayerNorm -> fc1 -> GELU -> fc2 -> residual in ONE kernel, the M x 4C hidden '
            'activation never in HBM; every launch of the step (depth-4 blocks, depth-5 blocks, relay-token blocks)',
            'bound': 'mfma', 'unit': 'TFLOP/s', 'peak': MFMA_F16_PEAK_TFLOPS, 'achieved': round(issued / (ms * 1e-3) / 1e12, 1),
            'frac': round(issued / (ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4),
            'useful_tflops_fp32_equivalent': round(flops / (ms * 1e-3) / 1e12, 2), 'launches': n,
            'avg_launch_us': round(ms / n * 1e3, 2), 'by_launch_size': by,
            'algorithmic_bytes_per_launch': int(nbytes / n),
            'algorithmic_bytes': 'SURVEY 8(d): 8 B x (row, channel) = read x + write out in f32; 2 M 8 C^2 useful FLOP',
            'traffic': pmc_traffic('ln_mlp_fused_kernel'), 'traffic_source': 'profiles/r06_pmc_traffic.json (2 x FETCH_SIZE + '
            'WRITE_SIZE per launch, averaged over the launches of the step: separate rocprofv3 --pmc passes of this command)',
            'mfma_busy_pmc': pmc_mfma_busy('r05_mlp_counters.txt') or pmc_mfma_busy('r04_mlp_counters.txt'),
            'timing': 'HIP event pair around every launch (ops.KernelTimer), one-stream schedule, after the timed region'}


def x6_roofline(groups, total):
    """MFMA roofline of the matched-precision leg's dominant kernel, hfl_linear_x6 (csrc/gemm_x6.hip): every transformer-block
    Linear of the step.  Issued FLOP = 6 bf16 MFMA plane products per fp32 product of the 2 M K N useful ones; algorithmic bytes
    = x once + out once (+ residual once), f32."""
    if not total:
        return None
    n, ms, nbytes, flops, _ = total
    issued = 6.0 * flops
    by = [{'algorithmic_bytes': int(b), 'launches': c, 'avg_launch_us': round(t / c * 1e3, 2)} for b, c, t in (groups or [])[:8]]
    return {'kernel': 'hfl_linear_x6', 'what': 'fp32-grade Linear on the bf16 matrix cores: operands split into three bf16 planes '
            '(x on the way into LDS, W once per parameter), six plane products per k-step, f32 accumulation, bias / GELU / '
            'residual in the epilogue; every launch of the matched-precision step',
            'bound': 'mfma', 'unit': 'TFLOP/s', 'peak': MFMA_F16_PEAK_TFLOPS, 'achieved': round(issued / (ms * 1e-3) / 1e12, 1),
            'frac': round(issued / (ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4),
            'useful_tflops_fp32_equivalent': round(flops / (ms * 1e-3) / 1e12, 2),
            'useful_over_f32_mfma_peak': round(flops / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
            'launches': n, 'avg_launch_us': round(ms / n * 1e3, 2),
            'largest_launches': by, 'algorithmic_bytes_per_launch': int(nbytes / n),
            'algorithmic_bytes': 'x (M K 4 B) + out (M N 4 B) [+ residual (M N 4 B)]; 2 M K N useful FLOP',
            'mfma_busy_pmc': pmc_mfma_busy('r06_x6_counters.txt'),
            'traffic': pmc_traffic('gemm_x6_kernel'), 'traffic_source': 'profiles/r06_pmc_traffic.json when present (2 x FETCH_SIZE + '
            'WRITE_SIZE per launch, separate rocprofv3 --pmc passes)',
            'timing': 'HIP event pair around every launch (ops.KernelTimer), one-stream schedule, after the timed region'}


def pmc_traffic(kernel):
    """HBM bytes per launch of a kernel from the committed PMC passes of the default command (tools/pmc_summary.py), or None."""
    for name in ('r06_pmc_traffic.json', 'r05_pmc_traffic.json'):
        path = os.path.join(ROOT, 'profiles', name)
        if os.path.exists(path):
            v = json.load(open(path)).get(kernel)
            if isinstance(v, dict):
                return v.get('hbm_bytes_per_launch')
    return None


def pmc_mfma_busy(name):
    """MFMA-pipe busy fraction of a kernel from its committed SQ counter survey (profiles/<name>, written by
    tools/pmc_survey2.sh in separate --pmc passes): SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x CUs x kernel cycles), the kernel's
    cycles taken as GRBM_GUI_ACTIVE / 8 XCDs of the same survey.  None when the survey is not in the tree."""
    path = os.path.join(ROOT, 'profiles', name)
    if not os.path.exists(path):
        return None
    vals, stamps = {}, []
    for ln in open(path):
        f = ln.split()
        if len(f) == 3 and f[0] == 'kernel_source_sha1':
            stamps.append((f[1], f[2]))
        elif len(f) >= 2 and f[0].isupper():
            try:
                vals[f[0]] = float(f[1])
            except ValueError:
                pass
    busy, gui = vals.get('SQ_VALU_MFMA_BUSY_CYCLES'), vals.get('GRBM_GUI_ACTIVE')
    if not busy or not gui:
        return None
    # a committed survey is evidence for the kernel source it was taken on: it carries the sha1 of that source
    # (tools/stamp_sources.py); a survey whose source has changed since is reported as stale, not as a measurement
    fresh = bool(stamps) and all(source_sha1(rel) == sha for rel, sha in stamps)
    out = {'frac': round(busy / (4 * 256 * gui / 8.0), 4) if fresh else None, 'source': 'profiles/' + name,
           'how': 'SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), rocprofv3 --pmc survey of the kernel alone'}
    if not fresh:
        out['stale'] = ('kernel source changed since the survey' if stamps else 'survey carries no source stamp') + \
                       ' (value at survey time: %.4f)' % (busy / (4 * 256 * gui / 8.0))
    return out


def source_sha1(rel):
    import hashlib
    try:
        return hashlib.sha1(open(os.path.join(ROOT, rel), 'rb').read()).hexdigest()
    except OSError:
        return None


def cpe_l1_bound():
    """What bounds hfl_cpe_forward (the depth-wise octree conv + LayerNorm + residual, libs/dwconv/csrc/dwconv.cu:24-42 +
    models/layers/octformer_layers.py:138-142), from the committed L1 / TA / L2 counter survey of its depth-4 launch
    (tools/cpe_counters.sh -> profiles/r06_cpe_counters.txt): the vector L1 moves one 64-B line per clock and CU, so
    TCP_TOTAL_CACHE_ACCESSES / 256 CUs clocks is the launch's floor; `frac_of_l1_bound` = that floor / the kernel's clocks."""
    path = os.path.join(ROOT, 'profiles', 'r06_cpe_counters.txt')
    if not os.path.exists(path):
        return None
    vals, stamps = {}, []
    for ln in open(path):
        f = ln.split()
        if len(f) == 3 and f[0] == 'kernel_source_sha1':
            stamps.append((f[1], f[2]))
        elif len(f) >= 2 and (f[0].isupper() or f[0].startswith('T')):
            try:
                vals[f[0]] = float(f[1])
            except ValueError:
                pass
    need = ('TCP_TOTAL_CACHE_ACCESSES_sum', 'GRBM_GUI_ACTIVE', 'TCP_TCC_READ_REQ_sum', 'TCC_HIT_sum', 'TCC_MISS_sum',
            'TD_TD_BUSY_sum', 'TCP_GATE_EN1_sum', 'TA_TA_BUSY_sum')
    if any(k not in vals for k in need):
        return None
    clk = vals['GRBM_GUI_ACTIVE'] / 8.0
    fresh = bool(stamps) and all(source_sha1(rel) == sha for rel, sha in stamps)
    return {'bound': 'vector L1 line rate (64 B / clock / CU)', 'source': 'profiles/r06_cpe_counters.txt',
            'fresh': fresh,
            'frac_of_l1_bound': round(vals['TCP_TOTAL_CACHE_ACCESSES_sum'] / 256.0 / clk, 3),
            'l1_hit_rate': round(1.0 - vals['TCP_TCC_READ_REQ_sum'] / vals['TCP_TOTAL_CACHE_ACCESSES_sum'], 3),
            'l2_hit_rate': round(vals['TCC_HIT_sum'] / (vals['TCC_HIT_sum'] + vals['TCC_MISS_sum']), 3),
            'l1_busy': round(vals['TCP_GATE_EN1_sum'] / 256.0 / clk, 3), 'td_busy': round(vals['TD_TD_BUSY_sum'] / 256.0 / clk, 3),
            'ta_busy': round(vals['TA_TA_BUSY_sum'] / 256.0 / clk, 3)}


def roofline_block(kern, kern_overlapped, sizes, args, kernel_txt, mfma_peak_f32, serialised):
    rec = (kern or {}).get(ATTN)
    if not rec:
        return None
    n, ms, nbytes, flops, moved = rec
    gbs = nbytes / (ms * 1e-3) / 1e9
    traffic, traffic_src = None, None
    for cand in ('r06_pmc_traffic.json', 'r05_pmc_traffic.json', 'r04_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_traffic.json', 'r01_pmc_traffic.json'):
        pmc = os.path.join(ROOT, 'profiles', cand)
        if os.path.exists(pmc) and args.config == 'wild-places' and args.batch == 32:
            # HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command
            # (tools/pmc_summary.py: 2*FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md HBM section)
            j = json.load(open(pmc))
            hit = [v for k, v in sorted(j.items(), key=lambda kv: -kv[1].get('launches', 0) if isinstance(kv[1], dict) else 0)
                   if k.startswith('window_attn_kernel') and isinstance(v, dict)]     # the kernel the step runs most
            if hit:
                traffic, traffic_src = hit[0]['hbm_bytes_per_launch'], 'profiles/' + cand
                break
    useful_tf = flops / (ms * 1e-3) / 1e12
    f16 = 'fp16' in kernel_txt
    roof = {'kernel': ATTN, 'kernel_variant': kernel_txt, 'bound': 'hbm',
            'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(gbs / HBM_PEAK_GBS, 4), 'traffic': traffic, 'traffic_source': traffic_src,
            'launches': n, 'avg_launch_us': round(ms * 1e3 / n, 2),
            'algorithmic_bytes_per_launch': int(nbytes / n),
            'algorithmic_bytes': 'SURVEY 8(d): 16 B x (row, channel) = read q,k,v + write out in f32',
            'moved_bytes_per_launch': int(moved / n),
            # useful = 4 L^2 C flop per real window (QK^T + PV at head dim 16).  The fp16 kernel issues 2 K=32 MFMAs per
            # 16x16 score tile (all four hi/lo cross terms of a 16-dim product: 4x the useful flop) and 3 per PV tile
            # (3x): 3.5x overall, on the fp16 engine.  The fp32 kernel issues exactly the useful flop on the fp32 engine.
            'mfma': {'useful_tflops': round(useful_tf, 2),
                     'issued_tflops': round(useful_tf * (3.5 if f16 else 1.0), 2),
                     'mfma_peak_used': MFMA_F16_PEAK_TFLOPS if f16 else mfma_peak_f32,
                     'engine': 'v_mfma_f32_16x16x32_f16' if f16 else 'v_mfma_f32_16x16x4_f32',
                     'frac_issued_of_peak_used': round(useful_tf * (3.5 if f16 else 1.0) /
                                                       (MFMA_F16_PEAK_TFLOPS if f16 else mfma_peak_f32), 4),
                     'useful_over_f32_peak': round(useful_tf / MFMA_F32_PEAK_TFLOPS, 4),
                     'mfma_busy_pmc': (pmc_mfma_busy('r05_attn_counters.txt') or pmc_mfma_busy('r04_attn_counters.txt')) if f16 else None},
            'timing': ('HIP event pair around every launch (recorded by the library on the launch stream for the fp16 kernel, '
                       'by ops.KernelTimer for the fp32 one) over %d steps re-run right after the timed region, never inside '
                       'it, with the step\'s launch schedule on ONE stream; the kernel trace of that schedule is '
                       'profiles/*_serial_kernel_stats.csv (bench.py --serial-streams)' % args.steps)
                      if serialised else 'HIP events per launch, re-run after the timed region with the streams of the step'}
    if sizes:
        # the step's launches by size: depth 5 (octf stage), depth 4, and depths 3 + 2 together in one launch (24 per step
        # here); the coarse levels hold 15 % of the bytes and are latency-bound (a few windows per CU)
        roof['by_launch_size'] = [{'algorithmic_bytes_per_launch': b, 'launches': c,
                                   'avg_launch_us': round(ms_ * 1e3 / c, 2),
                                   'frac': round(b * c / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                                  for b, c, ms_ in sizes]
    rec_o = (kern_overlapped or {}).get(ATTN) if kern_overlapped is not kern else None
    if rec_o:
        nt_, mst, bt, _, _ = rec_o
        roof['overlapped_streams'] = {'frac': round(bt / (mst * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                      'avg_launch_us': round(mst * 1e3 / nt_, 2), 'launches': nt_,
                                      'note': 'same steps with the three pyramid streams of the timed step (events after the '
                                              'timed region): a launch shares the CUs with its neighbours'}
    return roof


if __name__ == '__main__':
    main()
