#!/usr/bin/env python3
"""bench.py -- clouds/s of the HOTFormerLoc encoder forward on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched as
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, one rank per
GPU over RCCL.  W untimed warm-up steps, then EXACTLY K timed steps bracketed by a barrier +
`torch.cuda.synchronize()`; max over ranks; rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1]): batch of 32 synthetic 4096-point clouds per GPU,
Wild-Places cfg (octree depth 7, cylindrical, K=48), forward only, eval mode, fp32, random-init
style closed-form weights; the batch octree is resident on the device with neighbour tables
built when the timed region starts (the reference's model boundary: `misc/torch_utils.py:47-51`
happens before `model(batch)`).  A step = `model(batch)['global']` on every rank followed, for
N > 1, by the RCCL all-gather of the (B_local,256) descriptors.  Weak scaling: each rank
encodes its own contiguous slice of the global batch.

Extra objects on the JSON line: `roofline` for the dominant hand-written kernel (windowed
attention) from HIP events recorded around every launch inside the timed region, and
`cpu_baseline` = the CPU oracle (a port of the reference forward) timed on this box's host cores
on a bounded sample of the same workload (rank 0, N=1 only).
"""

import argparse
import json
import os
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='wild-places')
    ap.add_argument('--batch', type=int, default=32, help='clouds per GPU')
    ap.add_argument('--points', type=int, default=4096)
    ap.add_argument('--gemm', default='bf16x3', choices=['bf16x3', 'fp32'],
                    help="Linear layers: 'bf16x3' = one bf16 GEMM over (hi|hi|lo)x(hi|lo|hi) operands, fp32 "
                         "accumulate/output (default); 'fp32' = hipBLASLt fp32 GEMMs")
    ap.add_argument('--no-streams', action='store_true', help='pyramid depths on one stream')
    ap.add_argument('--attn-variant', type=int, default=0, help='A/B: window attention kernel variant (0 = default)')
    ap.add_argument('--train', action='store_true', help='time forward+backward (BASELINE config 3) instead of forward')
    ap.add_argument('--multistaged', action='store_true',
                    help='with --train: the full multi-staged step (stage 1 no-grad encode, TruncatedSmoothAP on the '
                         'all-gathered descriptors, stage 3 forward+backward, gradient all-reduce, AdamW step)')
    ap.add_argument('--no-collective', action='store_true', help='A/B: skip the descriptor all-gather (N > 1 diagnostics)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample', type=int, default=16, help='clouds in the CPU baseline sample')
    ap.add_argument('--cpu-threads', type=int, default=16,
                    help='torch threads of the CPU baseline (8-16 is the optimum measured on the 2x64-core host; more threads are slower)')
    return ap.parse_args()


def cpu_baseline(params, depth, args):
    """Oracle forward (port of the reference, torch CPU fp32) on the first `cpu_sample` clouds
    of the same workload; returns the dict for the JSON line."""
    import torch
    from hotformerloc_amd import synthetic as syn
    from oracle import hotformer_ref
    from oracle.testing import oracle_octree, synthetic_state_dict
    cores = min(args.cpu_threads, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    sd = synthetic_state_dict(params, 'init')
    warm = syn.make_clouds(2, 2, args.points, params.coordinates)
    log('cpu baseline: warm-up on %d threads' % cores)
    hotformer_ref.forward(sd, params, oracle_octree(warm, depth))            # warm-up, B=2
    log('cpu baseline: timed sample')
    clouds = syn.make_clouds(2, args.cpu_sample, args.points, params.coordinates)
    per = 8                                                                  # clouds per CPU batch
    octrees = [oracle_octree(clouds[i:i + per], depth) for i in range(0, len(clouds), per)]   # boundary: prebuilt
    t0 = time.perf_counter()
    for octree in octrees:
        hotformer_ref.forward(sd, params, octree)
    dt = time.perf_counter() - t0
    return {'value': round(args.cpu_sample / dt, 4), 'unit': 'clouds/s', 'cores': cores,
            'kind': 'port',
            'sample': 'oracle forward, %d batch(es) of <=%d clouds x %d pts (first %d clouds of the GPU workload), '
                      '%s cfg, %.1f s, torch %d threads (host has %d logical CPUs)'
                      % (len(octrees), per, args.points, args.cpu_sample, args.config, dt,
                         torch.get_num_threads(), os.cpu_count() or 1)}


_T0 = time.perf_counter()


def log(*a):
    print('[bench %.1fs]' % (time.perf_counter() - _T0), *a, file=sys.stderr, flush=True)


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from hotformerloc_amd import build_batch_octree, load_config, model_factory, ops
    from hotformerloc_amd import synthetic as syn
    from hotformerloc_amd.model import set_gemm_mode, set_pyramid_streams
    set_gemm_mode(args.gemm)
    set_pyramid_streams(not args.no_streams)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit('--gpus %d needs torch.distributed.run with --nproc-per-node %d'
                         % (args.gpus, args.gpus))
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
    model = model_factory(params)
    syn.fill_synthetic_weights(model, 'init')
    model = model.to(dev).eval()
    log('model ready')

    # this rank's contiguous slice of the global batch (ordered)
    clouds = syn.make_clouds(2, args.batch, args.points, params.coordinates,
                             first_index=rank * args.batch)
    octree = build_batch_octree(clouds, depth, 2, dev, construct_neigh=True)
    batch = {'octree': octree}
    torch.cuda.synchronize()
    log('octree ready', octree.nnum_nempty.tolist())
    if args.attn_variant:
        from hotformerloc_amd import _native
        _native.load().hfl_set_variant(b'window_attention', args.attn_variant)
    from hotformerloc_amd.distributed import all_gather_descriptors

    if args.train:
        model.train()
        for m in model.modules():
            if hasattr(m, 'drop_prob'):
                m.drop_prob = 0.0
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

    def step():
        if args.train and args.multistaged:
            multistaged_training_step(model, [batch], pos_mask, neg_mask, loss_fn, optim, n_total=n_tot)
            return torch.ones(1, device=dev)
        if args.train:
            model.zero_grad(set_to_none=True)
            y = model(batch)['global']
            (y * proj).sum().backward()
            return y.detach()
        y = model(batch)['global']
        if use_dist and not args.no_collective:
            all_gather_descriptors(y, args.batch * world, force=True)      # (B_total, 256) on every rank
        return y

    with (torch.enable_grad() if args.train else torch.inference_mode()):
        # clock ramp-up, lazy code-object loads of every GEMM shape, allocator growth: a few untimed steps
        # on top of the W the caller asked for (a fresh box has shown 25 % slower first processes)
        for i in range(5):
            step()
        torch.cuda.synchronize()
        for i in range(args.warmup):
            step()
            torch.cuda.synchronize()
            log('warmup step', i)
        if use_dist:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()
        with ops.KernelTimer() as timer:
            t0 = time.perf_counter()
            for _ in range(args.steps):
                y = step()
            if use_dist:
                dist.barrier(device_ids=[local_rank])
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
        kern = timer.summary()
        # Roofline leg: the same K steps once more with the pyramid depths on ONE stream.  In the timed
        # region above three streams share the GPU, so a kernel's start-to-end HIP-event time includes
        # the CUs it lent to its neighbours; serialised, the events bracket the kernel alone (this is
        # also what rocprofv3 --kernel-trace reports: profiles/r01_c_summary.md).  Not part of `value`.
        kern_iso = None
        if not args.no_streams and not args.train:
            from hotformerloc_amd.model import set_pyramid_streams
            set_pyramid_streams(False)
            model(batch)
            torch.cuda.synchronize()
            with ops.KernelTimer() as timer_iso:
                for _ in range(args.steps):
                    model(batch)
                torch.cuda.synchronize()
            kern_iso = timer_iso.summary()
            set_pyramid_streams(True)
    log('timed region done: %.3f s for %d steps' % (elapsed, args.steps))
    assert torch.isfinite(y).all()

    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        total_clouds = args.batch * world * args.steps
        n_c, ms_c, nbytes_c, _ = kern.get('hfl_window_attention_fwd', (0, 0.0, 0, 0))
        n, ms, nbytes, flops = (kern_iso or kern).get('hfl_window_attention_fwd', (0, 0.0, 0, 0))
        roof = None
        if n:
            gbs = nbytes / (ms * 1e-3) / 1e9
            traffic, traffic_src = None, None
            pmc = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')
            if os.path.exists(pmc) and args.config == 'wild-places' and args.batch == 32:
                # HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command
                # (tools/pmc_summary.py: 2*FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md HBM section)
                rec = json.load(open(pmc)).get('window_attn_kernel_v4') or json.load(open(pmc)).get('window_attn_kernel_v2')
                if rec:
                    traffic, traffic_src = rec['hbm_bytes_per_launch'], 'profiles/r01_pmc_traffic.json'
            roof = {'kernel': 'hfl_window_attention_fwd', 'bound': 'hbm',
                    'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(gbs / HBM_PEAK_GBS, 4), 'traffic': traffic, 'traffic_source': traffic_src,
                    'launches': n, 'avg_launch_us': round(ms * 1e3 / n, 2),
                    'algorithmic_bytes_per_launch': int(nbytes / n),
                    'mfma_tflops': round(flops / (ms * 1e-3) / 1e12, 2),
                    'mfma_frac': round(flops / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                    'timing': ('HIP events per launch over %d steps, pyramid streams serialised (re-run right '
                               'after the timed region)' % args.steps) if kern_iso else
                              'HIP events per launch over the timed region',
                    'frac_in_timed_region': round(nbytes_c / (ms_c * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if n_c else None}
        others = {}
        for name, (kn, kms, kb, kf) in kern.items():
            others[name] = {'launches_per_step': kn // args.steps,
                            'ms_per_step': round(kms / args.steps, 4),
                            'GBps': round(kb / (kms * 1e-3) / 1e9, 1) if kms > 0 else None}
        line = {
            'metric': 'point-clouds/sec (4096 pts, Wild-Places cfg)', 'value': round(total_clouds / elapsed, 2),
            'unit': 'clouds/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if args.gemm == 'fp32' else 'f32 (Linear products as 3-term bf16 split, f32 accumulate)',
            'data': 'synthetic',
            'config': {'workload': '%s cfg, batch=%d clouds/GPU x %d pts, octree depth %d, %s, '
                                   'octree+neighbours resident' % (args.config, args.batch, args.points, depth,
                                                                   ('multi-staged training step (stage 1 + TruncatedSmoothAP + stage 3 + grad all-reduce + AdamW)'
                                                                    if args.multistaged else 'forward+backward') if args.train else 'forward-only'),
                       'global_batch': args.batch * world, 'parallelism': 'dp%d' % world, 'gemm': args.gemm,
                       'collective': 'rccl all_gather (B_local,256) f32' if world > 1 else 'none'},
            'roofline': roof, 'kernels': others,
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(params, depth, args)
            line['gpu_over_cpu'] = round(line['value'] / line['cpu_baseline']['value'], 1)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
