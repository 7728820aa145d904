"""GPU time of one BASELINE config-3 training step (CS-Wild-Places cfg, B = 64, variable density, forward + backward) by
operator: torch.profiler over 2 steps, kernels grouped under the aten / autograd-Function op that launched them and by kernel
name.  Answers "what are the torch element-wise launches and the hipBLASLt fp32 GEMMs of the training step".
    python tools/train_ops_profile.py [--batch 64] > profile.txt"""
import argparse
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd import build_batch_octree, load_config, model_factory  # noqa: E402
from hotformerloc_amd import synthetic as syn  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--config', default='cs-wild-places')
    args = ap.parse_args()
    dev = torch.device('cuda')
    params, depth = load_config(args.config)
    model = model_factory(params)
    syn.fill_synthetic_weights(model, 'init')
    model = model.to(dev).train()
    clouds = []
    for i in range(args.batch):
        clouds += syn.make_clouds(3, 1, 4096, params.coordinates, kind='forest' if i % 2 == 0 else 'ball', n_points_max=32768,
                                  first_index=i)
    octree = build_batch_octree(clouds, depth, 2, dev, construct_neigh=True)
    proj = torch.from_numpy(syn.hash_uniform(99, args.batch * params.output_dim).reshape(args.batch, params.output_dim)
                            .astype('float32')).to(dev)

    def step():
        octree.drop_forward_caches()
        model.zero_grad(set_to_none=True)
        y = model({'octree': octree})['global']
        (y * proj).sum().backward()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    nsteps = 2
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for _ in range(nsteps):
            step()
        torch.cuda.synchronize()
    # kernels under their launching CPU op
    by_op = collections.defaultdict(lambda: [0.0, 0])
    by_kernel = collections.defaultdict(lambda: [0.0, 0])
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CUDA:
            continue
        ks = ev.kernels
        if not ks:
            continue
        # only leaf ops: an op whose children also carry kernels would double count
        t = sum(k.duration for k in ks)
        key = ev.name
        if ev.input_shapes:
            key += ' ' + str([tuple(s) for s in ev.input_shapes if s][:3])
        by_op[key][0] += t
        by_op[key][1] += len(ks)
        for k in ks:
            by_kernel[k.name[:90]][0] += k.duration
            by_kernel[k.name[:90]][1] += 1
    tot = sum(v[0] for v in by_kernel.values())
    print('total kernel time %.2f ms per step over %d steps' % (tot / nsteps / 1e3, nsteps))
    print('--- by launching op (us per step, launches per step); aten ops only, hfl: kernels are launched through ctypes and '
          'show under the enclosing autograd Function')
    for k, v in sorted(by_op.items(), key=lambda kv: -kv[1][0])[:70]:
        print('%10.1f %6.1f  %s' % (v[0] / nsteps, v[1] / nsteps, k[:150]))
    print('--- by kernel')
    for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1][0])[:60]:
        print('%10.1f %6.1f  %s' % (v[0] / nsteps, v[1] / nsteps, k))


if __name__ == '__main__':
    main()
