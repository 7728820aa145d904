"""Raw clouds -> what `Points(...)` receives, for a whole batch on the device.

The reference prepares every cloud on the host, one at a time, in the dataloader / evaluation loop
(`eval/pnv_evaluate.py:141-171`, `datasets/dataset_utils.py:84-90`, `datasets/augmentation.py:185-236`):
normalise -> |x| <= 1 mask -> (cylindrical configs) |xy| <= 1 mask -> cylindrical transform.  Here the batch goes to
the GPU raw (12 B per point) and ONE launch (`hfl_prepare_clouds`, a workgroup per cloud) normalises, masks and
compacts it; the result feeds `build_batch_octree` without leaving the device.

Parity: normalisation and both masks are bit-exact with the reference (tests/golden/preprocess.npz, produced by the
reference's own classes).  The cylindrical transform contains `torch.atan2` on the CPU, which the device's `atan2f`
matches only to 1-3 ulp -- enough to move a point that sits on an octree cell boundary (measured: about 1e-5 of the
points change their depth-7 cell, tests/test_gpu_preprocess.py; the reference's own float64 chain is not reproducible
across CPUs to better than 1 ulp either, tests/test_oracle_preprocess.py).  `cylindrical='device'` (default since
round 3) keeps the whole batch on the GPU under that <= 3 ulp contract; `cylindrical='host'` runs the transform on
the host exactly as the reference does (the masked points make one round trip) for callers that need its bits."""

from typing import List, Sequence

import numpy as np
import torch

from . import _native, ops
from ._native import check
from .synthetic import cylindrical as _cylindrical_host


def prepare_clouds(clouds: Sequence, coordinates: str = 'cartesian', normalize: bool = True,
                   scale_factor=None, unit_sphere_norm: bool = False, zero_mean: bool = True,
                   cylindrical: str = 'device', device='cuda') -> List[torch.Tensor]:
    """List of raw (n_i, 3) clouds (numpy / torch, any device) -> list of (m_i, 3) float32 CUDA tensors ready for
    `Points(...)` / `build_batch_octree`.  Arguments mirror `TrainingParams.normalize_points / scale_factor /
    unit_sphere_norm / zero_mean` (`misc/utils.py:210-213`) and `ModelParams.coordinates`."""
    if scale_factor is not None or unit_sphere_norm or not zero_mean:
        # fixed-scale and unit-sphere normalisation reduce with torch.mean / a division whose CPU summation order is
        # not reproducible on the device; no shipped config uses them (config/config_*.txt)
        raise NotImplementedError('only the bounding-box normalisation of the shipped configs runs on the device')
    if coordinates not in ('cartesian', 'cylindrical'):
        raise NotImplementedError('coordinates=%r' % coordinates)
    if cylindrical not in ('host', 'device'):
        raise ValueError("cylindrical must be 'host' or 'device'")
    device = torch.device(device)
    if device.type != 'cuda':
        raise _native.NativeLibraryError('prepare_clouds runs on the GPU (no CPU fallback)')
    if device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    if not clouds:
        return []
    ts = [torch.as_tensor(c, dtype=torch.float32).reshape(-1, 3) for c in clouds]
    sizes = [int(t.shape[0]) for t in ts]
    if min(sizes) < 1:
        raise ValueError('empty point cloud')
    off_host = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int64)
    pts = torch.cat([t.to(device, non_blocking=True) for t in ts]).contiguous()
    off = off_host.to(device, non_blocking=True)
    out = torch.empty_like(pts)
    counts = torch.empty(len(ts), dtype=torch.int32, device=device)
    cyl = coordinates == 'cylindrical'
    ops._dev(pts)
    check(_native.load().hfl_prepare_clouds(out.data_ptr(), counts.data_ptr(), pts.data_ptr(), off.data_ptr(),
                                            len(ts), int(bool(normalize)), int(cyl),
                                            int(cyl and cylindrical == 'device'), ops._stream()),
          'hfl_prepare_clouds')
    kept = counts.cpu().tolist()                               # the one host read: how many points survived
    starts = off_host.tolist()
    res = [out[s:s + k] for s, k in zip(starts, kept)]
    if cyl and cylindrical == 'host':
        res = [torch.from_numpy(_cylindrical_host(r.cpu().numpy())).to(device, non_blocking=True) for r in res]
    if any(k < 1 for k in kept):
        raise ValueError('a cloud has no point left inside the unit cube / cylinder')
    return res
