"""Deterministic synthetic inputs and weights.

There is no dataset and no checkpoint (`weights/.gitkeep`, `README.md:206-215` of
the reference), so benchmarks, parity tests and golden fixtures all use

  * clouds drawn from a counter-based integer hash owned by this repo (no
    `torch.Generator`, no libm in the sampling path -> identical bits on every box),
  * parameters filled from a closed-form generator keyed by the state_dict name,
    so 141 MB of weights never need shipping.

Everything here is host-side numpy; it is input preparation, not the hot path.
"""

from typing import Dict, Iterable, Optional

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over='ignore'):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        x = x ^ (x >> np.uint64(31))
    return x


def _fnv1a(name: str) -> int:
    h = 0xCBF29CE484222325
    for ch in name.encode('utf-8'):
        h = ((h ^ ch) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def hash_uniform(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """n float64 values in (-1, 1), exactly reproducible: value i depends only on
    (seed, offset + i)."""
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    with np.errstate(over='ignore'):
        h = _mix64(idx * np.uint64(0xD1342543DE82EF95) + np.uint64(seed & 0xFFFFFFFFFFFFFFFF))
    u24 = (h >> np.uint64(40)).astype(np.float64)          # top 24 bits
    return (2.0 * u24 + 1.0) / float(1 << 24) - 1.0


# ------------------------------------------------------------------------ clouds
def unit_ball_cloud(seed: int, n: int = 4096, radius: float = 0.999) -> np.ndarray:
    """n points uniform in the ball of `radius` (< 1 keeps every coordinate strictly
    inside (-1, 1), SURVEY Appendix A edge case).  Rejection from the cube with 12
    hash candidates per point (all-reject probability 1.5e-4 -> candidate shrunk
    into the ball)."""
    tries = 12
    c = hash_uniform(seed, n * tries * 3).reshape(n, tries, 3)
    inside = (c * c).sum(-1) < 1.0
    first = np.argmax(inside, axis=1)
    pts = c[np.arange(n), first]
    none = ~inside.any(axis=1)
    pts[none] *= 0.5
    return (pts * radius).astype(np.float32)


def forest_cloud(seed: int, n: int, radius: float = 0.999) -> np.ndarray:
    """Variable-density 'forest' cloud (CS-Wild-Places cfg):
    half the points in a thin ground slab, half on 40 thin vertical trunks."""
    u = hash_uniform(seed, n * 3 + 40 * 2 + 8).astype(np.float64)
    trunks = u[:80].reshape(40, 2) * 0.68
    p = u[88:88 + 3 * n].reshape(n, 3).copy()
    half = n // 2
    # ground slab: |z| small, xy over the inscribed square
    p[:half, 0] *= 0.70
    p[:half, 1] *= 0.70
    p[:half, 2] = p[:half, 2] * 0.02 - 0.30
    # trunks: xy jitter around a trunk centre, z in [-0.3, 0.3]
    t = (np.arange(n - half) % 40)
    p[half:, 0] = trunks[t, 0] + p[half:, 0] * 0.01
    p[half:, 1] = trunks[t, 1] + p[half:, 1] * 0.01
    p[half:, 2] = p[half:, 2] * 0.30
    return (p * radius).astype(np.float32)


def cloud_num_points(seed: int, lo: int, hi: int) -> int:
    """Hash-drawn point count in [lo, hi] (variable-density configs)."""
    u = (hash_uniform(seed ^ 0x5EED, 1)[0] + 1.0) * 0.5
    return int(lo + u * (hi - lo + 1)) if hi > lo else lo


def cylindrical(pc: np.ndarray) -> np.ndarray:
    """Cartesian (x,y,z) -> (rho,phi,z) rescaled to [-1,1] for the octree.

    Follows `datasets/coordinate_utils.py:30-45,68-91,104-116` of the reference:
    atan2/sqrt in float32, then a float64 `np.interp` of rho:[0,1]->[-1,1] and
    phi:[-pi,pi]->[-1,1] written back into the float32 array, then clamp."""
    pc = np.asarray(pc, dtype=np.float32)
    assert np.all(np.abs(pc) <= 1.0)
    t = torch.from_numpy(pc)
    phi = torch.atan2(t[:, 1], t[:, 0])
    rho = torch.sqrt(t[:, 0] ** 2 + t[:, 1] ** 2)
    out = torch.stack([rho, phi, t[:, 2]], dim=1)
    rho_s = torch.tensor(np.interp(out[:, 0].numpy(), [0, 1], [-1, 1]))
    phi_s = torch.tensor(np.interp(out[:, 1].numpy(), [-np.pi, np.pi], [-1, 1]))
    out[:, 0] = rho_s
    out[:, 1] = phi_s
    return torch.clamp(out, -1.0, 1.0).numpy()


def make_clouds(config_id: int, batch: int, n_points: int = 4096,
                coordinates: str = 'cartesian', kind: str = 'ball',
                n_points_max: Optional[int] = None, first_index: int = 0):
    """List of (n,3) float32 clouds, seed = 1000*config_id + cloud_index."""
    out = []
    for i in range(first_index, first_index + batch):
        seed = 1000 * config_id + i
        n = n_points if n_points_max is None else cloud_num_points(seed, n_points, n_points_max)
        pc = unit_ball_cloud(seed, n) if kind == 'ball' else forest_cloud(seed, n)
        if coordinates == 'cylindrical':
            pc = cylindrical(pc)
        out.append(pc)
    return out


# ----------------------------------------------------------------------- weights
_PROFILES = {
    # reference-like magnitudes (`hotformerloc_backbone.py:817-843`,
    # `octformer_layers.py:153-154`, `salsa.py:21`): trunc-normal 0.02 everywhere
    'init':   dict(linear=0.02, qkv=0.02, rpe=0.02, query=1.0, ln_w=0.10, bias=0.02),
    # peaky attention + visible RPE, so parity tests are sensitive to the bias path
    'stress': dict(linear=0.04, qkv=0.08, rpe=0.50, query=1.0, ln_w=0.10, bias=0.05),
}


def synthetic_tensor(name: str, shape: Iterable[int], profile: str = 'stress') -> np.ndarray:
    """Closed-form float32 tensor for the state_dict entry `name`."""
    p = _PROFILES[profile]
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if len(shape) else 1
    u = hash_uniform(_fnv1a(name), n)
    r3 = 3.0 ** 0.5                                       # uniform(-a,a) has std a/sqrt(3)
    if name.endswith('num_batches_tracked'):
        v = np.zeros(shape if shape else (1,))
    elif name.endswith('running_var'):                    # BatchNorm statistics of the alternative heads: positive
        v = 1.0 + u * 0.2
    elif name.endswith('running_mean'):
        v = u * 0.1
    elif name == 'p' or name.endswith('.p'):              # GeM exponent (models/layers/pooling.py:28,72)
        v = 3.0 + u * 0.5
    elif name.endswith('rpe_table'):
        v = u * (p['rpe'] * r3)
    elif name.endswith('.query'):
        v = u * (p['query'] * r3)
    elif name.endswith('.weights'):                       # ocnn conv: (kdim, Cin|1, Cout)
        fan_in = shape[0] * shape[1]
        v = u * (3.0 / fan_in) ** 0.5
    elif len(shape) == 1 and name.endswith('weight'):     # LayerNorm gain
        v = 1.0 + u * p['ln_w']
    elif len(shape) == 1:                                 # any bias
        v = u * p['bias']
    elif '.qkv.' in name:
        v = u * (p['qkv'] * r3)
    else:                                                 # Linear weight (out, in)
        v = u * (p['linear'] * r3)
    return v.astype(np.float32).reshape(shape)


def fill_synthetic_weights(model: torch.nn.Module, profile: str = 'stress') -> Dict[str, tuple]:
    """Overwrite every state_dict tensor of `model` in place; returns name->shape."""
    spec = {}
    with torch.no_grad():
        for name, t in model.state_dict().items():
            w = synthetic_tensor(name, t.shape, profile)
            t.copy_(torch.from_numpy(w).to(t.device, t.dtype))
            spec[name] = tuple(t.shape)
    return spec
