"""CPU restatement of the raw-cloud pre-steps in front of the octree build.  TEST INFRASTRUCTURE (oracle/__init__.py).

Follows `eval/pnv_evaluate.py:141-171` (evaluation) and `datasets/dataset_utils.py:84-90` (training collate):

    data = Normalize(scale_factor, unit_sphere_norm)(data)        datasets/augmentation.py:185-236
    data = data[all(|data| <= 1, dim=1)]                          pnv_evaluate.py:163-164
    cylindrical cfg only:
        data = data[norm(data[:, :2], dim=1) <= 1]                pnv_evaluate.py:166-169
        data = CylindricalCoordinates(use_octree=True)(data)      datasets/coordinate_utils.py:68-91,104-116

Pinned against the reference's own classes executed in the build container (`oracle/gen_golden_coords.py` ->
`tests/golden/preprocess.npz`), `tests/test_oracle_preprocess.py`."""

import numpy as np
import torch


def normalize_bbox(coords: torch.Tensor, zero_mean: bool = True, norm_range: float = 1.0) -> torch.Tensor:
    """`Normalize.__call__` with scale_factor=None, unit_sphere_norm=False (augmentation.py:213-223): fp32 torch ops
    in the reference's order."""
    bbmin = coords.min(dim=0).values
    bbmax = coords.max(dim=0).values
    if zero_mean:
        center = (bbmin + bbmax) * 0.5
        coords = coords - center
    box_size = (bbmax - bbmin).max() + 1.0e-6
    return coords * (2.0 * norm_range / box_size)


def cylindrical(pc: torch.Tensor) -> torch.Tensor:
    """`CylindricalCoordinates.__call__` (coordinate_utils.py:30-45,68-91,104-116): atan2 / sqrt in fp32, float64
    `np.interp` of rho [0,1] -> [-1,1] and phi [-pi,pi] -> [-1,1] written back into the fp32 tensor, clamp."""
    assert pc.ndim == 2 and pc.shape[1] == 3 and torch.all(abs(pc) <= 1.0)
    phi = torch.atan2(pc[:, 1], pc[:, 0])
    rho = torch.sqrt(pc[:, 0] ** 2 + pc[:, 1] ** 2)
    out = torch.stack([rho, phi, pc[:, 2]], dim=1)
    out[:, 0] = torch.tensor(np.interp(out[:, 0].numpy(), [0, 1], [-1, 1]))
    out[:, 1] = torch.tensor(np.interp(out[:, 1].numpy(), [-np.pi, np.pi], [-1, 1]))
    return torch.clamp(out, -1.0, 1.0)


def prepare_cloud(raw: torch.Tensor, normalize: bool, coordinates: str, stages: dict = None) -> torch.Tensor:
    """One raw (n,3) fp32 cloud -> the tensor handed to `Points(...)`."""
    data = raw
    if normalize:
        data = normalize_bbox(data)
    if stages is not None:
        stages['normalized'] = data.clone()
    data = data[torch.all(abs(data) <= 1.0, dim=1)]
    if coordinates == 'cylindrical':
        data_norm = torch.linalg.norm(data[:, :2], dim=1)[:, None]
        data = data[torch.all(data_norm <= 1.0, dim=1)]
        if stages is not None:
            stages['masked'] = data.clone()
        data = cylindrical(data)
    elif stages is not None:
        stages['masked'] = data.clone()
    return data
