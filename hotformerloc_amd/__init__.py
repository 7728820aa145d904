"""hotformerloc_amd: MI355X-native (gfx950) HOTFormerLoc hierarchical octree attention encoder.

Public surface (mirrors the reference's): `model_factory`, `ModelParams`, `Octree`, `Points`,
`merge_octrees`, and the `dwconv` op module.  See DESIGN.md / INTEGRATION.md.
"""

import os as _os

# hipBLASLt's default kernels for this model's GEMM shapes (tens of thousands of rows, N and K of a few
# hundred) are stream-K: workgroups wait on each other's partial tiles.  Two consequences measured on
# MI355X ("Linear layers"): (1) several such GEMMs running at once on different HIP
# streams can dead-lock once they fill the CUs (Oxford cfg, batch >= 48 with all pyramid depths on side
# streams), (2) the plain data-parallel schedule is ~6 % faster end to end on these skinny shapes.
# The library reads this switch when it is first used, i.e. at the first GEMM of the process.
_os.environ.setdefault('TENSILE_STREAMK_DATA_PARALLEL', '1')

from .params import ModelParams, load_config            # noqa: F401,E402
from .model_factory import model_factory                 # noqa: F401,E402
from .octree import Octree, Points, merge_octrees, build_batch_octree   # noqa: F401,E402

__version__ = '0.1.0'
