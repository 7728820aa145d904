"""hotformerloc_amd: MI355X-native (gfx950) HOTFormerLoc hierarchical octree attention encoder.

Public surface (mirrors the reference's): `model_factory`, `ModelParams`, `Octree`, `Points`,
`merge_octrees`, and the `dwconv` op module.  See DESIGN.md / INTEGRATION.md.
"""

from .params import ModelParams, load_config            # noqa: F401
from .model_factory import model_factory                 # noqa: F401
from .octree import Octree, Points, merge_octrees, build_batch_octree   # noqa: F401

__version__ = '0.1.0'
