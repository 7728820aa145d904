"""Stand-in for the subset of `ocnn==2.2.2` that the HOTFormerLoc hot path uses.
TEST INFRASTRUCTURE (see oracle/__init__.py); restated from ocnn's published
behaviour, pinned by ocnn's own fixtures (tests/test_oracle_ocnn.py)."""

from . import octree, nn, modules  # noqa: F401
from .octree import Octree, Points, merge_octrees, key2xyz, xyz2key  # noqa: F401

# `ocnn.octree.*` names used by the reference are served by the submodule itself.
__version__ = '2.2.2-restated'
