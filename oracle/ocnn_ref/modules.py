"""`ocnn.modules.InputFeature` restated (call site `models/hotformerloc.py:28-31`).
TEST INFRASTRUCTURE, see oracle/__init__.py."""

import torch


class InputFeature(torch.nn.Module):
    def __init__(self, feature: str = 'NDF', nempty: bool = False):
        super().__init__()
        self.feature = feature
        self.nempty = nempty

    def forward(self, octree):
        return octree.get_input_feature(self.feature, self.nempty)
