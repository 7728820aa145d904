"""`ModelParams`: the reference's model-config object (`misc/utils.py:15-115`).

Parses the reference's INI format (`models/hotformerloc_*_cfg.txt`; the shipped
benchmark configs are restated key for key under `hotformerloc_amd/configs/*.ini`)
into the same attribute names with the same defaults, so `model_factory(ModelParams(path))` reads identically on both
sides of the drop-in boundary.
"""

import configparser
import os
from typing import Optional

from .synthetic import cylindrical

CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'configs')

# short name -> (model cfg file, octree depth of the matching training cfg)
# depths: config/config_wild-places.txt:22, config_cs-wild-places.txt:22,
#         config_oxford.txt:21, config_cs-campus3d.txt:22
KNOWN_CONFIGS = {
    'wild-places':    ('wild_places.ini', 7),
    'cs-wild-places': ('cs_wild_places.ini', 7),
    'oxford':         ('oxford.ini', 9),
    'cs-campus3d':    ('cs_campus3d.ini', 7),
}


class CylindricalCoordinates:
    """`datasets/coordinate_utils.py:68-91` (use_octree=True): callable on a torch
    (n,3) tensor, returns the rescaled cylindrical cloud."""

    def __init__(self, use_octree: bool = True):
        self.use_octree = use_octree

    def __call__(self, pc):
        import torch
        assert self.use_octree
        return torch.from_numpy(cylindrical(pc.detach().cpu().numpy()))


class ModelParams:
    def __init__(self, model_params_path: str):
        if not os.path.exists(model_params_path):
            raise FileNotFoundError(model_params_path)
        config = configparser.ConfigParser()
        config.read(model_params_path)
        params = config['MODEL']

        self.model_params_path = model_params_path
        self.model = params.get('model')
        self.output_dim = params.getint('output_dim', 256)

        self.coordinates = params.get('coordinates', 'polar')
        assert self.coordinates in ['polar', 'cartesian', 'cylindrical'], \
            f'Unsupported coordinates: {self.coordinates}'
        if 'cartesian' in self.coordinates:
            self.quantizer = None
        elif 'cylindrical' in self.coordinates:
            self.quantizer = CylindricalCoordinates(use_octree=True)
        else:
            raise NotImplementedError(f'Unsupported coordinates: {self.coordinates}')

        self.normalize_embeddings = params.getboolean('normalize_embeddings', False)
        self.feature_size = params.getint('feature_size', 256)
        self.pooling = params.get('pooling', 'OctGeM')
        self.num_top_down = params.getint('num_top_down', 1)

        def ints(key, default):
            if key in params:
                return tuple(int(e) for e in params[key].split(','))
            return default
        self.channels = ints('channels', (96, 192, 384, 384))
        self.num_blocks = ints('num_blocks', (2, 2, 6, 2))
        self.num_heads = ints('num_heads', None)
        self.patch_size = params.getint('patch_size', 32)
        self.dilation = params.getint('dilation', 4)
        self.ct_size = params.getint('ct_size', 1)
        self.ct_propagation = params.getboolean('ct_propagation', False)
        self.ct_propagation_scale = params.getfloat('ct_propagation_scale', None)
        self.ADaPE_mode: Optional[str] = params.get('ADaPE_mode', None)
        self.drop_path = params.getfloat('drop_path', 0.5)
        self.input_features = params.get('input_features', 'P')
        self.downsample_input_embeddings = params.getboolean('downsample_input_embeddings', True)
        self.num_input_downsamples = params.getint('num_input_downsamples', 2)
        self.disable_RPE = params.getboolean('disable_RPE', False)
        self.conv_norm = params.get('conv_norm', 'batchnorm')
        assert self.conv_norm in ['batchnorm', 'layernorm', 'powernorm']
        self.layer_scale = params.getfloat('layer_scale', None)
        self.grad_checkpoint = params.getboolean('grad_checkpoint', True)
        if 'qkv_init' in params:
            self.qkv_init = list(params['qkv_init'].split(','))
            if len(self.qkv_init) > 1:
                self.qkv_init[1] = None if self.qkv_init[1] == 'None' else float(self.qkv_init[1])
        else:
            self.qkv_init = ['trunc_normal', 0.02]
        self.xcpe = params.getboolean('xCPE', False)

        if 'hotformerloc' in self.model.lower():
            self.num_pyramid_levels = params.getint('num_pyramid_levels', 3)
            self.num_octf_levels = params.getint('num_octf_levels', 1)
            self.k_pooled_tokens = params.get('k_pooled_tokens', '64')
            self.disable_rt = params.getboolean('disable_rt', False)
            if self.k_pooled_tokens.isdigit():
                self.k_pooled_tokens = int(self.k_pooled_tokens)
            else:
                self.k_pooled_tokens = tuple(int(e) for e in params['k_pooled_tokens'].split(','))
        else:
            if 'ct_layers' in params:
                self.ct_layers = tuple(e == 'True' for e in params['ct_layers'].split(','))
            else:
                self.ct_layers = tuple([False] * len(self.channels))

    def print(self):
        print('Model parameters:')
        for k, v in vars(self).items():
            print('{}: {}'.format(k, v))
        print('')


def load_config(name: str):
    """(ModelParams, octree_depth) for one of the shipped benchmark configs."""
    fname, depth = KNOWN_CONFIGS[name]
    return ModelParams(os.path.join(CONFIG_DIR, fname)), depth
