"""`model_factory(ModelParams) -> nn.Module`: the drop-in boundary
(`models/model_factory.py:25-76` of the reference; same field reads, same asserts)."""

from .model import HOTFormer, HOTFormerLoc, PoolingWrapper
from .params import ModelParams

_CHANNELS_PER_FEATURE = {'L': 3, 'P': 3, 'D': 1, 'N': 3}


def get_in_channels(input_features: str) -> int:
    n = 0
    for f in input_features:
        assert f in _CHANNELS_PER_FEATURE, "Invalid input features specified, must be in ['L','P','D','N']"
        n += _CHANNELS_PER_FEATURE[f]
    assert n > 0, "Invalid input features specified, must be in ['L','P','D','N']"
    return n


def _unsupported(params: ModelParams):
    bad = []
    if params.ct_size != 1:
        # not an omission: the reference's own model raises on it (models/hotformerloc_backbone.py:354-357)
        bad.append('ct_size != 1')
    if params.qkv_init[0] not in ('trunc_normal', 'torch_default'):
        bad.append('qkv_init=%s' % params.qkv_init[0])
    if bad:
        raise NotImplementedError('options the reference model does not run either: ' + ', '.join(bad))


# constructor argument <- ModelParams field (the reads of models/model_factory.py:27-70)
_BACKBONE_ARGS = {
    'channels': 'channels', 'num_blocks': 'num_blocks', 'num_heads': 'num_heads',
    'num_pyramid_levels': 'num_pyramid_levels', 'num_octf_levels': 'num_octf_levels',
    'patch_size': 'patch_size', 'dilation': 'dilation', 'drop_path': 'drop_path',
    'stem_down': 'num_input_downsamples', 'ADaPE_mode': 'ADaPE_mode', 'disable_RPE': 'disable_RPE',
    'conv_norm': 'conv_norm', 'qkv_init': 'qkv_init', 'grad_checkpoint': 'grad_checkpoint',
    'disable_rt': 'disable_rt', 'layer_scale': 'layer_scale', 'xcpe': 'xcpe',
    'rt_size': 'ct_size', 'rt_propagation': 'ct_propagation', 'rt_propagation_scale': 'ct_propagation_scale',
    'downsample_input_embeddings': 'downsample_input_embeddings',
}
_POOLING_ARGS = {
    'pool_method': 'pooling', 'in_dim': 'feature_size', 'output_dim': 'output_dim',
    'num_pyramid_levels': 'num_pyramid_levels', 'k_pooled_tokens': 'k_pooled_tokens',
}


def model_factory(model_params: ModelParams):
    p = model_params
    if 'hotformerloc' not in p.model.lower():
        raise NotImplementedError('Model not implemented: {}'.format(p.model))
    _unsupported(p)
    backbone = HOTFormer(in_channels=get_in_channels(p.input_features),
                         **{arg: getattr(p, field) for arg, field in _BACKBONE_ARGS.items()})
    pooling = PoolingWrapper(channels=p.channels[p.num_octf_levels:],
                             **{arg: getattr(p, field) for arg, field in _POOLING_ARGS.items()})
    if p.disable_rt:
        assert pooling.pooled_feats != 'relaytokens', (
            'If relay tokens are disabled, a local feature pooling method must be used!')
    return HOTFormerLoc(backbone=backbone, pooling=pooling, normalize_embeddings=p.normalize_embeddings,
                        input_features=p.input_features)
