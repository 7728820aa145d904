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
    if not params.downsample_input_embeddings:
        bad.append('downsample_input_embeddings=False')
    if params.ct_size != 1:
        bad.append('ct_size != 1')
    if params.ct_propagation:
        bad.append('ct_propagation=True')
    if params.disable_rt:
        bad.append('disable_rt=True')
    if params.layer_scale is not None:
        bad.append('layer_scale')
    if params.xcpe:
        bad.append('xCPE=True')
    if params.qkv_init[0] not in ('trunc_normal', 'torch_default'):
        bad.append('qkv_init=%s' % params.qkv_init[0])
    if bad:
        raise NotImplementedError('options outside the shipped configs (SURVEY section 8f rank 4): '
                                  + ', '.join(bad))


def model_factory(model_params: ModelParams):
    if 'hotformerloc' not in model_params.model.lower():
        raise NotImplementedError('Model not implemented: {}'.format(model_params.model))
    _unsupported(model_params)
    backbone = HOTFormer(
        in_channels=get_in_channels(model_params.input_features),
        channels=model_params.channels,
        num_blocks=model_params.num_blocks,
        num_heads=model_params.num_heads,
        num_pyramid_levels=model_params.num_pyramid_levels,
        num_octf_levels=model_params.num_octf_levels,
        patch_size=model_params.patch_size,
        dilation=model_params.dilation,
        drop_path=model_params.drop_path,
        stem_down=model_params.num_input_downsamples,
        ADaPE_mode=model_params.ADaPE_mode,
        disable_RPE=model_params.disable_RPE,
        conv_norm=model_params.conv_norm,
        qkv_init=model_params.qkv_init,
    )
    pooling = PoolingWrapper(
        pool_method=model_params.pooling,
        in_dim=model_params.feature_size,
        output_dim=model_params.output_dim,
        num_pyramid_levels=model_params.num_pyramid_levels,
        channels=model_params.channels[model_params.num_octf_levels:],
        k_pooled_tokens=model_params.k_pooled_tokens,
    )
    return HOTFormerLoc(backbone=backbone, pooling=pooling,
                        normalize_embeddings=model_params.normalize_embeddings,
                        input_features=model_params.input_features)
