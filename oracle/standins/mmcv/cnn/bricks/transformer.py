"""Stand-in mmcv.cnn.bricks.transformer (test-only): restated MultiheadAttention / FFN wrappers."""
import torch.nn as nn

from mmcv.registry import Registry, build_from_cfg
from mmcv.cnn import build_activation_layer

TRANSFORMER_LAYER = Registry('transformerLayer')


def build_transformer_layer(cfg, default_args=None):
    return build_from_cfg(cfg, TRANSFORMER_LAYER, default_args)


class MultiheadAttention(nn.Module):
    """identity + proj_drop(nn.MultiheadAttention(q, k, v)[0]); seq-first unless batch_first."""

    def __init__(self, embed_dims, num_heads, attn_drop=0., proj_drop=0.,
                 dropout_layer=None, init_cfg=None, batch_first=False, **kwargs):
        super().__init__()
        if 'dropout' in kwargs:  # deprecated_api_warning: dropout -> attn_drop
            attn_drop = kwargs.pop('dropout')
        self.embed_dims = embed_dims
        self.num_heads = num_heads
        self.batch_first = batch_first
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop, **kwargs)
        self.proj_drop = nn.Dropout(proj_drop)
        self.dropout_layer = nn.Identity()

    def forward(self, query, key=None, value=None, identity=None, query_pos=None,
                key_pos=None, attn_mask=None, key_padding_mask=None, **kwargs):
        if key is None:
            key = query
        if value is None:
            value = key
        if identity is None:
            identity = query
        if key_pos is None and query_pos is not None and query_pos.shape == key.shape:
            key_pos = query_pos
        if query_pos is not None:
            query = query + query_pos
        if key_pos is not None:
            key = key + key_pos
        if self.batch_first:
            query, key, value = (t.transpose(0, 1) for t in (query, key, value))
        out = self.attn(query=query, key=key, value=value, attn_mask=attn_mask,
                        key_padding_mask=key_padding_mask)[0]
        if self.batch_first:
            out = out.transpose(0, 1)
        return identity + self.dropout_layer(self.proj_drop(out))


class FFN(nn.Module):
    """x + layers(x); layers = Seq(Seq(Linear, act, Drop) x (num_fcs-1), Linear, Drop)."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2,
                 act_cfg=dict(type='ReLU', inplace=True), ffn_drop=0., dropout_layer=None,
                 add_identity=True, init_cfg=None, **kwargs):
        super().__init__()
        if 'dropout' in kwargs:  # deprecated_api_warning: dropout -> ffn_drop
            ffn_drop = kwargs.pop('dropout')
        if 'add_residual' in kwargs:
            add_identity = kwargs.pop('add_residual')
        assert num_fcs >= 2
        layers = []
        in_channels = embed_dims
        for _ in range(num_fcs - 1):
            layers.append(nn.Sequential(nn.Linear(in_channels, feedforward_channels),
                                        build_activation_layer(act_cfg), nn.Dropout(ffn_drop)))
            in_channels = feedforward_channels
        layers.append(nn.Linear(feedforward_channels, embed_dims))
        layers.append(nn.Dropout(ffn_drop))
        self.layers = nn.Sequential(*layers)
        self.dropout_layer = nn.Identity()
        self.add_identity = add_identity

    def forward(self, x, identity=None):
        out = self.layers(x)
        if not self.add_identity:
            return self.dropout_layer(out)
        if identity is None:
            identity = x
        return identity + self.dropout_layer(out)
