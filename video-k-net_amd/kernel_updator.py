"""`KernelUpdator` — drop-in for the reference's adaptive kernel update (knet/kernel_updator.py:7-93).

Same registry (`TRANSFORMER_LAYER`), same ctor kwargs, same parameter names (=> same state-dict keys); the arithmetic runs in
libvkn.so (exact-fp32 MFMA GEMMs with fused LayerNorm/sigmoid/ReLU epilogues).
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib, ops
from .registry import register_transformer_layer


@register_transformer_layer
class KernelUpdator(nn.Module):

    def __init__(self, in_channels=256, feat_channels=64, out_channels=None, input_feat_shape=3, gate_sigmoid=True,
                 gate_norm_act=False, activate_out=False, act_cfg=dict(type='ReLU', inplace=True),
                 norm_cfg=dict(type='LN')):
        super().__init__()
        self.in_channels = in_channels
        self.feat_channels = feat_channels
        self.out_channels_raw = out_channels
        self.gate_sigmoid = gate_sigmoid
        self.gate_norm_act = gate_norm_act
        self.activate_out = activate_out
        if isinstance(input_feat_shape, int):
            input_feat_shape = [input_feat_shape] * 2
        self.input_feat_shape = input_feat_shape     # stored but unused, as in the reference (:27-29)
        self.act_cfg = act_cfg
        self.norm_cfg = norm_cfg
        self.out_channels = out_channels if out_channels else in_channels
        if not gate_sigmoid or gate_norm_act or activate_out:
            raise NotImplementedError('only gate_sigmoid=True, gate_norm_act=False, activate_out=False (the reference '
                                      'defaults, used by every shipped config) are built for MI355X')
        if act_cfg.get('type', 'ReLU') != 'ReLU' or norm_cfg.get('type', 'LN') != 'LN':
            raise NotImplementedError('KernelUpdator: only ReLU / LN (every shipped config)')
        if not (in_channels == feat_channels == self.out_channels):
            raise NotImplementedError('KernelUpdator: in_channels == feat_channels == out_channels (every shipped config)')
        self.num_params_in = self.feat_channels
        self.num_params_out = self.feat_channels
        C, Cf = in_channels, feat_channels
        self.dynamic_layer = nn.Linear(C, 2 * Cf)
        self.input_layer = nn.Linear(C, 2 * Cf, 1)
        self.input_gate = nn.Linear(C, Cf, 1)
        self.update_gate = nn.Linear(C, Cf, 1)
        self.norm_in = nn.LayerNorm(Cf)
        self.norm_out = nn.LayerNorm(Cf)
        self.input_norm_in = nn.LayerNorm(Cf)
        self.input_norm_out = nn.LayerNorm(Cf)
        self.activation = nn.ReLU(inplace=act_cfg.get('inplace', False))
        self.fc_layer = nn.Linear(Cf, self.out_channels, 1)
        self.fc_norm = nn.LayerNorm(self.out_channels)

    def forward(self, update_feature, input_feature):
        """update_feature [B,N,C] (or [B*N,C]); input_feature [B,N,K*K=1,C] -> [B*N, 1, C]  (reference :56-93)."""
        C = self.in_channels
        u = update_feature.reshape(-1, C)
        M = u.shape[0]
        k = input_feature.reshape(M, -1, C)
        if k.shape[1] != 1:
            raise NotImplementedError('conv_kernel_size != 1 is not built (no shipped config uses it)')
        if torch.is_grad_enabled() and (update_feature.requires_grad or input_feature.requires_grad
                                        or any(p.requires_grad for p in self.parameters())):
            # differentiable: the library's own kernels in both directions (chain_train.py: LinearFn / UpdatorCoreFn / LayerNormActFn)
            from . import chain_train
            return chain_train.kernel_updator(self, ops._req(u, 'update_feature'), ops._req(k.reshape(M, C), 'input_feature')).reshape(M, 1, C)
        u, k = ops._req(u, 'update_feature'), ops._req(k.reshape(M, C), 'input_feature')
        named = {'kernel_update_conv.' + n: p for n, p in self.named_parameters()}
        pack = ops.StagePack(named, u.device)
        L = _lib.lib()
        out = torch.empty((M, C), dtype=torch.float32, device=u.device)
        # rows are independent: process in slabs of <= 256 rows (the ABI's N limit), one frame each
        for r0 in range(0, M, 256):
            r1 = min(M, r0 + 256)
            dims = ops.make_dims(1, r1 - r0, C, 1, 1, 8, 32, 1, 0, 0, ln_eps=self.fc_norm.eps)
            pack.ensure_prepared(dims)
            ws = ops._workspace(L.vkn_stage_workspace_bytes(ctypes.byref(dims)), u.device)
            with torch.cuda.device(u.device):
                _lib.check(L.vkn_kernel_updator_f32(ctypes.byref(dims), ctypes.byref(pack.w), ops._ptr(u[r0:r1]),
                                                    ops._ptr(k[r0:r1]), ops._ptr(out[r0:r1]), ops._ptr(ws), ws.numel(),
                                                    ops._stream()))
        return out.reshape(M, 1, C)
