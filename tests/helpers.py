"""Shared test helpers: golden loading, formula weights/inputs, oracle runs.  Imports oracle/ (allowed in tests/)."""
import os

import numpy as np
import torch

from oracle import synth
from oracle.knet_oracle import HeadCfg, head_param_shapes, iter_head_mask_preds

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASE_FIELDS = ('C', 'heads', 'ffn', 'ncls', 'n_thing', 'n_stuff', 'S', 'up', 'nprop', 'N', 'H', 'W', 'B', 'seed', 'video')


def load_golden(name):
    g = dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))
    case = dict(zip(CASE_FIELDS, (int(v) for v in g['case'])))
    if 'plink' in g:   # the "update" video heads (previous_link / previous_type of the swin configs)
        case['plink'], case['ptype'] = (str(g['plink']) or None), str(g['ptype'])   # ('' = no previous_link, a non-'ffn' tracking link)
    return g, case


def cfg_of(case) -> HeadCfg:
    return HeadCfg(num_stages=case['S'], in_channels=case['C'], num_heads=case['heads'], num_classes=case['ncls'],
                   mask_upsample_stride=case['up'], feat_channels=case['C'],
                   previous_type=case.get('ptype', 'ffn') if case['video'] else '', previous_link=case.get('plink') or '',
                   extra=dict(feedforward_channels=case['ffn']))


def make_case(case, dtype=torch.float32):
    """(cfg, state_dict, x, proposal_feats, mask_preds, previous_obj_feats|None) regenerated from the hash formulas."""
    cfg = cfg_of(case)
    shapes = head_param_shapes(cfg)
    sd = {k: torch.from_numpy(v).to(dtype) for k, v in synth.state_dict_like(shapes, case['seed']).items()}
    x, pf, mp = (torch.from_numpy(a).to(dtype) for a in
                 synth.head_inputs(case['B'], case['N'], case['C'], case['H'], case['W'], case['seed']))
    prev = None
    if case['video']:
        prev = torch.from_numpy(synth.normalish((case['B'], case['N'], case['C'], 1, 1), 99 + case['seed'], 1.0)).to(dtype)
    return cfg, sd, x, pf, mp, prev


def run_oracle(case, dtype=torch.float32, traces=None):
    cfg, sd, x, pf, mp, prev = make_case(case, dtype)
    with torch.no_grad():
        return iter_head_mask_preds(sd, x, pf, mp, cfg, previous_obj_feats=prev, traces=traces)


def maxabs(a, b):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b))) if a.size else 0.0


INIT_FIELDS = ('C', 'nprop', 'ncls', 'n_thing', 'H', 'W', 'B', 'seed', 'sem', 'cat')


def load_init_golden(name):
    g = dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))
    return g, dict(zip(INIT_FIELDS, (int(v) for v in g['case'])))


def make_init_case(case):
    """(loc, sem|None, init_w, seg_w|None, seg_b|None) of a kernel-initialisation golden, from the hash formulas
    (same recipe as oracle/gen_golden.py:init_inputs)."""
    B, C, H, W, seed = case['B'], case['C'], case['H'], case['W'], case['seed']
    loc = torch.from_numpy(synth.normalish((B, C, H, W), 31 + 7 * seed, 1.0))
    sem = torch.from_numpy(synth.normalish((B, C, H, W), 32 + 7 * seed, 1.0)) if case['sem'] else None
    shapes = {'init_kernels.weight': (case['nprop'], C, 1, 1)}
    if case['sem']:
        shapes['conv_seg.weight'] = (case['ncls'], C, 1, 1)
        shapes['conv_seg.bias'] = (case['ncls'],)
    sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_like(shapes, seed).items()}
    return loc, sem, sd['init_kernels.weight'], sd.get('conv_seg.weight'), sd.get('conv_seg.bias')


PAN_FIELDS = ('B', 'N', 'Np', 'T', 'ncls', 'Hm', 'Wm', 'up', 'Hb', 'Wb', 'h', 'w', 'Ho', 'Wo', 'seed')
PAN_CFG = dict(instance_score_thr=0.25, overlap_thr=0.6)   # test_cfg.merge_stuff_thing of the shipped configs


def load_pan_golden(name):
    g = dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))
    return g, dict(zip(PAN_FIELDS, (int(v) for v in g['case'])))


def make_pan_case(case):
    cls, logits = synth.panoptic_inputs(case['B'], case['N'], case['Np'], case['ncls'], case['Hm'], case['Wm'], case['seed'])
    meta = dict(img_shape=(case['h'], case['w'], 3), batch_input_shape=(case['Hb'], case['Wb']), ori_shape=(case['Ho'], case['Wo'], 3))
    return torch.from_numpy(cls), torch.from_numpy(logits), meta


def run_pan_oracle(case, b):
    from oracle.knet_oracle import panoptic_joint
    cls, logits, meta = make_pan_case(case)
    with torch.no_grad():
        return panoptic_joint(cls[b], logits[b], case['Np'], case['T'], case['Np'], PAN_CFG['instance_score_thr'],
                              PAN_CFG['overlap_thr'], meta, upsample_stride=case['up'])


def pan_info_rows(segments_info):
    """segments_info (list of dicts) -> the golden's [id, isthing, category_id, instance_id|-1, score|nan, area|-1] rows."""
    return np.array([[s['id'], int(s['isthing']), s['category_id'], s.get('instance_id', -1), s.get('score', float('nan')),
                      s.get('area', -1)] for s in segments_info], dtype=np.float64).reshape(-1, 6)


ASSIGN_FIELDS = ('N', 'G', 'ncls', 'H', 'W', 'seed')


def load_assign_golden(name):
    g = dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))
    return g, dict(zip(ASSIGN_FIELDS, (int(v) for v in g['case'])))


def make_assign_case(case):
    return tuple(torch.from_numpy(a) for a in synth.assign_inputs(case['N'], case['G'], case['ncls'], case['H'], case['W'], case['seed']))


# ---- measured parity margins: the free-running tests record WHAT they measured, not only pass / fail; the file is
# rewritten after every record so that a partial run still leaves evidence.  Copied to profiles/r04_parity_margins.json by hand.
_MARGINS = {}


def record_margins(test_id, values):
    import json
    _MARGINS[test_id] = {k: (float(v) if isinstance(v, (int, float, np.floating, np.integer)) else v) for k, v in values.items()}
    path = os.environ.get('VKN_MARGINS_JSON', os.path.join(ROOT, 'gpurun_out', 'parity_margins.json'))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        old = {}
        if os.path.exists(path):
            with open(path) as f:
                old = json.load(f)
        old.update(_MARGINS)
        with open(path, 'w') as f:
            json.dump(old, f, indent=1, sort_keys=True)
    except OSError:
        pass
    print(f'[margins] {test_id}: ' + ', '.join(f'{k}={v:.3g}' if isinstance(v, float) else f'{k}={v}' for k, v in _MARGINS[test_id].items()))
