"""Call signatures: for every class on the path that the reference defines and this package mirrors, every method the reference's
class defines itself must exist here with the reference's parameter names in the reference's order (extra parameters here must
have defaults) — what "same names, argument meaning" of a drop-in means mechanically.  The reference's classes are imported
UNMODIFIED from /root/reference through the plumbing stand-ins of oracle/standins (as oracle/gen_golden.py does) in a child
process; skipped where the reference tree is absent."""
import inspect
import json
import os
import subprocess
import sys

import pytest

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason='the reference tree is not present')

# reference class -> (module, class here)
PAIRS = {
    'knet.kernel_updator:KernelUpdator': 'KernelUpdator',
    'knet.det.kernel_update_head:KernelUpdateHead': 'KernelUpdateHead',
    'knet.det.kernel_iter_head:KernelIterHead': 'KernelIterHead',
    'knet.video.kernel_update_head:VideoKernelUpdateHead': 'VideoKernelUpdateHead',
    'knet.video.kernel_iter_head:VideoKernelIterHead': 'VideoKernelIterHead',
    'knet.det.kernel_head:ConvKernelHead': 'ConvKernelHead',
    'knet.det.mask_hungarian_assigner:MaskHungarianAssigner': 'MaskHungarianAssigner',
    'knet.det.mask_pseudo_sampler:MaskPseudoSampler': 'MaskPseudoSampler',
    'knet_vis.tracker.kernel_head:ConvKernelHeadVideo': 'ConvKernelHeadVideo',
    'knet_vis.tracker.kernel_update_head:KernelUpdateHeadVideo': 'KernelUpdateHeadVideo',
    'knet_vis.tracker.kernel_iter_head:KernelIterHeadVideo': 'KernelIterHeadVideo',
    'knet_vis.tracker.kernel_frame_iter_head:KernelFrameIterHeadVideo': 'KernelFrameIterHeadVideo',
    'knet_vis.tracker.mask_hungarian_assigner:MaskHungarianAssignerVideo': 'MaskHungarianAssignerVideo',
    'knet.video.qdtrack.trackers.quasi_dense_embed_tracker:QuasiDenseEmbedTracker': 'QuasiDenseEmbedTracker',
    'knet.cross_entropy_loss:CrossEntropyLoss': 'losses.CrossEntropyLoss',
    'knet.video.track_heads:QuasiDenseMaskEmbedHeadGTMask': 'QuasiDenseMaskEmbedHeadGTMask',
}

CHILD = r'''
import sys, os, json, inspect, importlib
sys.dont_write_bytecode = True
root = sys.argv[1]
sys.path.insert(0, os.path.join(root, 'oracle', 'standins'))
sys.path.insert(1, '/root/reference')
out = {}
for key in sorted(json.loads(sys.argv[2]), key=lambda k: ('QuasiDenseEmbedTracker' in k) + 2 * ('QuasiDenseMaskEmbedHeadGTMask' in k)):
    mod, cls = key.split(':')
    try:
        if cls == 'QuasiDenseEmbedTracker':      # its package __init__ needs cv2: loaded by path, as the golden generator does (last: it
            sys.path.insert(0, root)             # replaces the `knet` package entries in sys.modules)
            from oracle.gen_golden_tracker import load_reference_tracker
            c = load_reference_tracker()
        elif cls == 'QuasiDenseMaskEmbedHeadGTMask':   # knet/video/track_heads.py also holds RoI-based heads: loaded by path with name shims
            sys.path.insert(0, root)
            from oracle.gen_golden_tracker import load_reference_embed_head
            c = load_reference_embed_head()
        else:
            c = getattr(importlib.import_module(mod), cls)
    except Exception as e:            # a module the stand-ins cannot carry
        out[key] = {'__error__': type(e).__name__ + ': ' + str(e)[:200]}
        continue
    ms = {}
    for name, fn in vars(c).items():
        if callable(fn) and (not name.startswith('_') or name in ('__init__', '_get_target_single', '_decode_init_proposals',
                                                                 '_mask_forward', '_mask_forward_train')):
            try:
                ps = inspect.signature(fn).parameters.values()
            except (TypeError, ValueError):
                continue
            ms[name] = [[p.name, p.default is not inspect._empty, p.kind.name] for p in ps]
    out[key] = ms
print('@@' + json.dumps(out))
'''

# methods of the reference this package deliberately does not carry (reason beside each)
NOT_BUILT = {
    'VideoKernelIterHead': {'merge_stuff_thing_stuff_first'},      # dead code in the reference: nothing calls it
    # the memo update (births, momentum embeddings, velocities, backdrops, expiry) happens INSIDE the device match kernel; the method exists
    # and raises NotImplementedError with that explanation — listed here so that the surface test does not count it as carried
    'QuasiDenseEmbedTracker': {'update_memo'},
}


def _reference_surface():
    r = subprocess.run([sys.executable, '-c', CHILD, ROOT, json.dumps(sorted(PAIRS))], capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith('@@')]
    assert line, r.stderr[-2000:]
    return json.loads(line[0][2:])


def test_methods_and_parameter_names_follow_the_reference(vkn):
    surf = _reference_surface()
    problems = []
    checked = 0
    for key, here in sorted(PAIRS.items()):
        ref = surf[key]
        if '__error__' in ref:
            problems.append(f'{key}: reference class not importable: {ref["__error__"]}')
            continue
        cls = vkn
        for part in here.split('.'):
            cls = getattr(cls, part, None) if cls is not None else None
        cls = cls or vkn.HEADS.get(here)
        assert cls is not None, here
        for name, rparams in sorted(ref.items()):
            if name in NOT_BUILT.get(here, ()):
                continue
            fn = getattr(cls, name, None)
            if fn is None:
                problems.append(f'{here}.{name}: missing')
                continue
            mine = list(inspect.signature(fn).parameters.values())
            if mine and mine[0].name == 'self' and not (rparams and rparams[0][0] == 'self'):
                mine = mine[1:]
            var_kw = any(p.kind.name == 'VAR_KEYWORD' for p in mine)
            var_pos = any(p.kind.name == 'VAR_POSITIONAL' for p in mine)
            names = [p.name for p in mine if p.kind.name not in ('VAR_KEYWORD', 'VAR_POSITIONAL')]
            rnames = [p[0] for p in rparams if p[2] not in ('VAR_KEYWORD', 'VAR_POSITIONAL')]
            for i, rn in enumerate(rnames):
                if i < len(names) and names[i] == rn:
                    continue
                if rn in names and rparams[i][1]:        # a defaulted parameter of the reference, present here by name (keyword use)
                    continue
                if (var_kw and rparams[i][1]) or (var_pos and not rparams[i][1] and i >= len(names)):
                    continue
                problems.append(f'{here}.{name}: reference parameter {i} `{rn}` vs here {names}')
                break
            extra = [p for p in mine if p.name not in rnames and p.kind.name not in ('VAR_KEYWORD', 'VAR_POSITIONAL')
                     and p.default is inspect._empty]
            if extra:
                problems.append(f'{here}.{name}: extra required parameters {[p.name for p in extra]}')
            checked += 1
    assert not problems, '\n'.join(problems)
    assert checked > 60
