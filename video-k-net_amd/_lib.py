"""ctypes binding of libvkn.so (C ABI in include/vkn.h) + the hipcc build recipe.

The library is built IN-TREE (`video-k-net_amd/lib/libvkn.so`) so that it travels with the repo snapshot to the GPU
box; there is no CPU fallback: if the library is missing every op raises `VknLibraryError`.
"""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIBPATH = os.path.join(LIBDIR, 'libvkn.so')
SOURCES = ('vkn_gather.hip', 'vkn_update.hip', 'vkn_decode.hip', 'vkn_fused.hip', 'vkn_init.hip', 'vkn_panoptic.hip', 'vkn_merge.hip', 'vkn_assign.hip', 'vkn_assign_lr.hip', 'vkn_tracker.hip', 'vkn_loss.hip', 'vkn_chain.hip', 'vkn_chain_h2.hip', 'vkn_ksplit.hip', 'vkn_train.hip', 'vkn_api.hip')
MAX_FCS = 4

# every symbol include/vkn.h declares
SYMBOLS = ('vkn_version', 'vkn_strerror', 'vkn_workspace_init', 'vkn_workspace_status', 'vkn_sizeof_dims', 'vkn_sizeof_stage_weights', 'vkn_gather_workspace_bytes', 'vkn_mask_gather_f32', 'vkn_mask_gather_real_f32',
           'vkn_decode_workspace_bytes', 'vkn_mask_decode_f32', 'vkn_mask_decode_scaled_f32', 'vkn_split_planes_f32', 'vkn_mask_decode_planes_f32',
           'vkn_decode_gather_supported', 'vkn_decode_gather_f32', 'vkn_mask_decode_planes_x', 'vkn_decode_gather_x',
           'vkn_track_link_f32', 'vkn_track_link_flags_f32', 'vkn_prepared_bytes', 'vkn_prepare_stage_f32', 'vkn_split_weight_f32', 'vkn_linear_f32', 'vkn_split_weight_t_f32', 'vkn_sizeof_split_item', 'vkn_split_weights_batch_f32', 'vkn_linear_dw_f32', 'vkn_sizeof_dw_item', 'vkn_sizeof_updator_norms', 'vkn_sizeof_updator_norm_grads', 'vkn_linear_dw_batch_f32', 'vkn_layernorm_act_fwd_f32', 'vkn_layernorm_act_bwd_f32', 'vkn_updator_gate_product_f32', 'vkn_updator_gate_product_bwd_f32', 'vkn_updator_mix_fwd_f32', 'vkn_updator_mix_bwd_f32', 'vkn_attention_f32', 'vkn_attention_bwd_f32', 'vkn_upsample_bilinear_f32', 'vkn_upsample_bilinear_f16out', 'vkn_upsample_bilinear_bwd_f32', 'vkn_kernel_updator_f32',
           'vkn_stage_workspace_bytes', 'vkn_stage_forward_f32', 'vkn_stage_chain_f32', 'vkn_head_workspace_bytes', 'vkn_head_forward_f32', 'vkn_focal_loss_blocks', 'vkn_focal_loss_f32',
           'vkn_head_forward_prof_f32', 'vkn_head_forward_link_f32', 'vkn_stage_forward_link_f32', 'vkn_link_block_f32',
           'vkn_query_merge_workspace_bytes', 'vkn_query_merge_f32',
           'vkn_kernel_init_workspace_bytes', 'vkn_kernel_init_f32',
           'vkn_sizeof_panoptic_cfg', 'vkn_panoptic_workspace_bytes', 'vkn_panoptic_joint_f32',
           'vkn_merge_workspace_bytes', 'vkn_panoptic_thing_first_u8',
           'vkn_sizeof_assign_cfg', 'vkn_assign_workspace_bytes', 'vkn_assign_costs_f32', 'vkn_sizeof_assign_problem', 'vkn_assign_costs_batch_f32', 'vkn_assign_lowres_workspace_bytes', 'vkn_assign_costs_lowres_batch_f32', 'vkn_lsap_f32',
           'vkn_sizeof_lsap_problem', 'vkn_lsap_batch_f32',
           'vkn_mask_losses_chunks', 'vkn_mask_losses_blocks', 'vkn_mask_losses_fwd_f32', 'vkn_mask_losses_bwd_f32',
           'vkn_sizeof_tail_image', 'vkn_sizeof_tail_cfg', 'vkn_stage_targets', 'vkn_mask_losses_fwd_bank_f32', 'vkn_stage_losses_final_f32',
           'vkn_mask_losses_bwd_bank_f32', 'vkn_mask_losses_bwd_lowres_f32', 'vkn_mask_losses_lowres_chunks', 'vkn_mask_losses_fwd_lowres_f32', 'vkn_scale_by_f32', 'vkn_sgd_momentum_f32', 'vkn_check_range_i64',
           'vkn_pow2_scale_f32', 'vkn_scale_pad_rows_f32', 'vkn_transpose_pad_f32', 'vkn_threshold_rows_f16', 'vkn_unscale_rows_f32', 'vkn_sum_n_f32',
           'vkn_sizeof_tracker_cfg', 'vkn_qd_tracker_state_bytes', 'vkn_qd_tracker_workspace_bytes', 'vkn_qd_tracker_state_layout',
           'vkn_qd_tracker_reset', 'vkn_qd_tracker_match_f32')


class VknPanopticCfg(ctypes.Structure):
    """Mirror of include/vkn.h: VknPanopticCfg."""
    _fields_ = [('num_proposals', ctypes.c_int), ('num_thing_classes', ctypes.c_int), ('max_per_img', ctypes.c_int),
                ('instance_score_thr', ctypes.c_float), ('overlap_thr', ctypes.c_double), ('up', ctypes.c_int),
                ('Hm', ctypes.c_int), ('Wm', ctypes.c_int), ('Hb', ctypes.c_int), ('Wb', ctypes.c_int),
                ('h', ctypes.c_int), ('w', ctypes.c_int), ('Ho', ctypes.c_int), ('Wo', ctypes.c_int)]


class VknAssignCfg(ctypes.Structure):
    """Mirror of include/vkn.h: VknAssignCfg."""
    _fields_ = [('cls_weight', ctypes.c_float), ('dice_weight', ctypes.c_float), ('mask_weight', ctypes.c_float),
                ('focal_alpha', ctypes.c_float), ('focal_gamma', ctypes.c_float), ('focal_eps', ctypes.c_float),
                ('dice_eps', ctypes.c_float), ('dice_pred_min', ctypes.c_float), ('mask_pred_min', ctypes.c_float)]


class VknAssignProblem(ctypes.Structure):
    """Mirror of include/vkn.h: VknAssignProblem (device pointers as integers)."""
    _fields_ = [('mask_logits', ctypes.c_void_p), ('cls_logits', ctypes.c_void_p), ('gt_masks', ctypes.c_void_p),
                ('gt_labels', ctypes.c_void_p), ('G', ctypes.c_int), ('cost_out', ctypes.c_void_p)]


class VknLsapProblem(ctypes.Structure):
    """Mirror of include/vkn.h: VknLsapProblem (device pointers as integers)."""
    _fields_ = [('cost', ctypes.c_void_p), ('nr', ctypes.c_int), ('nc', ctypes.c_int), ('gt_inds', ctypes.c_void_p),
                ('row_ind', ctypes.c_void_p), ('col_ind', ctypes.c_void_p)]


class VknTailImage(ctypes.Structure):
    """Mirror of include/vkn.h: VknTailImage (device pointers as integers)."""
    _fields_ = [('row_ind', ctypes.c_void_p), ('col_ind', ctypes.c_void_p), ('gt_labels', ctypes.c_void_p), ('sem_cls', ctypes.c_void_p),
                ('k', ctypes.c_int), ('n_sem', ctypes.c_int), ('gt_row0', ctypes.c_int), ('sem_row0', ctypes.c_int),
                ('pos0', ctypes.c_int), ('reserved', ctypes.c_int)]


class VknTailCfg(ctypes.Structure):
    """Mirror of include/vkn.h: VknTailCfg."""
    _fields_ = [(n, ctypes.c_float) for n in ('w_cls', 'w_mask', 'w_dice', 'dice_eps', 'w_rank', 'avg_factor')] + [('with_rank', ctypes.c_int)]


class VknTrackerCfg(ctypes.Structure):
    """Mirror of include/vkn.h: VknTrackerCfg."""
    _fields_ = ([(n, ctypes.c_float) for n in ('init_score_thr', 'obj_score_thr', 'match_score_thr', 'memo_momentum', 'memo_keep',
                                               'nms_conf_thr', 'nms_backdrop_iou_thr', 'nms_class_iou_thr')]
                + [(n, ctypes.c_int) for n in ('memo_tracklet_frames', 'memo_backdrop_frames', 'with_cats', 'match_metric', 'max_dets',
                                               'max_tracklets', 'embed_dim')])


class VknLibraryError(RuntimeError):
    pass


class VknError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f'libvkn error {code}: {msg}')
        self.code = code


_fp = ctypes.c_void_p  # device pointers travel as integers


class VknSplitItem(ctypes.Structure):
    """include/vkn.h: one matrix of vkn_split_weights_batch_f32"""
    _fields_ = [('W', ctypes.c_void_p), ('images', ctypes.c_void_p), ('ldn', ctypes.c_longlong), ('ldk', ctypes.c_longlong),
                ('Nout', ctypes.c_int), ('K', ctypes.c_int), ('kvalid', ctypes.c_int), ('reserved', ctypes.c_int)]

SPLIT_MAX_ITEMS = 64


class VknDwItem(ctypes.Structure):
    """include/vkn.h: one weight gradient of vkn_linear_dw_batch_f32"""
    _fields_ = [('dY', ctypes.c_void_p), ('A', ctypes.c_void_p), ('dW', ctypes.c_void_p), ('db', ctypes.c_void_p),
                ('ldy', ctypes.c_int), ('lda', ctypes.c_int), ('Nout', ctypes.c_int), ('K', ctypes.c_int)]

DW_MAX_ITEMS = 48


class VknUpdatorNorms(ctypes.Structure):
    """include/vkn.h: LayerNorm vectors (and gate biases) of vkn_updator_mix_*"""
    _fields_ = [(n, ctypes.c_void_p) for n in ('norm_in_w', 'norm_in_b', 'norm_out_w', 'norm_out_b', 'input_norm_in_w', 'input_norm_in_b',
                                               'input_norm_out_w', 'input_norm_out_b', 'input_gate_b', 'update_gate_b')]


class VknUpdatorNormGrads(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ('norm_in_w', 'norm_in_b', 'norm_out_w', 'norm_out_b', 'input_norm_in_w', 'input_norm_in_b',
                                               'input_norm_out_w', 'input_norm_out_b')]


class VknDims(ctypes.Structure):
    _fields_ = [('B', ctypes.c_int), ('N', ctypes.c_int), ('C', ctypes.c_int), ('H', ctypes.c_int), ('W', ctypes.c_int),
                ('heads', ctypes.c_int), ('ff', ctypes.c_int), ('ncls', ctypes.c_int), ('n_cls_fcs', ctypes.c_int),
                ('n_mask_fcs', ctypes.c_int), ('thr_logit', ctypes.c_float), ('ln_eps', ctypes.c_float)]


_W_SCALAR_1 = ['ft_w', 'ft_b', 'ft_wT', 'dyn_w', 'dyn_b', 'inp_w', 'inp_b', 'ig_w', 'ig_b', 'ug_w', 'ug_b',
               'norm_in_w', 'norm_in_b', 'norm_out_w', 'norm_out_b', 'inorm_in_w', 'inorm_in_b', 'inorm_out_w',
               'inorm_out_b', 'fc_w', 'fc_b', 'fc_norm_w', 'fc_norm_b', 'attn_in_w', 'attn_in_b', 'attn_out_w',
               'attn_out_b', 'attn_norm_w', 'attn_norm_b', 'ffn1_w', 'ffn1_b', 'ffn2_w', 'ffn2_b', 'ffn_norm_w',
               'ffn_norm_b']
_W_TAIL = ['pa_in_w', 'pa_in_b', 'pa_out_w', 'pa_out_b', 'pa_norm_w', 'pa_norm_b', 'lffn1_w', 'lffn1_b', 'lffn2_w',
           'lffn2_b', 'lffn_norm_w', 'lffn_norm_b']


class VknStageWeights(ctypes.Structure):
    """Field order mirrors `struct VknStageWeights` in include/vkn.h exactly."""
    _fields_ = ([(n, _fp) for n in _W_SCALAR_1]
                + [('cls_fc_w', _fp * MAX_FCS), ('cls_ln_w', _fp * MAX_FCS), ('cls_ln_b', _fp * MAX_FCS),
                   ('fc_cls_w', _fp), ('fc_cls_b', _fp),
                   ('mask_fc_w', _fp * MAX_FCS), ('mask_ln_w', _fp * MAX_FCS), ('mask_ln_b', _fp * MAX_FCS),
                   ('fc_mask_w', _fp), ('fc_mask_b', _fp)]
                + [(n, _fp) for n in _W_TAIL]
                + [('prepared', _fp), ('prepared_bytes', ctypes.c_size_t)])


def hipcc_command(out=LIBPATH, extra=()):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    return [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', *extra,
            *[os.path.join(CSRC, s) for s in SOURCES], '-o', out]


DEBUG_LIBPATH = os.path.join(LIBDIR, 'libvkn_debug.so')


def _stale(path=None):
    path = path or LIBPATH
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(os.path.dirname(HERE), 'include', 'vkn.h')]
    exp = os.path.join(os.path.dirname(HERE), 'tools', 'experiments')     # debug-only kernel variants (#include'd under VKN_DEBUG)
    if path == DEBUG_LIBPATH and os.path.isdir(exp):
        deps += [os.path.join(exp, f) for f in os.listdir(exp)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _compile_objects(objdir, extra=(), force=False, verbose=False):
    """One `hipcc -c` per source, in parallel, re-using objects newer than every header and their own source (no cross-file device
    symbols exist, so plain separate compilation links)."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')] + [os.path.join(os.path.dirname(HERE), 'include', 'vkn.h')]
    exp = os.path.join(os.path.dirname(HERE), 'tools', 'experiments')
    if '-DVKN_DEBUG' in extra and os.path.isdir(exp):
        hdrs += [os.path.join(exp, f) for f in os.listdir(exp)]
    hdrs.append(os.path.join(CSRC, 'vkn_chain.hip'))              # vkn_chain_h2.hip #includes it
    t_h = max(os.path.getmtime(h) for h in hdrs if os.path.exists(h))
    jobs, objs = [], []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(objdir, s[:-4] + '.o')
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(t_h, os.path.getmtime(src)):
            jobs.append([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', *extra, '-c', src, '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise VknLibraryError('hipcc failed:\n' + r.stdout + r.stderr)
    with ThreadPoolExecutor(max_workers=max(1, min(int(os.environ.get('VKN_BUILD_JOBS', '6')), os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    return objs


def _link(objs, out, verbose=False):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', out]
    if verbose:
        print(' '.join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise VknLibraryError('hipcc link failed:\n' + r.stdout + r.stderr)


def build(force=False, verbose=False):
    """Compile csrc/*.hip for gfx950 into lib/libvkn.so (cross-compiles without a GPU): one object per source under lib/obj/,
    compiled in parallel, then one link."""
    os.makedirs(LIBDIR, exist_ok=True)
    if not force and not _stale():
        return LIBPATH
    _link(_compile_objects(os.path.join(LIBDIR, 'obj'), force=force, verbose=verbose), LIBPATH, verbose)
    return LIBPATH


def build_debug(force=False):
    """The same sources with -DVKN_DEBUG -> lib/libvkn_debug.so: the ONLY build that reads VKN_* environment knobs and contains
    the time-attribution kernel variants (tools/ only; never loaded by the package unless `use_debug()` is called first)."""
    os.makedirs(LIBDIR, exist_ok=True)
    if not force and not _stale(DEBUG_LIBPATH):
        return DEBUG_LIBPATH
    _link(_compile_objects(os.path.join(LIBDIR, 'obj_debug'), extra=('-DVKN_DEBUG',), force=force), DEBUG_LIBPATH)
    return DEBUG_LIBPATH


_LIB = None
_USE_DEBUG = False


def use_debug():
    """Measurement tools: load lib/libvkn_debug.so instead of the release library (must be called before the first op)."""
    global _USE_DEBUG
    if _LIB is not None:
        raise VknLibraryError('use_debug() must be called before the library is first used')
    _USE_DEBUG = True


def lib():
    """The loaded library (ctypes.CDLL) with argtypes set.  Raises VknLibraryError when it is not built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = DEBUG_LIBPATH if _USE_DEBUG else LIBPATH
    if not os.path.exists(path):
        raise VknLibraryError(f'{path} is missing — run `python -c "import __graft_entry__ as g; g.build()"` '
                              '(there is deliberately no CPU fallback)')
    L = ctypes.CDLL(path)
    c_int, c_size, c_uint, c_float = ctypes.c_int, ctypes.c_size_t, ctypes.c_uint, ctypes.c_float
    pD, pW = ctypes.POINTER(VknDims), ctypes.POINTER(VknStageWeights)
    L.vkn_version.restype = c_int
    L.vkn_version.argtypes = []
    L.vkn_strerror.restype = ctypes.c_char_p
    L.vkn_strerror.argtypes = [c_int]
    L.vkn_sizeof_dims.restype = c_size
    L.vkn_sizeof_dims.argtypes = []
    L.vkn_sizeof_stage_weights.restype = c_size
    L.vkn_sizeof_stage_weights.argtypes = []
    if L.vkn_sizeof_dims() != ctypes.sizeof(VknDims) or L.vkn_sizeof_stage_weights() != ctypes.sizeof(VknStageWeights):
        raise VknLibraryError('ctypes mirror of include/vkn.h structs is out of date (size mismatch)')
    for fn in (L.vkn_workspace_init, L.vkn_workspace_status):
        fn.restype = c_int
        fn.argtypes = [_fp, c_size, _fp]
    L.vkn_gather_workspace_bytes.restype = c_size
    L.vkn_gather_workspace_bytes.argtypes = [c_int] * 4
    L.vkn_mask_gather_f32.restype = c_int
    L.vkn_mask_gather_f32.argtypes = [_fp, _fp, c_float, _fp, _fp, c_int, c_int, c_int, c_int, _fp, c_size, c_uint, _fp]
    L.vkn_mask_gather_real_f32.restype = c_int
    L.vkn_mask_gather_real_f32.argtypes = [_fp, _fp, _fp, _fp, c_int, c_int, c_int, c_int, _fp, c_size, _fp]
    L.vkn_decode_workspace_bytes.restype = c_size
    L.vkn_decode_workspace_bytes.argtypes = [c_int] * 3
    L.vkn_mask_decode_f32.restype = c_int
    L.vkn_mask_decode_f32.argtypes = [_fp, _fp, _fp, _fp, c_int, c_int, c_int, c_int, _fp, c_size, c_uint, _fp]
    L.vkn_mask_decode_scaled_f32.restype = c_int
    L.vkn_mask_decode_scaled_f32.argtypes = [_fp, _fp, _fp, _fp, _fp, c_int, c_int, c_int, c_int, _fp, c_size, c_uint, _fp]
    L.vkn_split_planes_f32.restype = c_int
    L.vkn_split_planes_f32.argtypes = [_fp, _fp, _fp, c_int, c_int, c_int, _fp]
    L.vkn_mask_decode_planes_f32.restype = c_int
    L.vkn_mask_decode_planes_f32.argtypes = [_fp, _fp, _fp, _fp, _fp, c_int, c_int, c_int, c_int, _fp]
    L.vkn_decode_gather_supported.restype = c_int
    L.vkn_decode_gather_supported.argtypes = [c_int, c_int]
    L.vkn_mask_decode_planes_x.restype = c_int
    L.vkn_mask_decode_planes_x.argtypes = [_fp, c_int, _fp, _fp, _fp, _fp, c_int, c_int, c_int, c_int, _fp]
    L.vkn_decode_gather_x.restype = c_int
    L.vkn_decode_gather_x.argtypes = [_fp, c_int, _fp, _fp, _fp, c_float, _fp, _fp, c_int, c_int, c_int, c_int, _fp, c_size, _fp]
    L.vkn_decode_gather_f32.restype = c_int
    L.vkn_decode_gather_f32.argtypes = [_fp, _fp, _fp, _fp, c_float, _fp, _fp, c_int, c_int, c_int, c_int, _fp, c_size, _fp]
    L.vkn_track_link_f32.restype = c_int
    L.vkn_track_link_f32.argtypes = [pD, pW, _fp, _fp, _fp, _fp, c_size, _fp]
    L.vkn_track_link_flags_f32.restype = c_int
    L.vkn_track_link_flags_f32.argtypes = [pD, pW, _fp, _fp, _fp, _fp, c_size, c_uint, _fp]
    L.vkn_upsample_bilinear_f32.restype = c_int
    L.vkn_upsample_bilinear_f32.argtypes = [_fp, _fp, c_int, c_int, c_int, c_int, _fp]
    L.vkn_upsample_bilinear_f16out.restype = c_int
    L.vkn_upsample_bilinear_f16out.argtypes = [_fp, _fp, c_int, c_int, c_int, c_int, _fp]
    L.vkn_prepared_bytes.restype = c_size
    L.vkn_prepared_bytes.argtypes = [pD, pW]
    L.vkn_prepare_stage_f32.restype = c_int
    L.vkn_prepare_stage_f32.argtypes = [pD, pW, _fp, c_size, _fp]
    L.vkn_split_weight_f32.restype = c_int
    L.vkn_split_weight_f32.argtypes = [_fp, _fp, c_int, c_int, _fp]
    L.vkn_linear_f32.restype = c_int
    L.vkn_linear_f32.argtypes = [_fp, _fp, _fp, _fp, _fp, c_int, c_int, c_int, c_int, c_int, _fp, c_size, _fp]
    L.vkn_split_weight_t_f32.restype = c_int
    L.vkn_split_weight_t_f32.argtypes = [_fp, _fp, c_int, c_int, _fp]
    L.vkn_sizeof_split_item.restype = c_size
    L.vkn_sizeof_split_item.argtypes = []
    L.vkn_split_weights_batch_f32.restype = c_int
    L.vkn_split_weights_batch_f32.argtypes = [ctypes.POINTER(VknSplitItem), c_int, _fp]
    L.vkn_sizeof_dw_item.restype = c_size
    L.vkn_sizeof_dw_item.argtypes = []
    L.vkn_linear_dw_batch_f32.restype = c_int
    L.vkn_linear_dw_batch_f32.argtypes = [ctypes.POINTER(VknDwItem), c_int, c_int, _fp]
    for fn in (L.vkn_sizeof_updator_norms, L.vkn_sizeof_updator_norm_grads):
        fn.restype = c_size
        fn.argtypes = []
    # the host-side item arrays / parameter blocks are handed to the kernels verbatim: a header edit without its mirror is garbage pointers
    for have, want, nm in ((L.vkn_sizeof_split_item(), ctypes.sizeof(VknSplitItem), 'VknSplitItem'), (L.vkn_sizeof_dw_item(), ctypes.sizeof(VknDwItem), 'VknDwItem'),
                           (L.vkn_sizeof_updator_norms(), ctypes.sizeof(VknUpdatorNorms), 'VknUpdatorNorms'),
                           (L.vkn_sizeof_updator_norm_grads(), ctypes.sizeof(VknUpdatorNormGrads), 'VknUpdatorNormGrads')):
        if have != want:
            raise VknLibraryError(f'ctypes mirror of include/vkn.h struct {nm} is out of date ({want} bytes here, {have} in the library)')
    L.vkn_linear_dw_f32.restype = c_int
    L.vkn_linear_dw_f32.argtypes = [_fp, c_int, _fp, c_int, _fp, _fp, c_int, c_int, c_int, c_int, _fp]
    L.vkn_layernorm_act_fwd_f32.restype = c_int
    L.vkn_layernorm_act_fwd_f32.argtypes = [_fp, c_int, _fp, c_int, _fp, _fp, c_float, c_int, _fp, c_int, _fp, c_int, c_int, _fp]
    L.vkn_layernorm_act_bwd_f32.restype = c_int
    L.vkn_layernorm_act_bwd_f32.argtypes = [_fp, c_int, _fp, c_int, _fp, c_int, _fp, _fp, _fp, c_int, _fp, c_int, _fp, _fp, c_int, c_int, _fp]
    L.vkn_updator_gate_product_f32.restype = c_int
    L.vkn_updator_gate_product_f32.argtypes = [_fp, _fp, _fp, c_int, c_int, _fp]
    L.vkn_updator_gate_product_bwd_f32.restype = c_int
    L.vkn_updator_gate_product_bwd_f32.argtypes = [_fp, _fp, _fp, _fp, _fp, c_int, c_int, _fp]
    L.vkn_updator_mix_fwd_f32.restype = c_int
    L.vkn_updator_mix_fwd_f32.argtypes = [_fp, _fp, _fp, ctypes.POINTER(VknUpdatorNorms), c_float, _fp, _fp, c_int, c_int, _fp]
    L.vkn_updator_mix_bwd_f32.restype = c_int
    L.vkn_updator_mix_bwd_f32.argtypes = [_fp, _fp, _fp, _fp, ctypes.POINTER(VknUpdatorNorms), _fp, _fp, _fp, _fp,
                                          ctypes.POINTER(VknUpdatorNormGrads), c_int, c_int, _fp]
    L.vkn_attention_f32.restype = c_int
    L.vkn_attention_f32.argtypes = [_fp, c_int, _fp, _fp, c_int, _fp, c_int, c_int, c_int, c_int, c_int, c_int, _fp]
    L.vkn_attention_bwd_f32.restype = c_int
    L.vkn_attention_bwd_f32.argtypes = [_fp, c_int, _fp, _fp, c_int, _fp, c_int, _fp, c_int, _fp, c_int, _fp, _fp, c_int, c_int, c_int,
                                        c_int, c_int, c_int, _fp]
    L.vkn_kernel_updator_f32.restype = c_int
    L.vkn_kernel_updator_f32.argtypes = [pD, pW, _fp, _fp, _fp, _fp, c_size, _fp]
    L.vkn_stage_workspace_bytes.restype = c_size
    L.vkn_stage_workspace_bytes.argtypes = [pD]
    L.vkn_stage_forward_f32.restype = c_int
    L.vkn_stage_forward_f32.argtypes = [pD, pW] + [_fp] * 9 + [_fp, c_size, c_uint, _fp]
    L.vkn_stage_chain_f32.restype = c_int
    L.vkn_stage_chain_f32.argtypes = [pD, pW] + [_fp] * 6 + [_fp, c_size, c_uint, _fp]
    L.vkn_head_workspace_bytes.restype = c_size
    L.vkn_head_workspace_bytes.argtypes = [pD]
    L.vkn_focal_loss_blocks.restype = c_int
    L.vkn_focal_loss_blocks.argtypes = [c_int, c_int]
    L.vkn_focal_loss_f32.restype = c_int
    L.vkn_focal_loss_f32.argtypes = [_fp, _fp, _fp, c_int, c_int, c_int, ctypes.c_float, ctypes.c_float, _fp, _fp, _fp]
    L.vkn_head_forward_f32.restype = c_int
    L.vkn_head_forward_f32.argtypes = [pD, c_int, pW] + [_fp] * 8 + [c_int, _fp, _fp, c_size, c_uint, _fp]
    L.vkn_head_forward_prof_f32.restype = c_int
    L.vkn_head_forward_prof_f32.argtypes = [pD, c_int, pW] + [_fp] * 8 + [c_int, _fp, _fp, c_size, c_uint, _fp, _fp, _fp]
    L.vkn_head_forward_link_f32.restype = c_int
    L.vkn_head_forward_link_f32.argtypes = [pD, c_int, pW, pW, pW, c_int] + [_fp] * 8 + [c_int, _fp, _fp, c_size, c_uint, _fp]
    L.vkn_stage_forward_link_f32.restype = c_int
    L.vkn_stage_forward_link_f32.argtypes = [pD, pW, pW, pW, c_int] + [_fp] * 9 + [_fp, c_size, c_uint, _fp]
    L.vkn_link_block_f32.restype = c_int
    L.vkn_link_block_f32.argtypes = [pD, pW, _fp, _fp, _fp, _fp, _fp, c_size, _fp]
    L.vkn_query_merge_workspace_bytes.restype = c_size
    L.vkn_query_merge_workspace_bytes.argtypes = [pD, c_int]
    L.vkn_query_merge_f32.restype = c_int
    L.vkn_query_merge_f32.argtypes = [pD, c_int, pW, _fp, _fp, _fp, _fp, _fp, c_size, _fp]
    L.vkn_kernel_init_workspace_bytes.restype = c_size
    L.vkn_kernel_init_workspace_bytes.argtypes = [c_int] * 5
    L.vkn_kernel_init_f32.restype = c_int
    L.vkn_kernel_init_f32.argtypes = [_fp] * 5 + [c_int, c_int, c_int, c_float] + [_fp] * 4 + [c_int] * 5 + [_fp, c_size, c_uint, _fp]
    pP = ctypes.POINTER(VknPanopticCfg)
    L.vkn_sizeof_panoptic_cfg.restype = c_size
    L.vkn_sizeof_panoptic_cfg.argtypes = []
    if L.vkn_sizeof_panoptic_cfg() != ctypes.sizeof(VknPanopticCfg):
        raise VknLibraryError('ctypes mirror of VknPanopticCfg is out of date (size mismatch)')
    L.vkn_panoptic_workspace_bytes.restype = c_size
    L.vkn_panoptic_workspace_bytes.argtypes = [pP, c_int, c_int]
    L.vkn_panoptic_joint_f32.restype = c_int
    L.vkn_panoptic_joint_f32.argtypes = [pP, _fp, _fp, c_int, c_int, c_int, _fp, _fp, _fp, _fp, _fp, c_size, _fp]
    L.vkn_merge_workspace_bytes.restype = c_size
    L.vkn_merge_workspace_bytes.argtypes = [c_int, c_int]
    L.vkn_panoptic_thing_first_u8.restype = c_int
    L.vkn_panoptic_thing_first_u8.argtypes = [_fp, _fp, _fp, _fp, c_int, _fp, _fp, _fp, c_int, c_int, ctypes.c_double, ctypes.c_double,
                                              c_int, _fp, _fp, _fp, _fp, c_size, _fp]
    pA = ctypes.POINTER(VknAssignCfg)
    L.vkn_sizeof_assign_cfg.restype = c_size
    L.vkn_sizeof_assign_cfg.argtypes = []
    if L.vkn_sizeof_assign_cfg() != ctypes.sizeof(VknAssignCfg):
        raise VknLibraryError('ctypes mirror of VknAssignCfg is out of date (size mismatch)')
    L.vkn_assign_workspace_bytes.restype = c_size
    L.vkn_assign_workspace_bytes.argtypes = [c_int] * 3
    L.vkn_assign_costs_f32.restype = c_int
    L.vkn_assign_costs_f32.argtypes = [pA, _fp, _fp, _fp, _fp, c_int, c_int, c_int, c_int, _fp, _fp, c_size, _fp]
    L.vkn_upsample_bilinear_bwd_f32.restype = c_int
    L.vkn_upsample_bilinear_bwd_f32.argtypes = [_fp, _fp, c_int, c_int, c_int, c_int, _fp]
    L.vkn_mask_losses_chunks.restype = c_int
    L.vkn_mask_losses_chunks.argtypes = [c_int]
    L.vkn_mask_losses_blocks.restype = c_int
    L.vkn_mask_losses_blocks.argtypes = [c_int]
    L.vkn_mask_losses_fwd_f32.restype = c_int
    L.vkn_mask_losses_fwd_f32.argtypes = [_fp, _fp, _fp, _fp, c_int, c_int, c_int, c_int, c_int, _fp, _fp, _fp, _fp, _fp]
    L.vkn_mask_losses_bwd_f32.restype = c_int
    L.vkn_mask_losses_bwd_f32.argtypes = [_fp, _fp, _fp, _fp, _fp, _fp, _fp, c_int, c_int, c_int, c_int, _fp, _fp]
    for fn, st in ((L.vkn_sizeof_tail_image, VknTailImage), (L.vkn_sizeof_tail_cfg, VknTailCfg)):
        fn.restype = c_size
        fn.argtypes = []
        if fn() != ctypes.sizeof(st):
            raise VknLibraryError(f'{st.__name__} layout mismatch between include/vkn.h and _lib.py')
    L.vkn_stage_targets.restype = c_int
    L.vkn_stage_targets.argtypes = [ctypes.POINTER(VknTailImage), c_int, c_int, c_int, c_int, c_int, ctypes.c_float] + [_fp] * 8
    L.vkn_mask_losses_fwd_bank_f32.restype = c_int
    L.vkn_mask_losses_fwd_bank_f32.argtypes = [_fp] * 5 + [c_int] * 5 + [_fp] * 5
    L.vkn_stage_losses_final_f32.restype = c_int
    L.vkn_stage_losses_final_f32.argtypes = [ctypes.POINTER(VknTailCfg), _fp, _fp, c_int, _fp, c_int, c_int, _fp, c_int, _fp, _fp, _fp,
                                             c_int, c_int, c_int, _fp, _fp, _fp, _fp]
    L.vkn_mask_losses_bwd_bank_f32.restype = c_int
    L.vkn_mask_losses_bwd_bank_f32.argtypes = [_fp] * 9 + [ctypes.c_float] * 3 + [c_int, _fp, _fp, c_int, c_int, c_int, c_int, _fp, _fp]
    L.vkn_mask_losses_lowres_chunks.restype = c_int
    L.vkn_mask_losses_lowres_chunks.argtypes = [c_int, c_int]
    L.vkn_mask_losses_fwd_lowres_f32.restype = c_int
    L.vkn_mask_losses_fwd_lowres_f32.argtypes = [_fp] * 4 + [c_int] * 7 + [_fp] * 5
    L.vkn_mask_losses_bwd_lowres_f32.restype = c_int
    L.vkn_mask_losses_bwd_lowres_f32.argtypes = [_fp] * 9 + [ctypes.c_float] * 3 + [c_int, _fp, _fp] + [c_int] * 6 + [_fp, _fp]
    L.vkn_scale_by_f32.restype = c_int
    L.vkn_scale_by_f32.argtypes = [_fp, _fp, _fp, ctypes.c_float, _fp, c_size, _fp]
    L.vkn_sgd_momentum_f32.restype = c_int
    L.vkn_sgd_momentum_f32.argtypes = [_fp, _fp, _fp, c_size] + [ctypes.c_float] * 4 + [_fp]
    L.vkn_check_range_i64.restype = c_int
    L.vkn_check_range_i64.argtypes = [_fp, c_size, ctypes.c_longlong, ctypes.c_longlong, c_int, _fp, _fp]
    L.vkn_sum_n_f32.restype = c_int
    L.vkn_sum_n_f32.argtypes = [ctypes.POINTER(ctypes.c_void_p), c_int, c_size, _fp, _fp]
    L.vkn_pow2_scale_f32.restype = c_int
    L.vkn_pow2_scale_f32.argtypes = [_fp, c_size, c_int, _fp, _fp, _fp]
    L.vkn_scale_pad_rows_f32.restype = c_int
    L.vkn_scale_pad_rows_f32.argtypes = [_fp, _fp, c_int, c_int, c_int, c_size, _fp, _fp]
    L.vkn_transpose_pad_f32.restype = c_int
    L.vkn_transpose_pad_f32.argtypes = [_fp, _fp, c_int, c_int, c_int, c_int, _fp, _fp]
    L.vkn_threshold_rows_f16.restype = c_int
    L.vkn_threshold_rows_f16.argtypes = [_fp, ctypes.c_float, c_int, c_int, c_int, c_size, _fp, _fp]
    L.vkn_unscale_rows_f32.restype = c_int
    L.vkn_unscale_rows_f32.argtypes = [_fp, _fp, _fp, c_int, c_int, c_int, c_int, _fp, _fp, _fp]
    L.vkn_sizeof_assign_problem.restype = c_size
    L.vkn_sizeof_assign_problem.argtypes = []
    if L.vkn_sizeof_assign_problem() != ctypes.sizeof(VknAssignProblem):
        raise VknLibraryError('VknAssignProblem layout mismatch between include/vkn.h and _lib.py')
    L.vkn_assign_costs_batch_f32.restype = c_int
    L.vkn_assign_costs_batch_f32.argtypes = [pA, ctypes.POINTER(VknAssignProblem), c_int, c_int, c_int, c_int, _fp, c_size, _fp]
    L.vkn_assign_lowres_workspace_bytes.restype = c_size
    L.vkn_assign_lowres_workspace_bytes.argtypes = [c_int] * 6
    L.vkn_assign_costs_lowres_batch_f32.restype = c_int
    L.vkn_assign_costs_lowres_batch_f32.argtypes = [pA, ctypes.POINTER(VknAssignProblem)] + [c_int] * 6 + [_fp, c_size, _fp]
    L.vkn_sizeof_lsap_problem.restype = c_size
    L.vkn_sizeof_lsap_problem.argtypes = []
    if L.vkn_sizeof_lsap_problem() != ctypes.sizeof(VknLsapProblem):
        raise VknLibraryError('VknLsapProblem layout mismatch between include/vkn.h and _lib.py')
    L.vkn_lsap_batch_f32.restype = c_int
    L.vkn_lsap_batch_f32.argtypes = [ctypes.POINTER(VknLsapProblem), c_int, _fp, _fp]
    L.vkn_lsap_f32.restype = c_int
    L.vkn_lsap_f32.argtypes = [_fp, c_int, c_int, _fp, _fp]
    pT = ctypes.POINTER(VknTrackerCfg)
    L.vkn_sizeof_tracker_cfg.restype = c_size
    L.vkn_sizeof_tracker_cfg.argtypes = []
    if L.vkn_sizeof_tracker_cfg() != ctypes.sizeof(VknTrackerCfg):
        raise VknLibraryError('ctypes mirror of VknTrackerCfg is out of date (size mismatch)')
    for fn in (L.vkn_qd_tracker_state_bytes, L.vkn_qd_tracker_workspace_bytes):
        fn.restype = c_size
        fn.argtypes = [pT]
    L.vkn_qd_tracker_state_layout.restype = c_int
    L.vkn_qd_tracker_state_layout.argtypes = [pT, ctypes.POINTER(ctypes.c_size_t)]
    L.vkn_qd_tracker_reset.restype = c_int
    L.vkn_qd_tracker_reset.argtypes = [pT, _fp, c_size, _fp]
    L.vkn_qd_tracker_match_f32.restype = c_int
    L.vkn_qd_tracker_match_f32.argtypes = [pT, _fp, c_size, _fp, _fp, _fp, c_int, c_int, _fp, _fp, _fp, _fp, _fp, c_size, _fp]
    _LIB = L
    return L


def check(code):
    if code != 0:
        raise VknError(code, lib().vkn_strerror(code).decode())
