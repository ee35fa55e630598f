"""The reference's `roi_head` config dicts, parameterised — what a user's config file passes to `build_head`
(configs/det/_base_/models/knet_kitti_step_s3_r50_fpn.py:79-138 (det), configs/det/video_knet_kitti_step/
video_knet_s3_r50_*_link_ffn_joint_train.py:78-137 (video), configs/det/_base_/models/knet_vipseg_s3_r50_fpn.py (VIP-Seg),
configs/video_knet_vis/_base_/models/knet_track_r50.py (YouTube-VIS)).  Used by bench.py, tools/ and the tests so that none of
them depends on another's tree."""
import copy


def mask_head_cfg(video=False, C=256, heads=8, ffn=2048, ncls=19, n_thing=2, n_stuff=17, up=2, head_type=None, **over):
    mh = dict(type=head_type or ('VideoKernelUpdateHead' if video else 'KernelUpdateHead'), num_classes=ncls,
              num_thing_classes=n_thing, num_stuff_classes=n_stuff, num_ffn_fcs=2, num_heads=heads, num_cls_fcs=1,
              num_mask_fcs=1, feedforward_channels=ffn, in_channels=C, out_channels=C, dropout=0.0, mask_thr=0.5,
              conv_kernel_size=1, mask_upsample_stride=up, ffn_act_cfg=dict(type='ReLU', inplace=True), with_ffn=True,
              feat_transform_cfg=dict(conv_cfg=dict(type='Conv2d'), act_cfg=None),
              kernel_updator_cfg=dict(type='KernelUpdator', in_channels=C, feat_channels=C, out_channels=C,
                                      input_feat_shape=3, act_cfg=dict(type='ReLU', inplace=True), norm_cfg=dict(type='LN')),
              loss_rank=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=0.1),
              loss_mask=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
              loss_dice=dict(type='DiceLoss', loss_weight=4.0),
              loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=2.0))
    if video:
        mh.update(previous='placeholder', previous_type='ffn')
    mh.update(over)
    return mh


def roi_head_cfg(video=False, C=256, heads=8, ffn=2048, ncls=19, n_thing=2, n_stuff=17, S=3, up=2, nprop=100, train_cfg=None,
                 mask_over=None, **over):
    """`roi_head=dict(type='KernelIterHead' | 'VideoKernelIterHead', mask_head=[...] * S)`; `train_cfg=None` = test time.
    `mask_over`: overrides of every stage's mask_head dict — e.g. the "update" video configs
    (configs/det/video_knet_kitti_step/video_knet_s3_swinb_*_joint_update.py:98-100):
    `mask_over=dict(previous_link='update_dynamic_cov', previous_type='update')`."""
    mh = mask_head_cfg(video, C, heads, ffn, ncls, n_thing, n_stuff, up, **(mask_over or {}))
    cfg = dict(type='VideoKernelIterHead' if video else 'KernelIterHead', num_thing_classes=n_thing,
               num_stuff_classes=n_stuff, num_stages=S, stage_loss_weights=[1] * S, proposal_feature_channel=C,
               num_proposals=nprop, mask_head=[copy.deepcopy(mh) for _ in range(S)])
    cfg.update(dict(with_track=True, merge_joint=True) if video else dict(do_panoptic=True))
    if train_cfg is not None:
        cfg['train_cfg'] = train_cfg
    cfg.update(over)
    return cfg


def rcnn_train_cfg(S=3, mask_size=1):
    """`train_cfg.rcnn` of the shipped configs (knet_kitti_step_s3_r50_fpn.py:150-165): one entry per stage."""
    return [dict(assigner=dict(type='MaskHungarianAssigner',
                               cls_cost=dict(type='FocalLossCost', weight=2.0),
                               dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True),
                               mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True)),
                 sampler=dict(type='MaskPseudoSampler'), pos_weight=1) for _ in range(S)]
