INSTANCE_OFFSET = 1000  # mmdet.datasets.coco_panoptic (constant only)
