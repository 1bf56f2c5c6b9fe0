from .keypoint_utils import apply_nms_maxpool, select_topk_keypoints

__all__ = ["apply_nms_maxpool", "select_topk_keypoints"]
