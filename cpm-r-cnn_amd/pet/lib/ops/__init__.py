from .nms import nms, ml_nms, nms_segments
from .roi_align import roi_align, ROIAlign
