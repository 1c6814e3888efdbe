"""Constants of the reference (utils/const.py)."""
VFEAT_DIM = 4352          # ResNet-2048 || SlowFast-2304
MAX_FRM_SEQ_LEN = 100
VCMR_IOU_THDS = (0.5, 0.7)
