"""Constants of the reference's config.py that are part of the hot-path contract (values only;
reference config.py:15-37,143). The reference module also builds albumentations pipelines and class
name lists at import time -- data-prep concerns that are out of scope here.
"""
import torch

FIRST_OUT = 48            # config.py:15
CLS_PW = 1.0              # config.py:17
OBJ_PW = 1.0              # config.py:18
LEARNING_RATE = 5e-4      # config.py:20
WEIGHT_DECAY = 5e-4       # config.py:21
DEVICE = "cuda" if torch.cuda.is_available() else "cpu"   # config.py:23
IMAGE_SIZE = 640          # config.py:24
CONF_THRESHOLD = 0.01     # config.py:26
NMS_IOU_THRESH = 0.6      # config.py:27
MAP_IOU_THRESH = 0.5      # config.py:29

ANCHORS = [               # config.py:33-37 (pixels)
    [(10, 13), (16, 30), (33, 23)],       # P3/8
    [(30, 61), (62, 45), (59, 119)],      # P4/16
    [(116, 90), (156, 198), (373, 326)],  # P5/32
]
STRIDES = [8, 16, 32]     # model.py:153
nc = 80                   # config.py:143 (len(COCO))
