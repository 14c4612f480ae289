"""Constants the path reads (reference constants.py:12-29)."""
import time

import numpy as np
from torch import nn


def nonlinearity():
    return nn.ReLU(inplace=True)


NONLINEARITY = nonlinearity          # constants.py:20
TIME_STR = time.strftime("%Y_%m_%d_%H_%M_%S")   # dg_util.misc_util.get_time_str() stand-in (constants.py:23)
BASE_LOG_DIR = "logs"
IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32) * 255   # constants.py:28
IMAGENET_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32) * 255    # constants.py:29
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
