"""ORACLE -- TEST INFRASTRUCTURE ONLY.

Value map at the image edge: upscaling/upscaler/data.py:253-256 (convert_array_to_image) and
:266-270 (convert_image_series_to_array)."""
import numpy as np


def convert_uint8_to_array(frames_u8):
    """uint8 [N,H,W,3] -> float64 NHWC in [-1, 1]: v/127.5 - 1 (data.py:266-270)."""
    return (np.asarray(frames_u8) / 127.5) - 1


def convert_array_to_uint8(array):
    """inverse: uint8(around((a+1)*127.5)) (data.py:253-256)."""
    return np.uint8(np.around((np.asarray(array) + 1) * 127.5))
