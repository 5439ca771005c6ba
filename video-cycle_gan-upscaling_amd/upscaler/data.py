"""Frame-edge conversions of the reference's ``upscaler.data`` that define the tensor layout and value
range of the hot path (upscaling/upscaler/data.py:253-277).  Image file I/O, cropping and the pandas
plumbing of that module are out of scope."""
import numpy as np
import torch

from . import _engine as E
from . import _lib as L


def convert_array_to_image(array):
    """data.py:253-256: uint8(around((a+1)*127.5)); returns a PIL image when PIL is available, else
    the uint8 array."""
    a = np.uint8(np.around((np.asarray(array) + 1) * 127.5))
    try:
        from PIL import Image
        return Image.fromarray(a)
    except ImportError:  # pragma: no cover
        return a


def convert_image_to_array(img):
    """data.py:259-263"""
    return (np.array(img) / 127.5) - 1


def convert_image_series_to_array(image_series):
    """data.py:266-270: list/Series of HxWx3 uint8 images -> float64 NHWC in [-1, 1]."""
    array = np.array([np.array(img) for img in image_series])
    return (array / 127.5) - 1


def frames_u8_to_device(frames_u8):
    """Device-side version of convert_image_series_to_array: uint8 NHWC (numpy or torch) -> fp32 NCHW on the
    GPU in one fused kernel (vcg_frames_u8_to_nchw); same value map v/127.5 - 1."""
    rt = E.Runtime.get()
    t = frames_u8 if isinstance(frames_u8, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(frames_u8))
    if t.dtype != torch.uint8 or t.dim() != 4:
        raise ValueError("expected uint8 [N,H,W,C]")
    t = t.to(rt.device).contiguous()
    n, h, w, c = t.shape
    out = rt.empty(n, c, h, w)
    L.check(rt.lib.vcg_frames_u8_to_nchw(t.data_ptr(), out.data_ptr(), n, h, w, c, rt.stream), "vcg_frames_u8_to_nchw")
    return out


def device_to_frames_u8(x_nchw):
    """Device-side convert_array_to_image: fp32 NCHW on the GPU -> uint8 NHWC torch tensor (still on GPU)."""
    rt = E.Runtime.get()
    n, c, h, w = x_nchw.shape
    out = torch.empty((n, h, w, c), dtype=torch.uint8, device=rt.device)
    L.check(rt.lib.vcg_nchw_to_frames_u8(x_nchw.data_ptr(), out.data_ptr(), n, h, w, c, rt.stream), "vcg_nchw_to_frames_u8")
    return out


def assemble_training_batch(cropped_hd, cropped_gen1, cropped_gen2, cropped_scaled):
    """The batch assembly of the training loop on the device (upscaling/train_gan3.py:341-345): the loop concatenates the
    three down-scaled variants of each crop into one low-res batch of 3B frames and repeats the high-res crops three times
    (``pd.concat([hd, hd, hd])`` / ``pd.concat([gen1, gen2, scaled])``), then maps uint8 -> [-1, 1].

    Arguments: uint8 NHWC stacks (numpy / torch / sequences of HxWx3 images) of B frames each.  Returns
    (image_batch_lr, image_batch_hr): device fp32 NCHW tensors of 3B frames, ready for ``GanTrainer.train_step`` --
    each uint8 frame crosses PCIe once (1 byte per sample instead of the loop's 8-byte float64 arrays), the value map and
    the layout change run in vcg_frames_u8_to_nchw, and the three copies of the high-res frames are written by the device."""
    rt = E.Runtime.get()

    def stack(x):
        if isinstance(x, torch.Tensor):
            return x
        return torch.from_numpy(np.ascontiguousarray(np.stack([np.asarray(f) for f in x]) if not isinstance(x, np.ndarray) else x))
    hd, g1, g2, sc = (stack(v) for v in (cropped_hd, cropped_gen1, cropped_gen2, cropped_scaled))
    if not (g1.shape == g2.shape == sc.shape) or hd.shape[0] != g1.shape[0]:
        raise ValueError("the three low-res variants must share a shape and the batch size of the high-res crops")
    lr = frames_u8_to_device(torch.cat([g1, g2, sc], 0))
    hr1 = frames_u8_to_device(hd)
    b = hr1.shape[0]
    hr = rt.empty(3 * b, *hr1.shape[1:])
    for i in range(3):
        hr[i * b:(i + 1) * b].copy_(hr1)
    return lr, hr
