"""ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.

CPU restatement (torch CPU, fp32 or fp64) of the Keras 2.2.x / TF 1.14 layer semantics that the
reference's hot path instantiates (reference imports: upscaling/upscaler/model.py:1-11).  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; the product path (``video-cycle_gan-upscaling_amd/``) never does.

"Parity unpinned": the reference holds no numeric golden vectors for this path and Keras/TF are not
installable here, so these functions are pinned only by
  * the shape / parameter-count known answers of the reference's notebooks (SURVEY.md Appendix C,
    checked in tests/test_oracle_known_answers.py),
  * adjoint / finite-difference identities and cross-checks against torch.nn.functional
    (tests/test_oracle_identities.py).

Layout conventions: activations are NHWC at the API edge and NCHW inside
these functions; kernels keep Keras' own layouts --
  Conv2D kernel          (kh, kw, in, out)   "HWIO"
  Conv2DTranspose kernel (kh, kw, out, in)   "HWOI"
  Dense kernel           (in, out)
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-3        # keras.layers.BatchNormalization default epsilon
BN_MOMENTUM = 0.99   # keras.layers.BatchNormalization default momentum


# ----------------------------------------------------------------------------------------------
# padding arithmetic (TF "SAME"): SURVEY.md Appendix A, pinned by cnn_test.ipynb cell 12
# ----------------------------------------------------------------------------------------------
def same_pads(size, k, s):
    """TF SAME: out = ceil(in/s); total = max((out-1)*s + k - in, 0); before = total//2."""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    before = total // 2
    return out, before, total - before


def conv2d(x, w_hwio, b, stride=1, padding="same"):
    """keras.layers.Conv2D forward. x: [N,C,H,W]; w: (kh,kw,in,out); padding 'same' | 'valid' | int.

    An int padding p is symmetric explicit zero padding (used only by the PatchGAN extension,
    which has no reference counterpart -- SURVEY.md section 8 row a11)."""
    kh, kw = w_hwio.shape[0], w_hwio.shape[1]
    w = w_hwio.permute(3, 2, 0, 1).contiguous()          # -> (out, in, kh, kw)
    if padding == "same":
        _, pt, pb = same_pads(x.shape[2], kh, stride)
        _, pl, pr = same_pads(x.shape[3], kw, stride)
    elif padding == "valid":
        pt = pb = pl = pr = 0
    else:
        pt = pb = pl = pr = int(padding)
    x = F.pad(x, (pl, pr, pt, pb))
    return _conv2d_banded(x, w, b, stride)


_BAND_BYTES = 1 << 23        # im2col buffer per band: cache-sized (64 MB bands: 1.4x slower, page faults on every fresh buffer)
_POOL = None


def _pool():
    """im2col / col2im of doubles are single-threaded in torch: the bands of a convolution run on a few threads (the ops release the GIL)"""
    global _POOL
    if _POOL is None:
        import os
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=max(1, min(8, (os.cpu_count() or 2) // 2)))
    return _POOL


class _ConvGemmF64(torch.autograd.Function):
    """float64 convolution of an already padded input as im2col + dgemm, in bands of output rows.

    torch's own float64 CPU convolution is the same arithmetic through a slow path (no oneDNN for doubles: measured 7x slower than
    unfold + matmul on this build) and materialises the whole n*cin*kh*kw*oh*ow buffer (87 GB for final/conv at 512x512, batch 2).
    The bands are recomputed in the backward pass instead of being kept; partial results are summed in band order (deterministic)."""

    @staticmethod
    def _bands(x, kh, kw, stride):
        n, cin, hp, wp = x.shape
        oh, ow = (hp - kh) // stride + 1, (wp - kw) // stride + 1
        per_row = cin * kh * kw * ow * 8
        rows = max(1, min(oh, _BAND_BYTES // max(per_row, 1)))
        return oh, ow, [(i, r0, min(r0 + rows, oh)) for i in range(n) for r0 in range(0, oh, rows)]

    @staticmethod
    def forward(ctx, x, w, stride):
        o, cin, kh, kw = w.shape
        oh, ow, bands = _ConvGemmF64._bands(x, kh, kw, stride)
        wm = w.reshape(o, -1)
        y = x.new_empty(x.shape[0], o, oh, ow)

        @torch.no_grad()                     # grad mode is thread-local: the pool's threads do not inherit Function.forward's
        def one(band):
            i, r0, r1 = band
            cols = F.unfold(x[i:i + 1, :, r0 * stride:(r1 - 1) * stride + kh], (kh, kw), stride=stride)[0]
            y[i, :, r0:r1] = (wm @ cols).view(o, r1 - r0, ow)

        list(_pool().map(one, bands))
        ctx.save_for_backward(x, w)
        ctx.stride = stride
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride = ctx.stride
        o, cin, kh, kw = w.shape
        oh, ow, bands = _ConvGemmF64._bands(x, kh, kw, stride)
        wm = w.reshape(o, -1)
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]

        @torch.no_grad()
        def one(band):
            i, r0, r1 = band
            d = dy[i, :, r0:r1].reshape(o, -1)
            lo, hi = r0 * stride, (r1 - 1) * stride + kh
            pdw = d @ F.unfold(x[i:i + 1, :, lo:hi], (kh, kw), stride=stride)[0].t() if need_dw else None
            pdx = F.fold((wm.t() @ d).unsqueeze(0), (hi - lo, x.shape[3]), (kh, kw), stride=stride) if need_dx else None
            return pdw, pdx

        dx = torch.zeros_like(x) if need_dx else None
        dw = torch.zeros_like(wm) if need_dw else None
        for (i, r0, r1), (pdw, pdx) in zip(bands, _pool().map(one, bands)):
            if need_dw:
                dw += pdw
            if need_dx:
                dx[i:i + 1, :, r0 * stride:(r1 - 1) * stride + kh] += pdx
        return dx, (dw.view_as(w) if need_dw else None), None


def _conv2d_banded(x, w, b, stride):
    """convolution of an already padded input: float64 through _ConvGemmF64, anything else through F.conv2d"""
    if x.dtype != torch.float64:
        return F.conv2d(x, w, b, stride=stride)
    y = _ConvGemmF64.apply(x, w, stride)
    return y if b is None else y + b.view(1, -1, 1, 1)


class _ConvTGemmF64(torch.autograd.Function):
    """float64 full transposed convolution (length (in-1)*s + k) as dgemm + col2im, per sample; w: (in, out, kh, kw)"""

    @staticmethod
    def forward(ctx, x, w, stride):
        n, cin, h, wd = x.shape
        _, o, kh, kw = w.shape
        wm = w.reshape(cin, -1)                                    # [in, out*kh*kw]
        size = ((h - 1) * stride + kh, (wd - 1) * stride + kw)
        y = x.new_empty(n, o, *size)
        for i in range(n):
            y[i] = F.fold((wm.t() @ x[i].reshape(cin, -1)).unsqueeze(0), size, (kh, kw), stride=stride)[0]
        ctx.save_for_backward(x, w)
        ctx.stride = stride
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride = ctx.stride
        n, cin, h, wd = x.shape
        _, o, kh, kw = w.shape
        wm = w.reshape(cin, -1)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.zeros_like(wm) if ctx.needs_input_grad[1] else None
        for i in range(n):
            dcols = F.unfold(dy[i:i + 1], (kh, kw), stride=stride)[0]          # [out*kh*kw, h*w]
            if dx is not None:
                dx[i] = (wm @ dcols).view(cin, h, wd)
            if dw is not None:
                dw += x[i].reshape(cin, -1) @ dcols.t()
        return dx, (dw.view_as(w) if dw is not None else None), None


def conv2d_transpose_same(x, w_hwoi, b, stride=2):
    """keras.layers.Conv2DTranspose(padding='same') forward (reference: model.py:72).

    out = in*stride; equals the full transposed convolution (length (in-1)*s + k) cropped by
    before = floor((k-s)/2), after = ceil((k-s)/2)  (SURVEY.md Appendix A)."""
    kh, kw = w_hwoi.shape[0], w_hwoi.shape[1]
    w = w_hwoi.permute(3, 2, 0, 1).contiguous()          # (in, out, kh, kw) == torch conv_transpose
    full = _ConvTGemmF64.apply(x, w, stride) if x.dtype == torch.float64 else F.conv_transpose2d(x, w, None, stride=stride)
    oh, ow = x.shape[2] * stride, x.shape[3] * stride
    ct = max(kh - stride, 0) // 2
    cl = max(kw - stride, 0) // 2
    y = full[:, :, ct:ct + oh, cl:cl + ow]
    if b is not None:
        y = y + b.view(1, -1, 1, 1)
    return y


def dense(x, w_io, b):
    """keras.layers.Dense: x [N,in] @ (in,out) + b."""
    y = x @ w_io
    return y + b if b is not None else y


def flatten_nhwc(x):
    """keras.layers.Flatten of an NHWC tensor is (h, w, c)-major; x is NCHW here."""
    return x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)


# ----------------------------------------------------------------------------------------------
# normalisation
# ----------------------------------------------------------------------------------------------
def batchnorm(x, gamma, beta, moving_mean, moving_var, training, eps=BN_EPS, momentum=BN_MOMENTUM):
    """keras.layers.BatchNormalization (axis=-1 in NHWC == channel dim 1 here; 2-D input: dim 1).

    training: normalise with the biased batch variance.  Moving statistics are updated as
    ``moving*momentum + batch*(1-momentum)``; for 4-D inputs Keras' TF backend takes the fused path
    (tf.nn.fused_batch_norm) whose returned variance carries Bessel's correction, for 2-D inputs it
    takes the tf.nn.moments path (biased) -- SURVEY.md Appendix A.
    Returns (y, new_moving_mean, new_moving_var)."""
    if x.dim() == 4:
        axes, shape = (0, 2, 3), (1, -1, 1, 1)
    else:
        axes, shape = (0,), (1, -1)
    if training:
        mean = x.mean(dim=axes)
        var = x.var(dim=axes, unbiased=False)
        m = x.numel() // x.shape[1]
        var_upd = var * (m / max(m - 1, 1)) if x.dim() == 4 else var
        new_mm = moving_mean * momentum + mean.detach() * (1 - momentum)
        new_mv = moving_var * momentum + var_upd.detach() * (1 - momentum)
    else:
        mean, var = moving_mean, moving_var
        new_mm, new_mv = moving_mean, moving_var
    y = (x - mean.view(shape)) * torch.rsqrt(var.view(shape) + eps)
    if gamma is not None:
        y = y * gamma.view(shape)
    if beta is not None:
        y = y + beta.view(shape)
    return y, new_mm, new_mv


def instancenorm(x, gamma=None, beta=None, eps=1e-5):
    """Instance normalisation over (H, W) per (n, c) -- north_star extension, no reference
    counterpart; eps follows the canonical CycleGAN (1e-5)."""
    mean = x.mean(dim=(2, 3), keepdim=True)
    var = x.var(dim=(2, 3), unbiased=False, keepdim=True)
    y = (x - mean) * torch.rsqrt(var + eps)
    if gamma is not None:
        y = y * gamma.view(1, -1, 1, 1)
    if beta is not None:
        y = y + beta.view(1, -1, 1, 1)
    return y


# ----------------------------------------------------------------------------------------------
# activations (reference: model.py:21,66,73,276,841)
# ----------------------------------------------------------------------------------------------
def prelu(x, alpha):
    """PReLU(shared_axes=[1,2]): max(x,0) + alpha_c * min(x,0)."""
    a = alpha.view(1, -1, 1, 1) if x.dim() == 4 else alpha.view(1, -1)
    return torch.clamp(x, min=0) + a * torch.clamp(x, max=0)


def leaky_relu(x, alpha):
    return torch.where(x >= 0, x, x * alpha)


def head_activation(x, name):
    """Optional discriminator output squashing (reference: model.py:885-892) and GanLosses
    loss_activation (model.py:172-181)."""
    if name == "sigmoid":
        return torch.sigmoid(x)
    if name == "log-sigm":
        return torch.log(torch.sigmoid(x))
    if name == "tanh":
        return torch.tanh(x)
    if name == "bi-log":
        return (x / (1 + x.abs())) * torch.log(x.abs() + 2)
    return x


# ----------------------------------------------------------------------------------------------
# optimizer: keras.optimizers.Adam() defaults
# ----------------------------------------------------------------------------------------------
def adam_keras_step(p, g, m, v, t, lr=1e-3, beta_1=0.9, beta_2=0.999, eps=1e-7):
    """One Keras-form Adam update; t counts from 1; eps sits OUTSIDE the bias correction."""
    lr_t = lr * math.sqrt(1.0 - beta_2 ** t) / (1.0 - beta_1 ** t)
    m_t = beta_1 * m + (1.0 - beta_1) * g
    v_t = beta_2 * v + (1.0 - beta_2) * g * g
    p_t = p - lr_t * m_t / (torch.sqrt(v_t) + eps)
    return p_t, m_t, v_t


def glorot_uniform(rng, shape, fan_in, fan_out):
    """Keras default kernel initialiser: U(-l, l), l = sqrt(6/(fan_in+fan_out)).  rng is a
    numpy RandomState (frozen stream)."""
    import numpy as np
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape).astype(np.float32)


# ---------------------------------------------------------------------------------------------------------------
# bf16 storage emulation (BASELINE.json configs C3-C5): the product keeps the trunk's activations and their gradients
# in bf16 while all arithmetic is fp32.  These ops mark the storage points in the restatement so that parity of the bf16
# path can be asserted tightly (same roundings at the same places) instead of through loose "bf16-sized" bounds.
# ---------------------------------------------------------------------------------------------------------------
def _r(x):
    return x.to(torch.bfloat16).to(x.dtype)


class _Bf16Store(torch.autograd.Function):
    """tensor stored in bf16: value rounded forward, its gradient rounded backward"""
    @staticmethod
    def forward(ctx, x):
        return _r(x)

    @staticmethod
    def backward(ctx, g):
        return _r(g)


class _Bf16RoundFwd(torch.autograd.Function):
    """value rounded to bf16, gradient passed through (bf16 copy of an fp32 master tensor)"""
    @staticmethod
    def forward(ctx, x):
        return _r(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _Bf16RoundGrad(torch.autograd.Function):
    """identity forward, gradient rounded to bf16 backward"""
    @staticmethod
    def forward(ctx, x):
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return _r(g)


bf16_store, bf16_round_fwd, bf16_round_grad = _Bf16Store.apply, _Bf16RoundFwd.apply, _Bf16RoundGrad.apply
