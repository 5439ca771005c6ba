"""MI355X-native mirror of the reference's ``upscaler.model`` module for the GAN hot path.

Same names, signatures and call protocol as upscaling/upscaler/model.py (reference) for
  make_upscaler_orig (:267), make_discriminator_simple_512 (:836), make_discriminator_thin_512 (:901),
  wasserstein_loss (:159), GanLosses/WassersteinLosses/RelativisticLosses (:166-261),
  make_and_compile_gan (:1017), make_and_compile_gan2 (:1057), compile_training_model (:1130), Adam,
and the Keras ``Model`` methods its callers use: predict / train_on_batch / save / trainable /
input_shape / output_shape / name (train_gan3.py:346-368).  Arrays cross this API as NHWC (numpy or
torch), exactly as in the reference; on the device everything is fp32 NCHW and every arithmetic step
is a hand-written gfx950 kernel from libvcg_hip.so (include/vcg.h).  There is no CPU fallback.

Extensions named by north_star with no reference counterpart:
  make_discriminator_patchgan_70, ``norm='instance'``.
"""
import math
import os
from abc import ABCMeta, abstractmethod
from collections import OrderedDict

import numpy as np
import torch

from . import _engine as E
from . import _lib as L


# =================================================================================================
# optimizer + losses (host-side objects; the arithmetic runs in vcg_adam_keras_multi / vcg_mean_reduce)
# =================================================================================================
class Adam:
    """keras.optimizers.Adam defaults (lr 1e-3, beta 0.9/0.999, epsilon 1e-7, no decay).  One instance
    shared by several compiled models shares ``iterations`` like Keras does (model.py:1026,1066)."""

    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, decay=0.0, amsgrad=False):
        if decay != 0.0 or amsgrad:
            raise NotImplementedError("Adam(decay/amsgrad) is outside the hot path")
        self.lr, self.beta_1, self.beta_2, self.epsilon = lr, beta_1, beta_2, epsilon
        self.iterations = 0

    def lr_t(self):
        """bias-corrected step size, evaluated in double on the float32 values of lr / beta_1 / beta_2 (the values the
        kernels -- and Keras' float32 graph -- work with); vcg_adam_keras_multi_dev evaluates the same expression on the
        device, so eagerly launched and graph-replayed steps agree bit for bit"""
        t = self.iterations + 1
        lr, b1, b2 = (float(np.float32(v)) for v in (self.lr, self.beta_1, self.beta_2))
        return lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)


_DEFAULT_ADAM = Adam()     # mirrors the reference's default-argument instance


def wasserstein_loss(y_true, y_pred):
    """K.mean(y_true * y_pred) (model.py:159-160) evaluated on host arrays."""
    return float(np.mean(np.asarray(y_true, np.float64).reshape(-1, 1) * np.asarray(y_pred, np.float64).reshape(len(y_pred), -1)))


def mean_squared_error(y_true, y_pred):
    return float(np.mean((np.asarray(y_pred, np.float64) - np.asarray(y_true, np.float64)) ** 2))


def mean_absolute_error(y_true, y_pred):
    return float(np.mean(np.abs(np.asarray(y_pred, np.float64) - np.asarray(y_true, np.float64))))


class PixelLoss:
    """Pixel content loss = the non-VGG term of VGG_MSE_LOSS / VGG_MAE_LOSS (model.py:137,157)."""

    def __init__(self, kind="mse"):
        if kind not in ("mse", "mae"):
            raise ValueError(kind)
        self.kind = kind

    def loss(self, y_true, y_pred):
        return mean_squared_error(y_true, y_pred) if self.kind == "mse" else mean_absolute_error(y_true, y_pred)


VGG19_BLOCKS = ((1, 2, 64), (2, 2, 128), (3, 4, 256), (4, 4, 512), (5, 4, 512))


class _VggLossBase:
    """Shared constructor of VGG_LOSS / VGG_MSE_LOSS / VGG_MAE_LOSS (model.py:101-157).

    The reference builds ``VGG19(include_top=False, weights='imagenet')`` -- a download this offline image cannot
    make.  ``vgg19`` therefore names the weights instead of being a Keras model: a ``VGG19Features`` instance, a
    path to a .safetensors / .npz archive or a dict with Keras' layer names (``block1_conv1/kernel`` (3,3,3,64) ...
    ``block5_conv4/bias``), or the string ``'random'`` (seeded He-uniform: throughput runs and parity tests).
    ``vgg19=None`` raises, because silently training against random features is not what the caller asked for."""
    kind, rate = "vgg", 0.0

    def __init__(self, image_shape, vgg19=None):
        self.image_shape = tuple(image_shape)
        if vgg19 is None:
            raise RuntimeError("VGG19 ImageNet weights cannot be downloaded here: pass vgg19=<path to a .safetensors/.npz with "
                               "Keras' VGG19 layer names> (or a dict of arrays), or vgg19='random' for synthetic runs")
        self.model = vgg19 if isinstance(vgg19, VGG19Features) else VGG19Features(self.image_shape, vgg19)
        self.model.trainable = False

    def loss(self, y_true, y_pred):
        """host evaluation on NHWC arrays (the training models evaluate it on the device)"""
        ft, fp = self.model.predict(np.asarray(y_true)), self.model.predict(np.asarray(y_pred))
        d, dp = ft - fp, np.asarray(y_true) - np.asarray(y_pred)
        if self.kind == "vgg_mae":
            return float(np.mean(np.abs(d)) + self.rate * np.mean(np.abs(dp)))
        return float(np.mean(d * d) + self.rate * np.mean(dp * dp))


class VGG_LOSS(_VggLossBase):
    """mean((VGG(y_true) - VGG(y_pred))^2) at block5_conv4 (model.py:101-116)"""


class VGG_MSE_LOSS(_VggLossBase):
    """VGG_LOSS + mse_loss_rate * pixel MSE (model.py:119-137)"""
    kind = "vgg_mse"

    def __init__(self, image_shape, mse_loss_rate=0.1, vgg19=None):
        super().__init__(image_shape, vgg19)
        self.mse_loss_rate = self.rate = float(mse_loss_rate)


class VGG_MAE_LOSS(_VggLossBase):
    """mean|VGG(y_true) - VGG(y_pred)| + mae_loss_rate * pixel MAE (model.py:139-157)"""
    kind = "vgg_mae"

    def __init__(self, image_shape, mae_loss_rate=0.1, vgg19=None):
        super().__init__(image_shape, vgg19)
        self.mae_loss_rate = self.rate = float(mae_loss_rate)


def _content_kind(content_loss):
    owner = getattr(content_loss, "__self__", None)
    if isinstance(owner, _VggLossBase):
        return owner
    if isinstance(content_loss, _VggLossBase):
        return content_loss
    if isinstance(content_loss, str):
        k = {"mse": "mse", "mean_squared_error": "mse", "mae": "mae", "mean_absolute_error": "mae"}.get(content_loss)
        if k is None:
            raise ValueError("unsupported content loss %r" % (content_loss,))
        return k
    if content_loss is mean_squared_error:
        return "mse"
    if content_loss is mean_absolute_error:
        return "mae"
    owner = getattr(content_loss, "__self__", None)
    if isinstance(owner, PixelLoss):
        return owner.kind
    if isinstance(content_loss, PixelLoss):
        return content_loss.kind
    raise NotImplementedError(
        "content loss %r is not on the MI355X hot path: use 'mse'/'mae', PixelLoss(kind).loss or "
        "VGG_LOSS / VGG_MSE_LOSS / VGG_MAE_LOSS(image_shape, ..., vgg19=<weights>).loss" % (content_loss,))


def _act_value_and_grad(name, x):
    """loss_activation(x) and its derivative for a python float x (model.py:172-181)."""
    if name == "sigmoid":
        s = 1.0 / (1.0 + math.exp(-x))
        return s, s * (1 - s)
    if name == "log-sigm":
        # log(sigmoid(x)) ; derivative 1 - sigmoid(x)
        s = 1.0 / (1.0 + math.exp(-x)) if x > -700 else 0.0
        val = -math.log1p(math.exp(-x)) if x > -30 else x
        return val, 1.0 - s
    if name == "tanh":
        t = math.tanh(x)
        return t, 1 - t * t
    if name == "bi-log":
        a = abs(x)
        sgn = 1.0 if x >= 0 else -1.0
        u, v = x / (1 + a), math.log(a + 2)
        du = 1.0 / (1 + a) ** 2
        dv = sgn / (a + 2)
        return u * v, du * v + u * dv
    return x, 1.0


class GanLosses(metaclass=ABCMeta):
    """model.py:166-210.  ``real_output`` / ``fake_output`` hold discriminator outputs (host arrays when
    the closures are evaluated by hand; the training models read them on the device)."""

    def __init__(self, loss_activation="log-sigm", real_output=None, fake_output=None):
        self._real_output = real_output
        self._fake_output = fake_output
        self.loss_activation_name = loss_activation if loss_activation in ("sigmoid", "log-sigm", "tanh", "bi-log") else "none"
        self.loss_activation = lambda x: _act_value_and_grad(self.loss_activation_name, float(x))[0]

    @property
    def real_output(self):
        return self._real_output

    @real_output.setter
    def real_output(self, real_output):
        self._real_output = real_output

    @property
    def fake_output(self):
        return self._fake_output

    @fake_output.setter
    def fake_output(self, fake_output):
        self._fake_output = fake_output

    @property
    @abstractmethod
    def discriminator_loss(self):
        pass

    @property
    @abstractmethod
    def generator_loss(self):
        pass

    # what the device trainer needs to know
    relativistic = False


class WassersteinLosses(GanLosses):
    """model.py:215-235"""

    @property
    def discriminator_loss(self):
        def loss(y_true, y_pred):
            return float(np.mean(self._real_output) - np.mean(self._fake_output))
        return loss

    @property
    def generator_loss(self):
        def loss(y_true, y_pred):
            return float(np.mean(self._fake_output))
        return loss


class RelativisticLosses(GanLosses):
    """model.py:239-261"""
    relativistic = True

    @property
    def discriminator_loss(self):
        def loss(y_true, y_pred):
            return self.loss_activation(np.mean(self._real_output) - np.mean(self._fake_output))
        return loss

    @property
    def generator_loss(self):
        def loss(y_true, y_pred):
            return self.loss_activation(np.mean(self._fake_output) - np.mean(self._real_output))
        return loss


# =================================================================================================
# networks
# =================================================================================================
class Model:
    """Device-resident network with the Keras ``Model`` surface the reference's callers use."""

    def __init__(self, name, input_shape, seed):
        self.name = name
        self.trainable = True
        self._in_shape = tuple(input_shape)
        self.rt = E.Runtime.get()
        self.ps = E.ParamStore()
        self.layers = []
        self._seed = seed

    # -- construction helpers ---------------------------------------------------------------------
    def _add(self, layer):
        layer.declare(self.ps)
        self.layers.append(layer)
        return layer

    def _finish(self):
        self.ps.materialize(self.rt)
        rng = np.random.RandomState(self._seed)
        w = {}
        for l in self.layers:
            l.bind(self.rt, self.ps)
            w.update(l.init_weights(rng))
        self.ps.set_weights(w)

    # -- Keras surface ------------------------------------------------------------------------------
    @property
    def input_shape(self):
        return (None,) + self._in_shape

    @property
    def output_shape(self):
        return (None,) + tuple(self._out_shape(self._in_shape))

    def count_params(self):
        return self.ps.count_params()

    def get_weights_dict(self):
        return self.ps.get_weights()

    def set_weights_dict(self, weights):
        missing = self.ps.set_weights(weights)
        if missing:
            raise KeyError("missing weights: %s" % missing[:5])
        self.refresh()

    # the dict uses the reference's layer/weight names and Keras layouts, so a reference checkpoint
    # converted to {name: array} loads unchanged
    from_reference_weights = set_weights_dict

    def refresh(self):
        for l in self.layers:
            l.refresh()

    def save(self, path):
        """Model.save(path) (train_gan3.py:367-368): flat archive keyed by the reference's layer names."""
        from safetensors.numpy import save_file
        save_file({k: v for k, v in self.ps.get_weights().items()}, path,
                  metadata={"format": "vcg-amd", "name": str(self.name), "input_shape": repr(self._in_shape)})

    def load_weights(self, path):
        from safetensors.numpy import load_file
        self.set_weights_dict(load_file(path))

    def predict(self, x, batch_size=32):
        """learning phase 0 (BN uses moving statistics); NHWC in -> NHWC out, same container type."""
        rt = self.rt
        is_torch = isinstance(x, torch.Tensor)
        n = x.shape[0]
        outs = []
        for i in range(0, n, batch_size):
            xb = E.to_device_nchw(rt, x[i:i + batch_size])
            y, _ = self.forward(xb, training=False)
            outs.append(self._export(y))
        out = outs[0] if len(outs) == 1 else torch.cat(outs, 0)
        return out if is_torch else out.cpu().numpy()

    def _export(self, y):
        return E.to_nhwc(self.rt, y) if y.dim() == 4 else y

    def __call__(self, x):
        return self.predict(x)


class VGG19Features(Model):
    """keras.applications.VGG19(include_top=False) up to block5_conv4 -- the frozen feature extractor of the perceptual
    losses (model.py:108-112).  3x3 'same' convolutions with fused ReLU, 2x2 max pooling after blocks 1-4; only the
    data gradient is ever needed (``backward_data``)."""

    def __init__(self, image_shape, weights="random", seed=19):
        super().__init__("vgg19_block5_conv4", image_shape, seed)
        self.trainable = False
        self.stages = []
        cin = image_shape[2]
        for b, nconv, f in VGG19_BLOCKS:
            convs = []
            for i in range(nconv):
                convs.append(self._add(E.Conv2D("block%d_conv%d" % (b, i + 1), cin, f, 3, act=L.ACT_LRELU, alpha=0.0)))
                cin = f
            self.stages.append((convs, b < 5))
        self._finish()
        if isinstance(weights, str) and weights == "random":
            rng = np.random.RandomState(seed)            # He-uniform (oracle/models.py:init_vgg19_features draws the same)
            w, cin = {}, image_shape[2]
            for b, nconv, f in VGG19_BLOCKS:
                for i in range(nconv):
                    lim = np.sqrt(6.0 / (9 * cin))
                    w["block%d_conv%d/kernel" % (b, i + 1)] = rng.uniform(-lim, lim, (3, 3, cin, f)).astype(np.float32)
                    w["block%d_conv%d/bias" % (b, i + 1)] = rng.uniform(-0.05, 0.05, (f,)).astype(np.float32)
                    cin = f
            self.set_weights_dict(w)
        elif isinstance(weights, dict):
            self.set_weights_dict({k: np.asarray(v, np.float32) for k, v in weights.items()})
        else:
            path = str(weights)
            if path.endswith(".npz"):
                self.set_weights_dict({k: v.astype(np.float32) for k, v in np.load(path).items()})
            else:
                self.load_weights(path)

    def _out_shape(self, s):
        return (s[0] // 16, s[1] // 16, 512)

    def forward(self, x, training=False):
        tape = []
        h = x
        for convs, pool in self.stages:
            for c in convs:
                h, a = c.forward(h, tag="vgg_conv")
                tape.append(a)
            if pool:
                tape.append(h)
                h = E.maxpool2x2(self.rt, h)
        return h, tape

    def backward_data(self, tape, dy):
        """dL/d(input image) from dL/d(features); no parameter gradients (the network is frozen)"""
        tape = list(tape)
        d = dy
        for convs, pool in reversed(self.stages):
            if pool:
                d = E.maxpool2x2_bwd(self.rt, tape.pop(), d)
            for c in reversed(convs):
                d = c.backward(tape.pop(), d, True, False, 0, tag="vgg_conv")
        return d


# A/B switches of the bench scripts ("0" = the fp32 layers on layout-converted copies that the bf16 edge layers replaced)
INIT_BF16 = os.environ.get("VCG_INIT_BF16", "1") != "0"      # generator: initial/conv + PReLU straight into bf16 NHWC
HEAD_BF16 = os.environ.get("VCG_HEAD_BF16", "1") != "0"      # PatchGAN: the one-channel last convolution on bf16 NHWC
FIRST_BF16 = os.environ.get("VCG_FIRST_BF16", "1") != "0"    # critics: block 1 (3 input channels) straight into bf16 NHWC


class UpscalerOrig(Model):
    """make_upscaler_orig topology (model.py:267-295)."""

    def __init__(self, output_image_shape, kernel_size, filters, upscale_factor, res_block_num, norm, seed, trunk_dtype="fp32"):
        f = upscale_factor
        super().__init__("upscaler_orig", (output_image_shape[0] // f, output_image_shape[1] // f, output_image_shape[2]), seed)
        if trunk_dtype not in ("fp32", "bf16", "bf16+tail"):
            raise ValueError(trunk_dtype)
        # 'bf16+tail': also the up-sampling block and final/conv on bf16 activations (forward entirely, backward where the
        # bf16 gradient kernels exist: _engine.ConvT3x3Bf16 / FinalConv9x9Bf16); needs the x2 topology with 64 -> 256 -> 3
        self.tail_bf16 = trunk_dtype == "bf16+tail"
        if self.tail_bf16:
            trunk_dtype = "bf16"
            if upscale_factor != 2:
                raise NotImplementedError("the bf16 tail is instantiated for upscale_factor=2")
        if trunk_dtype == "bf16" and (kernel_size != 3 or filters != 64):
            raise NotImplementedError("the bf16 trunk is instantiated for kernel_size=3, filters=64")
        self.trunk_dtype = trunk_dtype
        bf = trunk_dtype == "bf16"
        conv = (lambda n, ci, co, kk: E.Conv3x3Bf16(n, ci, co)) if bf else (lambda n, ci, co, kk: E.Conv2D(n, ci, co, kk))
        normact = E.NormActBf16 if bf else E.NormAct
        self.upscale_times = int(math.log(f, 2))
        self.factor = 2 ** self.upscale_times
        k = kernel_size
        nrm = {"batch": "batch", "instance": "instance"}[norm]
        # 'bf16+tail': initial/conv + PReLU write the bf16 NHWC trunk input directly (one launch; E.InitialConv9x9Bf16)
        self.init_bf16 = bool(self.tail_bf16 and output_image_shape[2] == 3 and INIT_BF16)
        self.c_init = self._add((E.InitialConv9x9Bf16 if self.init_bf16 else E.Conv2D)("initial/conv", output_image_shape[2], filters, 9))
        self.a_init = self._add(E.NormAct("initial/prelu_op", filters, None, L.ACT_PRELU, prelu_name="initial/prelu"))
        self.blocks = []
        for i in range(res_block_num):
            n = "res_block/%d" % i
            self.blocks.append((
                self._add(conv(n + "/conv_pre", filters, filters, k)),
                self._add(normact(n + "/batch_norm_pre", filters, nrm, L.ACT_PRELU, prelu_name=n + "/prelu")),
                self._add(conv(n + "/conv_post", filters, filters, k)),
                self._add(normact(n + "/batch_norm_post", filters, nrm)),
            ))
        self.c_pre = self._add(conv("prefinal/conv2d", filters, 64, k))               # 64: model.py:283
        self.n_pre = self._add(normact("prefinal/batch_norm", 64, nrm))
        self.ups = []
        cin = 64
        upc = E.ConvT3x3Bf16 if self.tail_bf16 else E.ConvT2D
        for i in range(self.upscale_times):
            self.ups.append(self._add(upc("upscaling/%d/block/conv_transp" % i, cin, 256, k, L.ACT_LRELU, 0.2)))
            cin = 256                                                                 # 256: model.py:288
        self.c_fin = self._add(E.FinalConv9x9Bf16("final/conv", cin, 3, 9) if self.tail_bf16 else E.Conv2D("final/conv", cin, 3, 9, act=L.ACT_TANH))
        self._finish()
        if bf and len(self.blocks) * 2 + 1 <= 48:
            E.PackGroup3x3([c for b in self.blocks for c in (b[0], b[2])] + [self.c_pre])

    def _out_shape(self, s):
        return (s[0] * self.factor, s[1] * self.factor, 3)

    def to_inference_bf16(self):
        """inference engine on the bf16-storage kernels (BN folded, one hipGraph per input shape): ``_infer.py``"""
        from ._infer import Bf16Generator
        return Bf16Generator(self)

    def forward(self, x, training):
        tape = []
        if self.init_bf16:
            h, c = self.c_init.forward_prelu(x, self.ps[self.a_init.prelu_name + "/alpha"], training); tape.extend((c, None))
        else:
            h, c = self.c_init.forward(x); tape.append(c)
            h, c = self.a_init.forward(h, training); tape.append(c)
            if self.trunk_dtype == "bf16":          # the trunk (2*res+1 convolutions, norms, adds) on bf16 NHWC; fp32 outside
                h = E.to_bf16_nhwc(self.rt, h)
        skip = h
        bf = self.trunk_dtype == "bf16"
        folds = {}
        if bf and not training and self.n_pre.norm == "batch" and E.FOLD_PREDICT and len(self.blocks) * 2 + 1 <= 48:
            # every folded BatchNormalization of the pass in one launch
            folds = E.fold_batch(self.rt, [p for b in self.blocks for p in ((b[0], b[1]), (b[2], b[3]))] + [(self.c_pre, self.n_pre)])

        def conv_norm(cv, nm, h, residual=None):
            """conv -> norm[-> act][+ residual]; on the bf16 trunk the convolution's epilogue hands the norm its statistics, and in
            learning phase 0 (predict) the whole group is one launch (BatchNormalization folded into the epilogue)"""
            if bf and not training and nm.norm == "batch" and E.FOLD_PREDICT:
                tape.extend((None, None))
                return cv.forward_folded(h, nm, residual=residual, tag="trunk_conv", folded=folds.get(id(cv)))
            if nm.needs_stats(training):          # (both dtypes: the convolution's epilogue hands the norm its statistics)
                h, a, st = cv.forward_stats(h, nm.norm == "instance", tag="trunk_conv"); tape.append(a)
                h, a = nm.forward(h, training, residual=residual, stats=st); tape.append(a)
            else:
                h, a = cv.forward(h, tag="trunk_conv"); tape.append(a)
                h, a = nm.forward(h, training, residual=residual); tape.append(a)
            return h

        for (c1, n1, c2, n2) in self.blocks:
            gen = h
            h = conv_norm(c1, n1, h)
            h = conv_norm(c2, n2, h, residual=gen)
        h = conv_norm(self.c_pre, self.n_pre, h, residual=skip)
        if self.trunk_dtype == "bf16" and not self.tail_bf16:
            h = E.from_bf16_nhwc(self.rt, h)
        if self.tail_bf16 and len(self.ups) == 1:
            # up-sampling block -> final/conv in chunks of frames small enough for the Infinity Cache (E.tail_chunk): final/conv reads the
            # 256-channel tensor right after it was written.  Training keeps the whole tensor for the backward pass (each chunk is a slice of
            # it); predict re-uses one chunk buffer.
            n, hh, ww, _ = h.shape
            ch = E.tail_chunk(n, hh, ww, self.ups[0].cout)
            if ch < n:
                up = self.ups[0]
                y = self.rt.empty(n, 3, 2 * hh, 2 * ww)
                uall = torch.empty(n, 2 * hh, 2 * ww, up.cout, dtype=torch.bfloat16, device=self.rt.device) if training else None
                for i in range(0, n, ch):
                    c = min(ch, n - i)
                    u, _ = up.forward(h[i:i + c], tag="convt", out=uall[i:i + c] if training else None)
                    self.c_fin.forward(u, tag="final_conv", out=y[i:i + c])
                    del u
                if training:
                    tape.append((h, uall, up.desc(n, hh, ww)))
                    tape.append((uall, y, self.c_fin.desc(n, 2 * hh, 2 * ww)))
                else:
                    tape.extend((None, None))
                return y, tape
        for u in self.ups:
            h, a = u.forward(h, tag="convt"); tape.append(a)
        h, a = self.c_fin.forward(h, tag="final_conv"); tape.append(a)
        return h, tape

    def backward(self, tape, dy, which=0):
        """gradient of all trainables into grads[which]; dy = dL/d(output) NCHW."""
        rt = self.rt
        tape = list(tape)
        if self.tail_bf16:
            # final/conv's bf16 data gradient applies the up-sampling block's LeakyReLU derivative; the block's own gradients run
            # on the bf16 kernels and hand back bf16 NHWC, which is what the trunk's backward consumes
            d, sums = self.c_fin.backward(tape.pop(), dy, True, True, which, tag="final_conv", input_lrelu_slope=self.ups[-1].alpha,
                                          want_channel_sums=True)
            d = self.ups[-1].backward(tape.pop(), d, True, True, which, tag="convt", dz_channel_sums=sums)
        else:
            d = self.c_fin.backward(tape.pop(), dy, True, True, which, tag="final_conv")
            for u in reversed(self.ups):
                d = u.backward(tape.pop(), d, True, True, which, tag="convt")
        # s = skip + BN(conv(h)):  d flows to both
        if self.init_bf16:
            dskip = d                              # bf16 NHWC: added to the trunk's gradient inside the PReLU-backward pass
        elif self.tail_bf16:
            dskip = E.from_bf16_nhwc(rt, d)        # the long skip joins the fp32 initial/prelu output
        else:
            dskip = d
            if self.trunk_dtype == "bf16":
                d = E.to_bf16_nhwc(rt, d)
        d = self.n_pre.backward(tape.pop(), d, True, which)
        d = self.c_pre.backward(tape.pop(), d, True, True, which, tag="trunk_conv")
        for (c1, n1, c2, n2) in reversed(self.blocks):
            dres = d                                  # gradient wrt the block output = wrt `gen` branch too
            d = n2.backward(tape.pop(), d, True, which)
            d = c2.backward(tape.pop(), d, True, True, which, tag="trunk_conv")
            d = n1.backward(tape.pop(), d, True, which)
            d = c1.backward(tape.pop(), d, True, True, which, dx_residual=dres, tag="trunk_conv")
        # d is now dL/d(a_init output) from the trunk; add the long-skip gradient
        if self.init_bf16:
            tape.pop()
            self.c_init.backward_prelu(tape.pop(), d, dskip, self.ps[self.a_init.prelu_name + "/alpha"],
                                       self.ps.grad(self.a_init.prelu_name + "/alpha", which), which)
            return None
        if self.trunk_dtype == "bf16":
            d = E.from_bf16_nhwc(rt, d)
        E.axpby(rt, dskip, d, 1.0, 1.0)
        d = self.a_init.backward(tape.pop(), d, True, which)
        self.c_init.backward(tape.pop(), d, False, True, which)
        return None


class DiscriminatorStack(Model):
    """Strided-conv critic with Flatten/Dense head: make_discriminator_simple_512 (model.py:836-896) and
    make_discriminator_thin_512 (:901-961).

    ``dtype='bf16'`` (BASELINE.json configs C3/C4): every block keeps its activations in bf16 NHWC (block 1: FirstConvBf16 from the fp32
    NCHW frames; blocks 2..9: Conv2DBf16; NormActBf16: fp32 accumulation, statistics, master weights and gradients); the Dense head stays fp32.  Flatten of
    the NHWC tensor is its memory order, so entering the head is a flat bf16 -> fp32 conversion."""

    def __init__(self, input_shape, filters, activation, seed, name, dtype="fp32", kernel=3, strides=None, padding="same", dense1=1024):
        super().__init__(name, input_shape, seed)
        if dtype not in ("fp32", "bf16"):
            raise ValueError(dtype)
        strides = strides or (1,) + (2,) * (len(filters) - 1)
        self.dtype = dtype
        self.activation = activation
        self.head = L.HEAD_KINDS.get(activation, L.HEAD_NONE)     # any other string: no activation, like the reference's if/elif chain
        self.convs = []
        cin = input_shape[2]
        h, w = input_shape[0], input_shape[1]
        for i, f in enumerate(filters):
            n = "discriminator/block_%d" % (i + 1)
            s = strides[i]
            first_bf = dtype == "bf16" and i == 0 and FIRST_BF16 and cin == 3 and f % 64 == 0 and (kernel, s) == (3, 1)
            bf = dtype == "bf16" and (i > 0 or first_bf)
            conv = E.FirstConvBf16 if first_bf else E.Conv2DBf16 if bf else E.Conv2D
            norm = E.NormActBf16 if bf else E.NormAct
            cv = self._add(conv(n + "/Conv2d", cin, f, kernel, s, padding))
            self.convs.append((cv, self._add(norm(n + "/BatchNorm", f, "batch", L.ACT_LRELU, 0.1))))
            h, w, _, _ = cv.out_hw(h, w)
            if h < 1 or w < 1:
                raise ValueError("input %s is too small for %s" % (tuple(input_shape), name))
            cin = f
        self.first_bf16 = isinstance(self.convs[0][0], E.FirstConvBf16)
        self.flat = (h, w, cin)
        self.d1 = self._add(E.Dense("discriminator/final/Dense_1", h * w * cin, dense1))
        self.b1 = self._add(E.NormAct("discriminator/final/BatchNorm_1", dense1, "batch", L.ACT_LRELU, 0.1))
        self.d2 = self._add(E.Dense("discriminator/final/Dense_2", dense1, 32))
        self.b2 = self._add(E.NormAct("discriminator/final/BatchNorm_2", 32, "batch", L.ACT_LRELU, 0.1))
        self.d3 = self._add(E.Dense("discriminator/final/Dense_3", 32, 1))
        self._finish()

    def _out_shape(self, s):
        return (1,)

    @property
    def output_shape(self):
        return (None, 1)

    def forward(self, x, training, update_moving=True):
        if tuple(x.shape[2:]) != tuple(self._in_shape[:2]):
            raise ValueError("discriminator built for %s, got %s" % (self._in_shape[:2], tuple(x.shape[2:])))
        rt = self.rt
        bf = self.dtype == "bf16"
        tape = []
        h = x
        for i, (cv, na) in enumerate(self.convs):
            if bf and i == 1 and not self.first_bf16:
                h = E.to_bf16_nhwc(rt, h)
            if i > 0 and na.needs_stats(training) and cv.act == L.ACT_NONE:
                h, a, st = cv.forward_stats(h, False, tag="d_conv"); tape.append(a)
                h, a = na.forward(h, training, update_moving=update_moving, stats=st); tape.append(a)
                continue
            h, a = cv.forward(h, tag="d_conv"); tape.append(a)
            h, a = na.forward(h, training, update_moving=update_moving); tape.append(a)
        if bf and len(self.convs) > 1:
            tape.append(tuple(h.shape))
            hf = E.bf16_to_f32(rt, h)                  # Flatten of NHWC is (h,w,c)-major (Appendix A) = this tensor's memory order
        else:
            hf = E.to_nhwc(rt, h)
            tape.append(tuple(h.shape))
        h = hf.view(hf.shape[0], -1)
        for dn, bn in ((self.d1, self.b1), (self.d2, self.b2)):
            h, a = dn.forward(h); tape.append(a)
            h, a = bn.forward(h, training, update_moving=update_moving); tape.append(a)
        h, a = self.d3.forward(h); tape.append(a)
        if self.head:                                  # model.py:885-892: optional squashing of the critic's output
            tape.append(h)
            h = E.head_act_fwd(self.rt, h, self.head)
        return h, tape

    def backward(self, tape, dy, need_dx=False, param_grads=True, which=0):
        rt = self.rt
        bf = self.dtype == "bf16" and len(self.convs) > 1
        tape = list(tape)
        if self.head:
            dy = E.head_act_bwd(rt, tape.pop(), dy, self.head)
        d = self.d3.backward(tape.pop(), dy, True, param_grads, which)
        for dn, bn in ((self.d2, self.b2), (self.d1, self.b1)):
            d = bn.backward(tape.pop(), d, param_grads, which)
            d = dn.backward(tape.pop(), d, True, param_grads, which)
        shp = tape.pop()
        if bf:
            d = E.f32_to_bf16(rt, d).view(*shp)        # [n, h*w*c] -> bf16 NHWC
        else:
            n, c, h, w = shp
            dn_ = rt.empty(n, c, h, w)
            L.check(rt.lib.vcg_nhwc_to_nchw(d.data_ptr(), dn_.data_ptr(), n, h, w, c, rt.stream), "vcg_nhwc_to_nchw")
            d = dn_
        for i, (cv, na) in reversed(list(enumerate(self.convs))):
            d = na.backward(tape.pop(), d, param_grads, which)
            d = cv.backward(tape.pop(), d, need_dx or i > 0, param_grads, which, tag="d_conv")
            if bf and i == 1 and not self.first_bf16:
                d = E.from_bf16_nhwc(rt, d)
        return d


class DiscriminatorPatchGAN(Model):
    """70x70 PatchGAN (north_star extension): C64-C128-C256 (k4 s2),
    C512 (k4 s1), C1 (k4 s1), zero padding 1, LeakyReLU 0.2, instance (default) or batch norm on the
    three middle blocks.  ``dtype='bf16'``: the three middle blocks (99 % of its FLOPs) on bf16 NHWC activations
    (Conv2DBf16 / NormActBf16), the 1-channel last convolution reads them as they are (ConvCout1Bf16: fp32 weights and output), the
    3-channel first convolution + LeakyReLU writes them from the fp32 NCHW frames (FirstConvBf16)."""
    SPEC = ((64, 2, False), (128, 2, True), (256, 2, True), (512, 1, True), (1, 1, False))

    def __init__(self, input_shape, activation, norm, seed, dtype="fp32"):
        super().__init__("discriminator_patchgan_70", input_shape, seed)
        if dtype not in ("fp32", "bf16"):
            raise ValueError(dtype)
        self.dtype = dtype
        self.activation = activation
        self.head = L.HEAD_KINDS.get(activation, L.HEAD_NONE)
        self.convs = []
        cin = input_shape[2]
        for i, (f, s, has_norm) in enumerate(self.SPEC):
            n = "discriminator/block_%d" % (i + 1)
            last = i == len(self.SPEC) - 1
            if has_norm:
                bf = dtype == "bf16"
                cv = self._add((E.Conv2DBf16 if bf else E.Conv2D)(n + "/Conv2d", cin, f, 4, s, 1))
                na = self._add((E.NormActBf16 if bf else E.NormAct)(n + "/BatchNorm", f, norm, L.ACT_LRELU, 0.2))
            elif i == 0 and dtype == "bf16" and FIRST_BF16 and cin == 3:
                cv = self._add(E.FirstConvBf16(n + "/Conv2d", cin, f, 4, s, 1, L.ACT_LRELU, 0.2))   # fp32 NCHW frames -> bf16 NHWC in one launch
                na = None
            elif last and dtype == "bf16" and HEAD_BF16:
                cv = self._add(E.ConvCout1Bf16(n + "/Conv2d", cin, f, 4, s, 1))         # reads / writes the bf16 NHWC tensor directly
                na = None
            else:
                cv = self._add(E.Conv2D(n + "/Conv2d", cin, f, 4, s, 1, L.ACT_NONE if last else L.ACT_LRELU, 0.2))
                na = None
            self.convs.append((cv, na))
            cin = f
        self.first_bf16 = isinstance(self.convs[0][0], E.FirstConvBf16)
        self._finish()

    def _out_shape(self, s):
        h, w = s[0], s[1]
        for cv, _ in self.convs:
            h, w, _, _ = cv.out_hw(h, w)
        return (h, w, 1)

    def forward(self, x, training, update_moving=True):
        rt, bf = self.rt, self.dtype == "bf16"
        tape = []
        h = x
        last = len(self.convs) - 1
        for i, (cv, na) in enumerate(self.convs):
            if bf and i == 1 and not self.first_bf16:
                h = E.to_bf16_nhwc(rt, h)
            if bf and i == last and not isinstance(cv, E.ConvCout1Bf16):
                h = E.from_bf16_nhwc(rt, h)
            if na is not None and na.needs_stats(training) and (bf or (0 < i < last and cv.act == L.ACT_NONE)):
                h, a, st = cv.forward_stats(h, na.norm == "instance", tag="d_conv"); tape.append(a)
                h, a = na.forward(h, training, update_moving=update_moving, stats=st); tape.append(a)
                continue
            h, a = cv.forward(h, tag="d_conv"); tape.append(a)
            if na is not None:
                h, a = na.forward(h, training, update_moving=update_moving); tape.append(a)
        if self.head:
            tape.append(h)
            h = E.head_act_fwd(self.rt, h, self.head)
        return h, tape

    def backward(self, tape, dy, need_dx=False, param_grads=True, which=0):
        rt, bf = self.rt, self.dtype == "bf16"
        tape = list(tape)
        d = dy
        if self.head:
            d = E.head_act_bwd(self.rt, tape.pop(), d, self.head)
        last = len(self.convs) - 1
        for i, (cv, na) in reversed(list(enumerate(self.convs))):
            if na is not None:
                d = na.backward(tape.pop(), d, param_grads, which)
            if bf and i == 1 and self.first_bf16:
                # block 2's data gradient applies the derivative of block 1's LeakyReLU (its input IS that activation's output)
                d = cv.backward(tape.pop(), d, True, param_grads, which, tag="d_conv", input_lrelu_slope=self.convs[0][0].alpha)
            else:
                d = cv.backward(tape.pop(), d, need_dx or i > 0, param_grads, which, tag="d_conv")
            if bf and i == last and not isinstance(cv, E.ConvCout1Bf16):
                d = E.to_bf16_nhwc(rt, d)
            if bf and i == 1 and not self.first_bf16:
                d = E.from_bf16_nhwc(rt, d)
        return d


# =================================================================================================
# factories -- reference signatures
# =================================================================================================
def make_upscaler_orig(output_image_shape, kernel_size=5, filters=64, upscale_factor=4, res_block_num=16,
                       norm="batch", seed=7, trunk_dtype="fp32"):
    """model.py:267-295.  ``norm='instance'``, ``seed`` and ``trunk_dtype`` are extensions (keyword-only in spirit);
    ``trunk_dtype='bf16'`` keeps the residual trunk's activations in bf16 (fp32 master weights, gradients, statistics)."""
    if upscale_factor < 1 or (upscale_factor & (upscale_factor - 1)) != 0:
        raise ValueError("upscale_factor must be a power of two (train_gan3.py:120-122)")
    return UpscalerOrig(tuple(output_image_shape), kernel_size, filters, upscale_factor, res_block_num, norm, seed, trunk_dtype)


def make_discriminator_simple_512(input_shape, activation="none", seed=11, dtype="fp32"):
    """model.py:836-896.  ``seed`` / ``dtype`` are extensions: dtype='bf16' keeps the activations of blocks 2..9 in bf16."""
    return DiscriminatorStack(tuple(input_shape), (64, 128, 256, 512, 512, 512, 512, 512, 512), activation, seed,
                              "discriminator_simple_512", dtype)


def make_discriminator_thin_512(input_shape, activation="none", seed=11, dtype="fp32"):
    """model.py:901-961"""
    return DiscriminatorStack(tuple(input_shape), (64,) + (128,) * 8, activation, seed, "discriminator_thin_512", dtype)


def make_discriminator_sparse_512(input_shape, activation="none", seed=11):
    """model.py:964-1012: 5x5 'valid' convolutions, stride 1 then 3 (64-128-256-256-256-256), Dense 128 / 32 / 1.  fp32 only."""
    return DiscriminatorStack(tuple(input_shape), (64, 128, 256, 256, 256, 256), activation, seed, "discriminator_sparse_512",
                              kernel=5, strides=(1, 3, 3, 3, 3, 3), padding="valid", dense1=128)


def make_discriminator_patchgan_70(input_shape, activation="none", norm="instance", seed=11, dtype="fp32"):
    """north_star extension following the reference's factory naming pattern."""
    return DiscriminatorPatchGAN(tuple(input_shape), activation, norm, seed, dtype)


# =================================================================================================
# training wiring -- make_and_compile_gan / make_and_compile_gan2 / compile_training_model
# =================================================================================================
def _pixel_loss(rt, pred, target, kind, scale, out=None):
    """(value [1], d(scale * value)/d pred) of the mean squared / absolute difference of two device tensors"""
    val, dpred = (rt.empty(1) if out is None else out), rt.empty(*pred.shape)
    ws, wsn = rt.workspace(rt.lib.vcg_mean_reduce_workspace_bytes(pred.numel()))
    L.check(rt.lib.vcg_pixel_loss(pred.data_ptr(), target.data_ptr(), pred.numel(), L.LOSS_MSE if kind == "mse" else L.LOSS_MAE,
                                  float(scale), val.data_ptr(), dpred.data_ptr(), ws, wsn, rt.stream), "vcg_pixel_loss")
    return val, dpred


def _content_loss_and_grad(rt, kind, weight, fake, hr, out=None):
    """content loss value (device scalar, unweighted) and d(weight * loss)/d fake for 'mse' / 'mae' or a VGG19
    perceptual loss object (model.py:101-157): features of hr and fake at block5_conv4, their mean squared / absolute
    difference, its gradient back through the frozen VGG19 to the fake frames, plus ``rate`` x the pixel term."""
    if isinstance(kind, str):
        return _pixel_loss(rt, fake, hr, kind, weight, out)
    vgg = kind.model
    f_real, _ = vgg.forward(hr)
    f_fake, tape = vgg.forward(fake)
    pk = "mae" if kind.kind == "vgg_mae" else "mse"
    val, dfeat = _pixel_loss(rt, f_fake, f_real, pk, weight, out)
    dfake = vgg.backward_data(tape, dfeat)
    if kind.rate:
        pval, dpix = _pixel_loss(rt, fake, hr, pk, weight * kind.rate)
        E.axpby(rt, dpix, dfake, 1.0, 1.0)
        E.axpby(rt, pval, val, kind.rate, 1.0)
    return val, dfake


class _Slots:
    """Adam m/v for one compiled model (Keras creates separate slots per get_updates call)."""

    def __init__(self, model):
        rt = model.rt
        self.m = rt.zeros(max(model.ps.n_trainable, 1))
        self.v = rt.zeros(max(model.ps.n_trainable, 1))


class GanTrainer:
    """Shared state behind the three training models: who updates what, with which loss.

    The loop body (train_gan3.py:346-354) is held as a *plan*: kernel-only segments separated by the collectives of data
    parallelism.  Run eagerly it is the three reference calls; recorded, every run of consecutive segments is one hipGraph
    (one graph per step on a single GPU).  No segment reads a device value on the host: the GAN losses -- including the
    relativistic ones' non-linearity and its derivative (model.py:244-259) -- are evaluated by vcg_gan_loss from the two
    mean(D(.)) scalars where they lie.  The four reported losses are left in one packed device buffer
        lossbuf = [d_a, d_b, content, adversarial]     (loss_disc = wa*d_a - wb*d_b)
    and read (one D2H copy; under DP one 4-float all-reduce) when the caller asks for the Keras return values."""

    def __init__(self, generator, discriminator, wiring, content_kind, content_w, losses, disc_w, optimizer,
                 process_group=None):
        self.G, self.D = generator, discriminator
        self.rt = generator.rt
        self.wiring = wiring
        self.content_kind, self.cw, self.dw = content_kind, float(content_w), float(disc_w)
        self.losses = losses            # GanLosses instance (gan2) or None (v1: wasserstein_loss)
        self.opt = optimizer
        self.g_slots, self.d_slots = _Slots(generator), _Slots(discriminator)
        self.pg = process_group
        from . import _dist
        self.world = _dist.world_size(process_group)
        self._t_dev = None          # device copy of optimizer.iterations (graph-replayable Adam)
        self._graph = None
        self._lossbuf = self.rt.zeros(4)
        self._means = self.rt.zeros(2)
        self._loss_w = (1.0, 0.0)
        self.relativistic = bool(losses is not None and losses.relativistic)
        self.loss_kind = L.HEAD_KINDS[losses.loss_activation_name] if self.relativistic else L.HEAD_NONE
        # fused=True (extension, SURVEY.md section 7 "three-graph semantics"): the step WITHOUT the reference's separate
        # learning-phase-0 generator pass (train_gan3.py:346) -- the fakes the critic is trained on are those of the
        # generator's one training-mode forward, which the generator step then re-uses: 3 G + 9 D forward-equivalents
        # instead of 4 G + 9 D.  Default False: the faithful three-call step.
        self.fused = False
        self.coll_prof = None       # bench.py: {name: [(event0, event1), ...]} around every collective of data parallelism
        self._pending = None

    # -- collectives (data parallelism; never inside a recorded segment) -------------------------------
    def _reduce_grads(self, model, which=0):
        """DP: SUM of the model's flat gradient bucket over the ranks (one RCCL all-reduce); Adam applies the 1/ranks."""
        if self.pg is not None:
            from . import _dist
            _dist.allreduce_sum(model.ps.grads if which == 0 else model.ps.grads2, self.pg)

    _sync_grads = _reduce_grads

    def _timed_coll(self, name, fn):
        """run one collective; with coll_prof set, bracket it with HIP events on the compute stream (the stream that waits
        for it), so that event1 - event0 is the time the step is exposed to the collective"""
        if self.coll_prof is None or not torch.cuda.is_available():
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        self.coll_prof.setdefault(name, []).append((e0, e1))
        return r

    def _reduce_grads_start(self, model, which=0):
        """DP: issue the bucket's all-reduce without waiting for it -- RCCL runs it on its own stream behind the work already
        queued, the compute stream goes on with whatever does not read the bucket (the generator's forward, below)"""
        if self.pg is not None:
            from . import _dist
            self._pending = _dist.allreduce_sum_start(model.ps.grads if which == 0 else model.ps.grads2, self.pg)

    def _reduce_grads_wait(self):
        h, self._pending = self._pending, None
        if h is not None:
            h.wait()

    def _reduce_means(self):
        """DP, relativistic losses: the two mean(D(.)) scalars pass through a non-linearity before back-prop, so they
        must be the GLOBAL batch's: one 2-float all-reduce, its 1/ranks folded into vcg_gan_loss."""
        if self.pg is not None:
            from . import _dist
            _dist.allreduce_sum(self._means, self.pg)

    @property
    def _mean_scale(self):
        return 1.0 / self.world if (self.relativistic and self.pg is not None) else 1.0

    # -- optimizer ------------------------------------------------------------------------------------
    def _adam(self, model, slots, which=0):
        self._reduce_grads(model, which)
        self._apply_adam(model, slots, which)

    def _apply_adam(self, model, slots, which=0):
        rt = self.rt
        ps = model.ps
        g = ps.grads if which == 0 else ps.grads2
        gs = 1.0 / self.world
        if self._t_dev is not None:
            L.check(rt.lib.vcg_adam_keras_multi_dev(ps.params.data_ptr(), g.data_ptr(), slots.m.data_ptr(), slots.v.data_ptr(),
                                                    ps.n_trainable, float(self.opt.lr), float(self.opt.beta_1),
                                                    float(self.opt.beta_2), float(self.opt.epsilon), gs, self._t_dev.data_ptr(),
                                                    rt.stream), "vcg_adam_keras_multi_dev")
        else:
            L.check(rt.lib.vcg_adam_keras_multi(ps.params.data_ptr(), g.data_ptr(), slots.m.data_ptr(), slots.v.data_ptr(),
                                                ps.n_trainable, float(self.opt.lr_t()), float(self.opt.beta_1),
                                                float(self.opt.beta_2), float(self.opt.epsilon), gs, rt.stream),
                    "vcg_adam_keras_multi")
        self.opt.iterations += 1
        model.refresh()

    # -- segments of the loop body ---------------------------------------------------------------------
    def predict(self, lr_nchw):
        y, _ = self.G.forward(lr_nchw, training=False)
        return y

    def _disc_forward(self, s, hr, fake):
        """D on the real and on the generated batch; their output means (model.py:220-224,244-248 / train_gan.py:305-315)"""
        rt, D = self.rt, self.D
        if self.wiring == "gan2":
            s["out_r"], s["tape_r"] = D.forward(hr, True, True)
            s["out_f"], s["tape_f"] = D.forward(fake, True, True)
            E.mean_scalar(rt, s["out_r"], out=self._means[0:1])
            E.mean_scalar(rt, s["out_f"], out=self._means[1:2])
            self._loss_w = (1.0, 0.0)
        else:
            x = torch.cat([hr, fake], 0)       # device-side concatenation of the two batches (memory op)
            out, s["tape"] = D.forward(x, True, True)
            nb = hr.shape[0]
            per = out.numel() // out.shape[0]
            tot = out.numel()
            s["out"], s["nreal"] = out, nb * per
            E.mean_scalar(rt, out[:nb], out=self._lossbuf[0:1])
            E.mean_scalar(rt, out[nb:], out=self._lossbuf[1:2])
            self._loss_w = (nb * per / tot, (tot - nb * per) / tot)

    def _disc_backward(self, s):
        """loss (device scalar) and its broadcast gradients, D backward on both applications into two gradient buffers"""
        rt, D = self.rt, self.D
        if self.wiring == "gan2":
            out_r, out_f = s.pop("out_r"), s.pop("out_f")
            dyr, dyf = rt.empty(*out_r.shape), rt.empty(*out_f.shape)
            E.gan_loss(rt, self._means[0:1], self._means[1:2], self._mean_scale, self.loss_kind, self._lossbuf[0:1],
                       dyr, 1.0 / out_r.numel(), dyf, -1.0 / out_f.numel())
            D.backward(s.pop("tape_r"), dyr, False, True, 0)
            D.backward(s.pop("tape_f"), dyf, False, True, 1)
            E.axpby(rt, D.ps.grads2, D.ps.grads, 1.0, 1.0)
        else:
            out, nreal = s.pop("out"), s.pop("nreal")
            tot = out.numel()
            dy = rt.empty(*out.shape)
            L.check(rt.lib.vcg_fill(dy.data_ptr(), nreal, 1.0 / tot, rt.stream), "vcg_fill")
            L.check(rt.lib.vcg_fill(dy.data_ptr() + 4 * nreal, tot - nreal, -1.0 / tot, rt.stream), "vcg_fill")
            D.backward(s.pop("tape"), dy, False, True, 0)

    def _gan_forward_g(self, s, lr):
        """G forward with batch statistics (model.py:1103): independent of the critic's update, so under data parallelism it
        runs while the critic's gradient bucket is being all-reduced"""
        s["fake"], s["gtape"] = self.G.forward(lr, True)

    def _gan_forward(self, s, lr, hr):
        """G forward with batch statistics, frozen D on the fakes (and on the real batch for the relativistic loss),
        content loss value + gradient (model.py:1103-1123)"""
        rt, G, D = self.rt, self.G, self.D
        if "fake" not in s:
            self._gan_forward_g(s, lr)
        fake = s.pop("fake")
        s["out_f"], s["dtape"] = D.forward(fake, True, False)          # frozen D: batch stats, no moving update
        _, s["dfake"] = _content_loss_and_grad(rt, self.content_kind, self.cw, fake, hr, out=self._lossbuf[2:3])
        E.mean_scalar(rt, s["out_f"], out=self._means[0:1])
        if self.relativistic:
            out_r, _ = D.forward(hr, True, False)
            E.mean_scalar(rt, out_r, out=self._means[1:2])

    def _gan_backward(self, s):
        rt, G, D = self.rt, self.G, self.D
        out_f = s.pop("out_f")
        dyf = rt.empty(*out_f.shape)
        E.gan_loss(rt, self._means[0:1], self._means[1:2] if self.relativistic else None, self._mean_scale, self.loss_kind,
                   self._lossbuf[3:4], dyf, self.dw / out_f.numel())
        d_adv = D.backward(s.pop("dtape"), dyf, True, False, 0)
        dfake = s.pop("dfake")
        E.axpby(rt, d_adv, dfake, 1.0, 1.0)
        G.backward(s.pop("gtape"), dfake, 0)

    def _plan(self, lr, hr):
        """the loop body as [(is_collective, callable)]"""
        s = {}
        dp = self.pg is not None
        rel_dp = dp and self.relativistic
        tc = self._timed_coll
        if self.fused:      # one training-mode generator forward serves the critic's step and the generator's
            P = [(False, lambda: (self._gan_forward_g(s, lr), self._disc_forward(s, hr, s["fake"])))]
        else:
            P = [(False, lambda: self._disc_forward(s, hr, self.predict(lr)))]
        if rel_dp:
            P.append((True, lambda: tc("means_d_step", self._reduce_means)))
        P.append((False, lambda: self._disc_backward(s)))
        if dp:
            # the critic's bucket travels while the generator's training-mode forward (which reads no critic weight) runs
            P.append((True, lambda: tc("d_bucket_issue", lambda: self._reduce_grads_start(self.D))))
            if not self.fused:
                P.append((False, lambda: self._gan_forward_g(s, lr)))
            P.append((True, lambda: tc("d_bucket_wait", self._reduce_grads_wait)))
        P.append((False, lambda: (self._apply_adam(self.D, self.d_slots), self._gan_forward(s, lr, hr))))
        if rel_dp:
            P.append((True, lambda: tc("means_g_step", self._reduce_means)))
        P.append((False, lambda: self._gan_backward(s)))
        if dp:
            P.append((True, lambda: tc("g_bucket", lambda: self._reduce_grads(self.G))))
        P.append((False, lambda: self._apply_adam(self.G, self.g_slots)))
        return P

    # -- the reference's calls --------------------------------------------------------------------------
    def disc_step(self, hr, fake, apply=True):
        """disc_train.train_on_batch (train_gan3.py:353 / train_gan.py:315).  hr, fake: device NCHW.
        apply=False stops after the local gradients (the caller syncs and applies them)."""
        s = {}
        self._disc_forward(s, hr, fake)
        if self.relativistic:
            self._reduce_means()
        self._disc_backward(s)
        if apply:
            self._adam(self.D, self.d_slots)

    def gan_step(self, lr, hr, apply=True):
        """gan_train.train_on_batch (train_gan3.py:354 / train_gan.py:317)."""
        s = {}
        self._gan_forward(s, lr, hr)
        if self.relativistic:
            self._reduce_means()
        self._gan_backward(s)
        if apply:
            self._adam(self.G, self.g_slots)

    # -- checkpoint / resume (the reference only ever saves the generator: train_gan3.py:364-368) -----------
    def save_state(self, path):
        """everything a resumed run needs: G and D weights (reference layer names), both models' Adam moments and the
        shared iteration counter, in one .safetensors archive"""
        from safetensors.numpy import save_file
        if self._t_dev is not None:
            self.opt.iterations = int(self._t_dev[0].item())
        out = {"G/" + k: v for k, v in self.G.get_weights_dict().items()}
        out.update({"D/" + k: v for k, v in self.D.get_weights_dict().items()})
        for tag, s in (("G", self.g_slots), ("D", self.d_slots)):
            out["adam/%s/m" % tag] = s.m.cpu().numpy()
            out["adam/%s/v" % tag] = s.v.cpu().numpy()
        out["adam/iterations"] = np.asarray([self.opt.iterations], np.int64)
        save_file(out, path, metadata={"format": "vcg-amd-trainer", "wiring": self.wiring})

    def load_state(self, path):
        from safetensors.numpy import load_file
        d = load_file(path)
        self.G.set_weights_dict({k[2:]: v for k, v in d.items() if k.startswith("G/")})
        self.D.set_weights_dict({k[2:]: v for k, v in d.items() if k.startswith("D/")})
        for tag, s in (("G", self.g_slots), ("D", self.d_slots)):
            s.m.copy_(torch.from_numpy(d["adam/%s/m" % tag]))
            s.v.copy_(torch.from_numpy(d["adam/%s/v" % tag]))
        self.opt.iterations = int(d["adam/iterations"][0])
        if self._t_dev is not None:
            self._t_dev[0] = self.opt.iterations

    # -- loss read-back ---------------------------------------------------------------------------------
    def read_losses(self):
        """(loss_disc, loss_gan, loss_gan_gen, loss_gan_disc) of the last step as python floats: ONE device->host copy
        of the packed buffer; under DP one 4-float all-reduce in front of it (rank-local means over equal shards average
        to the global batch's; already-global relativistic values are unchanged by the average)."""
        v = self._lossbuf
        if self.pg is not None:
            from . import _dist
            v = _dist.allreduce_sum(v.clone(), self.pg)
        a, b, c, adv = (x / self.world for x in v.tolist())
        wa, wb = self._loss_w
        return wa * a - wb * b, self.cw * c + self.dw * adv, c, adv

    def disc_loss_value(self):
        return self.read_losses()[0]

    def gan_loss_values(self):
        return list(self.read_losses()[1:])

    def train_step(self, lr, hr, read_losses=True, fused=None):
        """One loop-body iteration (train_gan3.py:346-354) without host round trips between the three
        calls; lr/hr are device NCHW tensors.  Returns (loss_disc, loss_gan, loss_gan_gen, loss_gan_disc).
        fused=True/False switches ``self.fused`` (see __init__) for this and later steps."""
        if fused is not None:
            self.fused = bool(fused)
        for _, fn in self._plan(lr, hr):
            fn()
        return self.read_losses() if read_losses else None

    # -- hipGraph: the whole loop body as ONE graph launch (more under DP, cut at its collectives) ----------
    def capture_train_step(self, lr, hr, fused=None):
        """Record the loop body for these (static-shape) device batches (fused: see train_step; recorded as set here).  Later ``train_step_graph(lr, hr)`` copies the
        new frames into the captured input buffers and replays: ~700 kernel launches become one graph launch (no
        per-kernel host cost, no launch gaps) -- for every loss the reference offers, the relativistic ones included.

        Under data parallelism the plan is cut at its collectives into several graphs (Wasserstein: four --
            A: predict, D forward x2, D backward x2        -> all-reduce(D gradient bucket) issued, not waited for
            B: G forward (training mode)                   -> wait for the bucket (it travelled under B)
            C: Adam(D), D forward, D/G backward            -> all-reduce(G gradient bucket)
            D: Adam(G)
        relativistic: six, with the two 2-float mean all-reduces in addition) and the RCCL calls are issued eagerly
        between the replays: nothing about RCCL graph capture is assumed.

        Every lazily cached derived weight (per-tap transposed kernels, packed bf16 copies) is invalidated before the
        recording, so its derivation is part of the graph at each point of use: weights changed from outside between two
        replays (load_state, set_weights_dict, a broadcast) are picked up by the next replay."""
        rt = self.rt
        if fused is not None:
            self.fused = bool(fused)
        if self._t_dev is None:
            self._t_dev = torch.tensor([self.opt.iterations, 0], dtype=torch.int32, device=rt.device)    # {t, lr_t scratch}
        self._g_lr, self._g_hr = lr.clone(), hr.clone()
        self.train_step(self._g_lr, self._g_hr)        # eager warm-up with the device-side counter (lazy buffers exist)
        torch.cuda.synchronize()
        it0 = self.opt.iterations
        for m in (self.G, self.D) + ((self.content_kind.model,) if isinstance(self.content_kind, _VggLossBase) else ()):
            m.refresh()
        plan = self._plan(self._g_lr, self._g_hr)
        items, run = [], []
        for is_coll, fn in plan + [(True, None)]:
            if not is_coll:
                run.append(fn)
                continue
            if run:
                items.append(("graph", list(run)))
                run = []
            if fn is not None:
                items.append(("coll", fn))
        graphs, pool, seq = [], None, []
        # capture_error_mode="thread_local": the RCCL watchdog thread may poll its events while this thread records
        for kind, what in items:
            if kind == "coll":
                seq.append(what)
                continue
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                for fn in what:
                    fn()
            pool = g.pool()
            graphs.append(g)
            seq.append(g)
        # capture only records: undo the host-side counter advance of the recording pass
        self.opt.iterations = it0
        self._graph, self._seq = graphs, seq
        return graphs[0] if len(graphs) == 1 else graphs

    def train_step_graph(self, lr=None, hr=None, read_losses=True):
        """Replay the captured loop body (optionally on new frames of the captured shape)."""
        if self._graph is None:
            raise RuntimeError("call capture_train_step first")
        if lr is not None:
            self._g_lr.copy_(lr)
        if hr is not None:
            self._g_hr.copy_(hr)
        for item in self._seq:
            if isinstance(item, torch.cuda.CUDAGraph):
                item.replay()
            else:
                item()
        self.opt.iterations += 2
        return self.read_losses() if read_losses else None


class TrainingModel:
    """One of the three compiled models returned by make_and_compile_gan[2]."""

    def __init__(self, trainer, kind, name):
        self.trainer, self.kind, self.name = trainer, kind, name
        self.rt = trainer.rt

    def predict(self, x, batch_size=32):
        t = self.trainer
        if self.kind == "gen":
            return t.G.predict(x, batch_size)
        if self.kind == "disc":
            xs = x if isinstance(x, (list, tuple)) else [x]
            return t.D.predict(xs[-1], batch_size)
        xs = x if isinstance(x, (list, tuple)) else [x]
        g = t.G.predict(xs[0], batch_size)
        return [g, t.D.predict(g, batch_size)]

    def train_on_batch(self, x, y=None):
        t = self.trainer
        rt = self.rt
        if self.kind == "disc":
            if t.wiring == "gan2":
                hr, fake = x
                t.disc_step(E.to_device_nchw(rt, hr), E.to_device_nchw(rt, fake))
            else:
                xs = E.to_device_nchw(rt, x)
                yv = np.asarray(y).reshape(-1)
                nb = int((yv > 0).sum())
                if not (np.all(yv[:nb] == 1) and np.all(yv[nb:] == -1)):
                    raise NotImplementedError("v1 discriminator targets must be (+1...,-1...) as in train_gan.py:309-313")
                t.disc_step(xs[:nb], xs[nb:])
            return t.disc_loss_value()
        if self.kind == "gan":
            if t.wiring == "gan2":
                lr, hr = x
            else:
                lr, hr = x, (y[0] if isinstance(y, (list, tuple)) else y)
                # wasserstein_loss = mean(y_true * y_pred) (model.py:159-160): sign and scale of the adversarial term come from
                # the caller's targets; the reference passes +1 (train_gan.py:310,317) and that is what the device step computes
                if isinstance(y, (list, tuple)) and len(y) > 1 and not np.all(np.asarray(y[1]) == 1):
                    raise NotImplementedError("v1 gan_train targets for the discriminator output must be +1 (train_gan.py:310-317)")
            t.gan_step(E.to_device_nchw(rt, lr), E.to_device_nchw(rt, hr))
            return t.gan_loss_values()
        raise NotImplementedError("gen_train.train_on_batch is never called by the reference's GAN loop; "
                                  "use compile_training_model for generator-only training")

    def train_step(self, lr, hr):
        """fused loop body (extension): NHWC arrays in, four python floats out."""
        rt = self.rt
        return self.trainer.train_step(E.to_device_nchw(rt, lr), E.to_device_nchw(rt, hr))


def _make(generator, discriminator, wiring, content_loss, content_loss_weight, losses, discriminator_loss_weight,
          optimizer, process_group, fused_step=False):
    trainer = GanTrainer(generator, discriminator, wiring, _content_kind(content_loss), content_loss_weight, losses,
                         discriminator_loss_weight, optimizer, process_group)
    trainer.fused = bool(fused_step)
    discriminator.trainable = False     # state the reference leaves behind (model.py:1040,1101)
    return (TrainingModel(trainer, "gen", "generator_training_model"),
            TrainingModel(trainer, "disc", "discriminator_training_model"),
            TrainingModel(trainer, "gan", "gan_training_model"))


def make_and_compile_gan(generator, discriminator, input_shape, output_shape, content_loss, content_loss_weight,
                         discriminator_loss, discriminator_loss_weight, optimizer=_DEFAULT_ADAM, process_group=None,
                         fused_step=False):
    """model.py:1017-1051 (v1 wiring; discriminator_loss must be ``wasserstein_loss``).  ``process_group`` (data parallelism) and
    ``fused_step`` (GanTrainer.fused: the train_step()/hipGraph fast path without the separate predict pass; the three Keras-style
    calls are unaffected) are extensions."""
    if discriminator_loss is not wasserstein_loss:
        raise NotImplementedError("v1 wiring is implemented for discriminator_loss=wasserstein_loss (train_gan.py:267)")
    _check_shapes(generator, discriminator, input_shape, output_shape)
    return _make(generator, discriminator, "v1", content_loss, content_loss_weight, None, discriminator_loss_weight,
                 optimizer, process_group, fused_step)


def make_and_compile_gan2(generator, discriminator, input_shape, output_shape, content_loss, content_loss_weight,
                          discriminator_losses, discriminator_loss_weight, optimizer=_DEFAULT_ADAM, process_group=None,
                          fused_step=False):
    """model.py:1057-1125.  ``discriminator_losses`` is a zero-argument factory, called twice like the
    reference does (:1085,:1110).  ``process_group`` / ``fused_step``: extensions, see make_and_compile_gan."""
    _check_shapes(generator, discriminator, input_shape, output_shape)
    losses = discriminator_losses()
    discriminator_losses()
    if not isinstance(losses, GanLosses):
        raise TypeError("discriminator_losses() must return a GanLosses instance")
    return _make(generator, discriminator, "gan2", content_loss, content_loss_weight, losses,
                 discriminator_loss_weight, optimizer, process_group, fused_step)


def _check_shapes(generator, discriminator, input_shape, output_shape):
    if tuple(input_shape) != tuple(generator._in_shape):
        raise ValueError("input_shape %s does not match the generator's %s" % (tuple(input_shape), generator._in_shape))
    if tuple(output_shape) != tuple(discriminator._in_shape):
        raise ValueError("output_shape %s does not match the discriminator's %s" % (tuple(output_shape), discriminator._in_shape))


class _GeneratorOnlyTrainer:
    def __init__(self, upscaler, kind, optimizer):
        self.G, self.kind, self.opt = upscaler, kind, optimizer
        self.slots = _Slots(upscaler)


class GeneratorTrainingModel:
    """compile_training_model(upscaler, loss) (model.py:1130-1137): generator-only training with a pixel or
    VGG19 perceptual loss."""

    def __init__(self, upscaler, loss, optimizer):
        self.G, self.kind, self.opt = upscaler, _content_kind(loss), optimizer
        self.slots = _Slots(upscaler)
        self.rt = upscaler.rt
        self.name = "upscaler_training_model"

    def predict(self, x, batch_size=32):
        return self.G.predict(x, batch_size)

    def train_on_batch(self, x, y):
        rt, G = self.rt, self.G
        lr, hr = E.to_device_nchw(rt, x), E.to_device_nchw(rt, y)
        fake, tape = G.forward(lr, True)
        val, dfake = _content_loss_and_grad(rt, self.kind, 1.0, fake, hr)
        G.backward(tape, dfake, 0)
        ps = G.ps
        L.check(rt.lib.vcg_adam_keras_multi(ps.params.data_ptr(), ps.grads.data_ptr(), self.slots.m.data_ptr(),
                                            self.slots.v.data_ptr(), ps.n_trainable, float(self.opt.lr_t()),
                                            float(self.opt.beta_1), float(self.opt.beta_2), float(self.opt.epsilon), 1.0, rt.stream),
                "vcg_adam_keras_multi")
        self.opt.iterations += 1
        G.refresh()
        return float(val.item())


def compile_training_model(upscaler, loss, optimizer=_DEFAULT_ADAM):
    """model.py:1130-1137"""
    return GeneratorTrainingModel(upscaler, loss, optimizer)


# =================================================================================================
# functional block API (model.py:15-27, 63-75) -- see _graph.py
# =================================================================================================
from ._graph import (Input, activation, add, atanh_scaled, batch_norm, batch_norm_prelu, build_model, concatenate, conv2d,  # noqa: E402,F401
                     conv2d_transpose, cropping2d, downsampling_block, dropout, leaky_relu, make_generator_cyclegan, make_upscaler_attention,
                     make_upscaler_orig_functional, multiply_sigmoid, prelu, residual_block, residual_block_attention, resize_images,
                     upsampling_block, upsampling_block_attention)
# the other generators train_gan3.py offers behind -gm (model.py:332-363, 505-827) -- see _generators.py
from ._generators import (concatenate_layers, downsampling_unetish_block, find_crop_shape, inception_mini_resblock,  # noqa: E402,F401
                          inception_resblock_2path, inception_resblock_3path, make_upscaler_incep_resnet, make_upscaler_skip_con,
                          make_upscaler_unetish, make_upscaler_unetish_add, make_upscaler_unetish_complex, same_size_unetish_block,
                          sum_layers, upsampling_unetish_block)
