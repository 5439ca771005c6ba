"""ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/keras_ops.py header).

CPU restatement of the reference's GAN train step:
  losses        upscaling/upscaler/model.py:159-261  (wasserstein_loss, GanLosses, Wasserstein/Relativistic)
  wiring v1     make_and_compile_gan    model.py:1017-1051  + loop body train_gan.py:298-317
  wiring gan2   make_and_compile_gan2   model.py:1057-1125  + loop body train_gan3.py:339-354
Gradients come from torch autograd (independent of the product's hand-written backward kernels).

Keras facts encoded here:
  * predict runs BN with moving statistics, train_on_batch with batch statistics;
  * disc_train updates D only, gan_train updates G only (D frozen but its BN still uses batch stats
    and, being frozen, does not update its moving statistics there);
  * one shared Adam() instance -> shared iteration counter, separate m/v slots per compiled model;
  * the gan2 loss closures ignore y_true / y_pred.
Ill-defined in the reference and fixed here by choice (DESIGN.md "Deviations"): when D is applied
twice inside disc_train (gan2), its BN moving statistics are updated sequentially, real batch
first, then fake batch.  They never enter the train step's arithmetic.
"""
from collections import OrderedDict

import torch

from . import keras_ops as K
from .models import is_trainable


class SharedAdam:
    """keras.optimizers.Adam() default instance shared by the compiled models (model.py:1026,1066)."""

    def __init__(self, lr=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.lr, self.beta_1, self.beta_2, self.epsilon = lr, beta_1, beta_2, epsilon
        self.iterations = 0


class _Slots:
    def __init__(self, weights, v0=0.0):
        self.m = OrderedDict((k, torch.zeros_like(v)) for k, v in weights.items() if is_trainable(k))
        self.v = OrderedDict((k, torch.full_like(v, v0)) for k, v in weights.items() if is_trainable(k))


def content_loss_value(kind, y_true, y_pred, vgg_w=None):
    """'mse' / 'mae' pixel losses, or the VGG19 perceptual losses of the reference (model.py:101-157):
    ('vgg',) = VGG_LOSS, ('vgg_mse', rate) = VGG_MSE_LOSS, ('vgg_mae', rate) = VGG_MAE_LOSS; vgg_w = VGG19 weights."""
    if kind in ("mse", "mean_squared_error"):
        return ((y_pred - y_true) ** 2).mean()
    if kind in ("mae", "mean_absolute_error"):
        return (y_pred - y_true).abs().mean()
    if isinstance(kind, tuple) and kind[0] in ("vgg", "vgg_mse", "vgg_mae"):
        from . import models as M
        ft, fp = M.vgg19_block5_conv4(vgg_w, y_true), M.vgg19_block5_conv4(vgg_w, y_pred)
        if kind[0] == "vgg":
            return ((ft - fp) ** 2).mean()                                                      # model.py:116
        if kind[0] == "vgg_mse":
            return ((ft - fp) ** 2).mean() + kind[1] * ((y_true - y_pred) ** 2).mean()          # model.py:137
        return (ft - fp).abs().mean() + kind[1] * (y_true - y_pred).abs().mean()                # model.py:157
    raise ValueError(kind)


class GanOracle:
    """Holds G and D weights (dicts of torch tensors in Keras layouts) and performs the three
    reference calls.  g_forward(w, x_nhwc, training) / d_forward(w, x_nhwc, training) return
    (output, bn_updates)."""

    def __init__(self, g_forward, g_w, d_forward, d_w, wiring="gan2", content="mse",
                 content_loss_weight=1.0, losses="wass", loss_activation="log-sigm",
                 discriminator_loss_weight=1e-5, optimizer=None, adam_v0=0.0, vgg_w=None):
        """adam_v0: initial value of Adam's second-moment slots.  Keras starts at 0, which makes the first
        updates sign-like (|step| = lr whatever the gradient) and any fp32-vs-fp64 comparison after one step
        chaotic; parity runs that look past the first update prime the slots instead (tests only)."""
        self.g_forward, self.d_forward = g_forward, d_forward
        self.g_w, self.d_w = g_w, d_w
        self.wiring, self.content = wiring, content
        self.vgg_w = vgg_w
        self.cw, self.dw = content_loss_weight, discriminator_loss_weight
        self.losses, self.loss_activation = losses, loss_activation
        self.opt = optimizer or SharedAdam()
        self.d_slots, self.g_slots = _Slots(d_w, adam_v0), _Slots(g_w, adam_v0)
        self.last_d_grads = None
        self.last_g_grads = None
        self.last_fake_train = None

    # -- helpers --------------------------------------------------------------------------------
    def _act(self, x):
        if self.losses == "rel":
            return K.head_activation(x, self.loss_activation)
        return x

    def _apply_adam(self, weights, slots, grads):
        t = self.opt.iterations + 1
        for k, g in grads.items():
            p, m, v = K.adam_keras_step(weights[k].detach(), g, slots.m[k], slots.v[k], t, self.opt.lr,
                                        self.opt.beta_1, self.opt.beta_2, self.opt.epsilon)
            weights[k] = p
            slots.m[k], slots.v[k] = m, v
        self.opt.iterations += 1

    @staticmethod
    def _leaf(weights, train):
        out = OrderedDict()
        for k, v in weights.items():
            t = v.detach().clone()
            if train and is_trainable(k):
                t.requires_grad_(True)
            out[k] = t
        return out

    @staticmethod
    def _grads(loss, leaf):
        names = [k for k, v in leaf.items() if v.requires_grad]
        gs = torch.autograd.grad(loss, [leaf[k] for k in names], allow_unused=True)
        return OrderedDict((k, g if g is not None else torch.zeros_like(leaf[k])) for k, g in zip(names, gs))

    # -- the three reference calls ----------------------------------------------------------------
    def predict(self, lr):
        """gen_train.predict(lr): generator forward, learning phase 0 (train_gan3.py:346)."""
        with torch.no_grad():
            y, _ = self.g_forward(self.g_w, lr, False)
        return y

    def disc_train_on_batch(self, hr, fake):
        """gan2: disc_train.train_on_batch([hr, fake], y) (train_gan3.py:353; losses model.py:220-224,
        244-248).  v1: train_on_batch(concat(hr, fake), concat(+1, -1)) with wasserstein_loss
        (train_gan.py:305-315, model.py:159-160)."""
        leaf = self._leaf(self.d_w, True)
        stat_upd = OrderedDict()
        if self.wiring == "gan2":
            d_real, upd_r = self.d_forward(leaf, hr, True)
            for k, v in upd_r.items():            # sequential moving-stat update: real first
                leaf[k] = v
            d_fake, upd_f = self.d_forward(leaf, fake, True)
            stat_upd.update(upd_r)
            stat_upd.update(upd_f)
            loss = self._act(d_real.mean() - d_fake.mean())
        else:
            x = torch.cat([hr, fake], 0)
            d_out, upd = self.d_forward(leaf, x, True)
            y = torch.cat([torch.ones(hr.shape[0], dtype=x.dtype), -torch.ones(fake.shape[0], dtype=x.dtype)])
            loss = (y.view(-1, *([1] * (d_out.dim() - 1))) * d_out).mean()
            stat_upd.update(upd)
        grads = self._grads(loss, leaf)
        self.last_d_grads = grads
        self._apply_adam(self.d_w, self.d_slots, grads)
        for k, v in stat_upd.items():
            self.d_w[k] = v.detach()
        return float(loss.detach())

    def gan_train_on_batch(self, lr, hr):
        """gan_train.train_on_batch([lr, hr], [hr, y]) (train_gan3.py:354, model.py:1103-1123);
        v1: gan_train.train_on_batch(lr, [hr, +1]) (train_gan.py:317, model.py:1040-1049).
        Returns [total, content, adversarial] like Keras."""
        g_leaf = self._leaf(self.g_w, True)
        d_frozen = self._leaf(self.d_w, False)
        fake, g_upd = self.g_forward(g_leaf, lr, True)
        self.last_fake_train = fake.detach()
        d_fake, _ = self.d_forward(d_frozen, fake, True)
        content = content_loss_value(self.content, hr, fake, self.vgg_w)
        if self.wiring == "gan2":
            if self.losses == "wass":
                adv = d_fake.mean()                                   # model.py:230-233
            else:
                d_real, _ = self.d_forward(d_frozen, hr, True)
                adv = self._act(d_fake.mean() - d_real.mean())        # model.py:255-259
        else:
            adv = d_fake.mean()                                       # wasserstein_loss(+1, D(G(x)))
        total = self.cw * content + self.dw * adv
        grads = self._grads(total, g_leaf)
        self.last_g_grads = grads
        self._apply_adam(self.g_w, self.g_slots, grads)
        for k, v in g_upd.items():
            self.g_w[k] = v.detach()
        return [float(total.detach()), float(content.detach()), float(adv.detach())]

    def train_step(self, lr, hr):
        """One iteration of the reference loop body (train_gan3.py:346-354 / train_gan.py:303-317)."""
        fake = self.predict(lr)
        loss_disc = self.disc_train_on_batch(hr, fake)
        loss_gan, loss_gan_gen, loss_gan_disc = self.gan_train_on_batch(lr, hr)
        return loss_disc, loss_gan, loss_gan_gen, loss_gan_disc

    def train_step_fused(self, lr, hr):
        """The documented deviation `fused=True` of the build ("three-graph semantics"; NOT in the reference): the
        loop body without its separate learning-phase-0 generator pass (train_gan3.py:346).  The critic is trained on the fakes of
        the generator's training-mode forward (batch statistics), and the generator step (model.py:1103-1123) re-uses that forward --
        the generator's weights do not change in between, so re-evaluating it here gives the same tensors."""
        with torch.no_grad():
            fake, _ = self.g_forward(self.g_w, lr, True)
        loss_disc = self.disc_train_on_batch(hr, fake)
        loss_gan, loss_gan_gen, loss_gan_disc = self.gan_train_on_batch(lr, hr)
        return loss_disc, loss_gan, loss_gan_gen, loss_gan_disc
