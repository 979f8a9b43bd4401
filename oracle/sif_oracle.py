"""CPU oracle for the SIF-CNN-SR hot path (ModelB_2 forward/backward + SIF loss terms + Adam).

TEST INFRASTRUCTURE ONLY.  This file is a plain-PyTorch, CPU, fp32 restatement of the reference's
algorithm.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it; the product path (the ``*_amd`` package) never does and fails loudly without its HIP
library.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference itself
(``/root/reference/model.py`` and the hot-path functions of ``/root/reference/utils.py``) in the
build container, checks every function below against it on seeded inputs, and commits the resulting
vectors under ``tests/golden/``; ``tests/test_oracle_golden.py`` re-checks this file against those
vectors on any machine.

Rows next to the hot path (SURVEY.md §8 f), same file, pin status per row:
  * Fourier-domain evaluation and the scale-invariance baseline's data transforms: PINNED by import
    (``tests/golden/make_golden_fourier.py`` / ``make_golden_si.py`` run the reference's own functions);
  * tile pipeline: the bicubic x4 of ``cv2.resize(INTER_CUBIC)`` is restated with ``F.interpolate`` and a
    written-out cubic convolution -- OpenCV is not installed, parity with cv2 itself is UNPINNED;
  * PSNR / SSIM: numpy/scipy restatement of scikit-image 0.22 -- scikit-image is not installed, UNPINNED;
  * ``BF16_CONVS``: emulation of the build's bf16-operand mode (BASELINE.json config 5); the reference has no
    mixed-precision code, so there is nothing to pin against.

Every function cites the reference file:line it restates (paths relative to /root/reference).
The restatement is *functional*: parameters and buffers live in plain dicts keyed by the
reference's ``state_dict`` names, so the 104-key layout (SURVEY.md §8 b) is the oracle's own
data model.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5        # nn.BatchNorm2d default, model.py:136
BN_MOMENTUM = 0.1    # nn.BatchNorm2d default, model.py:136

# ----------------------------------------------------------------------------------------------
# Layer table: (state_dict prefix of the conv, prefix of its BatchNorm, Cin, Cout)
# Order == reference state_dict order (model.py:596-605 construction order).
# ----------------------------------------------------------------------------------------------
CONV_BN_LAYERS = [
    ("inbloc.bloc.0", "inbloc.bloc.1", 2, 16),
    ("inbloc.bloc.3", "inbloc.bloc.4", 16, 16),
    ("db1.resblock.doubleconv.bloc.0", "db1.resblock.doubleconv.bloc.1", 16, 16),
    ("db1.resblock.doubleconv.bloc.3", "db1.resblock.doubleconv.bloc.4", 16, 16),
    ("db1.lastconv.0", "db1.lastconv.1", 16, 32),
    ("db2.resblock.doubleconv.bloc.0", "db2.resblock.doubleconv.bloc.1", 32, 32),
    ("db2.resblock.doubleconv.bloc.3", "db2.resblock.doubleconv.bloc.4", 32, 32),
    ("db2.lastconv.0", "db2.lastconv.1", 32, 64),
    ("db3.resblock.doubleconv.bloc.0", "db3.resblock.doubleconv.bloc.1", 64, 64),
    ("db3.resblock.doubleconv.bloc.3", "db3.resblock.doubleconv.bloc.4", 64, 64),
    ("db3.lastconv.0", "db3.lastconv.1", 64, 64),
    ("ub1.convbloc.bloc.0", "ub1.convbloc.bloc.1", 128, 64),
    ("ub1.convbloc.bloc.3", "ub1.convbloc.bloc.4", 64, 32),
    ("ub2.convbloc.bloc.0", "ub2.convbloc.bloc.1", 64, 32),
    ("ub2.convbloc.bloc.3", "ub2.convbloc.bloc.4", 32, 16),
    ("ub3.convbloc.bloc.0", "ub3.convbloc.bloc.1", 32, 16),
    ("ub3.convbloc.bloc.3", "ub3.convbloc.bloc.4", 16, 16),
]
OUTLAY = ("outlay", 16, 1)


def state_dict_spec():
    """(name, shape, dtype) for the 104 state_dict entries, in reference order (SURVEY.md §8 b)."""
    spec = []
    for conv, bn, cin, cout in CONV_BN_LAYERS:
        spec.append((conv + ".weight", (cout, cin, 3, 3), torch.float32))
        spec.append((bn + ".weight", (cout,), torch.float32))
        spec.append((bn + ".bias", (cout,), torch.float32))
        spec.append((bn + ".running_mean", (cout,), torch.float32))
        spec.append((bn + ".running_var", (cout,), torch.float32))
        spec.append((bn + ".num_batches_tracked", (), torch.int64))
    spec.append(("outlay.weight", (1, 16, 3, 3), torch.float32))
    spec.append(("outlay.bias", (1,), torch.float32))
    return spec


def param_names():
    """The 53 trainable tensors, in ``model.parameters()`` order."""
    return [n for n, _, _ in state_dict_spec()
            if not n.endswith(("running_mean", "running_var", "num_batches_tracked"))]


def synthetic_state(seed: int) -> "OrderedDict[str, torch.Tensor]":
    """Formula-generated weights from a bit-stable numpy stream (any machine regenerates them).

    Conv weights ~ N(0, 2/(9 Cin)); BN gamma ~ U(0.5,1.5); beta ~ N(0,0.1); running_mean ~ N(0,0.1);
    running_var ~ U(0.5,1.5) so that eval mode exercises non-trivial statistics.
    """
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    for name, shape, dtype in state_dict_spec():
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.zeros((), dtype=torch.int64)
        elif name.endswith("running_var"):
            sd[name] = torch.from_numpy(rs.uniform(0.5, 1.5, shape).astype(np.float32))
        elif name.endswith("running_mean"):
            sd[name] = torch.from_numpy((0.1 * rs.standard_normal(shape)).astype(np.float32))
        elif len(shape) == 4:
            fan_in = shape[1] * 9
            sd[name] = torch.from_numpy(
                (rs.standard_normal(shape) * math.sqrt(2.0 / fan_in)).astype(np.float32))
        elif name.endswith("bias") and name.startswith("outlay"):
            sd[name] = torch.from_numpy((0.1 * rs.standard_normal(shape)).astype(np.float32))
        elif name.endswith(".weight"):           # BN gamma
            sd[name] = torch.from_numpy(rs.uniform(0.5, 1.5, shape).astype(np.float32))
        else:                                     # BN beta
            sd[name] = torch.from_numpy((0.1 * rs.standard_normal(shape)).astype(np.float32))
    return sd


def matched_state(stats: dict, seed: int) -> "OrderedDict[str, torch.Tensor]":
    """A synthetic state_dict whose every tensor has the first two moments and the range of a TRAINED checkpoint of the
    reference (``stats[key] = [mean, std, min, max]``, measured by tests/golden/make_golden_real.py on
    models/modelB_2609 / modelB_1009 -- the weights themselves must not travel, their 104 x 4 summary numbers are data):
    N(mean, std) from a seeded numpy stream, clipped to [min, max]; ``num_batches_tracked`` takes the stored count.  Puts the
    parity checks at the reference's own operating point (BatchNorm gains near 1 with trained spreads, running variances
    well away from 1, conv weights 2-3x smaller than the He init of ``synthetic_state``)."""
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    for key, shape, dtype in state_dict_spec():
        mean, std, lo, hi = stats[key]
        if key.endswith("num_batches_tracked"):
            sd[key] = torch.tensor(int(round(mean)), dtype=torch.int64)
            continue
        v = mean + std * rs.standard_normal(tuple(shape)).astype(np.float64)
        sd[key] = torch.from_numpy(np.clip(v, lo, hi).astype(np.float32))
    return sd


def synthetic_batch(seed: int, batch: int, hr: int = 256):
    """Seeded (lst, lst_up, ndvi) with the ModisDatasetB.__getitem__ shapes (dataset.py:101-142).

    BASELINE.md §3: lst ~ N(0,1) (B,1,hr/4,hr/4); lst_up = bicubic x4 of lst
    (F.interpolate, align_corners=False); ndvi ~ N(0,1) clipped to +-3.  numpy stream => bit-stable.
    """
    rs = np.random.RandomState(seed)
    lr = hr // 4
    lst = torch.from_numpy(rs.standard_normal((batch, 1, lr, lr)).astype(np.float32))
    ndvi = torch.from_numpy(np.clip(rs.standard_normal((batch, 1, hr, hr)), -3, 3).astype(np.float32))
    lst_up = F.interpolate(lst, scale_factor=4, mode="bicubic", align_corners=False)
    return lst, lst_up, ndvi


# ----------------------------------------------------------------------------------------------
# Model (model.py)
# ----------------------------------------------------------------------------------------------
def _conv3x3_rep(x, w, b=None):
    """nn.Conv2d(k=3, stride=1, padding=1, padding_mode='replicate') -- model.py:135,138,507,605."""
    return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="replicate"), w, b)


# Test hook: {bn prefix: bool mask (B,C,H,W)}.  When set, relu(z) is evaluated as z*mask, i.e. on the
# linear region an implementation under test actually took.  Two fp32 implementations disagree on the
# sign of a handful of |z| < 1e-6 pre-activations; each such flip moves a gradient by ~1/sqrt(N) of
# its norm, so gradient parity is only meaningful at equal masks.
RELU_MASKS = None
# Test hook: when a dict, every _bn_relu call stores the mask it took, {bn prefix: (z > 0)}.
RECORD_MASKS = None


def _bn_relu(x, sd, bn, training):
    """nn.BatchNorm2d (+ running-stat update in training) then the shared nn.ReLU -- model.py:136-137."""
    y = F.batch_norm(x, sd[bn + ".running_mean"], sd[bn + ".running_var"],
                     sd[bn + ".weight"], sd[bn + ".bias"],
                     training=training, momentum=BN_MOMENTUM, eps=BN_EPS)
    if training:
        sd[bn + ".num_batches_tracked"] += 1
    if RECORD_MASKS is not None:
        RECORD_MASKS[bn] = (y > 0).detach()
    if RELU_MASKS is not None:
        return y * RELU_MASKS[bn].to(y.dtype)
    return F.relu(y)


# BASELINE.json config 5 ("bf16 mixed precision, MFMA-bf16 conv tiles") as the build implements it: the operands of
# the sixteen 3x3 convs that run on the matrix cores (everything but inbloc.bloc.0 and outlay) are rounded to bf16
# (round-to-nearest-even), products accumulate in fp32.  Backward: the input gradient contracts bf16(dy) with bf16(W); the
# weight gradient contracts bf16(x) with bf16(dy).  The reference has no mixed-precision code at all;
# torch.autocast(bfloat16) is the looser yardstick the tests also report.
BF16_CONVS = False
# ... and since round 3 every activation-like tensor the build STORES is bf16 as well ("bf16
# activations"): raw conv outputs (BatchNorm then sees the rounded values), the pooled, residual-sum and upsampled tensors, and
# the gradients with respect to all of these.  Emulated with a straight-through rounding at each of those points whose backward
# rounds the gradient that passes.  Weights, BatchNorm statistics / coefficients, the model's input and output stay fp32.
BF16_STORE = False


def _rbf(t):
    return t.to(torch.bfloat16).to(t.dtype)


class _StoreBf16(torch.autograd.Function):
    """y = bf16(x) as stored; the gradient w.r.t. a stored tensor is itself a stored (bf16) tensor."""

    @staticmethod
    def forward(ctx, x):
        return _rbf(x)

    @staticmethod
    def backward(ctx, g):
        return _rbf(g)


def _st(x):
    return _StoreBf16.apply(x) if BF16_STORE else x


class _ConvBf16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return _conv3x3_rep(_rbf(x), _rbf(w))

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        with torch.enable_grad():
            xa = x.detach().requires_grad_(True)
            (dx,) = torch.autograd.grad(_conv3x3_rep(xa, _rbf(w).detach()), xa, _rbf(dy))
            wa = w.detach().requires_grad_(True)
            (dw,) = torch.autograd.grad(_conv3x3_rep(_rbf(x).detach(), wa), wa, _rbf(dy))
        return dx, dw


def _conv_bn_relu(x, sd, conv, bn, training):
    w = sd[conv + ".weight"]
    y = _ConvBf16.apply(x, w) if (BF16_CONVS and w.shape[1] >= 16) else _conv3x3_rep(x, w)
    return _bn_relu(_st(y), sd, bn, training)


def _double_conv(x, sd, prefix, training):
    """DoubleConvolution.forward -- model.py:134-141,159."""
    x = _conv_bn_relu(x, sd, prefix + ".0", prefix + ".1", training)
    return _conv_bn_relu(x, sd, prefix + ".3", prefix + ".4", training)


def _down_block_pool(x, sd, name, training):
    """DownBlock_pool.forward -- model.py:504,528-531 with ResidualConnection.forward :311-312."""
    x = _st(F.avg_pool2d(x, kernel_size=2, stride=2))
    x = _st(x + _double_conv(x, sd, name + ".resblock.doubleconv.bloc", training))
    return _conv_bn_relu(x, sd, name + ".lastconv.0", name + ".lastconv.1", training)


def _up_block(x, skip, sd, name, training):
    """UpBlock.forward, bilinear branch -- model.py:205-208,235-248 (F.pad is a no-op here)."""
    x = _st(F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True))
    assert x.shape[-2:] == skip.shape[-2:]
    x = torch.cat([x, skip], dim=1)
    return _double_conv(x, sd, name + ".convbloc.bloc", training)


def modelb2_forward(sd, x, training: bool):
    """ModelB_2.forward -- model.py:608-645.  ``sd`` is mutated (BN buffers) when ``training``."""
    s0 = _double_conv(x, sd, "inbloc.bloc", training)
    s1 = _down_block_pool(s0, sd, "db1", training)
    s2 = _down_block_pool(s1, sd, "db2", training)
    s3 = _down_block_pool(s2, sd, "db3", training)
    u = _up_block(s3, s2, sd, "ub1", training)
    u = _up_block(u, s1, sd, "ub2", training)
    u = _up_block(u, s0, sd, "ub3", training)
    return _conv3x3_rep(u, sd["outlay.weight"], sd["outlay.bias"])


# ----------------------------------------------------------------------------------------------
# SIF loss operators (utils.py, train_model_B_*.py)
# ----------------------------------------------------------------------------------------------
def generate_psf_kernel(res: float, mtf_res: float, mtf_fc: float, half_kernel_width=None):
    """utils.py:1615-1639 -- float64 Gaussian PSF, normalised, cast to float32."""
    fc = 0.5 / mtf_res
    sigma = math.sqrt(-math.log(mtf_fc) / 2) / (math.pi * fc)
    if half_kernel_width is None:
        half_kernel_width = int(math.ceil(mtf_res / res))
    h = half_kernel_width
    k = np.zeros((2 * h + 1, 2 * h + 1))
    for i in range(h + 1):
        for j in range(h + 1):
            dist = res * math.sqrt(i ** 2 + j ** 2)
            psf = np.exp(-(dist * dist) / (2 * sigma * sigma)) / (sigma * math.sqrt(2 * math.pi))
            k[h - i, h - j] = psf
            k[h - i, h + j] = psf
            k[h + i, h + j] = psf
            k[h + i, h - j] = psf
    k = k / np.sum(k)
    return k.astype(np.float32)


def psf_taps_1d(mtf: float, factor: float = 4.0):
    """Separable factor of generate_psf_kernel(1, factor, mtf): 9 float64 taps summing to 1.

    The 2-D kernel is exp(-(i^2+j^2)/2s^2)/Z = g_i g_j with g normalised (rank-1 to ~1e-8 after
    the float32 cast); the HIP path uses these taps (cast to fp32) separably.
    """
    fc = 0.5 / factor
    sigma = math.sqrt(-math.log(mtf) / 2) / (math.pi * fc)
    h = int(math.ceil(factor))
    g = np.exp(-(np.arange(-h, h + 1, dtype=np.float64) ** 2) / (2 * sigma * sigma))
    return g / g.sum()


def _psf_blur_padded(data, mtf, factor=4.0):
    """Common head of utils.py:1683-1697 and :1844-1858: reflect pad hw, depthwise 9x9, 'same'."""
    k = torch.tensor(generate_psf_kernel(1.0, factor, mtf, None), dtype=data.dtype)
    hw = int((k.shape[-1] - 1) / 2)
    data = F.pad(data, (hw, hw, hw, hw), mode="reflect")
    data = F.conv2d(data, k[None, None].expand(data.shape[1], -1, -1, -1),
                    groups=data.shape[1], padding="same")
    return data, hw


def downscale_LST_SR_to_LR(data, factor: float = 4, mtf: float = 0.1):
    """utils.py:1671-1706, deci_type='bic': blur on the reflect-padded image, bicubic /4, crop."""
    data, hw = _psf_blur_padded(data, mtf, factor)
    data = F.interpolate(data, scale_factor=1 / factor, mode="bicubic")
    sl = int(hw / factor)
    return data[:, :, sl:data.shape[-2] - sl, sl:data.shape[-1] - sl]


def get_output_ftm(data, factor: float = 4, mtf: float = 0.1):
    """utils.py:1833-1860: reflect-border Gaussian low-pass, same size as the input."""
    data, hw = _psf_blur_padded(data, mtf, factor)
    return data[:, :, hw:data.shape[-2] - hw, hw:data.shape[-1] - hw]


# train_model_B_predef_filters.py:38-42 (N-S, E-W and the two diagonals)
SOBEL_FILTERS = [[[1, 2, 1], [0, 0, 0], [-1, -2, -1]],
                 [[1, 0, -1], [2, 0, -2], [1, 0, -1]],
                 [[2, 1, 0], [1, 0, -1], [0, -1, -2]],
                 [[0, 1, 2], [-1, 0, 1], [-2, -1, 0]]]


def sobel_bank(x):
    """train_model_B_predef_filters.py:120-128: F.conv2d(x, (4,1,3,3), padding='same') (zero pad)."""
    f = torch.tensor(SOBEL_FILTERS, dtype=x.dtype)[:, None]
    return F.conv2d(x, f, padding="same")


def huber(a, b):
    """nn.HuberLoss(reduction='mean', delta=1.0) -- train_model_B_gradFTM.py:454."""
    return F.huber_loss(a, b, reduction="mean", delta=1.0)


def sr2_loss(sr, lst, ndvi, mean, std, alpha, gamma):
    """train_model_B_gradFTM.py:99-117 -> (ds_loss, percep_loss, loss)."""
    sr_un = sr * std + mean
    down = downscale_LST_SR_to_LR(sr_un)
    down = (down - mean) / std
    ds = huber(down, lst)
    g_lst = sr - get_output_ftm(sr, mtf=0.25)
    g_ndvi = ndvi - get_output_ftm(ndvi, mtf=0.25)
    pl = huber(g_lst, gamma * g_ndvi)
    return ds, pl, alpha * ds + (1 - alpha) * pl


def sr1_loss(sr, lst, ndvi, mean, std, alpha, gamma):
    """train_model_B_predef_filters.py:111-133 -> (ds_loss, percep_loss, loss)."""
    sr_un = sr * std + mean
    down = downscale_LST_SR_to_LR(sr_un)
    down = (down - mean) / std
    ds = huber(down, lst)
    pl = huber(sobel_bank(sr), gamma * sobel_bank(ndvi))
    return ds, pl, alpha * ds + (1 - alpha) * pl


def si_loss(sr, lst, ndvi, mean, std, alpha, gamma):
    """train_model_B_scale_invariance.py:98: plain Huber against a same-resolution target (passed as ``ndvi``
    here, so the three-tensor batch signature of the other two steps is kept) -> (loss, 0, loss)."""
    ds = huber(sr, ndvi)
    return ds, torch.zeros((), dtype=sr.dtype), ds


LOSSES = {"sr2": sr2_loss, "sr1": sr1_loss, "si": si_loss}


# ----------------------------------------------------------------------------------------------
# Train step (train_model_B_gradFTM.py:86-121 / train_model_B_predef_filters.py:98-137)
# ----------------------------------------------------------------------------------------------
def forward_backward(sd, lst, lst_up, ndvi, mean, std, alpha, gamma, kind="sr2"):
    """One fwd + loss + backward in training mode.  Returns (sr, (ds, pl, loss), grads dict).

    ``sd`` BN buffers are updated exactly as ``model.train(); model(x)`` would.
    """
    names = param_names()
    leaves = {n: sd[n].detach().clone().requires_grad_(True) for n in names}
    work = OrderedDict((k, leaves.get(k, v)) for k, v in sd.items())
    x = torch.cat((lst_up, ndvi), dim=1)                       # :94
    sr = modelb2_forward(work, x, training=True)               # :96
    ds, pl, loss = LOSSES[kind](sr, lst, ndvi, mean, std, alpha, gamma)
    grads = torch.autograd.grad(loss, [leaves[n] for n in names])
    for k in sd:                                               # carry BN buffer updates back
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            sd[k] = work[k].detach()
    return sr.detach(), (ds.detach(), pl.detach(), loss.detach()), dict(zip(names, grads))


class AdamState:
    """torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8, weight_decay=0) restated
    (train_model_B_gradFTM.py:453); single-tensor formula of torch/optim/adam.py."""

    def __init__(self, names, lr, beta1=0.9, beta2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.t = 0
        self.m = {n: None for n in names}
        self.v = {n: None for n in names}

    def step(self, sd, grads):
        self.t += 1
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        for n, g in grads.items():
            if self.m[n] is None:
                self.m[n] = torch.zeros_like(g)
                self.v[n] = torch.zeros_like(g)
            # same op sequence as torch.optim.adam._single_tensor_adam (bit-exact on CPU)
            self.m[n].lerp_(g, 1 - self.b1)
            self.v[n].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (self.v[n].sqrt() / (bc2 ** 0.5)).add_(self.eps)
            sd[n] = sd[n].clone().addcdiv_(self.m[n], denom, value=-(self.lr / bc1))


def train_step(sd, adam, lst, lst_up, ndvi, mean, std, alpha, gamma, kind="sr2"):
    """a11 / a12 of SURVEY.md §8: fwd, loss, backward, Adam.  Returns (ds, pl, loss) floats."""
    _, losses, grads = forward_backward(sd, lst, lst_up, ndvi, mean, std, alpha, gamma, kind)
    adam.step(sd, grads)
    return tuple(float(v) for v in losses)


def predict_tiles(sd, lst_up, ndvi, mean, std):
    """predict.py:100-101: eval-mode forward, de-normalised output."""
    with torch.inference_mode():
        return modelb2_forward(sd, torch.cat((lst_up, ndvi), dim=1), training=False) * std + mean


# ----------------------------------------------------------------------------------------------
# digests used by the golden fixtures
# ----------------------------------------------------------------------------------------------
def digest(t: torch.Tensor, nsamples: int = 64):
    """Order-independent + sampled summary of a tensor (float64 accumulation)."""
    d = t.detach().double().flatten()
    n = d.numel()
    idx = torch.linspace(0, n - 1, min(nsamples, n)).long()
    return {"shape": list(t.shape), "sum": float(d.sum()), "abs_sum": float(d.abs().sum()),
            "l2": float(d.norm()), "samples": [float(v) for v in d[idx]]}


# ----------------------------------------------------------------------------------------------
# Input pipeline (dataset.py:134-142, predict.py:84-103) and train-time metrics (utils.py:548-578)
# ----------------------------------------------------------------------------------------------
def prepare_tiles(lst, ndvi, stats=None, clip_ndvi=False):
    """dataset.py:134-142 / predict.py:88-99 per tile: z-score, us.upsampling (cv2.resize INTER_CUBIC x4,
    utils.py:163-180 -- restated with F.interpolate bicubic, align_corners=False: same A = -0.75 kernel,
    half-pixel centres, edge clamp; OpenCV absent => parity with cv2 unpinned), NDVI clip + z-score, cat."""
    st = stats or {"mean_lst": 0.0, "std_lst": 1.0, "mean_ndvi": 0.0, "std_ndvi": 1.0}
    l = (lst - st["mean_lst"]) / st["std_lst"]
    lst_up = F.interpolate(l, scale_factor=4, mode="bicubic", align_corners=False)
    n = ndvi.clamp(-1, 1) if clip_ndvi else ndvi
    n = (n - st["mean_ndvi"]) / st["std_ndvi"]
    return torch.cat((lst_up, n), dim=1)


def predict_granule(sd, lst_g, ndvi_g, stats, window=64):
    """predict.py:84-103: block loop over the full 64x64 LST tiles of a granule, batch 1 per tile."""
    out = torch.zeros((ndvi_g.shape[0], ndvi_g.shape[1]), dtype=lst_g.dtype)
    for i in range(0, lst_g.shape[0], window):
        for j in range(0, lst_g.shape[1], window):
            lb = lst_g[i:i + window, j:j + window]
            if lb.shape != (window, window):
                continue
            nb = ndvi_g[4 * i:4 * (i + window), 4 * j:4 * (j + window)]
            x = prepare_tiles(lb[None, None], nb[None, None], stats, clip_ndvi=True)
            y = modelb2_forward(sd, x, training=False)
            out[4 * i:4 * (i + window), 4 * j:4 * (j + window)] = y[0, 0] * stats["std_lst"] + stats["mean_lst"]
    return out


def psnr_skimage(predictions, targets):
    """utils.py:548-552 with skimage.metrics.peak_signal_noise_ratio (scikit-image 0.22, not installed here --
    restated: float32 difference, mean of squares accumulated in float64, 10*log10(range^2 / mse))."""
    p, t = np.asarray(predictions, dtype=np.float32), np.asarray(targets, dtype=np.float32)
    rng = float(t.max() - t.min())
    vals = []
    for i in range(t.shape[0]):
        mse = np.mean((t[i, 0] - p[i, 0]) ** 2, dtype=np.float64)
        vals.append(10 * np.log10(rng ** 2 / mse))
    return float(np.mean(vals))


def ssim_skimage(predictions, targets):
    """utils.py:554-578 with skimage.metrics.structural_similarity defaults (scikit-image 0.22, restated):
    win 7, scipy.ndimage.uniform_filter (float64 accumulation, float32 result for float32 images), sample
    covariance NP/(NP-1), K1 = 0.01, K2 = 0.03, mean over the map cropped by (win-1)//2."""
    from scipy.ndimage import uniform_filter
    p, t = np.asarray(predictions, dtype=np.float32), np.asarray(targets, dtype=np.float32)
    R = np.float32(t.max() - t.min())
    C1, C2 = (np.float32(0.01) * R) ** 2, (np.float32(0.03) * R) ** 2
    cov_norm = np.float32(49.0 / 48.0)
    vals = []
    for i in range(t.shape[0]):
        im1, im2 = t[i, 0], p[i, 0]
        ux, uy = uniform_filter(im1, size=7), uniform_filter(im2, size=7)
        uxx, uyy, uxy = uniform_filter(im1 * im1, size=7), uniform_filter(im2 * im2, size=7), uniform_filter(im1 * im2, size=7)
        vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
        A1, A2 = 2 * ux * uy + C1, 2 * vxy + C2
        B1, B2 = ux ** 2 + uy ** 2 + C1, vx + vy + C2
        S = (A1 * A2) / (B1 * B2)
        vals.append(S[3:-3, 3:-3].mean(dtype=np.float64))
    return float(np.mean(vals))


# ----------------------------------------------------------------------------------------------
# Fourier-domain evaluation (compare_methods.py:312-324, utils.py:598-662)
# ----------------------------------------------------------------------------------------------
def fft2_magnitude_shifted(img):
    """compare_methods.py:312: np.fft.fftshift(np.abs(sp.fft.fft2(img))) -- float64 here."""
    return np.fft.fftshift(np.abs(np.fft.fft2(np.asarray(img, dtype=np.float64))))


def attenuation_spectrum(im):
    """utils.py:598-636 on the shifted magnitude ``im``: ring r = {r^2 < d^2 <= (r+1)^2} around
    (H//2, W//2), r < min(H//2, W//2) - 1; [f0/f0 = 1, 10*(log10(mean ring) - log10(f0)), ...]."""
    im = np.asarray(im, dtype=np.float64)
    c0, c1 = im.shape[0] // 2, im.shape[1] // 2
    ii, jj = np.meshgrid(np.arange(im.shape[0]), np.arange(im.shape[1]), indexing="ij")
    d2 = (ii - c0) ** 2 + (jj - c1) ** 2
    f0 = im[c0, c1]
    out = [f0 / f0]
    for r in range(0, min(c0 - 1, c1 - 1)):
        mask = (d2 <= (r + 1) ** 2) & ~(d2 <= r ** 2)
        out.append(10 * (np.log10(im[mask].sum() / mask.sum()) - np.log10(f0)))
    return out


def frr_fro_fru(pb, rb, xb):
    """utils.py:638-662 (pb prediction, rb ground truth, xb bicubic spectra) -> (FRR, FRO, FRU)."""
    pb, rb, xb = (np.asarray(v, dtype=np.float64) for v in (pb, rb, xb))
    pfr = np.maximum(rb - xb, 0).sum()
    t3 = np.minimum(rb, xb)
    afr = (np.maximum(np.minimum(pb, rb), np.minimum(xb, rb)) - t3).sum()
    fro = (rb - np.maximum(pb, rb)).sum() / rb.sum()
    fru = (xb - np.minimum(pb, xb)).sum() / xb.sum()
    return afr / pfr, fro, fru


# ----------------------------------------------------------------------------------------------
# Scale-invariance baseline (train_model_B_scale_invariance.py:86-103, dataset.py:240-263, utils.py:183-213,1716-1756)
# ----------------------------------------------------------------------------------------------
def downsampling_l4(img, scale=(4, 4)):
    """utils.py:183-213: norm-L4 pooling, (mean of x^4 over each scale block)^(1/4)."""
    img = img.unfold(dimension=3, size=scale[0], step=scale[0]).unfold(dimension=2, size=scale[1], step=scale[1])
    return torch.pow(torch.sum(torch.pow(img, 4), dim=(-1, -2)) / (scale[0] * scale[1]), 0.25)


def downscale_test(data, deci_type="bic"):
    """utils.py:1716-1756 on a (B,1,H,W) tensor: reflect-pad 4 (the PSF is generated but never applied),
    'bic': bicubic /4 + crop 1; 'norm-L4': crop the pad again + downsampling_l4."""
    d = F.pad(data, (4, 4, 4, 4), mode="reflect")
    if deci_type == "bic":
        return F.interpolate(d, scale_factor=0.25, mode="bicubic")[:, :, 1:-1, 1:-1]
    return downsampling_l4(d[:, :, 4:-4, 4:-4])


def scale_invariance_inputs(lst, ndvi, stats):
    """dataset.py:256-263 for a batch: normalised lst (B,1,64,64), ndvi (B,1,256,256) ->
    (lst_4km_up, ndvi_1km, lst)."""
    ndvi_1km = downscale_test(ndvi, "bic")
    lst_4km = downscale_test(lst * stats["std_lst"] + stats["mean_lst"], "norm-L4")
    up = F.interpolate(lst_4km, scale_factor=4, mode="bicubic", align_corners=False)     # us.upsampling (cv2 INTER_CUBIC)
    return (up - stats["mean_lst"]) / stats["std_lst"], ndvi_1km, lst
