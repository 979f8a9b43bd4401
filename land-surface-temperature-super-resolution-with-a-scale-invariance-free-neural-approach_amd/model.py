"""``ModelB_2`` -- drop-in for the reference's ``model.ModelB_2`` (model.py:533-645) on MI355X.

Same constructor signature, attributes and 104-key ``state_dict`` (SURVEY.md §8 b); the module tree
(``inbloc.bloc.{0,1,3,4}``, ``db{1,2,3}.resblock.doubleconv.bloc.*``, ``db{1,2,3}.lastconv.{0,1}``,
``ub{1,2,3}.convbloc.bloc.*``, ``outlay``) is kept ONLY as a parameter container built from stock
``torch.nn`` modules -- their ``forward`` is never called.  ``ModelB_2.forward`` runs the whole
network as one launch schedule of hand-written gfx950 kernels behind the C ABI
(``sifsr_model_forward`` / ``sifsr_model_backward``, include/sifsr_hip.h); PyTorch provides device
memory, the stream and the autograd edge, nothing else.

All parameters are views into ONE flat fp32 buffer in ``parameters()`` order (so the optimizer and
the data-parallel gradient all-reduce see a single contiguous tensor); gradients come back the same
way.  There is no CPU compute path: the parameters must live on a ROCm device.  A CPU *input* is accepted the way
the reference's own inference scripts pass it (predict.py:96-101 and model_perf_aster_formatds.py:182-203 build CPU
tensors and call ``.numpy()`` on the result): it is staged to the parameters' device, the network runs there, and the
output comes back on the caller's device -- a transfer, not a CPU path.
"""
from __future__ import annotations

import weakref

import torch
from torch import nn

from . import _lib

_DEFAULT_DOWN = [16, 32, 64, 128]


def _conv(cin, cout, padding_mode, bias=False):
    return nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1, padding_mode=padding_mode, bias=bias)


class _Bloc(nn.Module):
    """Parameter container with the reference's ``.bloc`` Sequential (Conv, BN, ReLU) x 2 -- model.py:134-141."""

    def __init__(self, cin, cout, mid, padding_mode):
        super().__init__()
        mid = mid or cout
        self.bloc = nn.Sequential(_conv(cin, mid, padding_mode), nn.BatchNorm2d(mid), nn.ReLU(),
                                  _conv(mid, cout, padding_mode), nn.BatchNorm2d(cout), nn.ReLU())


class _Res(nn.Module):
    def __init__(self, c, padding_mode):
        super().__init__()
        self.doubleconv = _Bloc(c, c, None, padding_mode)            # model.py:289


class _Down(nn.Module):
    def __init__(self, cin, cout, padding_mode):
        super().__init__()
        self.downsampling = nn.AvgPool2d(kernel_size=2, stride=2)     # model.py:504
        self.resblock = _Res(cin, padding_mode)                       # model.py:505
        self.lastconv = nn.Sequential(_conv(cin, cout, padding_mode), nn.BatchNorm2d(cout), nn.ReLU())  # :506-509


class _Up(nn.Module):
    def __init__(self, cin, cout, padding_mode):
        super().__init__()
        self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)   # model.py:207
        self.convbloc = _Bloc(cin, cout, cin // 2, padding_mode)                     # model.py:208


# compute modes of the 3x3 convolutions (sifsr_model_*_ex): "fp32" = fp32 MFMA (default, the parity configuration);
# "bf16" = bf16 operands, fp32 accumulation (BASELINE.json config 5)
_COMPUTE_MODES = {"fp32": 0, "bf16": 1}


class _ModelFn(torch.autograd.Function):
    """One autograd node for the whole network: forward / backward are single C-ABI calls."""

    @staticmethod
    def forward(ctx, module, x, *params):
        _lib.require_gpu(x, "ModelB_2 input")
        B, C, H, W = x.shape
        if C != 2:
            raise _lib.SifsrError(f"ModelB_2 expects (B,2,H,W) = cat(lst_up, ndvi); got {tuple(x.shape)}")
        if H % 8 or W % 8 or H < 24 or W < 24:
            raise _lib.SifsrError("H and W must be multiples of 8 (three 2x poolings, as in the reference) and >= 24; the "
                                  "reference's patches are 256x256, 64x64 in the scale-invariance baseline")
        flat_p, flat_r, flat_n = module._flat_state(x.device)
        training = bool(module.training)
        need_bwd = training and any(ctx.needs_input_grad)   # grad mode is off inside forward(); this is the caller's
        if torch.cuda.is_current_stream_capturing():
            # a still-alive autograd graph of an EARLIER, un-captured step keeps the parameters'
            # AccumulateGrad nodes (bound to the stream they were made on) alive; the captured backward would make torch
            # join that stream with the capture stream and hipStreamEndCapture fails.  Refuse up front.
            prev = module._live_node() if module._live_node is not None else None
            if need_bwd and prev is not None and not getattr(prev, "sifsr_captured", False):
                raise _lib.SifsrError(
                    "stream capture of a training step while the autograd graph of an earlier (un-captured) step is still "
                    "alive -- some tensor that requires grad (typically the previous step's loss) still references it. "
                    "Drop or .detach() the outputs of the earlier steps before capturing (train.GraphedTrainStep does).")
            ctx.sifsr_captured = True
        # training-mode forward without a backward (torch.no_grad(), frozen parameters: BN recalibration) needs only the
        # forward part of the workspace; the library checks the size against what the call will touch
        ws_bytes = _lib.call("sifsr_model_workspace_bytes", B, H, W, 1 if need_bwd else 0)
        if ws_bytes == 0:
            raise _lib.SifsrError(f"unsupported shape {tuple(x.shape)}")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        sr = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
        bn = module._bn_hyper
        compute = _COMPUTE_MODES.get(getattr(module, "compute_dtype", "fp32"))
        if compute is None:
            raise _lib.SifsrError(f"compute_dtype must be one of {sorted(_COMPUTE_MODES)}")
        _lib.call("sifsr_model_forward_ex", x, sr, flat_p, flat_r, flat_n, ws, ws_bytes, B, H, W,
                  1 if training else 0, bn[0], bn[1], compute, _lib.stream_ptr(x.device))
        ctx.compute = compute
        ctx.module = module
        ctx.can_bwd = need_bwd
        ctx.training = training
        if need_bwd:
            ctx.ws = ws
            ctx.x = x
            ctx.shape = (B, H, W)
            ctx.param_version = module._flat_version
            object.__setattr__(module, "_live_node", weakref.ref(ctx))
        return sr

    @staticmethod
    def backward(ctx, dsr):
        module = ctx.module
        if not ctx.training:
            raise NotImplementedError("backward through eval-mode (running-statistics) BatchNorm is not implemented; "
                                      "call model.train() for training, or torch.inference_mode() for prediction")
        if not ctx.can_bwd:
            raise _lib.SifsrError("backward called on a forward that did not keep its workspace")
        if ctx.ws is None:
            raise _lib.SifsrError("backward called twice on the same forward (workspace already released)")
        B, H, W = ctx.shape
        dsr = dsr.contiguous()
        flat_p, _, _ = module._flat_state(dsr.device)
        grads = torch.empty_like(flat_p)
        _lib.call("sifsr_model_backward_ex", ctx.x, dsr, flat_p, grads, ctx.ws, ctx.ws.numel(), B, H, W, ctx.compute,
                  _lib.stream_ptr(dsr.device))
        ctx.ws = None
        ctx.x = None
        module._last_flat_grad = grads
        outs = [grads[o:o + n].view(s) for (o, n, s) in module._param_slices]
        return (None, None, *outs)


class ModelB_2(nn.Module):
    """Drop-in for ``model.ModelB_2`` (model.py:563): same ctor args, attributes and state_dict.

    Only the configuration the reference ships is implemented natively: ``padding_mode='replicate'``,
    ``activation='ReLU'``, ``bilinear`` true, ``downchannels=[16,32,64,128]``, ``in_channels=2``;
    anything else raises ``NotImplementedError`` at construction.
    """

    def __init__(self, in_channels, downchannels=_DEFAULT_DOWN, padding_mode="replicate", activation="ReLU",
                 bilinear=True, n_bridge_blocks=1):
        super().__init__()
        if padding_mode != "replicate":
            raise NotImplementedError("only padding_mode='replicate' has gfx950 kernels")
        if activation != "ReLU":
            raise NotImplementedError("only activation='ReLU' has gfx950 kernels")
        if not bilinear:
            raise NotImplementedError("only the bilinear UpBlock (bilinear=1) has gfx950 kernels")
        if list(downchannels) != _DEFAULT_DOWN or in_channels != 2:
            raise NotImplementedError("the native schedule is specialised for in_channels=2, downchannels=[16,32,64,128]")
        # attributes set by the reference ctor (model.py:587-592)
        self.in_channels = in_channels
        self.downchannels = downchannels
        self.padding = padding_mode
        self.activation = activation
        self.upfactor = 2 if bilinear else 1
        self.bridge = n_bridge_blocks
        # not in the reference: 'fp32' (default, the parity path) or 'bf16' = BASELINE.json config 5, bf16 MFMA operands in
        # the 3x3 convs (fp32 accumulation / storage / parameters) -- set the attribute, nothing else changes
        self.compute_dtype = "fp32"
        d, uf = downchannels, self.upfactor
        self.inbloc = _Bloc(in_channels, d[0], None, padding_mode)          # model.py:596
        self.db1 = _Down(d[0], d[1], padding_mode)                           # :597
        self.db2 = _Down(d[1], d[2], padding_mode)                           # :598
        self.db3 = _Down(d[2], d[3] // uf, padding_mode)                     # :599
        self.ub1 = _Up(d[3], d[2] // uf, padding_mode)                       # :601
        self.ub2 = _Up(d[2], d[1] // uf, padding_mode)                       # :602
        self.ub3 = _Up(d[1], d[0], padding_mode)                             # :603
        self.outlay = _conv(d[0], 1, padding_mode, bias=True)                # :605
        self._init_flat_bookkeeping()

    # ---- flat-buffer bookkeeping ---------------------------------------------------------------
    def _init_flat_bookkeeping(self):
        object.__setattr__(self, "_flat", None)          # (params, running, nbt) tensors; not registered
        object.__setattr__(self, "_flat_version", 0)
        object.__setattr__(self, "_last_flat_grad", None)
        object.__setattr__(self, "_live_node", None)     # weakref to the autograd node of the last training forward
        off, slices = 0, []
        for p in self.parameters():
            slices.append((off, p.numel(), tuple(p.shape)))
            off += p.numel()
        object.__setattr__(self, "_param_slices", slices)
        object.__setattr__(self, "_n_params", off)
        bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
        object.__setattr__(self, "_bns", bns)
        object.__setattr__(self, "_bn_hyper", (float(bns[0].momentum), float(bns[0].eps)))

    def __setstate__(self, state):                         # torch.load(full pickle): rebuild bookkeeping
        super().__setstate__(state)
        self._init_flat_bookkeeping()

    def __getstate__(self):
        state = self.__dict__.copy()
        for k in ("_flat", "_flat_version", "_last_flat_grad", "_live_node", "_param_slices", "_n_params", "_bns", "_bn_hyper"):
            state.pop(k, None)
        return state

    def _is_flat(self, device):
        f = self._flat
        if f is None or f[0].device != device:
            return False
        fp, fr, fn = f
        esz = fp.element_size()
        for p, (o, n, _) in zip(self.parameters(), self._param_slices):
            if p.device != device or p.dtype != torch.float32 or p.data_ptr() != fp.data_ptr() + o * esz:
                return False
        ro = 0
        for i, bn in enumerate(self._bns):
            c = bn.num_features
            if (bn.running_mean.data_ptr() != fr.data_ptr() + ro * 4
                    or bn.running_var.data_ptr() != fr.data_ptr() + (ro + c) * 4
                    or bn.num_batches_tracked.data_ptr() != fn.data_ptr() + i * 8):
                return False
            ro += 2 * c
        return True

    def _flat_state(self, device):
        """Return (flat_params, flat_running, flat_nbt) on ``device``, re-pointing the nn.Parameters /
        BN buffers into them if something (``.to()``, ``load_state_dict(assign=True)``...) moved them."""
        if self._is_flat(device):
            return self._flat
        params = list(self.parameters())
        for p in params:
            if p.device != device:
                raise _lib.SifsrError(f"model parameters are on {p.device} but the input is on {device}; call model.to(device)")
        with torch.no_grad():
            fp = torch.empty(self._n_params, dtype=torch.float32, device=device)
            for p, (o, n, s) in zip(params, self._param_slices):
                fp[o:o + n].copy_(p.detach().reshape(-1).float())
                p.data = fp[o:o + n].view(s)
            nrun = 2 * sum(bn.num_features for bn in self._bns)
            fr = torch.empty(nrun, dtype=torch.float32, device=device)
            fn = torch.empty(len(self._bns), dtype=torch.int64, device=device)
            ro = 0
            for i, bn in enumerate(self._bns):
                c = bn.num_features
                fr[ro:ro + c].copy_(bn.running_mean)
                fr[ro + c:ro + 2 * c].copy_(bn.running_var)
                fn[i] = bn.num_batches_tracked.to(device)
                bn.running_mean = fr[ro:ro + c]
                bn.running_var = fr[ro + c:ro + 2 * c]
                bn.num_batches_tracked = fn[i]
                ro += 2 * c
        assert nrun == _lib.call("sifsr_num_running") and self._n_params == _lib.call("sifsr_num_params")
        object.__setattr__(self, "_flat", (fp, fr, fn))
        object.__setattr__(self, "_flat_version", self._flat_version + 1)
        return self._flat

    def flat_parameters(self):
        """The single contiguous fp32 tensor all parameters alias (after the first GPU forward)."""
        dev = next(self.parameters()).device
        return self._flat_state(dev)[0]

    def flat_grad(self):
        """The contiguous gradient buffer written by the last backward (None before any)."""
        return self._last_flat_grad

    # ---- forward -----------------------------------------------------------------------------
    def forward(self, x_lst_ndvi):
        """model.py:608-645: (B,2,H,W) = cat(lst_up, ndvi) -> (B,1,H,W) super-resolved LST."""
        home = x_lst_ndvi.device
        pdev = next(self.parameters()).device
        if pdev.type != "cuda":
            raise _lib.SifsrError("ModelB_2 (MI355X build) has no CPU compute path: the parameters are on "
                                  f"{pdev}; call model.to('cuda') (predict.py: --prediction_device cuda)")
        x = x_lst_ndvi
        if home != pdev:
            # the reference's inference scripts hand over CPU tensors (predict.py:96-100) and read the result with
            # .numpy() (:101): stage the input to the parameters' device, bring the output back.  Inference only --
            # a training graph across the copy would hide the device mismatch the reference itself raises on.
            if torch.is_grad_enabled() and self.training:
                raise _lib.SifsrError(f"training-mode input on {home} but the parameters are on {pdev}: move the batch "
                                      "with .to(device) as train_model_B_gradFTM.py:89 does")
            x = x.to(pdev)
        x = x.contiguous().float()
        sr = _ModelFn.apply(self, x, *self.parameters())
        return sr if home == pdev else sr.to(home)
