"""``FlatAdam`` -- torch.optim.Adam semantics (train_model_B_gradFTM.py:453: Adam(lr), betas
(0.9, 0.999), eps 1e-8, weight_decay 0) as ONE gfx950 kernel launch over the flat parameter buffer
of ``ModelB_2`` instead of 53 per-tensor updates."""
from __future__ import annotations

import torch

from . import _lib


class FlatAdam(torch.optim.Optimizer):
    """Use as ``FlatAdam(model.parameters(), lr=...)`` -- same call shape as ``torch.optim.Adam``.

    ``model`` may be given to let the optimizer address the module's flat buffers directly; without
    it the flat buffers are recovered from the parameters' storage (they are contiguous views).
    ``grad_scale`` (set by ``distributed.allreduce_gradients``) multiplies the gradient inside the
    kernel, e.g. 1/world_size after a sum all-reduce.
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise ValueError("FlatAdam supports a single parameter group (the whole ModelB_2)")
        self.grad_scale = 1.0
        self._step = 0
        self._m = None
        self._v = None
        # capturable=True (as torch.optim.Adam's flag): the step count lives on the device, so step() can be part of a
        # captured hipGraph (train.GraphedTrainStep); ``_step`` then mirrors it only when state_dict() is taken
        self.capturable = bool(capturable)
        self._step_dev = None
        self._coef = None

    def _flat_views(self):
        ps = self.param_groups[0]["params"]
        base = ps[0]
        n = sum(p.numel() for p in ps)
        esz = 4
        off = 0
        contiguous_p = contiguous_g = True
        g0 = ps[0].grad
        for p in ps:
            if p.grad is None:
                raise _lib.SifsrError("FlatAdam.step(): a parameter has no gradient")
            if p.data_ptr() != base.data_ptr() + off * esz:
                contiguous_p = False
            if p.grad.data_ptr() != g0.data_ptr() + off * esz:
                contiguous_g = False
            off += p.numel()
        if not contiguous_p:
            raise _lib.SifsrError("parameters are not views of one flat buffer; run a forward pass on the GPU first "
                                  "(ModelB_2 flattens its parameters lazily)")
        flat_p = torch.as_strided(base.detach(), (n,), (1,), base.storage_offset())
        if contiguous_g:
            flat_g = torch.as_strided(g0, (n,), (1,), g0.storage_offset())
        else:   # gradients were accumulated/cloned by autograd: gather them (slow path, still exact)
            flat_g = torch.cat([p.grad.reshape(-1) for p in ps])
        return flat_p, flat_g, n

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g = self.param_groups[0]
        flat_p, flat_g, n = self._flat_views()
        _lib.require_gpu(flat_p, "parameters")
        if self._m is None or self._m.device != flat_p.device:
            self._m = torch.zeros(n, dtype=torch.float32, device=flat_p.device)
            self._v = torch.zeros(n, dtype=torch.float32, device=flat_p.device)
        if self.capturable:
            if self._step_dev is None or self._step_dev.device != flat_p.device:
                self._step_dev = torch.full((1,), self._step, dtype=torch.int64, device=flat_p.device)
                self._coef = torch.zeros(2, dtype=torch.float32, device=flat_p.device)
            _lib.call("sifsr_adam_flat_dev", flat_p, flat_g, self._m, self._v, n, float(g["lr"]), float(g["betas"][0]),
                      float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), self._step_dev, self._coef,
                      float(self.grad_scale), _lib.stream_ptr(flat_p.device))
            return loss
        self._step += 1
        _lib.call("sifsr_adam_flat", flat_p, flat_g, self._m, self._v, n, float(g["lr"]), float(g["betas"][0]),
                  float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), self._step, float(self.grad_scale),
                  _lib.stream_ptr(flat_p.device))
        return loss

    def state_dict(self):
        sd = super().state_dict()
        if self.capturable and self._step_dev is not None:
            self._step = int(self._step_dev.item())
        sd["flat"] = {"step": self._step, "exp_avg": self._m, "exp_avg_sq": self._v}
        return sd

    def load_state_dict(self, sd):
        flat = sd.pop("flat", None)
        super().load_state_dict(sd)
        if flat is not None:
            self._step, self._m, self._v = flat["step"], flat["exp_avg"], flat["exp_avg_sq"]
            self._step_dev = None
