"""Parity checks shared by ``tests/`` and ``__graft_entry__.smoke()``.  TEST INFRASTRUCTURE ONLY (like the rest of
``oracle/``): nothing in the product package imports this file.

``update_parity`` is the check of a whole optimisation step (SURVEY.md §8 a11 / a12,
train_model_B_gradFTM.py:117-121: ``loss.backward(); optimizer.step()`` with ``torch.optim.Adam``).
Adam's first update is ``-lr * g / (|g| + eps)`` = ``-lr * sign(g)`` wherever ``|g| >> eps``, so comparing
*parameters* with an absolute bound of a few ``lr`` cannot tell a correct update from one with the wrong sign
(2*lr away).  The comparison is therefore made on the *update* ``p_after - p_before``:

  * on the elements whose reference gradient is well above the ReLU-flip noise of DESIGN.md §6
    (``|g| >= sig_frac * max|g|`` of their tensor, for every step taken so far) the update must agree with the
    reference's in value (relative L2 <= ``max_rel_l2``: 1e-3 for the first step, 1e-1 later -- see MAX_REL_L2_LATER) and -- where the reference update itself is not a near
    cancellation of the first moment (``|upd_ref| >= SIGN_FRAC * lr``; from the second step on Adam's update is
    ``lr * m_hat / sqrt(v_hat)`` and changes sign where ``0.9 g_1 ~ -g_2``) -- in sign (>= ``min_sign_agree``);
  * the remaining elements (gradient ~ rounding noise, where either implementation's sign is arbitrary) keep the
    absolute bound ``|p - p_ref| <= 2.5 * lr * steps_taken``.
"""
from __future__ import annotations

import numpy as np
import torch

from .sif_oracle import CONV_BN_LAYERS

SIG_FRAC = 1e-2          # |g| >= 1e-2 * max|g| of the tensor: ten times the worst gradient disagreement measured between
                         # two correct fp32 implementations (~1e-3 of max|g|)
SIGN_FRAC = 0.2          # sign agreement is asked where |reference update| >= 0.2 * lr (first step: everywhere significant)
MIN_SIGN_AGREE = 0.999
MAX_REL_L2 = 1e-3        # first step (the update is -lr*sign(g) on the significant elements): measured 2e-6 .. 5e-6
# Later steps: Adam's lr * m_hat / sqrt(v_hat) is a smooth but ill-conditioned function of the gradient history where
# the steps' gradients disagree, so the ReLU-flip noise of DESIGN.md §6 (~1e-3 of max|g| between any two correct fp32
# implementations) shows up amplified.  Yardstick: the reference's own CPU path against itself with 1 thread instead
# of 8 (different reduction orders only) gives 4e-4 / 5e-3 (SR2 / SR1) at step 2 and 2e-3 / 1.8e-2 at step 3; the HIP
# path against it measures 1.2e-2 / 4.5e-2 at step 2.  A wrong sign gives ~2, a wrong bias correction ~0.4.
MAX_REL_L2_LATER = 1e-1


def flat(named, names):
    """Concatenate ``named[n]`` over ``names`` (parameters() order) into one float64 vector."""
    return torch.cat([named[n].detach().reshape(-1).double().cpu() for n in names])


def significant_mask(grads_per_step, names, sig_frac=SIG_FRAC):
    """Elements whose reference gradient was >= sig_frac * max|g| of their tensor in EVERY step of ``grads_per_step``
    (a list of {name: grad}).  Returns a flat bool vector in parameters() order."""
    mask = None
    for grads in grads_per_step:
        parts = []
        for n in names:
            g = grads[n].detach().double().cpu()
            parts.append((g.abs() >= sig_frac * g.abs().max()).reshape(-1))
        m = torch.cat(parts)
        mask = m if mask is None else (mask & m)
    return mask


def update_parity(upd, upd_ref, sig, p_after, p_after_ref, lr, steps_taken, what="",
                  min_sign_agree=MIN_SIGN_AGREE, max_rel_l2=None, verbose=True):
    """Assert the update criteria described in the module docstring; returns (sign_agreement, rel_l2, n_sig)."""
    if max_rel_l2 is None:
        max_rel_l2 = MAX_REL_L2 if steps_taken == 1 else MAX_REL_L2_LATER
    upd, upd_ref = upd.double().cpu(), upd_ref.double().cpu()
    sig = sig.cpu()
    n_sig = int(sig.sum())
    assert n_sig > 0.5 * sig.numel(), (what, "too few significant elements", n_sig, sig.numel())
    a, b = upd[sig], upd_ref[sig]
    big = b.abs() >= SIGN_FRAC * lr
    assert int(big.sum()) > 0.5 * n_sig, (what, "too few elements with a full-size update", int(big.sum()), n_sig)
    agree = float((torch.sign(a[big]) == torch.sign(b[big])).double().mean())
    rel_l2 = float((a - b).norm() / b.norm().clamp_min(1e-300))
    if verbose:
        print(f"[{what}] update: sign agreement {agree:.5f} on {int(big.sum())}, rel L2 {rel_l2:.3e} on {n_sig} of {sig.numel()}")
    assert agree >= min_sign_agree, (what, "update sign agreement", agree, n_sig)
    assert rel_l2 <= max_rel_l2, (what, "update relative L2", rel_l2, n_sig)
    rest = ~sig
    if int(rest.sum()):
        worst = float((p_after.double().cpu()[rest] - p_after_ref.double().cpu()[rest]).abs().max())
        assert worst <= 2.5 * lr * steps_taken, (what, "parameter bound on the noise-level elements", worst)
    return agree, rel_l2, n_sig


def pack_bits(b: torch.Tensor) -> np.ndarray:
    return np.packbits(b.cpu().numpy().astype(np.uint8))


def unpack_bits(a: np.ndarray, n: int) -> torch.Tensor:
    return torch.from_numpy(np.unpackbits(a)[:n].astype(bool))


def random_masks(seed, B, hr=256):
    """Seeded Bernoulli(1/2) masks for every BatchNorm layer (any machine regenerates them)."""
    rs = np.random.RandomState(seed)
    lv = [0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 2, 2, 1, 1, 0, 0]
    out = {}
    for (conv, bn, cin, cout), l in zip(CONV_BN_LAYERS, lv):
        h = hr >> l
        out[bn] = torch.from_numpy(rs.randint(0, 2, size=(B, cout, h, h)).astype(bool))
    return out


class OracleTrajectory:
    """The oracle's own optimisation trajectory for a golden training case ``c`` (bit-equal to the reference's when
    generated: asserted by tests/golden/make_golden_steps.py), one step at a time.  ``step()`` returns
    (update, params_after, significant_mask, losses) as flat float64 vectors in parameters() order."""

    def __init__(self, c, kind, mean, std):
        from . import sif_oracle as O
        self.O, self.c, self.kind, self.names, self.mean, self.std = O, c, kind, O.param_names(), mean, std
        self.sd = O.synthetic_state(c["wseed"])
        self.batch = O.synthetic_batch(c["bseed"], c["B"])
        self.adam = O.AdamState(self.names, c["lr"])
        self.hist = []

    def step(self):
        c = self.c
        before = flat(self.sd, self.names)
        _, losses, grads = self.O.forward_backward(self.sd, *self.batch, self.mean, self.std, c["alpha"], c["gamma"], self.kind)
        self.adam.step(self.sd, grads)
        self.hist.append(grads)
        after = flat(self.sd, self.names)
        return after - before, after, significant_mask(self.hist, self.names), losses
