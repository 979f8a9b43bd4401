#!/usr/bin/env python3
"""Golden vectors that make the step-level parity checks able to fail.

Run in the BUILD container only (imports /root/reference; nothing of it travels):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_steps.py

Writes
  * ``golden_steps_v1.npz`` -- for the two training cases of ``golden_v1.json`` (same seeds) and each of their three
    optimisation steps: the SIGN of the reference's parameter update ``p_after - p_before`` for all 282,705 elements
    (packed bits, parameters() order), the mask of elements whose reference gradient was >= 1e-2 * max|g| of their
    tensor in every step so far ("significant": above the ReLU-flip noise, oracle/checks.py), the subset of those
    whose reference update is at least 0.2 * lr in size ("big": where the update's SIGN is meaningful -- from the
    second step on Adam's update changes sign where 0.9 g_1 ~ -g_2), and the L2 norm of the reference update over
    the significant mask.  The run ASSERTS that the oracle's trajectory equals the reference's.
  * ``golden_masked_v1.json`` -- pins ``oracle.RELU_MASKS`` (the oracle's imposed-mask mode, which the GPU tests use
    for their tight 1e-4 gradient check): the reference model itself is run with given masks imposed on its own
    (shared) ``nn.ReLU`` through a forward hook, and the masked oracle must reproduce its output, losses and all 53
    gradients.  Three mask sets: the reference's own masks (must equal the un-hooked reference, too), those masks
    with 32 flips per layer, and seeded Bernoulli(1/2) masks -- the last is stored as digests because it does not
    depend on bit-exact forward values, so ``tests/test_oracle_golden.py`` can re-check it on any machine.

Data only (no reference source text).
"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np
import torch

from make_golden import import_reference, ref_loss, rel   # noqa: E402
from oracle import checks as C   # noqa: E402
from oracle import sif_oracle as O   # noqa: E402

MEAN, STD = 307.2378, 5.5698
CASES = (("sr2", 0.5, -0.25, 1e-4, 31, 41, 2), ("sr1", 0.99, -0.5, 1e-3, 32, 42, 2))


def main():
    torch.set_num_threads(8)
    ref_model, ref_utils = import_reference()
    names = O.param_names()

    def new_ref(sd):
        m = ref_model.ModelB_2(in_channels=2, downchannels=[16, 32, 64, 128], padding_mode="replicate",
                               activation="ReLU", bilinear=1, n_bridge_blocks=1)
        m.load_state_dict(sd, strict=True)
        return m

    # ------------------------------------------------------------------------------------------
    # 1. update signs / significance masks along the three-step trajectories
    # ------------------------------------------------------------------------------------------
    arrays, meta = {}, {}
    for kind, alpha, gamma, lr, wseed, bseed, B in CASES:
        lst, lst_up, ndvi = O.synthetic_batch(bseed, B)
        m = new_ref(O.synthetic_state(wseed)).train()
        opt = torch.optim.Adam(m.parameters(), lr=lr)
        sd = O.synthetic_state(wseed)
        adam = O.AdamState(names, lr)
        grads_hist = []
        for step in range(3):
            before = {n: p.detach().clone() for n, p in m.named_parameters()}
            opt.zero_grad()
            sr = m(torch.cat((lst_up, ndvi), 1))
            ds, pl, loss = ref_loss(ref_utils, kind, sr, lst, ndvi, MEAN, STD, alpha, gamma)
            loss.backward()
            gref = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
            opt.step()
            after = {n: p.detach().clone() for n, p in m.named_parameters()}
            # the oracle, same step
            _, _, g_o = O.forward_backward(sd, lst, lst_up, ndvi, MEAN, STD, alpha, gamma, kind)
            adam.step(sd, g_o)
            e_g = max(rel(g_o[n], gref[n]) for n in names)
            e_p = max(float((sd[n] - after[n]).abs().max()) for n in names)
            print(f"[{kind}] step {step}: oracle vs reference  grads rel {e_g:.2e}  params abs {e_p:.2e}")
            assert e_g < 1e-6 and e_p <= 1e-9, (kind, step, e_g, e_p)
            grads_hist.append(gref)
            upd = C.flat(after, names) - C.flat(before, names)
            sig = C.significant_mask(grads_hist, names)
            arrays[f"{kind}_s{step}_sign"] = C.pack_bits(upd > 0)
            arrays[f"{kind}_s{step}_sig"] = C.pack_bits(sig)
            arrays[f"{kind}_s{step}_big"] = C.pack_bits(sig & (upd.abs() >= C.SIGN_FRAC * lr))
            meta[f"{kind}_s{step}"] = {"n": int(upd.numel()), "n_sig": int(sig.sum()), "upd_l2_sig": float(upd[sig].norm()),
                                       "upd_l2": float(upd.norm()), "n_zero_update": int((upd == 0).sum())}
            print("   ", meta[f"{kind}_s{step}"])
    arrays["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, "golden_steps_v1.npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path), "bytes")

    # ------------------------------------------------------------------------------------------
    # 2. the imposed-mask mode of the oracle against the reference with the same masks imposed on its ReLU
    # ------------------------------------------------------------------------------------------
    out = {"version": 1, "torch": torch.__version__, "cases": {}}
    bn_names = [bn for _, bn, _, _ in O.CONV_BN_LAYERS]
    relu = ref_model.activation_functions["ReLU"]          # the ONE nn.ReLU instance every block shares (model.py:78-81)

    def ref_run(kind, alpha, gamma, wseed, lst, lst_up, ndvi, masks=None, record=None):
        m = new_ref(O.synthetic_state(wseed)).train()
        calls = [0]

        def hook(mod, inp, outp):
            i = calls[0]
            calls[0] += 1
            if record is not None:
                record[bn_names[i]] = (inp[0] > 0).detach()
            if masks is not None:
                return inp[0] * masks[bn_names[i]].to(inp[0].dtype)
            return None

        h = relu.register_forward_hook(hook)
        try:
            sr = m(torch.cat((lst_up, ndvi), 1))
        finally:
            h.remove()
        assert calls[0] == 17
        ds, pl, loss = ref_loss(ref_utils, kind, sr, lst, ndvi, MEAN, STD, alpha, gamma)
        grads = torch.autograd.grad(loss, [p for _, p in m.named_parameters()])
        return sr.detach(), (float(ds), float(pl), float(loss)), dict(zip([n for n, _ in m.named_parameters()], grads))

    def ora_run(kind, alpha, gamma, wseed, lst, lst_up, ndvi, masks):
        O.RELU_MASKS = masks
        try:
            sr, (ds, pl, loss), g = O.forward_backward(O.synthetic_state(wseed), lst, lst_up, ndvi, MEAN, STD, alpha, gamma, kind)
        finally:
            O.RELU_MASKS = None
        return sr, (float(ds), float(pl), float(loss)), g

    def compare(tag, a, b):
        sr_a, l_a, g_a = a
        sr_b, l_b, g_b = b
        e = {"sr": rel(sr_a, sr_b), "loss": max(abs(x - y) / abs(y) for x, y in zip(l_a, l_b)),
             "grad": max(rel(g_a[n], g_b[n]) for n in names)}
        print(f"   {tag}: {e}")
        assert max(e.values()) < 1e-6, (tag, e)
        return e

    for kind, alpha, gamma, lr, wseed, bseed, B in CASES:
        lst, lst_up, ndvi = O.synthetic_batch(bseed, B)
        args = (kind, alpha, gamma, wseed, lst, lst_up, ndvi)
        print(f"[{kind}] imposed masks")
        natural = {}
        plain = ref_run(*args, record=natural)
        rec_o = {}
        O.RECORD_MASKS = rec_o
        try:
            O.forward_backward(O.synthetic_state(wseed), lst, lst_up, ndvi, MEAN, STD, alpha, gamma, kind)
        finally:
            O.RECORD_MASKS = None
        for bn in bn_names:
            assert torch.equal(natural[bn], rec_o[bn]), bn          # same forward, bit for bit -> same masks
        worst = {}
        # (i) the reference's own masks: hooked reference == plain reference == masked oracle
        worst["natural_hook_vs_plain"] = compare("reference hooked(natural) vs plain", ref_run(*args, masks=natural), plain)
        worst["natural"] = compare("oracle masked(natural) vs plain reference", ora_run(*args, natural), plain)
        # (ii) 32 flips per layer: a linear region neither forward would take by itself
        flipped = {}
        for i, bn in enumerate(bn_names):
            mk = natural[bn].clone().reshape(-1)
            idx = torch.from_numpy(np.random.RandomState(1000 + i).choice(mk.numel(), 32, replace=False))
            mk[idx] = ~mk[idx]
            flipped[bn] = mk.view_as(natural[bn])
        worst["flipped"] = compare("oracle masked(32 flips/layer) vs hooked reference", ora_run(*args, flipped), ref_run(*args, masks=flipped))
        # (iii) seeded random masks -> digests (machine-independent)
        rnd = C.random_masks(500 + wseed, B)
        r = ref_run(*args, masks=rnd)
        worst["random"] = compare("oracle masked(random) vs hooked reference", ora_run(*args, rnd), r)
        out["cases"][f"masked_{kind}"] = {
            "kind": kind, "alpha": alpha, "gamma": gamma, "wseed": wseed, "bseed": bseed, "B": B, "mask_seed": 500 + wseed,
            "sr": O.digest(r[0]), "ds": r[1][0], "pl": r[1][1], "loss": r[1][2],
            "grads": {n: O.digest(g, 8) for n, g in r[2].items()}, "oracle_vs_reference_worst_rel": worst}
    # ------------------------------------------------------------------------------------------
    # 3. early stopping: sifsr.train.ModelCheckpoint against the reference's us.model_checkpoint (utils.py:667-714) on
    #    random validation-loss sequences (generator-time assertion; host-side bookkeeping, nothing to store)
    # ------------------------------------------------------------------------------------------
    import random
    import sifsr  # noqa: F401  (CPU import: the HIP library is only loaded on first use)
    from sifsr.train import ModelCheckpoint
    random.seed(1)
    lin = torch.nn.Linear(1, 1)
    for _ in range(300):
        n, pat = random.randint(1, 12), random.randint(1, 5)
        a, b = ref_utils.model_checkpoint(n, pat), ModelCheckpoint(n, pat)
        hist = {"v": []}
        for e in range(1, n + 1):
            hist["v"].append(round(random.random(), 1))
            a.test_update(lin, hist, "v", e); b.test_update(lin, hist, "v", e)
            assert (a.best_epoch, a.curr_patience, a.train_state, a.saved_best_value) == \
                   (b.best_epoch, b.curr_patience, b.train_state, b.saved_best_value), (hist, pat, e)
            if a.train_state == "break":
                break
    print("ModelCheckpoint == reference model_checkpoint on 300 random sequences")
    out["model_checkpoint_checked_sequences"] = 300

    path = os.path.join(HERE, "golden_masked_v1.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
