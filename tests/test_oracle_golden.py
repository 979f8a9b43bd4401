"""CPU: the oracle (oracle/sif_oracle.py) against the committed golden vectors, which are the
REFERENCE's own outputs captured by tests/golden/make_golden.py (SURVEY.md §8 c)."""
import numpy as np
import torch

from oracle import sif_oracle as O
from tests.conftest import check_digest

TOL = 2e-5   # the oracle was bit-exact vs the reference when generated; slack for other BLAS/threads


def test_state_dict_layout(golden):
    spec = [(k, list(s), str(d)) for k, s, d in O.state_dict_spec()]
    assert spec == [tuple(x) if not isinstance(x, list) else (x[0], x[1], x[2]) for x in golden["state_dict_spec"]]
    assert len(spec) == 104
    assert len(O.param_names()) == 53
    sd = O.synthetic_state(0)
    assert sum(sd[n].numel() for n in O.param_names()) == golden["n_params"] == 282705


def test_psf_kernels(golden):
    for mtf in (0.1, 0.25):
        k = O.generate_psf_kernel(1.0, 4, mtf, None)
        assert k.shape == (9, 9) and k.dtype == np.float32
        np.testing.assert_array_equal(k.flatten(), np.array(golden["cases"][f"psf_{mtf}"]["kernel9x9"], dtype=np.float32))
        t = O.psf_taps_1d(mtf)
        np.testing.assert_allclose(t, golden["cases"][f"psf_{mtf}"]["taps1d"], rtol=0, atol=1e-15)
        assert np.abs(np.outer(t, t) - k).max() < 2e-8      # rank-1 to fp32 rounding


def test_eval_forward(golden):
    for name, c in golden["cases"].items():
        if not name.startswith("eval_"):
            continue
        sd = O.synthetic_state(c["wseed"])
        lst, lst_up, ndvi = O.synthetic_batch(c["bseed"], c["B"])
        y = O.modelb2_forward(sd, torch.cat((lst_up, ndvi), 1), training=False)
        check_digest(y, c["y"], TOL)
        check_digest(O.predict_tiles(sd, lst_up, ndvi, 307.2378, 5.5698), c["y_denorm"], TOL)


def test_loss_operators(golden):
    rs = np.random.RandomState(5)
    x = torch.from_numpy(rs.standard_normal((2, 1, 256, 256)).astype(np.float32))
    xk = x * 5.5698 + 307.2378
    for name, fn, inp in (("downscale_mtf0.1", O.downscale_LST_SR_to_LR, xk),
                          ("ftm_mtf0.25", lambda t: O.get_output_ftm(t, mtf=0.25), x),
                          ("sobel", O.sobel_bank, x)):
        a = inp.clone().requires_grad_(True)
        y = fn(a)
        w = torch.from_numpy(np.random.RandomState(6).standard_normal(tuple(y.shape)).astype(np.float32))
        (g,) = torch.autograd.grad((y * w).sum(), a)
        check_digest(y, golden["cases"]["op_" + name]["y"], TOL)
        check_digest(g, golden["cases"]["op_" + name]["gx"], TOL)


def _train_case(golden, kind):
    c = golden["cases"][f"train_{kind}"]
    sd = O.synthetic_state(c["wseed"])
    lst, lst_up, ndvi = O.synthetic_batch(c["bseed"], c["B"])
    adam = O.AdamState(O.param_names(), c["lr"])
    for i, rec in enumerate(c["steps"]):
        sr, (ds, pl, loss), grads = O.forward_backward(sd, lst, lst_up, ndvi, c["mean"], c["std"],
                                                       c["alpha"], c["gamma"], kind)
        # later steps inherit ~sqrt(N) amplified differences through the cancelling gradient sums
        tol = TOL if i == 0 else 2e-3
        check_digest(sr, rec["sr"], tol)
        for got, key in ((ds, "ds"), (pl, "pl"), (loss, "loss")):
            assert abs(float(got) - rec[key]) <= tol * abs(rec[key])
        if i == 0:
            for n, d in rec["grads"].items():
                check_digest(grads[n], d, 1e-4)
            for k, d in rec["bn_buffers"].items():
                check_digest(sd[k].float(), d, TOL)
        adam.step(sd, grads)
        for n, d in rec["params_after"].items():
            check_digest(sd[n], d, 1e-5)


def test_train_sr2(golden):
    _train_case(golden, "sr2")


def test_train_sr1(golden):
    _train_case(golden, "sr1")


# ---- the imposed-mask mode (oracle.RELU_MASKS) and the update signs, pinned by tests/golden/make_golden_steps.py -----
import json
import os

from oracle import checks as C

_GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_masked_mode_vs_reference_with_imposed_masks():
    """The reference model run with seeded random masks imposed on its own nn.ReLU (forward hook) -- digests in
    golden_masked_v1.json -- against oracle.RELU_MASKS on the same masks.  The masked network is smooth, so this holds on
    any machine; the generator additionally asserted bit-equality for the reference's natural and perturbed masks."""
    gm = json.load(open(os.path.join(_GOLD, "golden_masked_v1.json")))
    for name, c in gm["cases"].items():
        assert max(max(v.values()) for v in c["oracle_vs_reference_worst_rel"].values()) < 1e-6
        lst, lst_up, ndvi = O.synthetic_batch(c["bseed"], c["B"])
        O.RELU_MASKS = C.random_masks(c["mask_seed"], c["B"])
        try:
            sr, (ds, pl, loss), g = O.forward_backward(O.synthetic_state(c["wseed"]), lst, lst_up, ndvi, 307.2378, 5.5698,
                                                       c["alpha"], c["gamma"], c["kind"])
        finally:
            O.RELU_MASKS = None
        check_digest(sr, c["sr"], TOL)
        for got, key in ((ds, "ds"), (pl, "pl"), (loss, "loss")):
            assert abs(float(got) - c[key]) <= TOL * abs(c[key])
        for n, d in c["grads"].items():
            check_digest(g[n], d, TOL)


def test_update_signs_along_the_trajectory(golden):
    """Oracle trajectory (3 Adam steps) against the REFERENCE's update signs / significance masks
    (golden_steps_v1.npz): the check the GPU tests apply to the HIP path, applied to the oracle itself."""
    z = np.load(os.path.join(_GOLD, "golden_steps_v1.npz"))
    meta = json.loads(bytes(z["meta_json"]).decode())
    names = O.param_names()
    for kind in ("sr2", "sr1"):
        c = golden["cases"][f"train_{kind}"]
        sd = O.synthetic_state(c["wseed"])
        lst, lst_up, ndvi = O.synthetic_batch(c["bseed"], c["B"])
        adam = O.AdamState(names, c["lr"])
        hist = []
        for i in range(3):
            before = C.flat(sd, names)
            _, _, grads = O.forward_backward(sd, lst, lst_up, ndvi, c["mean"], c["std"], c["alpha"], c["gamma"], kind)
            adam.step(sd, grads)
            hist.append(grads)
            upd = C.flat(sd, names) - before
            m = meta[f"{kind}_s{i}"]
            n = m["n"]
            sig_ref = C.unpack_bits(z[f"{kind}_s{i}_sig"], n)
            sign_ref = C.unpack_bits(z[f"{kind}_s{i}_sign"], n)
            assert int(sig_ref.sum()) == m["n_sig"]
            big_ref = C.unpack_bits(z[f"{kind}_s{i}_big"], n)
            agree = float(((upd > 0) == sign_ref)[big_ref].double().mean())
            assert agree >= 0.9999, (kind, i, agree)
            assert abs(float(upd[sig_ref].norm()) - m["upd_l2_sig"]) <= 1e-3 * m["upd_l2_sig"]
            sig = C.significant_mask(hist, names)
            assert float((sig == sig_ref).double().mean()) >= 0.999


def test_update_check_rejects_a_sign_error(golden):
    """The step-level check must be able to fail: a first Adam step taken with the gradient's sign flipped stays within
    2.5*lr of the reference parameters (the bound used in round 1) but is rejected by the update comparison."""
    import pytest
    c = golden["cases"]["train_sr2"]
    traj = C.OracleTrajectory(c, "sr2", c["mean"], c["std"])
    p0 = C.flat(traj.sd, traj.names)
    upd_ref, p_ref, sig, _ = traj.step()
    wrong = p0 - upd_ref                                   # every element moved the wrong way
    assert float((wrong - p_ref).abs().max()) <= 2.5 * c["lr"]
    with pytest.raises(AssertionError):
        C.update_parity(wrong - p0, upd_ref, sig, wrong, p_ref, c["lr"], 1)
    few = upd_ref.clone(); few[::200] *= -1                # 0.5 % of them
    with pytest.raises(AssertionError):
        C.update_parity(few, upd_ref, sig, p0 + few, p_ref, c["lr"], 1)
    C.update_parity(upd_ref, upd_ref, sig, p_ref, p_ref, c["lr"], 1)


def test_sobel_filters_are_the_references(golden):
    """golden["sobel_filters"] was parsed (ast.literal_eval) out of the reference's train_model_B_predef_filters.py:38-42 by
    tests/golden/make_golden.py; the oracle's bank, which the SR1 checks use everywhere, must be that literal."""
    assert O.SOBEL_FILTERS == golden["sobel_filters"] and np.array(golden["sobel_filters"]).shape == (4, 3, 3)


def _real():
    import json
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    with open(os.path.join(here, "golden_real_v1.json")) as f:
        g = json.load(f)
    with open(os.path.join(here, "real_weight_stats_v1.json")) as f:
        st = json.load(f)["checkpoints"]
    return g, st


def test_statistics_matched_state_vs_reference():
    """The oracle at the reference's own operating point: synthetic states with the per-tensor moments and ranges of the
    shipped trained checkpoints (oracle.matched_state from real_weight_stats_v1.json).  The generator asserted
    oracle == reference bit-for-bit under the REAL weights and under these states; here the oracle is re-checked against the
    reference's stored outputs (eval forward, train forward, losses, 53 gradients, BN buffers) on any machine."""
    g, st = _real()
    assert g["real_pairs"]["safe_loader"].startswith("refused")        # the real pairs were not unpickled (recorded)
    for kind in ("sr2", "sr1"):
        c = g["cases"][f"matched_{kind}"]
        stats = st[c["checkpoint"]]
        sd = O.matched_state(stats, c["wseed"])
        # the state really has the checkpoint's moments: per-tensor mean within 4 standard errors (+ what the clipping to
        # [min, max] moves), std within 25 % for tensors of at least 64 elements
        for k, v in sd.items():
            mean, std, lo, hi = stats[k]
            if k.endswith("num_batches_tracked"):
                assert int(v) == int(round(mean))
                continue
            v = v.double()
            assert float(v.min()) >= lo - 1e-6 * max(1.0, abs(lo)) and float(v.max()) <= hi + 1e-6 * max(1.0, abs(hi)), k
            if v.numel() >= 64 and std > 0:
                assert abs(float(v.mean()) - mean) < 5 * std / v.numel() ** 0.5 + 0.1 * std, k
                assert 0.7 * std < float(v.std()) < 1.3 * std, k
        lst, lst_up, ndvi = O.synthetic_batch(c["bseed"], c["B"])
        y = O.modelb2_forward({k: v.clone() for k, v in sd.items()}, torch.cat((lst_up, ndvi), 1), training=False)
        check_digest(y, c["y_eval"], TOL)
        sr, (ds, pl, loss), grads = O.forward_backward(sd, lst, lst_up, ndvi, g["mean"], g["std"], c["alpha"], c["gamma"], kind)
        check_digest(sr, c["sr"], TOL)
        for got, key in ((ds, "ds"), (pl, "pl"), (loss, "loss")):
            assert abs(float(got) - c[key]) <= 1e-5 * abs(c[key]), (key, float(got), c[key])
        for n, d in c["grads"].items():
            check_digest(grads[n], d, 5e-3)      # as test_train_*: other BLAS / thread counts may flip a ReLU
        for k, d in c["bn_buffers"].items():
            check_digest(sd[k].float(), d, 1e-5)
