#!/usr/bin/env python3
"""Generate the golden vectors that pin ``oracle/sif_oracle.py`` to the reference.

Run in the BUILD container only (needs /root/reference, which never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What it does
  1. imports the reference's ``model.py`` (torch only) and ``utils.py`` (after registering empty
     stub modules for cv2 / skimage / osgeo, which the hot-path functions never touch);
  2. runs reference and oracle side by side on seeded inputs and ASSERTS agreement
     (eval fwd, train fwd+bwd incl. BN buffers, both loss variants, PSF kernels, 3 Adam steps);
  3. writes the REFERENCE's outputs as digests (sum / abs-sum / L2 / 64 strided samples) to
     ``tests/golden/golden_v1.json``.  Inputs and weights are regenerated from numpy seeds by
     ``oracle.sif_oracle.synthetic_state / synthetic_batch`` so only digests are stored.

The fixture holds data only (no reference source text).
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.nn.functional as F

from oracle import sif_oracle as O

REF = "/root/reference"


def import_reference():
    for name in ("cv2", "skimage", "skimage.metrics", "osgeo", "osgeo.gdal"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["skimage.metrics"].structural_similarity = None
    sys.modules["skimage.metrics"].peak_signal_noise_ratio = None
    sys.modules["skimage"].metrics = sys.modules["skimage.metrics"]
    sys.modules["osgeo"].gdal = sys.modules["osgeo.gdal"]
    sys.path.insert(0, REF)
    import model as ref_model          # noqa
    import utils as ref_utils          # noqa
    sys.path.remove(REF)
    return ref_model, ref_utils


def reference_sobel_filters():
    """The literal ``filters = [...]`` of /root/reference/train_model_B_predef_filters.py:38-42, read as DATA: the script cannot
    be imported (it imports GDAL-backed modules and opens absent dataset files at import), so its source is parsed with
    ``ast`` and the module-level assignment is evaluated with ``ast.literal_eval`` -- nothing of the file is executed.  The
    SR1 goldens are therefore pinned by the reference's own filter bank, not by the oracle's copy of it."""
    import ast
    with open(os.path.join(REF, "train_model_B_predef_filters.py")) as f:
        tree = ast.parse(f.read())
    for node in tree.body:
        if isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id == "filters" for t in node.targets):
            return ast.literal_eval(node.value)
    raise RuntimeError("no module-level `filters = [...]` in the reference's train_model_B_predef_filters.py")


REF_FILTERS = None      # set by main() / the other generators (reference_sobel_filters())


def _filters():
    global REF_FILTERS
    if REF_FILTERS is None:
        REF_FILTERS = reference_sobel_filters()
        assert REF_FILTERS == O.SOBEL_FILTERS, "oracle.SOBEL_FILTERS differs from the reference's filter bank"
    return REF_FILTERS


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def ref_loss(ref_utils, kind, sr, lst, ndvi, mean, std, alpha, gamma):
    """train_model_B_gradFTM.py:99-117 / train_model_B_predef_filters.py:111-133 restated around
    the REFERENCE's own utils functions (the scripts' train_step cannot be imported: module-global
    train_ds and absent dataset files, SURVEY.md §8 c)."""
    loss_fn = torch.nn.HuberLoss(reduction="mean", delta=1.0)
    sr_un = sr * std + mean
    down = ref_utils.downscale_LST_SR_to_LR(sr_un)
    down = (down - mean) / std
    ds = loss_fn(down, lst)
    if kind == "sr2":
        g_l = sr - ref_utils.get_output_ftm(sr, mtf=0.25)
        g_n = ndvi - ref_utils.get_output_ftm(ndvi, mtf=0.25)
    else:
        filters = _filters()                        # the reference's own literal (train_model_B_predef_filters.py:38-42)
        ft = torch.zeros((len(filters), 1, 3, 3))
        for i in range(len(ft)):
            ft[i, 0] = torch.tensor(filters[i], dtype=torch.float)
        g_l = F.conv2d(sr, ft, padding="same")
        g_n = F.conv2d(ndvi, ft, padding="same")
    pl = loss_fn(g_l, gamma * g_n)
    return ds, pl, alpha * ds + (1 - alpha) * pl


def main():
    torch.set_num_threads(8)
    ref_model, ref_utils = import_reference()
    out = {"version": 1, "torch": torch.__version__, "cases": {}}
    MEAN, STD = 307.2378, 5.5698
    worst = {}

    def new_ref(sd):
        m = ref_model.ModelB_2(in_channels=2, downchannels=[16, 32, 64, 128], padding_mode="replicate",
                               activation="ReLU", bilinear=1, n_bridge_blocks=1)
        assert list(m.state_dict().keys()) == list(sd.keys())
        m.load_state_dict(sd, strict=True)
        return m

    # ---- state_dict layout -------------------------------------------------------------------
    m0 = ref_model.ModelB_2(2, [16, 32, 64, 128], "replicate", "ReLU", 1, 1)
    ref_spec = [(k, list(v.shape), str(v.dtype)) for k, v in m0.state_dict().items()]
    ora_spec = [(k, list(s), str(d)) for k, s, d in O.state_dict_spec()]
    assert ref_spec == ora_spec, "state_dict layout mismatch"
    assert [n for n, _ in m0.named_parameters()] == O.param_names()
    out["sobel_filters"] = _filters()            # parsed from the reference's script (data: 4 x 3 x 3 integers)
    out["state_dict_spec"] = ref_spec
    out["n_params"] = sum(p.numel() for p in m0.parameters())

    # ---- PSF kernels (a7) --------------------------------------------------------------------
    for mtf in (0.1, 0.25):
        kr = ref_utils.generate_psf_kernel(1.0, 4, mtf, None)
        ko = O.generate_psf_kernel(1.0, 4, mtf, None)
        assert np.array_equal(kr, ko)
        t = O.psf_taps_1d(mtf)
        sep = np.outer(t, t)
        worst[f"psf_rank1_{mtf}"] = float(np.abs(sep - kr).max())
        out["cases"][f"psf_{mtf}"] = {"kernel9x9": [float(v) for v in kr.flatten()],
                                      "taps1d": [float(v) for v in t]}

    # ---- eval forward (a1-a6, a14) ------------------------------------------------------------
    for wseed, bseed, B in ((11, 21, 2), (12, 22, 1)):
        sd = O.synthetic_state(wseed)
        lst, lst_up, ndvi = O.synthetic_batch(bseed, B)
        m = new_ref(sd).eval()
        with torch.inference_mode():
            y_ref = m(torch.cat((lst_up, ndvi), 1))
        y_ora = O.modelb2_forward(O.synthetic_state(wseed), torch.cat((lst_up, ndvi), 1), training=False)
        worst[f"eval_{wseed}"] = rel(y_ora, y_ref)
        assert worst[f"eval_{wseed}"] < 2e-6, worst
        out["cases"][f"eval_w{wseed}_b{bseed}_B{B}"] = {
            "wseed": wseed, "bseed": bseed, "B": B, "y": O.digest(y_ref),
            "y_denorm": O.digest(y_ref * STD + MEAN)}

    # ---- loss operators alone (a8-a10), value + input-grad ------------------------------------
    rs = np.random.RandomState(5)
    x = torch.from_numpy(rs.standard_normal((2, 1, 256, 256)).astype(np.float32))
    xk = x * STD + MEAN
    for name, fr, fo, inp in (
            ("downscale_mtf0.1", lambda t: ref_utils.downscale_LST_SR_to_LR(t), lambda t: O.downscale_LST_SR_to_LR(t), xk),
            ("ftm_mtf0.25", lambda t: ref_utils.get_output_ftm(t, mtf=0.25), lambda t: O.get_output_ftm(t, mtf=0.25), x),
            ("sobel", lambda t: F.conv2d(t, torch.tensor(_filters(), dtype=torch.float)[:, None], padding="same"),
             O.sobel_bank, x)):
        a = inp.clone().requires_grad_(True)
        yr = fr(a)
        wgt = torch.from_numpy(np.random.RandomState(6).standard_normal(tuple(yr.shape)).astype(np.float32))
        (gr,) = torch.autograd.grad((yr * wgt).sum(), a)
        b = inp.clone().requires_grad_(True)
        yo = fo(b)
        (go,) = torch.autograd.grad((yo * wgt).sum(), b)
        worst[name] = max(rel(yo, yr), rel(go, gr))
        assert worst[name] < 2e-6, worst
        out["cases"]["op_" + name] = {"y": O.digest(yr), "gx": O.digest(gr)}

    # ---- train fwd+bwd, both losses (a11, a12) + 3 Adam steps ----------------------------------
    for kind, alpha, gamma, lr, wseed, bseed, B in (("sr2", 0.5, -0.25, 1e-4, 31, 41, 2),
                                                    ("sr1", 0.99, -0.5, 1e-3, 32, 42, 2)):
        sd0 = O.synthetic_state(wseed)
        lst, lst_up, ndvi = O.synthetic_batch(bseed, B)
        m = new_ref(sd0).train()
        opt = torch.optim.Adam(m.parameters(), lr=lr)
        sd = O.synthetic_state(wseed)
        adam = O.AdamState(O.param_names(), lr)
        case = {"kind": kind, "alpha": alpha, "gamma": gamma, "lr": lr, "wseed": wseed, "bseed": bseed,
                "B": B, "mean": MEAN, "std": STD, "steps": []}
        for step in range(3):
            opt.zero_grad()
            sr = m(torch.cat((lst_up, ndvi), 1))
            ds, pl, loss = ref_loss(ref_utils, kind, sr, lst, ndvi, MEAN, STD, alpha, gamma)
            loss.backward()
            gref = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
            sr_o, (ds_o, pl_o, loss_o), g_o = O.forward_backward(sd, lst, lst_up, ndvi, MEAN, STD, alpha, gamma, kind)
            e = {"sr": rel(sr_o, sr), "ds": rel(ds_o, ds), "pl": rel(pl_o, pl), "loss": rel(loss_o, loss),
                 "grad": max(rel(g_o[n], gref[n]) for n in gref)}
            worst[f"{kind}_step{step}"] = e
            assert max(e.values()) < 5e-4, e
            rec = {"sr": O.digest(sr), "ds": float(ds), "pl": float(pl), "loss": float(loss)}
            if step == 0:
                rec["grads"] = {n: O.digest(g, 8) for n, g in gref.items()}
                msd = m.state_dict()
                rec["bn_buffers"] = {k: O.digest(msd[k].float(), 8) for k in msd
                                     if k.endswith(("running_mean", "running_var", "num_batches_tracked"))}
                for k in rec["bn_buffers"]:
                    assert rel(sd[k].float(), msd[k].float()) < 1e-5, k
            opt.step()
            adam.step(sd, g_o)
            msd = m.state_dict()
            rec["params_after"] = {n: O.digest(msd[n], 8) for n in O.param_names()}
            worst[f"{kind}_params{step}"] = max(rel(sd[n], msd[n]) for n in O.param_names())
            case["steps"].append(rec)
        out["cases"][f"train_{kind}"] = case

    out["oracle_vs_reference_worst_rel"] = worst
    path = os.path.join(HERE, "golden_v1.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(worst, indent=1))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
