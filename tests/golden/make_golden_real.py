#!/usr/bin/env python3
"""Goldens at the reference's own operating point (missing #3 / next 7).

Run in the BUILD container only (imports /root/reference and loads its shipped checkpoints; nothing of either travels):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_real.py

What it does
  1. loads the shipped trained weights ``models/modelB_2609/modelB_state_dict.pt`` (SIF-NN-SR2) and
     ``models/modelB_1009/modelB_state_dict.pt`` (SIF-NN-SR1) with ``torch.load(weights_only=True)`` and ASSERTS
     oracle == reference UNDER THOSE WEIGHTS: eval forward, and training-mode forward + SIF loss + all 53 gradients + BN
     buffers, with each checkpoint's own loss and hyper-parameters (``modelB_train_params.json``);
  2. measures per-tensor (mean, std, min, max) of all 104 state_dict entries of both checkpoints -- 2 x 104 x 4 numbers, data --
     and writes them to ``real_weight_stats_v1.json``: ``oracle.matched_state(stats, seed)`` regenerates from them, on any
     machine, a synthetic state with the trained checkpoints' moments and ranges (the weights themselves are CeCILL-C
     artefacts of the reference and are not copied);
  3. runs reference and oracle on that statistics-matched state (asserting equality again) and stores the REFERENCE's
     outputs as digests in ``golden_real_v1.json`` for the CPU oracle test and the HIP tests.

Inputs.  The 83 real (LST, NDVI) pairs under ``test_data_formatted/data/*_data_dict.pkl`` are pickles of dicts holding
``rasterio`` / ``affine`` objects next to the arrays.  The only loader this project may use on files that ship inside the
reference is one that executes nothing from the file; ``torch.load(weights_only=True)`` REFUSES them (numpy reconstruct
globals are not on its allow-list) -- recorded in the fixture (``real_pairs``) and in DESIGN.md -- so the inputs are the
seeded synthetic batches of ``oracle.synthetic_batch`` (z-scored LST / NDVI are ~N(0, 1) by construction of the
reference's own normalisation, dataset.py:134-142).

Data only (no reference source text, no weights).
"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np   # noqa: E402,F401
import torch   # noqa: E402

from make_golden import REF, import_reference, ref_loss, rel   # noqa: E402
from oracle import sif_oracle as O   # noqa: E402

MEAN, STD = 307.2378, 5.5698
CKPTS = {"modelB_2609": "sr2", "modelB_1009": "sr1"}      # model_perf_aster_formatds.py:65-67


def tensor_stats(v):
    v = v.double().flatten()
    return [float(v.mean()), float(v.std(unbiased=False)) if v.numel() > 1 else 0.0, float(v.min()), float(v.max())]


def main():
    torch.set_num_threads(8)
    ref_model, ref_utils = import_reference()
    names = O.param_names()

    def new_ref(sd):
        m = ref_model.ModelB_2(in_channels=2, downchannels=[16, 32, 64, 128], padding_mode="replicate",
                               activation="ReLU", bilinear=1, n_bridge_blocks=1)
        m.load_state_dict(sd, strict=True)
        return m

    def compare(sd, kind, alpha, gamma, bseed, B, tag, worst):
        """reference vs oracle under ``sd``: eval forward, then train forward + loss + backward.  Returns the reference's outputs."""
        lst, lst_up, ndvi = O.synthetic_batch(bseed, B)
        x = torch.cat((lst_up, ndvi), 1)
        m = new_ref(sd).eval()
        with torch.inference_mode():
            y_ref = m(x)
        y_ora = O.modelb2_forward({k: v.clone() for k, v in sd.items()}, x, training=False)
        worst[f"{tag}_eval"] = rel(y_ora, y_ref)
        assert worst[f"{tag}_eval"] < 2e-6, worst
        m = new_ref(sd).train()
        sr = m(x)
        ds, pl, loss = ref_loss(ref_utils, kind, sr, lst, ndvi, MEAN, STD, alpha, gamma)
        loss.backward()
        gref = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
        sd_o = {k: v.clone() for k, v in sd.items()}
        sr_o, (ds_o, pl_o, loss_o), g_o = O.forward_backward(sd_o, lst, lst_up, ndvi, MEAN, STD, alpha, gamma, kind)
        e = {"sr": rel(sr_o, sr), "ds": rel(ds_o, ds), "pl": rel(pl_o, pl), "loss": rel(loss_o, loss),
             "grad": max(rel(g_o[n], gref[n]) for n in names)}
        msd = m.state_dict()
        e["bn"] = max(rel(sd_o[k].float(), msd[k].float()) for k in msd if k.endswith(("running_mean", "running_var")))
        worst[f"{tag}_train"] = e
        assert max(e.values()) < 5e-4, e
        return {"y_eval": O.digest(y_ref), "y_denorm": O.digest(y_ref * STD + MEAN), "sr": O.digest(sr.detach()),
                "ds": float(ds), "pl": float(pl), "loss": float(loss),
                "grads": {n: O.digest(g, 8) for n, g in gref.items()},
                "bn_buffers": {k: O.digest(msd[k].float(), 8) for k in msd
                               if k.endswith(("running_mean", "running_var", "num_batches_tracked"))}}

    stats, out, worst = {}, {"version": 1, "torch": torch.__version__, "mean": MEAN, "std": STD, "cases": {}}, {}
    for ck, kind in CKPTS.items():
        path = os.path.join(REF, "models", ck, "modelB_state_dict.pt")
        sd = torch.load(path, map_location="cpu", weights_only=True)
        assert [k for k in sd] == [k for k, _, _ in O.state_dict_spec()], "state_dict layout"
        hp = json.load(open(os.path.join(REF, "models", ck, "modelB_train_params.json")))["hyperparameters"]
        alpha, gamma = float(hp["alpha"]), float(hp["gamma"])
        # 1. oracle == reference under the REAL weights (nothing stored but the agreement figures)
        compare(sd, kind, alpha, gamma, 51, 2, f"{ck}_real", worst)
        # 2. the checkpoint's per-tensor statistics
        stats[ck] = {k: tensor_stats(v) for k, v in sd.items()}
        # 3. the statistics-matched synthetic state: reference's outputs stored
        wseed = 71 if kind == "sr2" else 72
        msd = O.matched_state(stats[ck], wseed)
        rec = compare(msd, kind, alpha, gamma, 61, 2, f"{ck}_matched", worst)
        rec.update({"checkpoint": ck, "kind": kind, "alpha": alpha, "gamma": gamma, "wseed": wseed, "bseed": 61, "B": 2})
        out["cases"][f"matched_{kind}"] = rec

    # the real (LST, NDVI) pairs: the safe loader's verdict, recorded
    pkl = os.path.join(REF, "test_data_formatted", "data", "0_data_dict.pkl")
    try:
        torch.load(pkl, map_location="cpu", weights_only=True)
        verdict = "loaded"
    except Exception as e:   # noqa: BLE001
        verdict = f"refused by torch.load(weights_only=True): {type(e).__name__}"
    out["real_pairs"] = {"file_kind": "pickled dict with rasterio / affine objects", "safe_loader": verdict,
                         "fallback": "seeded synthetic batches (oracle.synthetic_batch)"}
    out["oracle_vs_reference_worst_rel"] = worst
    with open(os.path.join(HERE, "real_weight_stats_v1.json"), "w") as f:
        json.dump({"what": "per-tensor [mean, std (population), min, max] of the reference's shipped checkpoints; "
                           "oracle.matched_state() regenerates statistics-matched synthetic states from them",
                   "checkpoints": stats}, f, indent=0)
    with open(os.path.join(HERE, "golden_real_v1.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(worst, indent=1))
    print(out["real_pairs"])
    for n in ("real_weight_stats_v1.json", "golden_real_v1.json"):
        print("wrote", n, os.path.getsize(os.path.join(HERE, n)), "bytes")


if __name__ == "__main__":
    main()
