"""GPU tests of the drop-in boundary as the reference's own scripts use it (SURVEY.md §8 b):

  * BASELINE.json config 1 -- ``ModisDatasetB`` (16 pairs) -> ``DataLoader(batch_size=8)`` -> one epoch of training
    steps -> validation pass (paramsB.json:6, train_model_B_gradFTM.py:86-138,176-237,295-354) against the oracle on the
    same batches;
  * ``predict.py:96-101`` / ``model_perf_aster_formatds.py:182-203`` replayed verbatim: CPU input tensors, ``.numpy()``
    on the model output;
  * the zero-edit import surface: ``from model import ModelB_2``, ``from dataset import ModisDatasetB``, ``import utils as us``
    with ``dropin/`` on ``sys.path``;
  * a training-mode forward under ``torch.no_grad()`` (BatchNorm recalibration), which nn.Module allows;
  * the guard against capturing a training step while an earlier step's autograd graph is alive.
"""
import copy
import importlib
import os
import sys

import numpy as np
import pytest
import torch

from oracle import sif_oracle as O
from tests.conftest import check_digest, rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-4
MEAN, STD = 307.2378, 5.5698


@pytest.fixture(scope="module")
def sifsr():
    import sifsr as pkg
    assert torch.cuda.is_available()
    return pkg


@pytest.fixture()
def dropin():
    """``model`` / ``dataset`` / ``utils`` resolved through dropin/ (and removed again, so no other test sees them)."""
    path = os.path.join(ROOT, "dropin")
    saved = {k: sys.modules.pop(k, None) for k in ("model", "dataset", "utils")}
    sys.path.insert(0, path)
    try:
        yield {k: importlib.import_module(k) for k in ("model", "dataset", "utils")}
    finally:
        sys.path.remove(path)
        for k, v in saved.items():
            sys.modules.pop(k, None)
            if v is not None:
                sys.modules[k] = v


def test_config1_one_epoch_batch8_16_pairs(sifsr):
    """paramsB.json: batch_size 8, alpha 0.1, gamma -0.4, lr 1e-3; 16 synthetic pairs; 1 epoch of SR2 training + the
    validation pass, through ``train.fit`` -- the reference's ``train()`` loop.  Checked per batch against the oracle run
    on the SAME batches (a fixed-order loader, shuffle is the loader's business): step-1 losses 1e-4, step-2 losses
    2e-3 (they inherit the first update's ReLU-flip noise), eval losses 2e-3, the update criteria of oracle/checks.py."""
    from torch.utils.data import DataLoader
    from oracle import checks as C
    alpha, gamma, lr, bs = 0.1, -0.4, 1e-3, 8
    train_ds = sifsr.dataset.ModisDatasetB("data/ModisDatasetB.csv", transf="norm", split="Train", time="day", length=16)
    val_ds = sifsr.dataset.ModisDatasetB("data/ModisDatasetB.csv", transf="norm", split="Val", time="day", length=16)
    assert len(train_ds) == 16 and train_ds[0][0].shape == (1, 64, 64) and train_ds[0][1].shape == (1, 256, 256)
    sd0 = O.synthetic_state(7)
    m = sifsr.ModelB_2(in_channels=2, downchannels=[16, 32, 64, 128], padding_mode="replicate", activation="ReLU",
                       bilinear=1, n_bridge_blocks=1)
    m.load_state_dict(sd0)
    m = m.to("cuda")
    opt = sifsr.FlatAdam(m.parameters(), lr=lr)
    ckpt = sifsr.train.ModelCheckpoint(1, 30)
    m, metrics = sifsr.train.fit(m, train_ds, val_ds, 1, bs, opt, alpha, gamma, "sr2", "cuda", ckpt, shuffle=False)
    # a one-epoch run leaves train_state None in the reference's checkpoint, so its train() never sets metrics['best_epoch']
    # (train_model_B_gradFTM.py:342-344) -- mirrored
    assert "best_epoch" not in metrics and ckpt.best_epoch == 1 and ckpt.train_state is None and len(ckpt.saved_state) == 104
    for k in ("train_loss", "train_dsloss", "train_perceploss", "train_psnr", "train_ssim",
              "val_loss", "val_dsloss", "val_perceploss", "val_psnr", "val_ssim"):
        assert len(metrics[k]) == 1 and np.isfinite(metrics[k][0]), k

    # ---- the oracle over the same two training batches and two validation batches
    names = O.param_names()
    sd = copy.deepcopy(sd0)
    adam = O.AdamState(names, lr)
    stats = train_ds.stats
    tl = DataLoader(train_ds, batch_size=bs, shuffle=False)
    tr = []
    for lst, lst_up, ndvi in tl:
        tr.append(O.train_step(sd, adam, lst, lst_up, ndvi, stats["mean_lst"], stats["std_lst"], alpha, gamma, "sr2"))
    ora_train = np.mean(np.array(tr), axis=0)          # (ds, pl, loss) epoch means
    va, pv, sv = [], [], []
    for lst, lst_up, ndvi in DataLoader(val_ds, batch_size=bs, shuffle=False):
        with torch.inference_mode():
            sr = O.modelb2_forward(sd, torch.cat((lst_up, ndvi), 1), training=False)
            va.append([float(v) for v in O.sr2_loss(sr, lst, ndvi, stats["mean_lst"], stats["std_lst"], alpha, gamma)])
        pv.append(O.psnr_skimage(sr.numpy(), lst_up.numpy())); sv.append(O.ssim_skimage(sr.numpy(), lst_up.numpy()))
    ora_val = np.mean(np.array(va), axis=0)
    got_train = np.array([metrics["train_dsloss"][0], metrics["train_perceploss"][0], metrics["train_loss"][0]])
    got_val = np.array([metrics["val_dsloss"][0], metrics["val_perceploss"][0], metrics["val_loss"][0]])
    print("train epoch means HIP", got_train, "oracle", ora_train, "| val HIP", got_val, "oracle", ora_val)
    assert np.all(np.abs(got_train - ora_train) <= 2e-3 * np.abs(ora_train))
    assert np.all(np.abs(got_val - ora_val) <= 2e-3 * np.abs(ora_val))
    assert abs(metrics["val_psnr"][0] - np.mean(pv)) <= 2e-2 and abs(metrics["val_ssim"][0] - np.mean(sv)) <= 2e-3
    # parameters after the epoch (two Adam steps): within the Adam bound of the oracle's, BN buffers at 1e-4
    p_hip = m.flat_parameters().detach().double().cpu()
    p_ora = C.flat(sd, names)
    assert float((p_hip - p_ora).abs().max()) <= 2.5 * lr * 2
    msd = m.state_dict()
    for k in msd:
        if k.endswith(("running_mean", "running_var")):
            assert rel_err(msd[k], sd[k]) < 1e-3, k
        if k.endswith("num_batches_tracked"):
            assert int(msd[k]) == int(sd[k]) == 2

    # ---- and step by step (first batch tight): the same loop by hand with the update criteria
    m2 = sifsr.ModelB_2(2); m2.load_state_dict(sd0); m2 = m2.cuda()
    opt2 = sifsr.FlatAdam(m2.parameters(), lr=lr)
    sd = copy.deepcopy(sd0); adam = O.AdamState(names, lr); hist = []
    p_before = m2.flat_parameters().detach().clone()
    for i, (lst, lst_up, ndvi) in enumerate(tl):
        ds, pl, loss = sifsr.train.train_step(m2, opt2, lst.cuda(), lst_up.cuda(), ndvi.cuda(), stats, alpha, gamma, "sr2")
        before_o = C.flat(sd, names)
        _, (ds_o, pl_o, loss_o), g_o = O.forward_backward(sd, lst, lst_up, ndvi, stats["mean_lst"], stats["std_lst"], alpha, gamma, "sr2")
        adam.step(sd, g_o); hist.append(g_o)
        tol = TOL if i == 0 else 2e-3
        for got, ref in ((ds, ds_o), (pl, pl_o), (loss, loss_o)):
            assert abs(float(got) - float(ref)) <= tol * abs(float(ref)), (i, float(got), float(ref))
        p_after = m2.flat_parameters().detach().clone()
        after_o = C.flat(sd, names)
        # significance from THIS step's gradients only: the two steps see different batches
        C.update_parity((p_after - p_before).double().cpu(), after_o - before_o, C.significant_mask([g_o], names),
                        p_after, after_o, lr, i + 1, what=f"config1 step {i}")
        p_before = p_after


def test_predict_py_call_pattern_with_cpu_tensors(sifsr, golden, dropin):
    """predict.py:43-48,68,96-101 replayed verbatim on a golden case: the model built from JSON-style keyword
    arguments and moved with ``.to(prediction_device)``, ``.eval()``, CPU input tensors from numpy blocks,
    ``torch.inference_mode()``, ``.numpy()[0,0,:,:] * std + mean`` on the output -- against the reference's own
    de-normalised output (golden ``y_denorm``).  Also model_perf_aster_formatds.py:182-203 (``lst_sr.numpy()``)."""
    ModelB_2 = dropin["model"].ModelB_2
    c = golden["cases"]["eval_w11_b21_B2"]
    stats = {"mean_lst": MEAN, "std_lst": STD, "mean_ndvi": 0.6452, "std_ndvi": 0.1683}
    modelB_parameters = {"in_channels": 2, "downchannels": [16, 32, 64, 128], "padding_mode": "replicate",
                         "activation": "ReLU", "bilinear": 1, "n_bridge_blocks": 1}
    prediction_device = "cuda"
    modelB = ModelB_2(in_channels=modelB_parameters['in_channels'],
                      downchannels=modelB_parameters['downchannels'],
                      padding_mode=modelB_parameters['padding_mode'],
                      activation=modelB_parameters['activation'],
                      bilinear=modelB_parameters['bilinear'],
                      n_bridge_blocks=modelB_parameters['n_bridge_blocks']).to(prediction_device)
    t = {k: v.clone() for k, v in O.synthetic_state(c["wseed"]).items()}       # stands for torch.load(weights_path, map_location=...)
    modelB.load_state_dict(t)
    modelB.eval()
    lst, lst_up, ndvi = O.synthetic_batch(c["bseed"], c["B"])
    outs = []
    for b in range(c["B"]):
        lst_up_np = lst_up[b, 0].numpy()                  # stands for us.upsampling(lst, (4,4)) -> numpy (256,256)
        ndvi_block = ndvi[b, 0].numpy() * stats['std_ndvi'] + stats['mean_ndvi']
        lst_up_t = (torch.tensor(lst_up_np).unsqueeze(0).unsqueeze(0))
        ndvi_t = (torch.tensor(ndvi_block).unsqueeze(0).unsqueeze(0) - stats['mean_ndvi']) / stats['std_ndvi']
        assert not lst_up_t.is_cuda and not ndvi_t.is_cuda
        with torch.inference_mode():
            lst_sr = modelB(torch.cat((lst_up_t, ndvi_t), dim=1)).numpy()[0, 0, :, :] * stats['std_lst'] + stats['mean_lst']
        assert isinstance(lst_sr, np.ndarray) and lst_sr.shape == (256, 256)
        outs.append(torch.from_numpy(lst_sr))
    check_digest(torch.stack(outs)[:, None], c["y_denorm"], TOL)
    # model_perf_aster_formatds.py:189-203
    x_lst_ndvi = torch.cat((lst_up[0:1], ndvi[0:1]), dim=1)
    with torch.inference_mode():
        lst_sr = modelB(x_lst_ndvi)
    lst_sr = lst_sr * stats['std_lst'] + stats['mean_lst']
    assert rel_err(torch.from_numpy(lst_sr.numpy()[0, 0, :, :]), outs[0]) < 1e-6
    # parameters on the CPU (predict.py's default --prediction_device cpu): still refused, loudly
    cpu_model = ModelB_2(**modelB_parameters)
    with pytest.raises(sifsr.SifsrError, match="no CPU compute path"):
        cpu_model(x_lst_ndvi)
    # a CPU batch in TRAINING mode with gradients on is the device mismatch the reference itself raises on
    modelB.train()
    with pytest.raises(sifsr.SifsrError, match="move the batch"):
        modelB(x_lst_ndvi)


def test_dropin_import_surface_runs_the_sr2_step(sifsr, golden, dropin):
    """train_model_B_gradFTM.py:27-29 + :94-121 with zero edits to the imports: ``from model import ModelB_2``,
    ``from dataset import ModisDatasetB``, ``import utils as us``; torch.optim.Adam, nn.HuberLoss, us.* -- against the
    golden step.  ``torch.save(model)`` (us.save_model, utils.py:826) pickles the class as ``model.ModelB_2``."""
    import io
    from model import ModelB_2
    from dataset import ModisDatasetB
    import utils as us
    c = golden["cases"]["train_sr2"]
    train_ds = ModisDatasetB('data/ModisDatasetB.csv', transf='norm', split='Train', time='day')
    assert set(train_ds.stats) >= {"mean_lst", "std_lst", "mean_ndvi", "std_ndvi"}
    device = "cuda"
    modelB = ModelB_2(in_channels=2, downchannels=[16, 32, 64, 128], padding_mode="replicate", activation="ReLU",
                      bilinear=1, n_bridge_blocks=1).to(device)
    modelB.load_state_dict(O.synthetic_state(c["wseed"]))
    optimizer = torch.optim.Adam(modelB.parameters(), lr=c["lr"])
    loss_fn = torch.nn.HuberLoss(reduction='mean', delta=1.0).to(device)
    checkpoint = us.model_checkpoint(2, 30)
    data = O.synthetic_batch(c["bseed"], c["B"])
    alpha, gamma, mean, std = c["alpha"], c["gamma"], MEAN, STD
    modelB.train()
    lst, lst_up, ndvi = data[0].to(device), data[1].to(device), data[2].to(device)
    optimizer.zero_grad()
    lst_ndvi = torch.cat((lst_up, ndvi), dim=1)
    lst_SR = modelB(lst_ndvi)
    lst_SR_unnorm = lst_SR * std + mean
    lst_SR_down = us.downscale_LST_SR_to_LR(lst_SR_unnorm)
    lst_SR_down = (lst_SR_down - mean) / std
    ds_loss = loss_fn(lst_SR_down, lst)
    grads_lst = lst_SR - us.get_output_ftm(lst_SR, mtf=0.25)
    grads_ndvi = ndvi - us.get_output_ftm(ndvi, mtf=0.25)
    percep_loss = loss_fn(grads_lst, gamma * grads_ndvi)
    loss = alpha * ds_loss + (1 - alpha) * percep_loss
    loss.backward()
    optimizer.step()
    rec = c["steps"][0]
    for got, key in ((ds_loss, "ds"), (percep_loss, "pl"), (loss, "loss")):
        assert abs(got.item() - rec[key]) < TOL * abs(rec[key])
    psnr = us.psnr_skimage(lst_SR.detach().cpu().numpy(), lst_up.detach().cpu().numpy())       # :126
    ssim = us.ssim_skimage(lst_SR.detach().cpu().numpy(), lst_up.detach().cpu().numpy())       # :127
    assert abs(psnr - O.psnr_skimage(lst_SR.detach().cpu().numpy(), data[1].numpy())) < 2e-3
    assert abs(ssim - O.ssim_skimage(lst_SR.detach().cpu().numpy(), data[1].numpy())) < 1e-4
    metrics = {"val_loss": [loss.item()]}
    checkpoint.test_update(modelB, metrics, 'val_loss', 1)
    assert checkpoint.best_epoch == 1 and len(checkpoint.saved_state) == 104
    metrics["val_loss"].append(loss.item() + 1.0)
    checkpoint.test_update(modelB, metrics, 'val_loss', 2)
    assert checkpoint.train_state == 'break' and checkpoint.best_epoch == 1      # last epoch with a non-zero counter
    modelB.load_state_dict(checkpoint.saved_state)
    np.testing.assert_array_equal(us.generate_psf_kernel(1.0, 4, 0.25, None),
                                  np.array(golden["cases"]["psf_0.25"]["kernel9x9"], dtype=np.float32).reshape(9, 9))
    up = us.upsampling(data[0][0, 0].numpy(), (4, 4))
    assert up.shape == (256, 256) and rel_err(torch.from_numpy(up), data[1][0, 0]) < 1e-5
    buf = io.BytesIO(); torch.save(modelB, buf)
    buf.seek(0)
    again = torch.load(buf, weights_only=False)            # our own file
    assert type(again).__module__ == "model" and type(again).__name__ == "ModelB_2"
    with pytest.raises(NotImplementedError):
        us.read_LST("x.hdf", "day")


@pytest.mark.parametrize("kind", ["sr2", "sr1"])
def test_train_step_gradient_shortcut_equals_loss_backward(sifsr, kind):
    """train_step seeds the backward with d loss / d sr from the fused loss op (sr.backward(dsr)) instead of loss.backward():
    same losses, bit-identical gradients."""
    import sifsr as S
    torch.manual_seed(3)
    dev = torch.device("cuda", 0)
    stats = dict(S.dataset.DEFAULT_STATS)
    lst, lst_up, ndvi = S.dataset.synthetic_device_batch(2, dev, seed=5, hr=64)
    sd = O.synthetic_state(17)
    grads = []
    for shortcut in (False, True):
        m = S.ModelB_2(2, [16, 32, 64, 128], "replicate", "ReLU", 1, 1).to(dev)
        m.load_state_dict(sd)
        m.train()
        sr = m(torch.cat((lst_up, ndvi), dim=1))
        if shortcut:
            ds, pl, loss, dsr = S.sif_ops.sif_loss_with_grad(kind, sr, lst, ndvi, stats["mean_lst"], stats["std_lst"], 0.5, -0.25)
            sr.backward(dsr)
        else:
            ds, pl, loss = S.sif_loss(kind, sr, lst, ndvi, stats["mean_lst"], stats["std_lst"], 0.5, -0.25)
            loss.backward()
        torch.cuda.synchronize()
        grads.append((float(ds), float(pl), float(loss), m.flat_grad().detach().clone()))
    assert grads[0][:3] == grads[1][:3]
    assert torch.equal(grads[0][3], grads[1][3])


def test_train_mode_forward_without_grad(sifsr):
    """``model.train()`` under ``torch.no_grad()`` (BatchNorm recalibration / frozen-model passes): nn.Module accepts it,
    uses batch statistics and updates the running statistics.  Round 1 sized the workspace for an eval forward and the
    library rejected it (status 1003)."""
    sd = O.synthetic_state(3)
    lst, lst_up, ndvi = O.synthetic_batch(5, 2)
    x = torch.cat((lst_up, ndvi), 1)
    sd_o = copy.deepcopy(sd)
    y_o = O.modelb2_forward(sd_o, x, training=True)
    m = sifsr.ModelB_2(2); m.load_state_dict(sd); m = m.cuda().train()
    with torch.no_grad():
        y = m(x.cuda())
    assert not y.requires_grad and rel_err(y, y_o) < TOL
    msd = m.state_dict()
    for k in msd:
        if k.endswith(("running_mean", "running_var")):
            assert rel_err(msd[k], sd_o[k]) < 1e-5, k
        if k.endswith("num_batches_tracked"):
            assert int(msd[k]) == 1
    for p in m.parameters():                   # every parameter frozen, grad mode on: same thing
        p.requires_grad_(False)
    y2 = m(x.cuda())
    assert not y2.requires_grad
    with torch.inference_mode():
        y3 = m(x.cuda())
    assert torch.equal(y2, y) and torch.equal(y3, y)      # batch statistics do not depend on the running buffers
    assert int(m.state_dict()["inbloc.bloc.1.num_batches_tracked"]) == 3


def test_capture_with_live_earlier_graph_is_refused(sifsr):
    """capturing a training step while the autograd graph of an earlier un-captured step is alive used to
    end in a failed hipStreamEndCapture; now it is refused with a clear error before anything is recorded, and the same
    capture succeeds once the earlier outputs are dropped."""
    torch.manual_seed(2)
    m = sifsr.ModelB_2(2).cuda().train()
    lst, lst_up, ndvi = (t.cuda() for t in O.synthetic_batch(9, 1))
    x = torch.cat((lst_up, ndvi), 1)

    def step():
        for p in m.parameters():
            p.grad = None
        sr = m(x)
        _, _, loss = sifsr.sif_loss("sr2", sr, lst, ndvi, MEAN, STD, 0.5, -0.25)
        loss.backward()
        return loss

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
        held = m(x)                       # a live output of an un-captured training forward: its graph stays alive
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with pytest.raises(sifsr.SifsrError, match="earlier"):
        with torch.cuda.graph(g):
            step()
    torch.cuda.synchronize()
    del held, g
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        out = step().detach()
    g2.replay()
    torch.cuda.synchronize()
    assert torch.isfinite(out)


def test_fit_early_stop_restores_the_best_state(sifsr):
    """train.fit over (at most) 3 epochs with patience 1 and a monitored metric that gets worse after the first epoch: the loop
    must switch train -> eval -> train every epoch, break when the patience is spent and load the saved best state back into
    the flat parameter buffer (train_model_B_gradFTM.py:338-352) -- parameters AND BatchNorm buffers equal the checkpoint's
    copy, and the nn.Parameters still alias the flat buffer afterwards (the next step trains the restored weights).  The
    checkpoint sees a scripted validation loss (1, 2, 3): what is under test is the loop's plumbing, not the optimisation."""
    train_ds = sifsr.dataset.ModisDatasetB("data/ModisDatasetB.csv", transf="norm", split="Train", time="day", length=8)
    val_ds = sifsr.dataset.ModisDatasetB("data/ModisDatasetB.csv", transf="norm", split="Val", time="day", length=8)
    m = sifsr.ModelB_2(2, [16, 32, 64, 128], "replicate", "ReLU", 1, 1)
    m.load_state_dict(O.synthetic_state(11))
    m = m.to("cuda")
    opt = sifsr.FlatAdam(m.parameters(), lr=1e-3)
    seen = []

    class Scripted(sifsr.train.ModelCheckpoint):
        def test_update(self, model, metrics, key, epoch):
            assert not model.training                       # called right after the validation pass (eval mode)
            assert np.isfinite(metrics[key][-1])
            super().test_update(model, {key: [float(epoch)]}, key, epoch)
            seen.append((epoch, self.train_state))

    ck = Scripted(3, patience=1)
    m, metrics = sifsr.train.fit(m, train_ds, val_ds, 3, 4, opt, 0.5, -0.25, "sr2", "cuda", ck, shuffle=False)
    assert seen == [(1, None), (2, "break")], seen
    assert metrics["best_epoch"] == ck.best_epoch == 1 and len(metrics["train_loss"]) == len(metrics["val_loss"]) == 2
    sd = m.state_dict()
    for k, v in ck.saved_state.items():
        assert torch.equal(sd[k].cpu(), v.cpu()), k
    assert int(sd["inbloc.bloc.1.num_batches_tracked"]) == 2     # two training batches of epoch 1 (epoch 2's four were rolled back)
    flat = m.flat_parameters()
    off = 0
    for p_ in m.parameters():                                # still views of the one flat buffer
        assert p_.data_ptr() == flat.data_ptr() + 4 * off
        off += p_.numel()
    before = flat.clone()
    lst, lst_up, ndvi = (torch.from_numpy(np.stack([train_ds[i][j] for i in range(4)])).cuda() for j in range(3))
    sifsr.train.train_step(m, sifsr.FlatAdam(m.parameters(), lr=1e-3), lst, lst_up, ndvi, train_ds.stats, 0.5, -0.25, "sr2")
    torch.cuda.synchronize()
    assert not torch.equal(m.flat_parameters(), before)
