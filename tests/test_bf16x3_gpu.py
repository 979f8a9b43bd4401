"""Split-bf16 compute mode ("fp32 on the bf16 matrix cores", ``model.compute_dtype = "bf16x3"``, compute mode 2 of the C ABI):
every conv operand is split exactly into three bf16 terms while staging and six of the nine cross products are accumulated
in fp32.  Unlike the bf16 mode this is not a reduced-precision path: the checks below use the SAME fp32 references and the
same 1e-4 bar as the fp32 kernels, and additionally ask that the error against a float64 reference is of the order of the
fp32 MFMA kernel's own."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sif_oracle as O
from tests.conftest import rel_err

pytestmark = pytest.mark.gpu
MEAN, STD = 307.2378, 5.5698
TOL = 1e-4


@pytest.fixture(scope="module")
def sifsr():
    import sifsr as pkg
    assert torch.cuda.is_available()
    return pkg


@pytest.mark.parametrize("shape", [(16, 16, 2, 48, 64), (32, 16, 1, 32, 32), (64, 32, 2, 24, 40), (16, 64, 1, 40, 24),
                                   (128, 64, 1, 16, 32), (64, 128, 1, 16, 16)])
def test_conv_ops_match_fp64_like_fp32_kernels(sifsr, shape):
    """forward and input gradient of one conv, all kernel variants (cout blocks 1, 2, 4, 8; partial tiles)"""
    from sifsr import _lib as L
    cin, cout, B, H, W = shape
    rs = np.random.RandomState(cin + cout + H)
    x = torch.from_numpy(rs.standard_normal((B, cin, H, W)).astype(np.float32))
    w = torch.from_numpy((rs.standard_normal((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5).astype(np.float32))
    dy = torch.from_numpy(rs.standard_normal((B, cout, H, W)).astype(np.float32))
    conv = lambda a, b: F.conv2d(F.pad(a, (1, 1, 1, 1), mode="replicate"), b)
    y64 = conv(x.double(), w.double())
    xa = x.double().clone().requires_grad_(True)
    (gx64,) = torch.autograd.grad(conv(xa, w.double()), xa, dy.double())
    S = torch.cuda.current_stream().cuda_stream
    wf = torch.empty(9 * cin * cout, device="cuda"); wd = torch.empty(4 * 9 * cin * cout, device="cuda")
    L.call("sifsr_pack_conv_weights", w.cuda(), cin, cout, wf, wd, S)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    nchw = lambda t: t.permute(0, 3, 1, 2).cpu()
    y3 = torch.empty(B, H, W, cout, device="cuda"); gx3 = torch.empty(B, H, W, cin, device="cuda")
    y1 = torch.empty_like(y3); gx1 = torch.empty_like(gx3)
    L.call("sifsr_conv3x3_fwd_bf16x3", nhwc(x), cin, None, None, None, 0, None, None, wd, y3, cout, None, B, H, W, S)
    L.call("sifsr_conv3x3_dgrad_bf16x3", nhwc(dy), cout, wd, cin, gx3, cin, None, 0, None, B, H, W, S)
    L.call("sifsr_conv3x3_fwd", nhwc(x), cin, None, None, None, 0, None, None, wf, y1, cout, None, B, H, W, S)
    L.call("sifsr_conv3x3_dgrad", nhwc(dy), cout, wd, w.cuda(), cin, gx1, cin, None, 0, None, B, H, W, S)
    torch.cuda.synchronize()
    e3, e1 = rel_err(nchw(y3), y64), rel_err(nchw(y1), y64)
    g3, g1 = rel_err(nchw(gx3), gx64), rel_err(nchw(gx1), gx64)
    print(f"{shape}: fwd err vs f64: split {e3:.2e} fp32-MFMA {e1:.2e} | dgrad: split {g3:.2e} fp32-MFMA {g1:.2e}")
    assert e3 < 2e-6 and g3 < 2e-6            # fp32-level (the bf16 mode sits at 3e-3 here)
    assert e3 < 8 * e1 + 1e-7 and g3 < 8 * g1 + 1e-7


def test_model_forward_backward_and_step_at_fp32_bar(sifsr):
    """eval forward, training forward + SIF loss + backward against the fp32 oracle at the fp32 path's own tolerance"""
    sd = O.synthetic_state(51)
    lst, lst_up, ndvi = O.synthetic_batch(53, 2)
    x = torch.cat((lst_up, ndvi), 1)
    y_ref = O.modelb2_forward(copy.deepcopy(sd), x, training=False)
    sr_o, (ds_o, pl_o, loss_o), g_o = O.forward_backward(copy.deepcopy(sd), lst, lst_up, ndvi, MEAN, STD, 0.5, -0.25, "sr2")

    def run(mode):
        m = sifsr.ModelB_2(2, [16, 32, 64, 128], "replicate", "ReLU", 1, 1)
        m.load_state_dict(copy.deepcopy(sd), strict=True)
        m.compute_dtype = mode
        m = m.cuda()
        with torch.inference_mode():
            y = m.eval()(x.cuda()).cpu()
        m.train()
        sr = m(x.cuda())
        ds, pl, loss = sifsr.sif_loss("sr2", sr, lst.cuda(), ndvi.cuda(), MEAN, STD, 0.5, -0.25)
        loss.backward()
        torch.cuda.synchronize()
        g = {n: p.grad.detach().cpu() for n, p in m.named_parameters()}
        return y, sr.detach().cpu(), float(loss.detach()), g

    y3, sr3, loss3, g3 = run("bf16x3")
    y1, sr1, loss1, g1 = run("fp32")
    print(f"eval fwd vs oracle: split {rel_err(y3, y_ref):.2e} fp32 {rel_err(y1, y_ref):.2e}; train sr: split {rel_err(sr3, sr_o):.2e} "
          f"fp32 {rel_err(sr1, sr_o):.2e}; loss rel: split {abs(loss3 - float(loss_o)) / abs(float(loss_o)):.2e}")
    assert rel_err(y3, y_ref) < TOL and rel_err(sr3, sr_o) < TOL
    assert abs(loss3 - float(loss_o)) < TOL * abs(float(loss_o))
    # gradients: parameter gradients are conditioned on the ReLU masks: two correct fp32 pipelines
    # differ by mask flips at pre-activations within rounding of zero, so the bar is the fp32 MFMA path's own distance
    # to the oracle on the same inputs, not 1e-4
    e3 = {n: rel_err(g3[n], g_o[n]) for n in g3 if n in g_o}
    e1 = {n: rel_err(g1[n], g_o[n]) for n in g1 if n in g_o}
    assert len(e3) > 40
    print(f"grads vs oracle: split worst {max(e3.values()):.2e} median {sorted(e3.values())[len(e3) // 2]:.2e}; "
          f"fp32 path worst {max(e1.values()):.2e} median {sorted(e1.values())[len(e1) // 2]:.2e}")
    assert max(e3.values()) < 2 * max(e1.values()) + TOL
    assert sorted(e3.values())[len(e3) // 2] < 2 * sorted(e1.values())[len(e1) // 2] + TOL


def test_rejects_unknown_mode(sifsr):
    m = sifsr.ModelB_2(2).cuda()
    m.compute_dtype = "fp16"
    with pytest.raises(sifsr.SifsrError):
        m(torch.zeros(1, 2, 64, 64, device="cuda"))
