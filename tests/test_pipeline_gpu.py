"""GPU parity of the rows next to the hot path: the on-device input pipeline / granule
block loop (dataset.py:134-142, predict.py:84-103) and the train-time metrics (utils.py:548-578), against
the oracle's restatement.  OpenCV and scikit-image are absent: these two rows are pinned against torch's bicubic
and a numpy/scipy restatement of scikit-image 0.22 -- parity with cv2 / skimage themselves is unpinned."""
import copy

import numpy as np
import pytest
import torch

from oracle import sif_oracle as O
from tests.conftest import rel_err

pytestmark = pytest.mark.gpu
STATS = {"mean_lst": 307.2378, "std_lst": 5.5698, "mean_ndvi": 0.6452, "std_ndvi": 0.1683}


@pytest.fixture(scope="module")
def sifsr():
    import sifsr as pkg
    assert torch.cuda.is_available()
    return pkg


@pytest.mark.parametrize("win", [64, 16])
def test_prepare_tiles(sifsr, win):
    rs = np.random.RandomState(win)
    T = 5
    lst = torch.from_numpy((rs.standard_normal((T, 1, win, win)) * 5.5 + 307).astype(np.float32))
    ndvi = torch.from_numpy((rs.standard_normal((T, 1, 4 * win, 4 * win)) * 0.5 + 0.6).astype(np.float32))
    for stats, clip in ((STATS, True), (None, False)):
        ref = O.prepare_tiles(lst, ndvi, stats, clip)
        x = sifsr.pipeline.prepare_tiles(lst.cuda(), ndvi.cuda(), stats, clip)
        assert x.shape == ref.shape
        assert (x.cpu() - ref).abs().max().item() < 1e-5 * max(1.0, ref.abs().max().item())


def test_predict_granule(sifsr):
    """200 x 136 LST granule: 3 x 2 full tiles, ragged right/bottom edges skipped (stay 0) as in predict.py."""
    rs = np.random.RandomState(9)
    h, w = 200, 136
    lst_g = torch.from_numpy((rs.standard_normal((h, w)) * 5.5 + 307).astype(np.float32))
    ndvi_g = torch.from_numpy((rs.standard_normal((4 * h, 4 * w)) * 0.6 + 0.5).astype(np.float32))
    sd = O.synthetic_state(3)
    ref = O.predict_granule(copy.deepcopy(sd), lst_g, ndvi_g, STATS)
    m = sifsr.ModelB_2(2, [16, 32, 64, 128], "replicate", "ReLU", 1, 1)
    m.load_state_dict(sd, strict=True)
    out = sifsr.predict.predict_granule(m.cuda(), lst_g.cuda(), ndvi_g.cuda(), STATS, batch=4)
    assert out.shape == ref.shape
    assert rel_err(out.cpu(), ref) < 1e-4
    assert out[4 * 192:, :].abs().max().item() == 0 and out[:, 4 * 128:].abs().max().item() == 0


@pytest.mark.parametrize("shape,kelvin", [((3, 1, 256, 256), False), ((2, 1, 96, 80), True)])
def test_psnr_ssim(sifsr, shape, kelvin):
    rs = np.random.RandomState(shape[2])
    t = rs.standard_normal(shape).astype(np.float32)
    # a smooth-ish target and a noisy prediction of it (what train_model_B_gradFTM.py:126-127 compares)
    t = (t + np.roll(t, 1, 2) + np.roll(t, 1, 3) + np.roll(t, (1, 1), (2, 3))) / 2
    p = t + 0.3 * rs.standard_normal(shape).astype(np.float32)
    if kelvin:
        t, p = t * 5.5 + 307, p * 5.5 + 307
    psnr_ref, ssim_ref = O.psnr_skimage(p, t), O.ssim_skimage(p, t)
    psnr, ssim = sifsr.metrics.psnr_ssim(torch.from_numpy(p).cuda(), torch.from_numpy(t).cuda())
    assert abs(float(psnr) - psnr_ref) < 1e-4 * abs(psnr_ref)
    # float32 variance terms cancel (uxx - ux*ux) at Kelvin magnitudes, in skimage as here: looser there
    assert abs(float(ssim) - ssim_ref) < (2e-3 if kelvin else 1e-4) * abs(ssim_ref)
    assert float(sifsr.metrics.psnr_skimage(torch.from_numpy(p).cuda(), torch.from_numpy(t).cuda())) == float(psnr)
