"""CPU checks of the oracle's restatements for the rows next to the hot path (input pipeline, PSNR/SSIM):
closed-form cases, since OpenCV / scikit-image are not installed (parity with them is unpinned)."""
import numpy as np
import torch

from oracle import sif_oracle as O


def test_bicubic_prepare_closed_form():
    # bicubic x4 reproduces constants and (away from the clamped border) linear ramps exactly
    T, w = 2, 16
    lst = torch.full((T, 1, w, w), 3.0)
    ndvi = torch.linspace(-2, 2, 4 * w).repeat(T, 1, 4 * w, 1)
    x = O.prepare_tiles(lst, ndvi, {"mean_lst": 1.0, "std_lst": 2.0, "mean_ndvi": 0.5, "std_ndvi": 0.25}, clip_ndvi=True)
    assert x.shape == (T, 2, 4 * w, 4 * w)
    assert torch.allclose(x[:, 0], torch.full_like(x[:, 0], 1.0), atol=1e-6)
    assert torch.allclose(x[:, 1], (ndvi[:, 0].clamp(-1, 1) - 0.5) / 0.25)
    # a row ramp against the cubic-convolution formula written out (A = -0.75, half-pixel centres, edge clamp:
    # the INTER_CUBIC / ATen kernel; it does NOT reproduce linear ramps exactly, only A = -0.5 would)
    ramp = torch.arange(w, dtype=torch.float32).repeat(1, 1, w, 1)
    up = O.prepare_tiles(ramp, torch.zeros(1, 1, 4 * w, 4 * w))[0, 0]
    A = -0.75
    k1 = lambda x: ((A + 2) * x - (A + 3)) * x * x + 1
    k2 = lambda x: ((A * x - 5 * A) * x + 8 * A) * x - 4 * A
    expect = np.zeros(4 * w)
    for X in range(4 * w):
        sx = (X + 0.5) / 4 - 0.5
        ix = int(np.floor(sx)); t = sx - ix
        c = [k2(t + 1), k1(t), k1(1 - t), k2(2 - t)]
        expect[X] = sum(c[j] * min(max(ix - 1 + j, 0), w - 1) for j in range(4))
    assert np.allclose(up[5].numpy(), expect, atol=1e-5)


def test_psnr_ssim_closed_form():
    rs = np.random.RandomState(0)
    t = rs.standard_normal((2, 1, 40, 48)).astype(np.float32)
    assert abs(O.ssim_skimage(t, t) - 1.0) < 1e-6
    d = 0.25
    rng = float(t.max() - t.min())
    assert abs(O.psnr_skimage(t + d, t) - 10 * np.log10(rng ** 2 / d ** 2)) < 1e-4
    # SSIM of an image against its negative-correlated copy is well below 1 and symmetric in its arguments
    p = t[:, :, ::-1, :].copy()
    assert O.ssim_skimage(p, t) < 0.5
    assert abs(O.ssim_skimage(p, t) - O.ssim_skimage(t, p)) < 1e-5


def test_predict_granule_skips_ragged_tiles():
    sd = O.synthetic_state(1)
    lst = torch.full((70, 64), 300.0)
    ndvi = torch.zeros(280, 256)
    out = O.predict_granule(sd, lst, ndvi, {"mean_lst": 300.0, "std_lst": 5.0, "mean_ndvi": 0.0, "std_ndvi": 1.0})
    assert out.shape == (280, 256)
    assert out[256:].abs().max() == 0 and out[:256].abs().max() > 0
