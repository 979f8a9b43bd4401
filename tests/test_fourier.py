"""Fourier-domain evaluation row: oracle vs the reference's golden vectors on CPU, and the
hand-written FFT / ring kernels vs both on the GPU."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import sif_oracle as O
from tests.golden.make_golden_fourier import CASES, fourier_case

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden_fourier_v1.json")))


def test_oracle_matches_reference_golden():
    assert [(c["seed"], c["H"], c["W"]) for c in GOLD["cases"]] == CASES
    for c in GOLD["cases"]:
        imgs = fourier_case(c["seed"], c["H"], c["W"])
        specs = [O.attenuation_spectrum(O.fft2_magnitude_shifted(im)) for im in imgs]
        for s, g in zip(specs, c["spectra"]):
            assert len(s) == len(g) == min(c["H"] // 2, c["W"] // 2)
            assert np.allclose(s, g, rtol=0, atol=1e-6)
        rb, xb, pb = specs
        assert np.allclose(O.frr_fro_fru(pb, rb, xb), [c["FRR"], c["FRO"], c["FRU"]], rtol=1e-7, atol=1e-9)


def test_host_scores_match_reference_golden():
    import sifsr
    for c in GOLD["cases"]:
        rb, xb, pb = c["spectra"]
        f = sifsr.fourier
        assert np.allclose([f.get_FRR(pb, rb, xb), f.get_FRO(pb, rb, xb), f.get_FRU(pb, rb, xb)],
                           [c["FRR"], c["FRO"], c["FRU"]], rtol=1e-9, atol=1e-12)
        # the drop-in named like utils.py:598 (takes the shifted magnitude), on CPU tensors
        im = fourier_case(c["seed"], c["H"], c["W"])[0]
        s = f.compute_2D_attenuation_spectra(torch.from_numpy(O.fft2_magnitude_shifted(im)))
        assert np.allclose(s, c["spectra"][0], rtol=0, atol=1e-6)


@pytest.mark.gpu
def test_hip_fft_and_spectra_vs_oracle_and_golden():
    import sifsr
    for c in GOLD["cases"]:
        imgs = fourier_case(c["seed"], c["H"], c["W"])
        x = torch.from_numpy(np.stack(imgs)).cuda()
        mag = sifsr.fourier.fft2_magnitude(x).cpu().numpy()
        spec = sifsr.fourier.attenuation_spectra(x).cpu().numpy()
        for i, im in enumerate(imgs):
            ref = O.fft2_magnitude_shifted(im)
            assert np.abs(mag[i] - ref).max() < 1e-6 * ref.max()          # float32 output of a float64 transform
            # dB values stored as float32: 1e-4 dB absolute (|dB| <= ~100)
            assert np.allclose(spec[i], c["spectra"][i], rtol=0, atol=2e-4), np.abs(spec[i] - c["spectra"][i]).max()
        rb, xb, pb = spec
        got = [sifsr.fourier.get_FRR(pb, rb, xb), sifsr.fourier.get_FRO(pb, rb, xb), sifsr.fourier.get_FRU(pb, rb, xb)]
        assert np.allclose(got, [c["FRR"], c["FRO"], c["FRU"]], rtol=1e-4, atol=1e-6)
        single = sifsr.fourier.attenuation_spectra(x[0])
        assert single.shape == (min(c["H"] // 2, c["W"] // 2),) and torch.equal(single.cpu(), torch.from_numpy(spec[0]))
    with pytest.raises(sifsr.SifsrError):
        sifsr.fourier.fft2_magnitude(torch.zeros(1, 100, 256).cuda())      # sides must be powers of two
