"""Error behaviour of the boundary (SURVEY.md §8 b): bad shapes / pointers / workspaces come back as the documented
status codes (and SifsrError in the Python layer), never as a fault; unsupported constructor options raise at
construction as in DESIGN.md."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
SHAPE, ARG, WORKSPACE = 1001, 1002, 1003


@pytest.fixture(scope="module")
def lib():
    import sifsr
    from sifsr import _lib
    assert torch.cuda.is_available()
    return _lib.lib()


def test_c_abi_status_codes(lib):
    S = torch.cuda.current_stream().cuda_stream
    x = torch.zeros(1, 2, 64, 64, device="cuda")
    sr = torch.zeros(1, 1, 64, 64, device="cuda")
    p = torch.zeros(282705, device="cuda"); r = torch.zeros(1184, device="cuda"); n = torch.zeros(17, dtype=torch.int64, device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    nbytes = lib.sifsr_model_workspace_bytes(1, 64, 64, 0)
    assert nbytes > 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    f = lib.sifsr_model_forward
    ok = f(P(x), P(sr), P(p), P(r), P(n), P(ws), nbytes, 1, 64, 64, 0, 0.1, 1e-5, ctypes.c_void_p(S))
    assert ok == 0
    assert f(None, P(sr), P(p), P(r), P(n), P(ws), nbytes, 1, 64, 64, 0, 0.1, 1e-5, ctypes.c_void_p(S)) == ARG
    assert f(P(x), P(sr), P(p), P(r), P(n), P(ws), nbytes // 2, 1, 64, 64, 0, 0.1, 1e-5, ctypes.c_void_p(S)) == WORKSPACE
    assert f(P(x), P(sr), P(p), P(r), P(n), P(ws), nbytes, 1, 60, 64, 0, 0.1, 1e-5, ctypes.c_void_p(S)) == SHAPE   # not a multiple of 8
    assert f(P(x), P(sr), P(p), P(r), P(n), P(ws), nbytes, 0, 64, 64, 0, 0.1, 1e-5, ctypes.c_void_p(S)) == SHAPE
    assert lib.sifsr_model_workspace_bytes(1, 20, 64, 1) == 0          # unsupported shape: no workspace size
    # backward needs the training workspace of a training forward
    g = torch.zeros_like(p)
    assert lib.sifsr_model_backward(P(x), P(sr), P(p), P(g), P(ws), nbytes, 1, 64, 64, ctypes.c_void_p(S)) == WORKSPACE
    # compute mode other than 0 / 1
    assert lib.sifsr_model_forward_ex(P(x), P(sr), P(p), P(r), P(n), P(ws), nbytes, 1, 64, 64, 0, 0.1, 1e-5, 7, ctypes.c_void_p(S)) == ARG
    # operators
    y = torch.zeros(1, 20, 20, 24, device="cuda")
    assert lib.sifsr_conv3x3_fwd(P(y), 24, None, None, None, 0, None, None, P(p), P(y), 16, None, 1, 20, 20, ctypes.c_void_p(S)) == SHAPE
    assert lib.sifsr_psnr_ssim(P(sr), P(sr), 1, 4, 4, P(ws), nbytes, P(p), ctypes.c_void_p(S)) == SHAPE       # smaller than the 7x7 window
    assert lib.sifsr_fft2_attenuation(P(sr), 1, 48, 64, P(ws), nbytes, P(p), None, ctypes.c_void_p(S)) == SHAPE   # not a power of two
    assert lib.sifsr_l4pool4(P(sr), P(p), 1, 6, 8, ctypes.c_void_p(S)) == SHAPE
    assert lib.sifsr_tiles_prepare(P(sr), P(sr), P(p), 1, 1, 128, 0, 0, 0, 0.0, 1.0, 0.0, 1.0, 0, ctypes.c_void_p(S)) == SHAPE   # window > 64
    torch.cuda.synchronize()


def test_python_layer_errors():
    import sifsr
    m = sifsr.ModelB_2(2).cuda()
    with pytest.raises(sifsr.SifsrError):
        m(torch.zeros(1, 2, 64, 64))                    # CPU tensor
    with pytest.raises(sifsr.SifsrError):
        m(torch.zeros(1, 3, 64, 64, device="cuda"))     # channels
    with pytest.raises(sifsr.SifsrError):
        m(torch.zeros(1, 2, 44, 64, device="cuda"))     # not a multiple of 8
    assert m(torch.zeros(1, 2, 64, 64, device="cuda", dtype=torch.float64)).dtype == torch.float32   # converted, as .float()
    with pytest.raises(NotImplementedError):
        sifsr.downscale_LST_SR_to_LR(torch.zeros(1, 1, 64, 64, device="cuda"), deci_type="norm-L4")
    with pytest.raises(sifsr.SifsrError):
        # the /4 decimation needs multiples of 4
        sifsr.sif_loss("sr2", torch.zeros(1, 1, 50, 48, device="cuda"), torch.zeros(1, 1, 12, 12, device="cuda"),
                       torch.zeros(1, 1, 50, 48, device="cuda"), 300.0, 5.0, 0.5, -0.25)
    # eval-mode backward is refused loudly, not silently wrong
    m.eval()
    y = m(torch.zeros(1, 2, 64, 64, device="cuda", requires_grad=True))
    if y.requires_grad:
        with pytest.raises(NotImplementedError):
            y.sum().backward()
    # empty batch
    with pytest.raises(sifsr.SifsrError):
        m(torch.zeros(0, 2, 64, 64, device="cuda"))
