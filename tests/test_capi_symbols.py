"""CPU: the C-ABI shared library loads and exports every symbol include/sifsr_hip.h declares; the
host-only introspection entry points agree with the oracle's view of the reference layout.
No compute call is made (there is no GPU here)."""
import ctypes
import os

import pytest

from oracle import sif_oracle as O


@pytest.fixture(scope="module")
def L():
    import sifsr  # noqa: F401
    from sifsr import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib


def test_every_declared_symbol_is_exported(L):
    names = L.declared_symbols()
    assert len(names) >= 40
    handle = ctypes.CDLL(L.LIB_PATH)
    missing = [n for n in names if not hasattr(handle, n)]
    assert not missing, missing
    # and nothing undeclared leaks out with our prefix
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln and "sifsr_" in ln}
    assert exported == set(names), exported ^ set(names)


def test_header_parser_types(L):
    d = L.parse_header()
    ret, args = d["sifsr_model_forward"]
    assert ret is ctypes.c_int
    assert [a for a, _ in args] == ["x", "sr", "params", "running", "nbt", "workspace", "workspace_bytes", "B", "H", "W",
                                    "training", "momentum", "eps", "stream"]
    assert args[6][1] is ctypes.c_size_t and args[11][1] is ctypes.c_float and args[0][1] is ctypes.c_void_p
    assert d["sifsr_model_workspace_bytes"][0] is ctypes.c_size_t
    assert d["sifsr_bn_finalize"][1][3][1] is ctypes.c_double


def test_layer_table_matches_reference_layout(L):
    assert L.call("sifsr_abi_version") == 3
    assert L.call("sifsr_num_params") == 282705
    tab = (ctypes.c_int * (17 * 8))()
    assert L.call("sifsr_layer_table", tab, 17) == 17
    off = run = ch = 0
    spec = {n: s for n, s, _ in O.state_dict_spec()}
    levels = [0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 2, 2, 1, 1, 0, 0]
    for l, (conv, bn, cin, cout) in enumerate(O.CONV_BN_LAYERS):
        row = list(tab[l * 8:(l + 1) * 8])
        assert row[:3] == [cin, cout, levels[l]]
        assert row[3] == off; off += cout * cin * 9            # conv weight, OIHW
        assert spec[conv + ".weight"] == (cout, cin, 3, 3)
        assert row[4] == off; off += cout                      # BN weight
        assert row[5] == off; off += cout                      # BN bias
        assert row[6] == run; run += 2 * cout
        assert row[7] == ch; ch += cout
    assert off + 144 + 1 == 282705
    assert run == L.call("sifsr_num_running") == 1184


def test_workspace_sizes(L):
    inf = L.call("sifsr_model_workspace_bytes", 1, 256, 256, 0)
    trn = L.call("sifsr_model_workspace_bytes", 1, 256, 256, 1)
    assert 0 < inf < trn
    # activations: conv outputs 27.5 MiB + pooled/residual/upsampled 10.5 MiB per patch (SURVEY.md §8 a)
    assert 38 * 2**20 < inf < 48 * 2**20
    assert L.call("sifsr_model_workspace_bytes", 64, 256, 256, 1) > 30 * trn   # fixed-size scratch (slabs) amortises
    assert L.call("sifsr_model_workspace_bytes", 1, 200, 256, 1) > 0           # any multiples of 8 (>= 24), as the reference
    assert L.call("sifsr_model_workspace_bytes", 1, 204, 256, 1) == 0          # not a multiple of 8
    assert L.call("sifsr_model_workspace_bytes", 1, 16, 256, 1) == 0           # too small for three poolings + a 3x3 window
    assert L.call("sifsr_model_workspace_bytes", 0, 256, 256, 1) == 0
    reg = (ctypes.c_size_t * 56)()
    assert L.call("sifsr_model_workspace_regions", 2, 256, 256, reg, 56) == 56
    r = list(reg)
    # distinct and 256-byte aligned; the three dyB slots (43..45) are one shared scratch since dL/dy is no longer stored
    assert len(set(r)) == 54 and r[43] == r[44] == r[45] and all(v % 64 == 0 for v in r)
    assert L.call("sifsr_sif_loss_workspace_bytes", 2, 2, 256, 256) > 2 * 256 * 256 * 4
