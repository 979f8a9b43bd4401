"""Every diagnostic build flavour of the MFMA kernels (csrc/diag.h, SIFSR_PK_MODE in common.h) must keep COMPILING: they are
never shipped and no GPU test runs them, so this CPU test builds each once (device code only, gfx950) -- a flavour that rots
is found here, not in the middle of a profiling session."""
import concurrent.futures as cf
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = glob.glob(os.path.join(ROOT, "*_amd", "csrc"))[0]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAVOURS = [
    ("conv_mfma.hip", "-DSIFSR_DIAG_NOMFMA"), ("conv_wino8.hip", "-DSIFSR_DIAG_NOMFMA"), ("conv_wgrad_wino.hip", "-DSIFSR_DIAG_NOMFMA"),
    ("conv_bwd16.hip", "-DSIFSR_DIAG_NOMFMA"),
    ("conv_wino8.hip", "-DSIFSR_DIAG_CLOCK"), ("conv_bwd16.hip", "-DSIFSR_DIAG_CLOCK"), ("conv_wino8.hip", "-DSIFSR_DIAG_W8_ABL=7"),
    ("conv_wino8.hip", "-DSIFSR_PK_MODE=1"), ("conv_wgrad_wino.hip", "-DSIFSR_PK_MODE=2"), ("conv_bwd16.hip", "-DSIFSR_PK_MODE=3"),
]


def _compile(job):
    src, flag = job
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-Wno-unused-function", flag,
                        "-c", os.path.join(CSRC, src), "-o", os.devnull], capture_output=True, text=True)
    return src, flag, r.returncode, r.stderr[-1500:]


def test_every_diagnostic_flavour_compiles():
    with cf.ThreadPoolExecutor(max_workers=4) as ex:
        results = list(ex.map(_compile, FLAVOURS))
    bad = [(s, f, err) for s, f, rc, err in results if rc != 0]
    assert not bad, bad


def test_shipped_sources_define_no_diagnostic_switch():
    """The library is built with none of the switches: no source may define one, and every #if on them lives in diag.h / common.h."""
    for path in glob.glob(os.path.join(CSRC, "*.hip")):
        text = open(path).read()
        assert "#define SIFSR_DIAG_" not in text and "#ifdef SIFSR_D" not in text and "#if SIFSR_PK_MODE" not in text, path
