"""CPU-side checks of the C-ABI library: it loads, exports every symbol the header declares,
validates arguments, and refuses to run without a GPU (no CPU fallback).  No compute calls."""
import ctypes as C
import os

import pytest

from helpers import ROOT, header_symbols, libpb


def test_library_exports_every_declared_symbol():
    lp = libpb()
    L = lp.load()
    names = header_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "libpb_hip.so lacks %s declared in include/pb_hip.h" % n
    assert set(lp.EXPORTS) == set(names)


def test_config_default_matches_reference_defaults():
    lp = libpb()
    L = lp.load()
    cfg = lp.PbConfig()
    L.pb_config_default(C.byref(cfg))
    # src/process_baseband.cu:343-354 of the reference: NBIT 2, npol 1, RFI_MODE 2
    assert (cfg.nbit, cfg.npol, cfg.rfi_mode, cfg.rows_per_seg) == (2, 1, 2, 1024)
    assert cfg.struct_size == C.sizeof(lp.PbConfig)


def test_create_rejects_bad_arguments_like_the_reference_getopt():
    lp = libpb()
    for kw in (dict(nbit=3), dict(npol=4), dict(rfi_mode=3), dict(taps=2), dict(rows_per_seg=12),
               dict(nant=0), dict(taps=4, fft_backend=lp.FFT_HIPFFT)):
        with pytest.raises(ValueError):
            lp.PbHandle(**kw)


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lp = libpb()
    with pytest.raises(lp.PbError, match="no HIP device"):
        lp.PbHandle()


def test_product_does_not_import_oracle():
    """The product path never includes, links, imports or dlopens anything under oracle/
    (comments may cite it as the specification)."""
    import re
    pkg = os.path.join(ROOT, "vlite-fast_amd")
    bad = re.compile(r"(^\s*#\s*include.*oracle|^\s*(import|from)\s+oracle|liboracle|CDLL\(.*oracle|-loracle)", re.M)
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")) or fn == "Makefile":
                txt = open(os.path.join(dp, fn)).read()
                assert not bad.search(txt), fn
