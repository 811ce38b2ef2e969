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


SHIPPED_SWITCHES = {"PB_FUSE_KURTOSIS", "PB_KUR_EARLY", "PB_OVERLAP_DETECT", "PB_DETECT_DEPTH", "PB_DAG_BANDS"}


def test_shipped_library_reads_exactly_the_documented_switches():
    """Switches whose effect is "results invalid" (PB_SKIP: a kernel left out; PB_PFB_DBG: flags from stale bytes) and
    the switches that only ever served one timing experiment (PB_COPY_DMA, PB_COPY_WGS, PB_DET_CUS, PB_DET_PRIO) exist in
    experiment builds only (`make exp` / tools/build_variants.sh, selected with PB_LIBPATH): the shipped library -- the
    ONE product library, there is no other flavour beside it -- and the native host program do not even contain the
    variables' names, so no environment can make a production host write garbage.  What libpb_hip.so does read is
    exactly SHIPPED_SWITCHES (scheduling choices that leave results unchanged, each covered by
    tests/test_gpu_schedules.py or tests/test_gpu_parity.py), which is also the table in INTEGRATION.md."""
    import re
    csrc = os.path.join(ROOT, "vlite-fast_amd", "csrc")
    forbidden = (b"PB_SKIP", b"PB_PFB_DBG", b"CH_ABL", b"FFT_ABL", b"D2_ABL", b"PB_COPY_DMA", b"PB_COPY_WGS", b"PB_DET_CUS",
                 b"PB_DET_PRIO", b"PB_FINE_GRAINED", b"PB_LEAN_LDS")
    for fn in ("libpb_hip.so", "process_baseband"):
        path = os.path.join(csrc, fn)
        assert os.path.exists(path), fn
        blob = open(path, "rb").read()
        for name in forbidden:
            assert name not in blob, "%s contains %s" % (fn, name.decode())
    blob = open(os.path.join(csrc, "libpb_hip.so"), "rb").read()
    in_lib = set(m.decode() for m in re.findall(rb"PB_[A-Z][A-Z0-9_]{3,}(?=\x00)", blob))
    env_like = set(n for n in in_lib if not n.startswith(("PB_ST_", "PB_FFT_", "PB_E", "PB_OK")))
    assert env_like == SHIPPED_SWITCHES, env_like ^ SHIPPED_SWITCHES
    assert not os.path.exists(os.path.join(csrc, "libpb_hip_fg.so")), "the fine-grained variant is a patch under tools/experiments"
    # and the sources read the environment only through names on this list (the second set: experiments build only)
    allowed = SHIPPED_SWITCHES | {"PB_LEAN_LDS", "PB_COPY_DMA", "PB_COPY_WGS", "PB_DET_CUS", "PB_DET_PRIO", "PB_SKIP"}
    seen = set()
    for fn in os.listdir(csrc):
        if fn.endswith((".hip", ".h")):
            seen |= set(re.findall(r'(?:getenv|env_int)\("([A-Z0-9_]+)"', open(os.path.join(csrc, fn)).read()))
    assert seen <= allowed, seen - allowed
    txt = open(os.path.join(csrc, "pb_api.hip")).read()
    i = txt.index('getenv("PB_SKIP")')
    assert "#if PB_EXPERIMENTS" in txt[max(0, i - 200):i]       # the one PB_SKIP read sits behind the build switch
    assert txt.count('getenv("PB_SKIP")') == 2                   # (both on that one line)
    for name in ("PB_COPY_DMA", "PB_COPY_WGS", "PB_DET_CUS", "PB_DET_PRIO"):
        i = txt.index('env_int("%s"' % name)
        assert "#if PB_EXPERIMENTS" in txt[max(0, i - 400):i], name
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name in SHIPPED_SWITCHES:
        assert "| `%s` |" % name in doc, name


def test_tree_order_of_the_library_equals_the_python_plan():
    """pb_coadd_tree_order (for hosts written in C) == coadd.tree_order for every leaf count; host
    code only, no GPU"""
    import importlib
    lp = libpb()
    L = lp.load()
    coadd = importlib.import_module("vlite-fast_amd.coadd")
    for n in range(1, 33):
        out = (C.c_int32 * n)()
        assert L.pb_coadd_tree_order(n, out) == 0
        assert list(out) == coadd.tree_order(range(n)), n
    assert L.pb_coadd_tree_order(33, (C.c_int32 * 33)()) == -22 and L.pb_coadd_tree_order(0, (C.c_int32 * 1)()) == -22
