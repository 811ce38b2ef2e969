import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden"),
          os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_python.npz"))


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def psrdada_mock(tmp_path_factory):
    """The REAL psrdada shim (vlite-fast_amd/csrc/pb_dada_shim.c, built by the package's own `make dada` with
    -Wall -Wextra -Werror) over the psrdada stand-in of tests/mock_psrdada (declarations written from the reference's
    call sites + shared-file rings; not psrdada, pins nothing about its ABI).  -> namespace: shim (ctypes, prototypes
    declared), shim_path, ctl (the stand-in's test controls: create / destroy / shutdown / counts), dir."""
    import ctypes as C
    import importlib
    import subprocess
    import types
    mdir = os.path.join(ROOT, "tests", "mock_psrdada")
    os.makedirs(os.path.join(mdir, "lib"), exist_ok=True)
    lib = os.path.join(mdir, "lib", "libpsrdada.so")
    subprocess.run(["gcc", "-O2", "-std=gnu11", "-shared", "-fPIC", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(mdir, "include"),
                    "-o", lib, os.path.join(mdir, "mock_psrdada.c"), "-lpthread"], check=True)
    shim = os.path.join(mdir, "lib", "libpb_dada.so")
    subprocess.run(["make", "-s", "-B", "-C", os.path.join(ROOT, "vlite-fast_amd", "csrc"), "dada", "PSRDADA=" + mdir,
                    "DADA_OUT=" + shim], check=True)
    d = str(tmp_path_factory.mktemp("rings"))
    os.environ["MOCK_PSRDADA_DIR"] = d
    os.environ.setdefault("MOCK_PSRDADA_TIMEOUT_S", "20")
    ctl = C.CDLL(lib)
    ctl.mock_psrdada_create.argtypes = [C.c_uint32, C.c_uint64, C.c_uint64]
    ctl.mock_psrdada_destroy.argtypes = [C.c_uint32]
    ctl.mock_psrdada_shutdown.argtypes = [C.c_uint32]
    ctl.mock_psrdada_counts.argtypes = [C.c_uint32] + [C.POINTER(C.c_uint64)] * 4
    dada = importlib.import_module("vlite-fast_amd.dada")
    L = dada.bind_shim(C.CDLL(shim))

    def counts(key):
        v = [C.c_uint64() for _ in range(4)]
        assert ctl.mock_psrdada_counts(key, *[C.byref(x) for x in v]) == 0
        return tuple(int(x.value) for x in v)          # data buffers handed over / handed back, headers posted / cleared
    return types.SimpleNamespace(shim=L, shim_path=shim, ctl=ctl, dir=d, mdir=mdir, counts=counts)
