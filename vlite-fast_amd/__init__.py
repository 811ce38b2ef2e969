"""vlite-fast_amd: MI355X-native VLITE-Fast baseband -> filterbank front end.

The directory name carries a hyphen (it mirrors the reference's name), so import it with
    importlib.import_module("vlite-fast_amd")      or      import vlite_fast_amd
(the latter is a two-line alias module at the repository root).

Only the hot path of the reference's process_baseband executable lives here:
  csrc/            hand-written gfx950 HIP kernels + the C ABI (include/pb_hip.h)
  libpb.py         ctypes binding of libpb_hip.so (fails loudly if it is missing)
  vdif.py          VDIF frames and psrdada-style ASCII headers (the input contract)
  sigproc.py       SIGPROC .fil header / file writer (the output contract)
  process_baseband.py   the executable's host side, same command-line flags
  genbase.py       genbase-style synthetic baseband (test-side producer)
  coadd.py         incoherent antenna sum across GPUs (torch.distributed / RCCL)
  search.py        dedispersion + boxcar search on the filterbank (stands where heimdall stands)
  triggers.py      trigger_t wire format towards the voltage dumper
"""
__version__ = "0.1"
