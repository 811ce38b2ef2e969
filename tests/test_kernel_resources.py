"""The hot kernels' compiled resource usage (no GPU: hipcc cross-compiles gfx950): no scratch memory, and the
occupancy the design counts on.  Guards against a change that silently costs the channeliser spilled registers --
round 4 measured one: twenty spilled VGPRs in k_channelize_kur showed up as +9 % HBM write traffic and +8 % reads in the
PMC counters while the step time stayed inside the noise (profiles/r04_notes.md)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vlite-fast_amd", "csrc")
FLAGS = ("-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt "
         "-fno-gpu-flush-denormals-to-zero -I../../include -I. -w --cuda-device-only -S -o -").split()


def _usage(src):
    """{kernel symbol: {NumVgprs, ScratchSize, Occupancy, LDSByteSize}} from the device assembly's resource comments"""
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [src], cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out, cur = {}, None
    for line in r.stdout.decode().splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            continue
        m = re.match(r"^; (NumVgprs|ScratchSize|Occupancy|LDSByteSize): (\d+)", line)
        if m and cur:
            out.setdefault(cur, {})[m.group(1)] = int(m.group(2))
    return out


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_channelisers_use_no_scratch_and_keep_three_workgroups_per_cu():
    u = _usage("k_channelize.hip")
    kur = next(v for k, v in u.items() if "k_channelize_kur" in k)
    plain = next(v for k, v in u.items() if k.startswith("_Z12k_channelize8"))
    for name, k in (("k_channelize_kur", kur), ("k_channelize", plain)):
        assert k["ScratchSize"] == 0, (name, k)
        assert k["NumVgprs"] <= 168 and k["Occupancy"] >= 3, (name, k)       # 3 waves per SIMD = 3 workgroups per CU
        assert k["LDSByteSize"] <= 163840 // 3, (name, k)


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_pfb_channeliser_uses_no_scratch_and_keeps_three_workgroups_per_cu():
    u = _usage("k_channelize_pfb.hip")
    assert not any("k_channelize_pfb_kur" in sym for sym in u)      # ONE taps = 4 channeliser (the fused one left in round 5)
    for name in ("k_channelize_pfb7",):
        k = next(v for sym, v in u.items() if name in sym)
        assert k["ScratchSize"] == 0 and k["NumVgprs"] <= 168 and k["Occupancy"] >= 3, (name, k)
        assert k["LDSByteSize"] <= 163840 // 3, (name, k)


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_detect_uses_no_scratch():
    u = _usage("k_detect2.hip")
    assert u and all(v["ScratchSize"] == 0 for v in u.values()), {k: v for k, v in u.items() if v["ScratchSize"]}
    # the three-chunk ring of the headline path: 58 KB, i.e. one workgroup beside two channeliser workgroups
    assert max(v["LDSByteSize"] for v in u.values()) <= 60 * 1024
