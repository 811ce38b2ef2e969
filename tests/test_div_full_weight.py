"""The channeliser divides the spectra of rows without flags by 1 + 2^-23 on the bits (div_full_weight,
k_channelize.hip); tools/check_div_full_weight.c compares that rule with the IEEE division for every positive
normal binary32 (2.1e9 values, a few seconds)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bit_rule_equals_ieee_division_for_every_normal_float(tmp_path):
    exe = str(tmp_path / "chk")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tools", "check_div_full_weight.c")], check=True)
    r = subprocess.run([exe], stdout=subprocess.PIPE, timeout=600)
    assert r.returncode == 0 and b" 0 mismatches" in r.stdout, r.stdout
    # the kernel's rule is the checked one
    src = open(os.path.join(ROOT, "vlite-fast_amd", "csrc", "k_channelize.hip")).read()
    assert "(b & 0x7fffffu) - 1u" in src and "m1 >= 0x400001u ? 2u : 1u" in src and "0x3f800001u" in src
