"""CPU test of the host side's two-bit codec for packed transfers (bwt_amd/csrc/pack2.h): tools/pack2_selftest.cc
checks it against its byte-by-byte definition, with symbols outside the range at every position."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pack2_selftest(tmp_path):
    exe = str(tmp_path / "pack2_selftest")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-o", exe, os.path.join(ROOT, "tools", "pack2_selftest.cc"),
                    os.path.join(ROOT, "bwt_amd", "csrc", "pack2.cc")], check=True)
    out = subprocess.run([exe, "3000001"], check=True, capture_output=True, text=True).stdout
    assert out.strip().endswith("OK"), out
