"""CPU dry test of the multi-process stop protocol: compiles and runs tests/cpp/
test_batch_protocol.cpp (g++ only), which drives clipper_amd/csrc/host_batch.hpp — the loop
clipper_hip_solve runs on a multi-process shard — with simulated ranks and collectives."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_batched_stop_protocol(tmp_path):
    exe = str(tmp_path / "test_batch_protocol")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-I", os.path.join(ROOT, "clipper_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "test_batch_protocol.cpp"), "-o", exe])
    out = subprocess.check_output([exe], timeout=300).decode()
    assert "batch protocol ok" in out
