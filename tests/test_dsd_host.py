"""CPU test of the host side of the exact densest-subgraph rounding (clipper_amd/csrc/dsd_host.h):
tests/cpp/test_dsd.cpp (g++ only) holds the certified replay of the reference's bisection (two maximum
flows when the peeling finds the optimum) against the reference's procedure as it stands (one flow per
bisection step, /root/reference/src/dsd.cpp:200-241) on random graph families — weighted, unweighted
(exact ties), planted clusters, equal cliques, two clusters, degenerate inputs: equal node sets."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_replay_equals_the_plain_bisection(tmp_path):
    exe = str(tmp_path / "test_dsd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror",
                           os.path.join(ROOT, "tests", "cpp", "test_dsd.cpp"), "-o", exe])
    out = subprocess.check_output([exe, "400"], timeout=600).decode()
    last = out.strip().splitlines()[-1]
    assert " 0 mismatches" in last, out
    cases, fell_back = int(last.split()[0]), int(last.split()[4])
    assert cases >= 2000 and fell_back <= cases // 100, last      # the fast path is the one that ran
