"""CPU test of the two host planners (clipper_amd/csrc/host_plan.hpp): the work list of the streaming
pass on the slices and the units / pieces of the resident solver are pure functions of the slice
directory; tests/cpp/test_planners.cpp (g++ only) checks on random directories that every plan covers
every step of every slice exactly once, every partial-sum slot is written once, and a resident unit
fits its share of LDS by the planner's own bound."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_planners_cover_every_step_once(tmp_path):
    exe = str(tmp_path / "test_planners")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "clipper_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "test_planners.cpp"), "-o", exe])
    out = subprocess.check_output([exe], timeout=300).decode()
    assert "planners ok" in out
