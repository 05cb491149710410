"""CPU test of the native host code on the hot path's input side: the compressed 8-wide BVH builder
(csrc/bvh_build.cpp) compiled with AddressSanitizer + UBSan and checked structurally
(tests/native/bvh_check.cpp).  No GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = tmp_path_factory.mktemp("bvh") / "bvh_check"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    os.path.join(ROOT, "tests", "native", "bvh_check.cpp"), os.path.join(ROOT, "raytracing_engine_amd", "csrc", "bvh_build.cpp"),
                    "-o", str(exe)], check=True)
    return str(exe)


@pytest.mark.parametrize("n,seed,edge", [(1, 1, 0.5), (2, 2, 0.5), (3, 3, 0.5), (4, 4, 0.5), (9, 5, 3.0), (100, 6, 0.5), (5000, 7, 0.1), (20000, 8, 2.0)])
def test_bvh_structure_under_sanitizers(checker, n, seed, edge):
    out = subprocess.run([checker, str(n), str(seed), str(edge)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr
