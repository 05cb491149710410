"""CPU test of the native host code on the hot path's input side: the compressed 8-wide BVH builder
(csrc/bvh_build.cpp) compiled with AddressSanitizer + UBSan and checked structurally
(tests/native/bvh_check.cpp); its threads under ThreadSanitizer, and its output must not depend on the
number of threads.  No GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = tmp_path_factory.mktemp("bvh") / "bvh_check"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    os.path.join(ROOT, "tests", "native", "bvh_check.cpp"), os.path.join(ROOT, "raytracing_engine_amd", "csrc", "bvh_build.cpp"), os.path.join(ROOT, "raytracing_engine_amd", "csrc", "bvh_two_level.cpp"),
                    "-o", str(exe)], check=True)
    return str(exe)


@pytest.mark.parametrize("n,seed,edge", [(1, 1, 0.5), (2, 2, 0.5), (3, 3, 0.5), (4, 4, 0.5), (9, 5, 3.0), (100, 6, 0.5), (5000, 7, 0.1), (20000, 8, 2.0)])
def test_bvh_structure_under_sanitizers(checker, n, seed, edge):
    out = subprocess.run([checker, str(n), str(seed), str(edge)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr


@pytest.fixture(scope="module")
def checker_tsan(tmp_path_factory):
    exe = tmp_path_factory.mktemp("bvh_tsan") / "bvh_check"
    subprocess.run(["g++", "-O2", "-g", "-std=c++17", "-pthread", "-fsanitize=thread",
                    os.path.join(ROOT, "tests", "native", "bvh_check.cpp"), os.path.join(ROOT, "raytracing_engine_amd", "csrc", "bvh_build.cpp"), os.path.join(ROOT, "raytracing_engine_amd", "csrc", "bvh_two_level.cpp"),
                    "-o", str(exe)], check=True)
    return str(exe)


def test_builder_threads_are_race_free_and_deterministic(checker_tsan):
    """150 000 triangles: subtree threads, the chunked binning of big nodes, the per-subtree dynamic program
    and the level-parallel node emission all run; ThreadSanitizer must stay silent and one thread must
    produce the very same nodes and leaf order as all of them."""
    many = subprocess.run([checker_tsan, "150000", "11", "0.2"], capture_output=True, text=True)
    assert many.returncode == 0 and many.stdout.startswith("OK") and "ThreadSanitizer" not in many.stderr, many.stdout + many.stderr
    one = subprocess.run(["taskset", "-c", "0", checker_tsan, "150000", "11", "0.2"], capture_output=True, text=True)
    assert one.returncode == 0 and one.stdout == many.stdout, one.stdout + one.stderr


def test_two_level_cut_is_race_free_and_deterministic(checker_tsan):
    """The SAH cut of the two-level build splits leaves side by side and bins big leaves on several threads: ThreadSanitizer
    must stay silent, and one CPU must produce the very same flattened tree (hash of nodes + leaf order) as all of them."""
    many = subprocess.run([checker_tsan, "150000", "11", "0.2", "64"], capture_output=True, text=True)
    assert many.returncode == 0 and many.stdout.startswith("OK") and "ThreadSanitizer" not in many.stderr, many.stdout + many.stderr
    one = subprocess.run(["taskset", "-c", "0", checker_tsan, "150000", "11", "0.2", "64"], capture_output=True, text=True)
    assert one.returncode == 0 and one.stdout == many.stdout, one.stdout + one.stderr
    odd = subprocess.run([checker_tsan, "150000", "11", "0.2", "37"], capture_output=True, text=True)  # not a power of two: the fullest leaves split first
    assert odd.returncode == 0 and odd.stdout.startswith("OK") and "ThreadSanitizer" not in odd.stderr, odd.stdout + odd.stderr


@pytest.fixture(scope="module")
def checker_limited(tmp_path_factory):
    """bvh_check with tests/native/thread_limit.cpp: pthread_create fails with EAGAIN after
    RT_TEST_THREAD_BUDGET calls (what a thread / process limit of the box looks like to std::thread)."""
    exe = tmp_path_factory.mktemp("bvh_lim") / "bvh_check"
    subprocess.run(["g++", "-O2", "-g", "-std=c++17", "-pthread", os.path.join(ROOT, "tests", "native", "bvh_check.cpp"),
                    os.path.join(ROOT, "tests", "native", "thread_limit.cpp"), os.path.join(ROOT, "raytracing_engine_amd", "csrc", "bvh_build.cpp"), os.path.join(ROOT, "raytracing_engine_amd", "csrc", "bvh_two_level.cpp"),
                    "-ldl", "-o", str(exe)], check=True)
    return str(exe)


@pytest.mark.parametrize("budget", [0, 1, 2, 5])
def test_builder_survives_a_thread_limit(checker_limited, budget):
    """std::thread construction that throws std::system_error (thread limit) must neither terminate the
    process nor change the tree: the chunks / subtrees whose thread could not start run on the caller."""
    free = subprocess.run([checker_limited, "150000", "11", "0.2"], capture_output=True, text=True)
    assert free.returncode == 0 and free.stdout.startswith("OK"), free.stdout + free.stderr
    lim = subprocess.run([checker_limited, "150000", "11", "0.2"], capture_output=True, text=True, env=dict(os.environ, RT_TEST_THREAD_BUDGET=str(budget)))
    assert lim.returncode == 0 and lim.stdout == free.stdout, lim.stdout + lim.stderr


@pytest.mark.parametrize("n,chunks,rebuild", [(5000, 64, None), (20000, 64, 3), (300, 64, 0), (9, 4, None), (40000, 7, 6)])
def test_two_level_bvh_structure_under_sanitizers(checker, n, chunks, rebuild):
    """Top level over bottom-level chunks, flattened into the single-level node format (csrc/bvh_two_level.cpp): the
    same structural checks as the single-level tree (every triangle reachable exactly once, inside its leaf box, child
    indexing consistent), also after one chunk's triangles moved and only that chunk was rebuilt."""
    args = [checker, str(n), "21", "0.4", str(chunks)] + ([str(rebuild)] if rebuild is not None else [])
    out = subprocess.run(args, capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr
