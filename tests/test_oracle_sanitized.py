"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer ("CPU oracle under
-fsanitize=address,undefined"): `make -C oracle asan` builds oracle/_build/liboracle_asan.so from the same sources, and the
oracle's own known-answer tests (tests/test_oracle_a.py, tests/test_oracle_b.py) run against it in a child interpreter with
the sanitizer runtime preloaded.  Any report (heap overflow in a level image, signed overflow, misaligned access ...) aborts the
child.  CPU only."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_kats_pass_under_asan_and_ubsan():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"], check=True)
    so = os.path.join(ROOT, "oracle", "_build", "liboracle_asan.so")
    assert os.path.exists(so)
    runtimes = [subprocess.run(["gcc", "-print-file-name=" + n], capture_output=True, text=True, check=True).stdout.strip() for n in ("libasan.so", "libubsan.so")]
    assert all(os.path.isabs(p) and os.path.exists(p) for p in runtimes), runtimes
    env = dict(os.environ, ORACLE_SO=so, LD_PRELOAD=":".join(runtimes),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="4")
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle_a.py"),
                          os.path.join(ROOT, "tests", "test_oracle_b.py")], capture_output=True, text=True, env=env, cwd=ROOT)
    tail = out.stdout[-3000:] + out.stderr[-3000:]
    assert out.returncode == 0, tail
    import re
    m = re.search(r"(\d+) passed", out.stdout)
    assert m and int(m.group(1)) >= 40 and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
