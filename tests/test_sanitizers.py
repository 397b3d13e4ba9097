"""The CPU-side code (host object API with its OBJ and PNG loaders, the CPU twin of the per-ray code, the oracle) under
AddressSanitizer and UndefinedBehaviorSanitizer: tests/sanitize/san_driver.cpp loads three presets, traces a small ray
set through tree and linear scan, renders a 9-tap frame, and compares the twin with the oracle.  GPU code cannot be
sanitized on this hardware pool; the twin compiles the same rt_core.h / rt_lbvh.h / rt_prepare.h for the host."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "ray_tracer_amd", "csrc", "host")
SOURCES = [os.path.join(ROOT, "tests", "sanitize", "san_driver.cpp"), os.path.join(ROOT, "tests", "cpu_twin", "twin.cpp")] + \
          [os.path.join(HOST, f) for f in ("scene.cpp", "presets.cpp", "obj_loader.cpp", "image.cpp", "png_decode.cpp")]
ORACLE = os.path.join(ROOT, "oracle", "rt_oracle.c")
DEPS = SOURCES + [ORACLE, os.path.join(ROOT, "oracle", "rt_oracle.h"), os.path.join(HOST, "scene.hpp"), os.path.join(HOST, "glmini.hpp"),
                  os.path.join(ROOT, "include", "rt_amd.h")] + \
       [os.path.join(ROOT, "ray_tracer_amd", "csrc", f) for f in ("rt_core.h", "rt_lbvh.h", "rt_prepare.h")]
OUT = os.path.join(ROOT, "tests", "sanitize", "san_driver")


def build():
    if os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in DEPS):
        return
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fopenmp", "-ffp-contract=off", "-fno-fast-math",
           "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-static-libasan",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "ray_tracer_amd", "csrc"), "-I" + HOST,
           "-I" + os.path.join(ROOT, "oracle"), "-o", OUT] + SOURCES + ["-x", "c", ORACLE, "-lm", "-lz"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        if "asan" in r.stderr and "cannot find" in r.stderr:
            pytest.skip("this toolchain has no sanitizer runtime")
        raise AssertionError(r.stderr[-3000:])


def test_cpu_code_runs_clean_under_asan_and_ubsan():
    build()
    env = dict(os.environ, OMP_NUM_THREADS="4", ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([OUT, os.path.join(ROOT, "assets")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "all presets ran clean" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
