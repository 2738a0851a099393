"""The host code of the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only, no HIP call is made):
tests/san/host_san_driver.cpp drives rt_plan_groups -- the Morton sort, greedy grouping and bisection that run on every
rt_upload_scene -- and the other host-only entry points over the built-in scenes and random ones (slabs, blobs, shells,
lines, coincident spheres, huge and tiny radii, NaN / inf entries, short output buffers).  rt_host.cpp is rebuilt with
g++ -fsanitize=address,undefined, rt_api.hip's HOST side with hipcc -fno-gpu-sanitize -fsanitize=address,undefined (device
code is not instrumented: GPU sanitizers are not available on this pool); the kernels are the product's objects."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raytracer_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_host_entry_points_under_asan_ubsan(tmp_path):
    objs = [os.path.join(CSRC, "build", f"rt_{n}.o") for n in
            ("kernels_exact", "sweep_small", "sweep_grouped", "sweep_grouped_fast", "sweep_grouped3", "sweep_chunked", "sweep_compact", "sweep_sorted")]
    missing = [o for o in objs if not os.path.exists(o)]
    if missing:
        pytest.skip("product objects not built (make -C raytracer_amd/csrc): " + ", ".join(os.path.basename(m) for m in missing))
    san = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
    inc = ["-I", os.path.join(ROOT, "include"), "-I", CSRC]
    host_o, api_o, drv_o = (str(tmp_path / n) for n in ("rt_host.o", "rt_api.o", "driver.o"))
    subprocess.check_call(["g++", "-std=c++17", "-ffp-contract=off", *san, *inc, "-c", "-o", host_o, os.path.join(CSRC, "rt_host.cpp")])
    subprocess.check_call(["g++", "-std=c++17", *san, *inc, "-c", "-o", drv_o, os.path.join(ROOT, "tests", "san", "host_san_driver.cpp")])
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fno-gpu-sanitize", *san, *inc, "-c", "-o", api_o,
                           os.path.join(CSRC, "rt_api.hip")])
    exe = str(tmp_path / "host_san")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-fno-gpu-sanitize", "-fsanitize=address,undefined", "-o", exe, drv_o, host_o, api_o, *objs])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:verify_asan_link_order=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    probe = subprocess.run([exe], env=dict(env, RT_SAN_SELFTEST="1"), capture_output=True, text=True, timeout=120)
    assert probe.returncode != 0 and "AddressSanitizer" in probe.stderr, "the build is not instrumented: a heap overflow in rt_split_bands went unnoticed"
    r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "clean under ASan + UBSan" in r.stdout, (r.stdout[-2000:], r.stderr[-6000:])
