"""CPU tests of the host-side plumbing of the library (no GPU): persistent per-device workers, lane lock, and the
exception-to-error-code mapping — under ThreadSanitizer and AddressSanitizer + UBSan as well (sanitizers run on the CPU
build only; the GPU pool has no sanitizer support).  Also the C++ host field / curve arithmetic the Horner tail uses."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host", "workers_test.cpp")
BASE = ["g++", "-O1", "-g", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-pthread"]
LINK = ["-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"]


@pytest.mark.parametrize("name,flags", [("plain", []), ("tsan", ["-fsanitize=thread"]), ("asan_ubsan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])])
def test_workers_lanes_and_error_mapping(name, flags, tmp_path):
    exe = str(tmp_path / f"workers_{name}")
    subprocess.check_call(BASE + flags + ["-o", exe, SRC] + LINK)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=0", HIP_VISIBLE_DEVICES="")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "workers OK" in r.stdout
