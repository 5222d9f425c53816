"""The dataset loader (host-only C++: JSON reader, PNG decoder, .npy reader) under AddressSanitizer + UBSan on corrupted files.

GPU sanitizers are not available on the pool, and the loader is the one part of librtxn.so that parses bytes it did not
write, so this is where a sanitizer run pays: tools/san/loader_fuzz.cpp is built together with rtx_nerf_amd/csrc/loader.cpp
by g++ -fsanitize=address,undefined and loads a few thousand damaged scenes (byte flips, truncation, extreme header fields,
structurally valid PNGs of every kind around random content).  Round 3's first runs of it found two things, both fixed:
a directory opened as a file (ftell = LONG_MAX -> a 2^63-byte allocation request) and misaligned double loads from a .npy
whose header is not a multiple of 8 bytes.  CPU only."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path_factory.mktemp("san") / "loader_fuzz")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-D__HIP_PLATFORM_AMD__",
           "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "rtx_nerf_amd", "csrc"),
           os.path.join(ROOT, "rtx_nerf_amd", "csrc", "loader.cpp"), os.path.join(ROOT, "tools", "san", "loader_fuzz.cpp"), "-lz", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        if "sanitize" in r.stderr or "asan" in r.stderr or "ubsan" in r.stderr:
            pytest.skip("this g++ has no sanitizer runtime")
        pytest.fail(r.stderr[-2000:])
    return exe


@pytest.mark.parametrize("seed", [1, 2])
def test_loader_has_no_sanitizer_report_on_corrupted_scenes(harness, tmp_path, seed):
    r = subprocess.run([harness, str(tmp_path), "4000", str(seed)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    assert "no sanitizer report" in r.stdout and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    accepted = int(r.stdout.split("corrupted loads,")[1].split("accepted")[0])
    assert 200 < accepted < 3800          # the campaign reaches both outcomes: files that still load and files that are refused
