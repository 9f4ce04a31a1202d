"""The C++ host mirror (include/beat_detector.hpp) compiles against the C ABI;
on a GPU the demo caller reports the oracle's beats."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "listen_demo")


def _build():
    src = os.path.join(ROOT, "tests", "cpp", "listen_demo.cpp")
    lib_dir = os.path.join(ROOT, "beat_detector_amd")
    if os.path.exists(EXE) and os.path.getmtime(EXE) > max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "include", "beat_detector.hpp")),
            os.path.getmtime(os.path.join(lib_dir, "libbeatdet_hip.so"))):
        return
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), src,
                           "-o", EXE, "-L", lib_dir, "-lbeatdet_hip", "-Wl,-rpath," + lib_dir,
                           "-Wl,-rpath-link,/opt/rocm/lib", "-pthread"])


def test_cpp_mirror_compiles_and_links(bd):
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [0, 1])
def test_cpp_demo_reports_the_oracle_beats(bd, oracle, kind):
    if not os.path.exists(EXE):
        _build()
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    out = subprocess.run([EXE, str(kind), "5", "7"], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    assert "err: Variable keep_recording is false from the beginning!?" in lines
    got = [int(l.split()[1]) for l in lines if l.startswith("beat ")]
    x = bd.synth_fill_host(1, 44100 * 5, first_stream=7, seed=123)[0]
    _, _, beats = oracle.run_stream(kind, 44100, x, 1024, full=False)
    want = [ms for _, ms in beats]
    assert got == want and len(want) > 0
    assert f"detector beats {len(want)}" in lines
    # the live engine with a callback length per stream and tick: stream 0 in callbacks of 1024, stream 1 in 512
    _, _, beats512 = oracle.run_stream(kind, 44100, x, 512, full=False)
    assert f"live ragged beats {len(want)} {len(beats512)}" in lines
