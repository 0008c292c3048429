"""GPU: the C ABI used without Python or torch — a stand-alone C++ program (tests/c_abi/c_abi_smoke.cpp) linked against
libpram_hip.so with nothing but the HIP runtime, checking linear (fp32 and split-fp16) and attention against fp64 loops, fused == split
attention bit for bit, and the error-code path."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def test_c_abi_standalone_program(hip_lib, tmp_path):
    exe = tmp_path / "c_abi_smoke"
    src = ROOT / "tests" / "c_abi" / "c_abi_smoke.cpp"
    lib_dir = ROOT / "pram_amd" / "csrc"
    build = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", str(src), "-I", str(ROOT / "include"),
                            "-L", str(lib_dir), "-lpram_hip", f"-Wl,-rpath,{lib_dir}", "-o", str(exe)],
                           capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    assert "c_abi_smoke ok" in run.stdout and "fused == split" in run.stdout and "pram_linear_x3_f32" in run.stdout
