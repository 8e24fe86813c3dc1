"""The C ABI is usable without Python: a gcc-built plain-C program (tests/c_abi/c_abi_smoke.c,
compiled by __graft_entry__.build()) links libpe_hip.so and checks known answers on the GPU."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
EXE = ROOT / "tests" / "c_abi" / "_bin" / "c_abi_smoke"


def test_c_program_is_built_and_links_the_library():
    if not EXE.exists():   # a fresh checkout: build() has not run yet (needs gcc + the library)
        import sys

        sys.path.insert(0, str(ROOT))
        import __graft_entry__ as entry
        from pose_estimation_amd.build import build as build_lib

        build_lib()
        entry._build_c_abi_smoke()
    assert EXE.exists()
    out = subprocess.run(["ldd", str(EXE)], capture_output=True, text=True).stdout
    assert "libpe_hip.so" in out and "not found" not in out.split("libpe_hip.so")[1].splitlines()[0]


@pytest.mark.gpu
def test_c_program_runs_on_the_gpu():
    r = subprocess.run([str(EXE)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "c_abi_smoke ok" in r.stdout
