"""Builds libpe_hip.so (gfx950) in-tree with hipcc.

    python -m pose_estimation_amd.build [--force]

hipcc cross-compiles without a GPU, so this runs on the CPU-only build host;
the resulting pose_estimation_amd/lib/libpe_hip.so travels to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
OBJ = CSRC / "_obj"
LIB = PKG / "lib" / "libpe_hip.so"
ARCH = "gfx950"

# -ffp-contract=off: the bandwidth-bound kernels promise a fixed float32 operation order
# (bit-for-bit parity with the CPU restatement); MFMA kernels do not depend on contraction.
CXXFLAGS = [
    f"--offload-arch={ARCH}",
    "-O3",
    "-fPIC",
    "-std=c++17",
    "-ffp-contract=off",
    "-Wall",
    "-Wno-unused-function",
    f"-I{ROOT / 'include'}",
    f"-I{CSRC}",
]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found; libpe_hip.so cannot be built")
    return exe


def sources() -> list[Path]:
    return sorted(CSRC.glob("*.hip"))


def _deps_mtime() -> float:
    files = list(CSRC.glob("*.h")) + list((ROOT / "include").glob("*.h")) + [Path(__file__)]
    return max(f.stat().st_mtime for f in files)


def _compile(src: Path, force: bool, verbose: bool) -> Path:
    obj = OBJ / (src.stem + ".o")
    newest = max(src.stat().st_mtime, _deps_mtime())
    if not force and obj.exists() and obj.stat().st_mtime >= newest:
        return obj
    cmd = [_hipcc(), *CXXFLAGS, "-c", str(src), "-o", str(obj)]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src.name}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)
    return obj


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile every csrc/*.hip for gfx950 and link lib/libpe_hip.so.  Returns its path."""
    OBJ.mkdir(parents=True, exist_ok=True)
    LIB.parent.mkdir(parents=True, exist_ok=True)
    srcs = sources()
    jobs = min(len(srcs), int(os.environ.get("MAX_JOBS", "8")), 16)
    with ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, verbose), srcs))
    if force or not LIB.exists() or any(o.stat().st_mtime > LIB.stat().st_mtime for o in objs):
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB)] + [
            str(o) for o in objs
        ] + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print(f"built {path}")
