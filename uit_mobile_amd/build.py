"""Build libuitk.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python -m uit_mobile_amd.build            # or: from uit_mobile_amd.build import build; build()

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
uit_mobile_amd/libuitk.so travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
LIB = HERE / "libuitk.so"
SOURCES = ["api.cpp", "pack.cpp", "frontend.hip", "encoder.hip", "epilogue.hip", "calibrate.hip"]
HEADERS = ["uitk_dev.h", "pack.h", "../../include/uitk.h"]
# -fno-slp-vectorize: the kernels write their packed math explicitly (float2 / ext_vector types); the SLP vectoriser's
# own packing costs more v_mov than it saves (same-box A/B: front-end -4 %)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize", "-Wall", "-Wno-unused-result", "-Wno-inline-asm"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found: libuitk.so cannot be built (and there is no fallback path)")
    return exe


def needs_build():
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any((CSRC / s).stat().st_mtime > t for s in SOURCES + HEADERS)


def build(force=False, verbose=False, extra_flags=()):
    if not force and not needs_build():
        return LIB
    cmd = [hipcc(), *FLAGS, *extra_flags, "-o", str(LIB), *[str(CSRC / s) for s in SOURCES]]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    env = dict(os.environ)
    subprocess.check_call(cmd, cwd=str(CSRC), env=env)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
