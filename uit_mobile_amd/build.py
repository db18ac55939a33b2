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
SOURCES = ["api.cpp", "pack.cpp", "frontend.hip", "encoder.hip", "epilogue.hip", "calibrate.hip", "general.hip"]
def _headers():
    """every header a translation unit can include: csrc/*.h (globbed: a new header cannot be forgotten -- naive_forward.h once was, and
    its structs travel by value between api.cpp and calibrate.hip) + the public C ABI"""
    return sorted(p.name for p in CSRC.glob("*.h")) + ["../../include/uitk.h"]


HEADERS = _headers()
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
    """One object per source under csrc/build/ (compiled in parallel, re-compiled only when the source or a header is newer), then one
    link: a change in api.cpp does not re-compile the 75 encoder instantiations.  Same flags as ever."""
    if not force and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    obj_dir = CSRC / "build"
    obj_dir.mkdir(exist_ok=True)
    compile_flags = [f for f in FLAGS if f != "-shared"]
    hdr_t = max((CSRC / h).stat().st_mtime for h in HEADERS)
    tag = obj_dir / "flags.txt"                               # other flags (lab builds) invalidate every object
    flag_key = " ".join([*compile_flags, *extra_flags])
    stale_all = force or not tag.exists() or tag.read_text() != flag_key
    jobs = []
    for s in SOURCES:
        obj = obj_dir / (s + ".o")
        if stale_all or not obj.exists() or obj.stat().st_mtime < max((CSRC / s).stat().st_mtime, hdr_t):
            jobs.append([hipcc(), *compile_flags, *extra_flags, "-c", "-o", str(obj), str(CSRC / s)])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd, cwd=str(CSRC), env=dict(os.environ))
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), 6))) as ex:
        list(ex.map(run, jobs))
    tag.write_text(flag_key)
    run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *[str(obj_dir / (s + ".o")) for s in SOURCES]])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
