"""CPU: AddressSanitizer + UndefinedBehaviorSanitizer over the two pieces of native HOST code (GPU ASan is not available on the
pool): the C oracle (every path: both attention layouts, GELU, short / multi-crop / odd-length clips) and the product's host-only
weight packer csrc/pack.cpp (every precision and layout).  Each runs in a child process."""
import os
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
GCC = shutil.which("gcc")


def _libasan():
    if not GCC:
        return None
    p = subprocess.run([GCC, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if p and os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(_libasan() is None, reason="gcc / libasan not available")
def test_oracle_under_sanitizers(tmp_path):
    subprocess.check_call(["make", "-C", str(ROOT / "oracle"), "-s", "liboracle_asan.so"])
    code = f"""
import sys
sys.path[:0] = [{str(ROOT)!r}, {str(ROOT / 'tests' / 'golden')!r}]
import numpy as np
from pathlib import Path
import oracle.oracle as O
O.build = lambda force=False: Path({str(ROOT / 'oracle' / 'liboracle_asan.so')!r})
from uit_mobile_amd.synth import make_state_dict, synth_pcm_i16
for name, att, act in (("uit_xxxs", "bneck", "relu"), ("audio_transformer_h128_d4_m3", "full", "gelu")):
    o = O.Oracle(make_state_dict(name, 3, attention=att), act=act)
    for T in (16000, 8000, 16384, 4099):
        r = o.forward(synth_pcm_i16(3, T, seed=T), want_stages=True)
        assert np.isfinite(r["probs"]).all()
    o.melspec_db(synth_pcm_i16(2, 16001, seed=1).astype(np.float32) / 32768.0)
# round 6: the general geometries (other n_fft -- 400 is no power of two: the radix-2 FFT once walked off its arrays there --, n_mels, patch
# size, widths, heads, poolings), short clips, one and several crops
from general_cases import GENERAL_CASES, SMALL_CASES, state_dict_args
from uit_mobile_amd.synth import make_general_state_dict
for tag in SMALL_CASES:
    case = GENERAL_CASES[tag]
    k = case["kwargs"]
    o = O.Oracle(make_general_state_dict(**state_dict_args(case)), target_length=k["target_length"], eval_avg=k.get("eval_avg", "mean"),
                 hop=k.get("hop_size", 160), heads=k["num_heads"], act=case["act"], pooling=k["pooling"], center=k.get("center", True))
    for T in (6000, 16000, 40000):
        r = o.forward(synth_pcm_i16(2, T, seed=T), want_stages=True)
        assert np.isfinite(r["probs"]).all(), (tag, T)
print("sanitizer run ok")
"""
    env = dict(os.environ, LD_PRELOAD=_libasan(), ASAN_OPTIONS="detect_leaks=0", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "sanitizer run ok" in out.stdout, (out.stdout + out.stderr)[-3000:]


@pytest.mark.skipif(_libasan() is None or not Path("/opt/rocm/include/hip/hip_runtime.h").exists(), reason="g++ / HIP headers not available")
def test_host_packer_under_sanitizers(tmp_path):
    exe = tmp_path / "asan_pack"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-pthread", "-o", str(exe),
                           str(ROOT / "tests" / "native" / "asan_pack_main.cpp"), str(ROOT / "uit_mobile_amd" / "csrc" / "pack.cpp")])
    out = subprocess.run([str(exe)], env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "0 failures" in out.stdout, (out.stdout + out.stderr)[-3000:]


def _libtsan():
    if not GCC:
        return None
    p = subprocess.run([GCC, "-print-file-name=libtsan.so"], capture_output=True, text=True).stdout.strip()
    return p if p and os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(_libtsan() is None or not Path("/opt/rocm/include/hip/hip_runtime.h").exists(), reason="g++ / libtsan / HIP headers not available")
def test_host_packer_thread_pool_under_thread_sanitizer(tmp_path):
    """The packer rounds the weight matrices of a calibrated image on worker threads (pack.cpp: the guarded pool of pack_image)."""
    exe = tmp_path / "tsan_pack"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-pthread",
                           "-o", str(exe), str(ROOT / "tests" / "native" / "asan_pack_main.cpp"), str(ROOT / "uit_mobile_amd" / "csrc" / "pack.cpp")])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "0 failures" in out.stdout and "ThreadSanitizer" not in out.stderr, (out.stdout + out.stderr)[-3000:]
