"""GPU box, -DUITK_LAB library (UITK_LIB_PATH=tools/ab/libuitk_lab.so): does every LDS-DMA piece of the f16l6 weight ring hold the
right data when its stage is consumed?  After each ring step lane p compares the last dword of piece p with the packed image."""
import ctypes, struct, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests"), str(ROOT / "tests" / "golden")]
from uit_mobile_amd.synth import make_state_dict, synth_pcm_i16
from uit_mobile_amd import models, _lib
from mfma_emu import HDR_FMT, HDR_FIELDS, L6_BLOCK, L6_QK, l6_mlp_offset, PE_STAGES, STAGE
name, depth = "uit_xxxs", 4
sd = make_state_dict(name, 5)
m = getattr(models, name)(outputdim=537, target_length=102, precision="f16l6")
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
img = m.eval().packed_image()
h = dict(zip(HDR_FIELDS, struct.unpack_from(HDR_FMT, img, 0)))
tab = np.zeros(32 + depth * 14 * 32, np.uint32)
for blk in range(depth):
    for k in range(14):
        off = 0 if k == 0 else L6_QK if k == 1 else l6_mlp_offset(k - 2)
        npieces = 16 if k == 0 else 24 if k == 1 else (31 if (k - 2) % 4 == 3 else 19)
        for p in range(npieces):
            o = h["off_wstream"] + PE_STAGES * STAGE + blk * L6_BLOCK + off + p * 1024 + 1020
            tab[32 + (blk * 14 + k) * 32 + p] = struct.unpack_from("<I", img, o)[0]
m = m.cuda()
x = torch.from_numpy(synth_pcm_i16(4096, 16000, seed=41)).cuda()
m(x); torch.cuda.synchronize()
buf = torch.from_numpy(tab.astype(np.int64).astype(np.uint32).view(np.int32).copy()).cuda()
hnd = m._handle(torch.device("cuda", 0))
_lib.check(_lib.load().uitk_set_timing_buffers(hnd, ctypes.c_void_p(buf.data_ptr()), None), "timing")
for rep in range(5):
    buf[:32] = 0
    m(x); torch.cuda.synchronize()
    c = buf[:32].cpu().numpy().view(np.uint32)
    print(f"run {rep}: stale/misplaced pieces by stage k = 0..13: {c[:14].tolist()}   by piece group (p/4): {c[16:24].tolist()}   total {int(c[:14].sum())}")
_lib.load().uitk_set_timing_buffers(hnd, None, None)
