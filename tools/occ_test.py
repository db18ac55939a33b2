"""GPU box: encoder time vs batch size for 4- and 8-wave workgroups (round quantisation).  python tools/occ_test.py"""
import os, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests" / "golden")]


def main():
    from synth_weights import make_state_dict
    from uit_mobile_amd import models
    from bench import synth_batch
    dev = torch.device("cuda", 0)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in make_state_dict("uit_xs", 13).items()}
    x = synth_batch(8192, 16000, 0, dev, False)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for waves in ("4", "8"):
        os.environ["UITK_ENC_WAVES"] = waves
        m = models.uit_xs(outputdim=537, target_length=102).to(dev)
        m.load_state_dict(sd); m.eval()
        for B in (1024, 2048, 4096, 8192):
            xb = x[:B]
            for _ in range(3): m(xb)
            en = []
            for _ in range(10):
                ev[0].record(); m(xb, amax_sync=lambda s: ev[1].record()); ev[2].record(); torch.cuda.synchronize()
                en.append(ev[1].elapsed_time(ev[2]))
            wgs = B // (2 * int(waves))
            print(f"waves/WG {waves} B {B:5d} WGs {wgs:5d} ({wgs/256:.1f}/CU) encoder {np.median(en):.4f} ms  per-clip {np.median(en)/B*1e6:.1f} ns")


if __name__ == "__main__":
    main()
