"""GPU box: encoder time vs batch size for 4- and 8-wave workgroups (round quantisation) + small-batch latency incl. B = 1.
python tools/occ_test.py [precision]"""
import os, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests" / "golden")]


def main():
    from uit_mobile_amd.synth import make_state_dict
    from uit_mobile_amd import models
    from bench import synth_batch
    dev = torch.device("cuda", 0)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in make_state_dict("uit_xs", 13).items()}
    x = synth_batch(8192, 16000, 0, dev, False)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
    for waves in ("4", "8") if prec in ("f16", "bf16", "fp8") else ("8",):
        m = models.uit_xs(outputdim=537, target_length=102, precision=prec).to(dev)
        m.load_state_dict(sd); m.eval()
        m.set_option("enc_waves", int(waves))
        for B in (1, 16, 256, 1024, 2048, 4096, 8192):
            xb = x[:B]
            for _ in range(3): m(xb)
            en = []
            for _ in range(10):
                ev[0].record(); m(xb, amax_sync=lambda s: ev[1].record()); ev[2].record(); torch.cuda.synchronize()
                en.append(ev[1].elapsed_time(ev[2]))
            wgs = B // (2 * int(waves))
            # host cost of the two-stage call (Python + ctypes, no GPU wait): what the removed per-call parameter scan used to dominate
            import time
            t0 = time.perf_counter()
            for _ in range(200):
                m.encode(m.frontend(xb))
            host_us = (time.perf_counter() - t0) / 200 * 1e6
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                m(xb); torch.cuda.synchronize()
            lat_ms = (time.perf_counter() - t0) / 50 * 1e3
            print(f"{prec} waves/WG {waves} B {B:5d} WGs {max(wgs, 1):5d} ({wgs/256:.1f}/CU) encoder {np.median(en):.4f} ms  per-clip {np.median(en)/B*1e6:.1f} ns  "
                  f"| host enqueue frontend()+encode() {host_us:.1f} us | synchronous call latency {lat_ms:.3f} ms")


if __name__ == "__main__":
    main()
