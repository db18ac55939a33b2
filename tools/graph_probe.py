"""Does a captured HIP graph of one step (front-end + encoder) shave the inter-kernel gaps?  python tools/graph_probe.py (GPU box)"""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests" / "golden")]
from synth_weights import make_state_dict
from uit_mobile_amd import models
from bench import synth_batch
dev = torch.device("cuda", 0)
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in make_state_dict("uit_xs", 13).items()}
x = synth_batch(8192, 16000, 0, dev, False)
m = models.uit_xs(outputdim=537, target_length=102).to(dev); m.load_state_dict(sd); m.eval()
m.reserve(8192, 16000, dev)
with torch.no_grad():
    for _ in range(30): m(x)
    torch.cuda.synchronize()
    def timed(fn, n=300):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    print(f"eager   m(x): {timed(lambda: m(x)):.4f} ms/step")
    st = m.frontend(x)
    def two():
        global st
        nxt = m.frontend(x); o = m.encode(st); st = nxt
    print(f"eager   frontend/encode pipeline: {timed(two):.4f} ms/step")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): m(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s):
            for _ in range(10): y = m(x)
        print(f"graph of 10 steps: {timed(lambda: g.replay(), 30) / 10:.4f} ms/step")
    except Exception as e:
        print("graph capture failed:", type(e).__name__, str(e)[:300])
