"""GPU: what does a HIP graph buy on the bench step?  The step is two kernels (front-end, encoder); eager, each forward is two
launches from Python (~26 us of host time against 770 us of GPU time: the queue never runs dry), so a graph can only remove what the
command processor spends BETWEEN dependent kernels.  Same model, same batch, same stream:

    eager          `steps` forwards, events around them                                   (bench.py's loop)
    graph x1       one forward captured; the graph replayed `steps` times
    graph x20      twenty forwards captured; the graph replayed steps / 20 times

    python tools/graph_step.py [steps=200] [model=uit_xs] [B=8192]      -> ms per step, median of 7 repeats; outputs compared bitwise
"""
import statistics
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from bench import make_state_dict, synth_batch  # noqa: E402
from uit_mobile_amd import models  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    name = sys.argv[2] if len(sys.argv) > 2 else "uit_xs"
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
    dev = torch.device("cuda", 0)
    sd = make_state_dict(name, 13)
    m = getattr(models, name)(outputdim=537, target_length=102, check_weights=False)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m = m.to(dev).eval()
    x = synth_batch(B, 16000, 0, dev, False)
    m.prepare(dev, sample=x)
    side = torch.cuda.Stream(dev)
    out = torch.empty((B, 537), device=dev)

    def timed(fn, n_calls):
        ms = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            with torch.cuda.stream(side):
                e0.record()
                for _ in range(n_calls):
                    fn()
                e1.record()
            torch.cuda.synchronize(dev)
            ms.append(e0.elapsed_time(e1) / steps)
        return statistics.median(ms), min(ms)

    with torch.no_grad():
        with torch.cuda.stream(side):
            for _ in range(20):
                eager_out = m(x)
        torch.cuda.synchronize(dev)
        graphs = {}
        for k in (1, 20):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for _ in range(k):
                    out.copy_(m(x)) if k == 1 else m(x)
            graphs[k] = g
        graphs[1].replay()
        torch.cuda.synchronize(dev)
        same = bool(torch.equal(out, eager_out))
        res = {"eager": timed(lambda: m(x), steps), "graph x1": timed(graphs[1].replay, steps), "graph x20": timed(graphs[20].replay, steps // 20)}
    print(f"{name} x {B}, {steps} steps per repeat, 7 repeats; graph output == eager output bitwise: {same}")
    for k, (med, mn) in res.items():
        print(f"  {k:10s} {med:.4f} ms per step (median), {mn:.4f} (best)   {B / med / 1e3:.2f} M clips/s")
    print("(graph x1 also copies the [B, 537] output inside the graph: ~17.6 MB written + read, ~5 us)")


if __name__ == "__main__":
    main()
