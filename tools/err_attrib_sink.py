"""CPU: what an ATTENTION SINK does to the fp16 activation floor of the default family (mean pooling).  synth.make_sink_trained_like_state_dict
puts a sink in every block and head (the six tokens of frequency row 0 carry a key component u, every query looks for u); fp64
restatement with exact weights (tools/err_attrib_centring.py), the product's rounding points switched on (k, v, attention output
centred as the product centres them), then k / v alone.

    python tools/err_attrib_sink.py uit_xs bneck relu            -> profiles/round5/err_attrib_sink.log

Finding: the floor grows with the sink -- uit_xs: 2.4e-4 (no sink) -> 5.9e-4 -> 9.9e-4 -> 1.2e-3 (gain 1 / 2 / 4) -- and the keys and values
carry it (gain 4: k alone 5.1e-4, v alone 3.0e-4): a sink key is an OUTLIER, centring on the all-token mean does not shrink it, and its
rounding error q . dk is proportional to the score it produces.  Split weights cannot help; the keys' and values' low-order terms
(k_lo . q, P . v_lo: what 'f16a' / 'f16ax' carry) would."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tools")]
import err_attrib_centring as E  # noqa: E402
import lo_term_study as L  # noqa: E402
from bench import synth_batch  # noqa: E402
from uit_mobile_amd.synth import DEPTH, make_sink_trained_like_state_dict  # noqa: E402

name, att, act = sys.argv[1], sys.argv[2], sys.argv[3]
x = (synth_batch(24, 16000, 0, "cpu", False).numpy() * 32768).round().astype(np.int16)
xcal = (synth_batch(32, 16000, 5000, "cpu", False).numpy() * 32768).round().astype(np.int16)
for gain in (0.0, 1.0, 2.0, 4.0):
    sd = make_sink_trained_like_state_dict(name, 212, attention=att, gain=gain)
    mb, mbc = L.mel_bn_of(sd, x), L.mel_bn_of(sd, xcal)
    rec = {}
    E.forward(sd, mbc, DEPTH[name], act, att, "mean", (), rec=rec)
    run = lambda pts: E.forward(sd, mb, DEPTH[name], act, att, "mean", pts, mus=rec)
    exact = run(())
    err = lambda pts: float((run(pts) - exact).abs().max())
    prod = ["patch", "ln1c" if att == "full" or act == "gelu" else "ln1", "q", "kc", "vc", "P", "Oc", "ln2", "hidden"]
    print(f"{name} {att} mean pooling, sink gain {gain}: |logit|max {float(exact.abs().max()):.1f}; all of the product's rounding points {err(prod):.2e}; "
          f"without k, v {err([p for p in prod if p not in ('kc', 'vc')]):.2e}; k alone {err(['kc']):.2e}; v alone {err(['vc']):.2e}; q alone {err(['q']):.2e}; P alone {err(['P']):.2e}", flush=True)
