"""Compare the residual streams two builds left for the eight clips of tools/ax_race.py (AX_TOKENS=1): where do they part?
    python tools/ax_tokens_cmp.py gpurun_out/ax_tokens_libuitk.npy gpurun_out/ax_tokens_libuitk_ax_chained.npy"""
import sys
import numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])            # [8, 1, depth + 1, 24, 128]
d = np.abs(a - b)
d = np.where(np.isfinite(d), d, 1e9)
print("shape", a.shape, "max |a|", float(np.nanmax(np.abs(a))))
for clip in range(a.shape[0]):
    for stage in range(a.shape[2]):
        x = d[clip, 0, stage]                                  # [24 tokens, 128]
        if x.max() > 1e-3:
            tok = np.where(x.max(1) > 1e-3)[0]
            ch = np.where(x.max(0) > 1e-3)[0]
            print(f"clip {clip} (wave {clip}) first differs after block {stage - 1 if stage else 'patch-embed'}: max {x.max():.3g}; tokens {tok.tolist()}; "
                  f"{len(ch)} channels" + (f" {ch[:16].tolist()}..." if len(ch) else ""))
            break
    else:
        print(f"clip {clip}: identical to 1e-3 through all blocks")
