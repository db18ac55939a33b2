"""uit_mobile_amd -- MI355X-native batched inference for the UiT audio models (log-mel front-end + UiT encoder).

Drop-in for the model API of RicherMans/UIT_Mobile (`models.uit_xs`, `models.PRETRAINED_CHECKPOINTS`,
`model.load_state_dict`, `model(wave[B,T]) -> probs[B,537]`), backed by hand-written HIP kernels for gfx950 behind
the C ABI in include/uitk.h.  There is no CPU compute path: `forward` on a non-HIP tensor raises.
"""
from . import models  # noqa: F401

__all__ = ["models"]
