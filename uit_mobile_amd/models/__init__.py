"""Mirror of the reference's `models` package surface for the hot path (models/__init__.py:1):
`models.uit_xs`, `models.uit_xxs`, `models.uit_xxxs`, `models.PRETRAINED_CHECKPOINTS`, `getattr(models, name)`.
(The MobileNetV2 teacher is training-only and out of scope.)"""
from .uit import *  # noqa: F401,F403
