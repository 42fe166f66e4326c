"""rtrec_amd -- MI355X-native SLIM online-update and scoring engine.

Drop-in for rtrec.models.SLIM / rtrec.recommender.Recommender (fit, partial_fit, bulk_fit,
recommend, recommend_batch, similar_items).  The arithmetic runs in hand-written HIP kernels
for gfx950 (rtrec_amd/csrc) behind the C-ABI of include/rtrec_amd.h; there is no CPU fallback.
"""
__version__ = "0.1.0"

from .models.slim import SLIM  # noqa: E402,F401
from .recommender import Recommender  # noqa: E402,F401

__all__ = ["SLIM", "Recommender", "__version__"]
