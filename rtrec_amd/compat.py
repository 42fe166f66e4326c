"""Loading model files written by the reference (rtrec.models.SLIM.save).

A reference pickle names rtrec's own classes (rtrec.models.internal.slim_elastic.SLIMElastic,
rtrec.utils.interactions.UserItemInteractions, ...).  The unpickler below redirects those names to
the rtrec_amd classes, whose __setstate__ methods accept the reference's attribute layout
(dict-of-dict interaction store, plain `item_similarity` attribute, IndexedSet internals) and
convert it to the columnar / device-backed form -- rtrec itself does not have to be installed.
"""
from __future__ import annotations

import io
import pickle
from typing import Any, Dict, Tuple


def _class_map() -> Dict[Tuple[str, str], Any]:
    from .models.internal.slim_elastic import SLIMElastic
    from .utils.features import FeatureStore, IndexedSet
    from .utils.identifiers import Identifier
    from .utils.interactions import UserItemInteractions
    from .utils.lru import LRUFreqSet
    return {
        ("rtrec.models.internal.slim_elastic", "SLIMElastic"): SLIMElastic,
        ("rtrec.utils.interactions", "UserItemInteractions"): UserItemInteractions,
        ("rtrec.utils.lru", "LRUFreqSet"): LRUFreqSet,
        ("rtrec.utils.identifiers", "Identifier"): Identifier,
        ("rtrec.utils.features", "FeatureStore"): FeatureStore,
        ("rtrec.utils.collections", "IndexedSet"): IndexedSet,
    }


class ReferenceUnpickler(pickle.Unpickler):
    """pickle.Unpickler that maps rtrec.* class references onto their rtrec_amd equivalents."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._map = _class_map()

    def find_class(self, module: str, name: str):
        return self._map.get((module, name)) or super().find_class(module, name)


def loads(data: bytes) -> Any:
    return ReferenceUnpickler(io.BytesIO(data)).load()
