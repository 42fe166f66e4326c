"""Raw id <-> dense internal id.

Same contract as rtrec.utils.identifiers.Identifier (/root/reference/rtrec/utils/identifiers.py:
11-90): integer ids pass through unmapped (and switch scoring to the sparse top-k path,
/root/reference/rtrec/models/slim.py:78,94), other hashables get consecutive ids in order of
first sight; mixing the two kinds raises ValueError; force_identify maps integers too.
"""
from __future__ import annotations

from typing import Any, Iterable, List, Optional

import numpy as np


class IdentifierError(Exception):
    def __init__(self, id_name: str, obj_id: int):
        super().__init__(f"Identifier not found for {id_name}: {obj_id}")


def _is_int(obj: Any) -> bool:
    return isinstance(obj, (int, np.integer))


class Identifier:
    def __init__(self, name: str = "ID", force_identify: bool = False, **kwargs: Any) -> None:
        self.name = name
        self.force_identify = force_identify
        self.obj_to_id: dict = {}
        self.id_to_obj: list = []
        # None: undecided, True: integer ids pass through, False: ids are mapped
        self.pass_through: Optional[bool] = False if force_identify else None

    def _mixed(self, obj: Any) -> ValueError:
        return ValueError(f"Mixed types detected for {self.name}: {obj}")

    def identify(self, obj: Any) -> int:
        if not self.force_identify and _is_int(obj):
            if self.pass_through is False:
                raise self._mixed(obj)
            self.pass_through = True
            return int(obj)
        if self.pass_through is True:
            raise self._mixed(obj)
        known = self.obj_to_id.get(obj)
        if known is not None:
            return known
        new_id = len(self.id_to_obj)
        self.obj_to_id[obj] = new_id
        self.id_to_obj.append(obj)
        self.pass_through = False
        return new_id

    def identify_many(self, objs: Iterable[Any]) -> np.ndarray:
        """identify() over a batch; integer arrays take a vectorised pass-through path."""
        if isinstance(objs, np.ndarray) and objs.dtype.kind in "iu" and not self.force_identify:
            if self.pass_through is False:
                raise self._mixed(objs[0] if len(objs) else None)
            if len(objs):
                self.pass_through = True
            return objs.astype(np.int64)
        return np.fromiter((self.identify(o) for o in objs), dtype=np.int64)

    def get_id(self, obj: Any) -> Optional[int]:
        if not self.force_identify and _is_int(obj):
            if not self.pass_through:
                raise self._mixed(obj)
            return int(obj)
        return self.obj_to_id.get(obj)

    def get(self, obj_id: int) -> Any:
        if self.pass_through:
            return obj_id
        if 0 <= obj_id < len(self.id_to_obj):
            return self.id_to_obj[obj_id]
        raise IdentifierError(self.name, obj_id)

    def get_or_default(self, obj_id: int, default: Optional[Any] = None) -> Any:
        if self.pass_through:
            return obj_id
        if 0 <= obj_id < len(self.id_to_obj):
            return self.id_to_obj[obj_id]
        return default

    def __getitem__(self, obj_id: int) -> Any:
        return self.get(obj_id)
