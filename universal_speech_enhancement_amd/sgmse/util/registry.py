"""Name -> class plug-in registry with the interface of the reference's
``sgmse/util/registry.py:5-36`` (``register(name)`` decorator, ``get_by_name``, ``get_all_names``;
a double registration only warns; an unknown name raises ``ValueError``)."""
import warnings
from typing import Callable, Dict, List


class Registry:
    def __init__(self, managed_thing: str):
        self.managed_thing = managed_thing
        self._items: Dict[str, type] = {}

    def register(self, name: str) -> Callable:
        def deco(cls):
            if name in self._items:
                warnings.warn(f"{self.managed_thing} with name '{name}' doubly registered, old class will be replaced.")
            self._items[name] = cls
            return cls
        return deco

    def get_by_name(self, name: str):
        try:
            return self._items[name]
        except KeyError:
            raise ValueError(f"{self.managed_thing} with name '{name}' unknown.") from None

    def get_all_names(self) -> List[str]:
        return list(self._items)
