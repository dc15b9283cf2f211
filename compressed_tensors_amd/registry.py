"""Name -> class registry used to look compressors up by their wire-format string.

Mirrors the three entry points of the reference's RegistryMixin that the hot path touches
(registry/registry.py:109-185): `register`, `get_value_from_registry`, `load_from_registry`,
with the same name standardisation (`_`/space -> `-`, lower-case; :28-42), the same
duplicate-name guard (RuntimeError, :215-223) and KeyError for unknown names (:250-257).
"""
from collections import defaultdict
from typing import Any, Optional, Union

__all__ = ["RegistryMixin", "standardize_lookup_name"]

_REGISTRY: dict = defaultdict(dict)
_ALIASES: dict = defaultdict(dict)


def standardize_lookup_name(name: str) -> str:
    return name.replace("_", "-").replace(" ", "-").lower()


class RegistryMixin:
    registry_requires_subclass: bool = False

    @classmethod
    def register(cls, name: Optional[str] = None, alias: Union[str, list, None] = None):
        def decorator(value: Any):
            cls.register_value(value, name=name, alias=alias)
            return value

        return decorator

    @classmethod
    def register_value(cls, value: Any, name: Optional[str] = None, alias: Union[str, list, None] = None):
        key = standardize_lookup_name(name if name is not None else value.__name__)
        if cls.registry_requires_subclass and not (isinstance(value, type) and issubclass(value, cls)):
            raise ValueError(f"class {value} is not a subclass of the class it is registered for: {cls}.")
        table = _REGISTRY[cls]
        if key in table and table[key] is not value:
            raise RuntimeError(
                f"Attempting to register name {key} as {value} however a value {table[key]} "
                f"has already been registered to {cls} under that name"
            )
        table[key] = value
        if alias is not None:
            for a in [alias] if isinstance(alias, str) else alias:
                _ALIASES[cls][standardize_lookup_name(a)] = key

    @classmethod
    def get_value_from_registry(cls, name: str):
        key = standardize_lookup_name(getattr(name, "value", name))
        key = _ALIASES[cls].get(key, key)
        table = _REGISTRY[cls]
        if key not in table:
            raise KeyError(
                f"Unable to find {name} registered under type {cls}.\n"
                f"Registered values for {cls}: {list(table)}\n"
            )
        return table[key]

    @classmethod
    def load_from_registry(cls, name: str, **constructor_kwargs):
        return cls.get_value_from_registry(name)(**constructor_kwargs)

    @classmethod
    def registered_names(cls) -> list:
        return list(_REGISTRY[cls])
