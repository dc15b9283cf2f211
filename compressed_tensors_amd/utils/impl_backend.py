"""Priority-based backend dispatch with the reference's interface (utils/impl_backend.py:23-134).

Kept so that code written against the reference's plug-in point keeps working: a function
decorated with `ImplBackend.entrypoint(name)` tries the backends registered under `name` in
ascending priority and runs its own body when none accepts the arguments.  In this package the
entrypoints' own bodies already are the HIP path; the registry exists for drop-in parity and for
`install()` (compressed_tensors_amd/install.py), which registers the HIP `_quantize` backend
into a live upstream install.  CT_ENFORCE_EAGER is honoured like upstream (:14,115).
"""
import functools
import os
from typing import Callable

__all__ = ["ImplBackend"]

ENFORCE_EAGER = os.environ.get("CT_ENFORCE_EAGER", False)


class ImplBackend:
    _backends: dict = {}
    _fn_registry: dict = {}

    @classmethod
    def register(cls, name: str, req: Callable[..., bool], priority):
        def decorator(backend_fn: Callable) -> Callable:
            if priority == "disable":
                return backend_fn
            cls._add_to_registry(backend_fn)
            cls._backends.setdefault(name, []).append((backend_fn, req, priority))
            cls._backends[name].sort(key=lambda entry: entry[2])
            return backend_fn

        return decorator

    @classmethod
    def call(cls, fn_name: str, *args, **kwargs):
        if fn_name not in cls._fn_registry:
            raise KeyError(f"No registered backend named '{fn_name}'. Available: {list(cls._fn_registry)}")
        return cls._fn_registry[fn_name](*args, **kwargs)

    @classmethod
    def entrypoint(cls, name: str) -> Callable:
        def decorator(fallback_fn: Callable) -> Callable:
            cls._add_to_registry(fallback_fn)

            @functools.wraps(fallback_fn)
            def wrapper(*args, **kwargs):
                if not ENFORCE_EAGER:
                    for backend_fn, req, _ in cls._backends.get(name, []):
                        if req(*args, **kwargs):
                            return backend_fn(*args, **kwargs)
                return fallback_fn(*args, **kwargs)

            return wrapper

        return decorator

    @classmethod
    def _add_to_registry(cls, fn: Callable):
        if fn.__name__ in cls._fn_registry:
            raise ValueError(
                f"A backend with function name '{fn.__name__}' is already registered. "
                "Backend function names must be unique across all ops."
            )
        cls._fn_registry[fn.__name__] = fn
