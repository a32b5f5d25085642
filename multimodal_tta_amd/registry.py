"""Name -> component tables: the drop-in boundary of the adaptation hot path.

Mirrors the behaviour of the reference plugin API (reference: src/registry.py:10-56 for the
``Registry`` class, :60-66 for the seven global tables, :68-124 for ``register_*`` / ``get_*``
and :127-167 for the ``list_*`` helpers).  Behaviour that callers rely on and that the tests
pin against a capture of the reference module (tests/golden/registry_behaviour.json):

* ``register(name)`` is usable as a decorator, ``register(name, obj)`` as a direct call;
  both return the registered object unchanged.
* a duplicate name prints ``Warning: <name> is already registered in <table>`` and overwrites.
* ``get`` of an unknown name raises ``KeyError("<name> is not registered in <table>")``.
* ``list_all`` keeps insertion order.

When the reference package itself is importable (``src.registry`` on ``sys.path``) the tables
below are *the reference's own objects*, so components registered here are visible to an
unmodified ``main.py`` / ``ExperimentManager`` (reference: src/core/experiment_manager.py:88-94).
"""
from __future__ import annotations

import sys
from typing import Any, Callable, Dict, List, Optional

_TABLE_NAMES = (
    "models",
    "datasets",
    "dataset_builders",
    "evaluation_strategies",
    "criteria",
    "providers",
    "plugins",
)


class Registry:
    """One named table.  Same observable behaviour as reference src/registry.py:10-56."""

    def __init__(self, name: str):
        self.name = name
        self._registry: Dict[str, Any] = {}

    def register(self, name: str, cls: Optional[Any] = None) -> Callable:
        def bind(obj):
            if name in self._registry:
                print(f"Warning: {name} is already registered in {self.name}")
            self._registry[name] = obj
            return obj

        return bind if cls is None else bind(cls)

    def get(self, name: str) -> Any:
        try:
            return self._registry[name]
        except KeyError:
            raise KeyError(f"{name} is not registered in {self.name}") from None

    def has(self, name: str) -> bool:
        return name in self._registry

    def list_all(self) -> List[str]:
        return list(self._registry)

    def clear(self) -> None:
        self._registry.clear()


def _adopt_reference_tables() -> Optional[Dict[str, Registry]]:
    """Reuse the reference's live tables if its module is already imported in this process."""
    ref = sys.modules.get("src.registry")
    if ref is None:
        return None
    found = {}
    for t in _TABLE_NAMES:
        obj = getattr(ref, t.upper(), None)
        if obj is None or not hasattr(obj, "register"):
            return None
        found[t] = obj
    return found


_tables = _adopt_reference_tables() or {t: Registry(t) for t in _TABLE_NAMES}

MODELS = _tables["models"]
DATASETS = _tables["datasets"]
DATASET_BUILDERS = _tables["dataset_builders"]
EVALUATION_STRATEGIES = _tables["evaluation_strategies"]
CRITERIA = _tables["criteria"]
PROVIDERS = _tables["providers"]
PLUGINS = _tables["plugins"]


def register_model(name: str):
    return MODELS.register(name)


def register_dataset(name: str):
    return DATASETS.register(name)


def register_dataset_builder(name: str):
    return DATASET_BUILDERS.register(name)


def register_evaluation_strategy(name: str):
    return EVALUATION_STRATEGIES.register(name)


def register_criterion(name: str):
    return CRITERIA.register(name)


def register_provider(name: str):
    return PROVIDERS.register(name)


def register_plugin(name: str):
    return PLUGINS.register(name)


def get_model(name: str):
    return MODELS.get(name)


def get_dataset(name: str):
    return DATASETS.get(name)


def get_dataset_builder(name: str):
    return DATASET_BUILDERS.get(name)


def get_evaluation_strategy(name: str):
    return EVALUATION_STRATEGIES.get(name)


def get_criterion(name: str):
    return CRITERIA.get(name)


def get_provider(name: str):
    return PROVIDERS.get(name)


def get_plugin(name: str):
    return PLUGINS.get(name)


def list_all_components() -> Dict[str, List[str]]:
    return {t: _tables[t].list_all() for t in _TABLE_NAMES}


def list_models():
    return MODELS.list_all()


def list_datasets():
    return DATASETS.list_all()


def list_dataset_builders():
    return DATASET_BUILDERS.list_all()


def list_evaluation_strategies():
    return EVALUATION_STRATEGIES.list_all()


def list_criteria():
    return CRITERIA.list_all()


def list_providers():
    return PROVIDERS.list_all()


def list_plugins():
    return PLUGINS.list_all()
