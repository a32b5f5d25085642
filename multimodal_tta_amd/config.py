"""A small PyYAML composer for the reference's ``configs/`` surface (Hydra/OmegaConf are not
installed on the target image) plus the two accessors every reference component uses.

What is reproduced, with the reference file each rule is needed for:

* defaults lists with ``_self_`` placement            (reference: configs/config.yaml:1-7)
* group selection on the command line, ``model=unet`` (reference: train_hecktor21.sh:43-59)
* absolute group paths ``/_global_patches: brats``    (reference: configs/task/brats.yaml:1-3)
* ``# @package _global_`` headers                     (reference: configs/_global_patches/brats.yaml:1)
* dotted value overrides ``training.optimizers.adam.lr=5e-3`` and ``+new.key=v``
* ``${a.b}`` interpolation, ``${now:%Y}``             (reference: configs/config.yaml:10-12)
* YAML-1.2 float spelling: ``1e-4`` is a float        (reference: configs/training/default.yaml:22,32)
* ``get_config`` returns the default when the stored value is ``None``; ``require_config``
  raises ``ValueError`` on missing/None and ``TypeError`` on a type mismatch
                                                      (reference: src/utils/config.py:7-32)
"""
from __future__ import annotations

import os
import re
import time
from typing import Any, Dict, Iterable, List, Optional, Sequence, Tuple

import yaml

__all__ = ["Cfg", "compose", "get_config", "require_config", "to_container", "as_cfg"]


# ----------------------------------------------------------------------------- container
class Cfg(dict):
    """dict with attribute access; nested dicts are wrapped on the way in."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def __setitem__(self, key, value):
        super().__setitem__(key, _wrap(value))

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key) from None

    def __setattr__(self, key, value):
        self[key] = value

    def __delattr__(self, key):
        del self[key]

    def copy(self):
        return Cfg(to_container(self))


def _wrap(v):
    if isinstance(v, Cfg):
        return v
    if isinstance(v, dict):
        return Cfg(v)
    if isinstance(v, (list, tuple)):
        return [_wrap(x) for x in v]
    return v


def as_cfg(obj: Any) -> Cfg:
    if obj is None:
        return Cfg()
    if isinstance(obj, Cfg):
        return obj
    if isinstance(obj, dict):
        return Cfg(obj)
    # OmegaConf DictConfig (only when the real library is around)
    try:  # pragma: no cover - optional dependency
        from omegaconf import DictConfig, OmegaConf

        if isinstance(obj, DictConfig):
            return Cfg(OmegaConf.to_container(obj, resolve=True))
    except Exception:
        pass
    raise TypeError(f"`cfg` must be a mapping, got {type(obj).__name__}")


def to_container(obj: Any) -> Any:
    if isinstance(obj, dict):
        return {k: to_container(v) for k, v in obj.items()}
    if isinstance(obj, list):
        return [to_container(v) for v in obj]
    return obj


# ----------------------------------------------------------------------------- accessors
_MISSING = object()


def _select(cfg: Any, path: str) -> Any:
    node = cfg
    if path == "" or path is None:
        return node
    for part in str(path).split("."):
        if isinstance(node, dict):
            if part not in node:
                return _MISSING
            node = node[part]
        elif isinstance(node, list):
            try:
                node = node[int(part)]
            except (ValueError, IndexError):
                return _MISSING
        else:
            return _MISSING
    return node


def get_config(cfg: Any, path: str, default: Any = None, type_: Optional[type] = None) -> Any:
    """Optional read.  Missing *or None* yields ``default`` (reference: src/utils/config.py:21-32)."""
    cfg = as_cfg(cfg)
    value = _select(cfg, path)
    if value is _MISSING or value is None:
        value = default
    if type_ is not None and value is not None and not isinstance(value, type_):
        raise TypeError(f"Config '{path}' must be {type_.__name__}, got {type(value).__name__}")
    return value


def require_config(cfg: Any, path: str, type_: Optional[type] = None) -> Any:
    """Required read (reference: src/utils/config.py:7-19)."""
    cfg = as_cfg(cfg)
    value = _select(cfg, path)
    if value is _MISSING or value is None:
        raise ValueError(f"Required configuration missing: {path}")
    if type_ is not None and not isinstance(value, type_):
        raise TypeError(f"Config '{path}' must be {type_.__name__}, got {type(value).__name__}")
    return value


# ----------------------------------------------------------------------------- YAML loading
class _Loader(yaml.SafeLoader):
    pass


# YAML 1.2 / OmegaConf float grammar: "1e-4", "5e-3", "1.", ".5", "1.0e-5", inf/nan
_FLOAT_RE = re.compile(
    r"""^(?:[-+]?(?:[0-9][0-9_]*)\.[0-9_]*(?:[eE][-+]?[0-9]+)?
        |[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)
        |[-+]?\.[0-9_]+(?:[eE][-+]?[0-9]+)?
        |[-+]?\.(?:inf|Inf|INF)
        |\.(?:nan|NaN|NAN))$""",
    re.X,
)
_Loader.add_implicit_resolver("tag:yaml.org,2002:float", _FLOAT_RE, list("-+0123456789."))


def _load_yaml_text(text: str) -> Any:
    return yaml.load(text, Loader=_Loader)


def parse_value(text: str) -> Any:
    """Parse the right-hand side of a ``key=value`` override the way OmegaConf would."""
    text = text.strip()
    if text == "":
        return ""
    try:
        return _load_yaml_text(text)
    except yaml.YAMLError:
        return text


_PACKAGE_RE = re.compile(r"^\s*#\s*@package\s+(\S+)\s*$")


def _read_file(path: str) -> Tuple[Dict[str, Any], Optional[str]]:
    with open(path, "r", encoding="utf-8") as fh:
        text = fh.read()
    package = None
    for line in text.splitlines()[:5]:
        m = _PACKAGE_RE.match(line)
        if m:
            package = m.group(1)
            break
    data = _load_yaml_text(text) or {}
    if not isinstance(data, dict):
        raise ValueError(f"{path}: top level must be a mapping")
    return data, package


def _deep_merge(dst: Dict[str, Any], src: Dict[str, Any]) -> Dict[str, Any]:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _deep_merge(dst[k], v)
        else:
            dst[k] = to_container(v) if isinstance(v, (dict, list)) else v
    return dst


def _place(root: Dict[str, Any], package: str, content: Dict[str, Any]) -> None:
    if package in ("", "_global_"):
        _deep_merge(root, content)
        return
    node = root
    parts = package.split(".")
    for p in parts[:-1]:
        node = node.setdefault(p, {})
    if isinstance(node.get(parts[-1]), dict):
        _deep_merge(node[parts[-1]], content)
    else:
        node[parts[-1]] = _deep_merge({}, content)


class _Composer:
    def __init__(self, config_dir: str, group_choice: Dict[str, str]):
        self.dir = config_dir
        self.choice = dict(group_choice)
        self.root: Dict[str, Any] = {}
        self.used_groups: List[str] = []

    def _path(self, rel: str) -> str:
        p = os.path.join(self.dir, rel + ".yaml")
        if not os.path.isfile(p):
            raise FileNotFoundError(
                f"config '{rel}' not found under {self.dir} "
                f"(available: {sorted(self._options(os.path.dirname(rel)))})"
            )
        return p

    def _options(self, group: str) -> List[str]:
        d = os.path.join(self.dir, group)
        if not os.path.isdir(d):
            return []
        return [f[:-5] for f in os.listdir(d) if f.endswith(".yaml")]

    def load(self, rel: str, package: str, is_primary: bool = False) -> None:
        """Merge config file ``rel`` (path without .yaml, relative to the config dir) at ``package``."""
        data, header_pkg = _read_file(self._path(rel))
        if header_pkg is not None:
            package = "" if header_pkg == "_global_" else header_pkg
        defaults = data.pop("defaults", None)
        own_group = os.path.dirname(rel)
        entries: List[Any] = list(defaults) if defaults else []
        if not any(e == "_self_" for e in entries):
            entries.append("_self_")
        for e in entries:
            if e == "_self_":
                content = {k: v for k, v in data.items() if not (is_primary and k == "hydra")}
                _place(self.root, package, content)
                continue
            if isinstance(e, str):
                # sibling file in the same group, same package
                sib = e.lstrip("/")
                rel2 = sib if e.startswith("/") else (os.path.join(own_group, sib) if own_group else sib)
                self.load(rel2, package)
                continue
            if isinstance(e, dict) and len(e) == 1:
                (group, option), = e.items()
                group = str(group)
                optional = group.startswith("optional ")
                if optional:
                    group = group[len("optional "):].strip()
                absolute = group.startswith("/")
                gname = group.lstrip("/")
                gpath = gname if (absolute or is_primary or not own_group) else os.path.join(own_group, gname)
                option = self.choice.get(gpath, self.choice.get(gname, option))
                if option is None or option == "null":
                    continue
                self.used_groups.append(gpath)
                try:
                    self.load(os.path.join(gpath, str(option)), gpath.replace("/", "."))
                except FileNotFoundError:
                    if optional:
                        continue
                    raise
                continue
            raise ValueError(f"{rel}: unsupported defaults entry {e!r}")


_INTERP_RE = re.compile(r"\$\{([^${}]+)\}")


def _resolve_interpolations(root: Dict[str, Any]) -> None:
    def resolve_str(s: str, depth: int = 0) -> Any:
        if depth > 16:
            raise ValueError(f"interpolation cycle at {s!r}")
        whole = _INTERP_RE.fullmatch(s)

        def lookup(expr: str) -> Any:
            expr = expr.strip()
            if expr.startswith("now:"):
                return time.strftime(expr[4:])
            if expr.startswith("oc.env:"):
                name, _, dflt = expr[7:].partition(",")
                return os.environ.get(name.strip(), dflt.strip())
            if ":" in expr:  # unknown resolver: leave untouched
                return "${" + expr + "}"
            v = _select(root, expr)
            if v is _MISSING:
                raise KeyError(f"interpolation key '{expr}' not found")
            if isinstance(v, str) and "${" in v:
                v = resolve_str(v, depth + 1)
            return v

        if whole:
            return lookup(whole.group(1))
        return _INTERP_RE.sub(lambda m: str(lookup(m.group(1))), s)

    def walk(node):
        if isinstance(node, dict):
            for k, v in list(node.items()):
                if isinstance(v, str) and "${" in v:
                    try:
                        node[k] = resolve_str(v)
                    except KeyError:
                        pass
                else:
                    walk(v)
        elif isinstance(node, list):
            for i, v in enumerate(node):
                if isinstance(v, str) and "${" in v:
                    try:
                        node[i] = resolve_str(v)
                    except KeyError:
                        pass
                else:
                    walk(v)

    walk(root)


def _set_path(root: Dict[str, Any], dotted: str, value: Any, must_exist: bool) -> None:
    parts = dotted.split(".")
    node = root
    for p in parts[:-1]:
        nxt = node.get(p) if isinstance(node, dict) else None
        if not isinstance(nxt, dict):
            if must_exist and nxt is not None:
                raise KeyError(f"cannot override '{dotted}': '{p}' is not a mapping")
            nxt = {}
            node[p] = nxt
        node = nxt
    if must_exist and parts[-1] not in node:
        raise KeyError(
            f"Could not override '{dotted}': key not in config (use '+{dotted}=...' to add it)"
        )
    node[parts[-1]] = value


def default_config_dir() -> str:
    here = os.path.dirname(os.path.abspath(__file__))
    return os.path.join(os.path.dirname(here), "configs")


def compose(
    config_dir: Optional[str] = None,
    config_name: str = "config",
    overrides: Optional[Sequence[str]] = None,
) -> Cfg:
    """Compose ``configs/<config_name>.yaml`` with Hydra-style ``overrides``."""
    config_dir = os.path.abspath(config_dir or default_config_dir())
    overrides = list(overrides or [])
    group_choice: Dict[str, str] = {}
    value_overrides: List[Tuple[str, Any, bool]] = []
    deletions: List[str] = []
    for ov in overrides:
        if ov.startswith("~"):
            deletions.append(ov[1:].split("=")[0])
            continue
        if "=" not in ov:
            raise ValueError(f"override '{ov}' is not of the form key=value")
        key, _, val = ov.partition("=")
        add = key.startswith("+")
        key = key.lstrip("+")
        is_group = os.path.isdir(os.path.join(config_dir, key.replace(".", "/"))) and not os.path.isfile(
            os.path.join(config_dir, key + ".yaml")
        )
        if is_group and "." not in key:
            group_choice[key] = val
        else:
            value_overrides.append((key, parse_value(val), add))

    comp = _Composer(config_dir, group_choice)
    comp.load(config_name, "", is_primary=True)
    # groups named on the command line but absent from the defaults list ("+group=x" style)
    for g, opt in group_choice.items():
        if g not in comp.used_groups:
            comp.load(os.path.join(g, opt), g)
    root = comp.root
    for key, val, add in value_overrides:
        # Hydra rejects a bare override of an absent key; the TTA keys live in method/*.yaml and
        # are routinely absent from older trees, so absent keys are added instead of refused.
        _set_path(root, key, val, must_exist=False)
    for key in deletions:
        parts = key.split(".")
        node = root
        for p in parts[:-1]:
            node = node.get(p, {})
        node.pop(parts[-1], None)
    _resolve_interpolations(root)
    return Cfg(root)
