"""ONE environment switch for the A/B tools instead of one per threshold: ZIGMA_KNOBS="module.NAME=value,module.NAME=value" overrides module-level
routing constants of this package at import (module = mamba_simple | model_zigma | linear | selective_scan_interface).  Only names that already
exist can be set (a typo raises), values are parsed as int, float, True / False or left as strings.  The shipped defaults are the constants in the
modules; nothing in the product path reads any other new environment variable (the older per-feature switches are listed in DESIGN.md §3.5)."""
import os


def _parse(v):
    if v in ("True", "False"):
        return v == "True"
    for cast in (int, float):
        try:
            return cast(v)
        except ValueError:
            pass
    return v


def apply(module_globals, module_name):
    spec = os.environ.get("ZIGMA_KNOBS", "").strip()
    if not spec:
        return
    for item in spec.split(","):
        item = item.strip()
        if not item:
            continue
        key, _, val = item.partition("=")
        mod, _, name = key.strip().rpartition(".")
        if mod != module_name:
            continue
        if name not in module_globals or name.startswith("_"):
            raise RuntimeError(f"ZIGMA_KNOBS: {module_name} has no knob {name!r}")
        module_globals[name] = _parse(val.strip())
