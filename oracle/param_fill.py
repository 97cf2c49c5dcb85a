"""Re-export of tests/golden/param_fill.py (the deterministic weight fill IS part of the round-2 / round-6 fixtures: their weights are not stored, both sides
regenerate them from the fixture's seed) for the generator scripts and tests that have always imported it from here."""
import importlib.util
import os

_spec = importlib.util.spec_from_file_location("zigma_golden_param_fill", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "param_fill.py"))
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
fill_state, _value = _mod.fill_state, _mod._value
