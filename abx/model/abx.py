"""`abx.model.abx` of the reference (abx/model/abx.py) -> MI355X implementation."""
from abx_amd.model.abx import ScoreNetwork, get_prev  # noqa: F401
