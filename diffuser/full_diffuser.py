"""`diffuser.full_diffuser` of the reference (diffuser/full_diffuser.py) -> MI355X implementation."""
from abx_amd.diffuser.full_diffuser import FullDiffuser, diffuser_obj_dict  # noqa: F401
