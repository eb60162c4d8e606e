"""Alias package: lets the reference's drivers (`from abx.model.abx import ScoreNetwork, get_prev`) import the
MI355X implementation unchanged.  Everything lives in abx_amd/."""
