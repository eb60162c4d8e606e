"""abx_amd — MI355X-native reverse-diffusion sampling hot path of AbX.

Public surface mirrors the reference:  abx_amd.model.abx.{ScoreNetwork,get_prev},
abx_amd.diffuser.full_diffuser.FullDiffuser (also importable as `abx.model.abx` / `diffuser.full_diffuser`
through the alias packages at the repo root).
"""
__version__ = '0.1.0'
