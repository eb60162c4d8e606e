"""Alias package: `from diffuser.full_diffuser import FullDiffuser` as in the reference's inference.py:24."""
