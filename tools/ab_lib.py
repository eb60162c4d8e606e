#!/usr/bin/env python
"""Run a tool against an alternative build of libabx_hip.so (A/B of kernel variants on the same box):
    python tools/ab_lib.py tools/probes/bin/libabx_hip_x.so tools/kbench.py --only tri"""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import abx_amd._lib as _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
