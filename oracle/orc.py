"""TEST INFRASTRUCTURE — loader for the CPU restatement (oracle/liborc.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the product package (renderer_amd/) never does.
"""
import os
import subprocess

from renderer_amd import _binding as B
from renderer_amd.api import Renderer

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liborc.so")


def build(force=False):
    """Compiles liborc.so with g++ (see oracle/Makefile)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s", "liborc.so"], check=True)
    return LIB_PATH


_capi = None


def capi():
    global _capi
    if _capi is None:
        if not os.path.exists(LIB_PATH):
            build()
        _capi = B.CApi(LIB_PATH, "orc_")
    return _capi


def oracle_renderer(width, height, settings=None):
    return Renderer(capi(), width, height, settings, 0)
