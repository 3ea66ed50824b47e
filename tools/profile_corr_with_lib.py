#!/usr/bin/env python3
"""tools/profile_corr.py on another build of the library (DEVO_LIB=devo_amd/lib/libdevo_<tag>.so, see tools/build_variant.sh)."""
import os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import devo_amd._lib as L
if os.environ.get("DEVO_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["DEVO_LIB"])
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profile_corr.py"), run_name="__main__")
