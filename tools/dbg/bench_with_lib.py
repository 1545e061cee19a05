#!/usr/bin/env python3
"""bench.py against another build of the library (A/B of compiler flags on one box):
    DBG_LIB=build_dbg/lib_x.so python3 tools/dbg/bench_with_lib.py --steps 400 --warmup 40 --no-cpu-baseline --no-secondary --no-roofline"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bsarec_amd import _lib as Lb
if os.environ.get("DBG_LIB"):
    Lb.LIB_PATH = os.path.join(ROOT, os.environ["DBG_LIB"])
import bench
bench.main()
