#!/usr/bin/env python3
"""Development aid (GPU): bench.py against lib/libcvd_hip_base.so, a library built from an EARLIER commit (A/B on one box).
The earlier cvd_solver_options is a prefix of the current one: the struct is cut at `last_field`.
usage: bench_base.py <last option field of the old header> [bench.py arguments]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robust_cvd_amd import api
last = sys.argv[1]
names = [n for n, _ in api.SolverOptions._fields_]
cut = names.index(last) + 1


class OldOptions(C.Structure):
    _fields_ = api.SolverOptions._fields_[:cut]


api.SolverOptions = OldOptions
api.load_library(variant="base")
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
