#!/usr/bin/env python3
"""tools/bench_with_lib.py <lib.so> [bench.py flags...] -- bench.py against an exploration build of the library
(tools/build_pair_variant.sh): A/B timing of kernel variants inside the real step.  Never the product library."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphmaker_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench  # noqa: E402

bench.main()
