#!/usr/bin/env python
"""Development aid: run a script of this repo against another build of the shared library (an A/B or instrumented build under ab_libs/).

    python tools/run_with_lib.py ab_libs/libslu_x.so bench.py --no-cpu-baseline ..."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semanticlidarunc_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
script = sys.argv[2]
sys.argv = sys.argv[2:]
sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
runpy.run_path(script, run_name="__main__")
