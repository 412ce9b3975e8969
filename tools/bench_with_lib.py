"""Runs bench.py against a variant build of the extension (kernel experiments): LSK_LIB=/path/to/variant.so python tools/bench_with_lib.py [bench flags]"""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layerskip_amd import _lib  # noqa: E402

_lib._LIBS["bf16"] = _lib.load(os.environ["LSK_LIB"])
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
