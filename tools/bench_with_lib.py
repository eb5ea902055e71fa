"""Diagnostic: bench.py with an alternative build of the library
(python tools/bench_with_lib.py path/to/lib.so [bench.py args...])."""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instaorder_amd import ext
ext._LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
