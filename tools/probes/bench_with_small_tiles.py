import os, sys, runpy
sys.path.insert(0, os.getcwd())
from dynadjust_amd import _lib
lib = _lib.load()
thr = int(os.environ["SMALL_TILES"])
if thr >= 0:
    lib.dnagpu_debug_set_small_tiles(thr)
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path("bench.py", run_name="__main__")
