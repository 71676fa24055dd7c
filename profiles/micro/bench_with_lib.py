"""bench.py with another build of the library (arg 1 = path of the .so): A/B of kernel variants on the same box."""
import os, runpy, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(R, "hmer-img2latex_amd"))
from img2latex_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(R, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
