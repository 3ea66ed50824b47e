import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import devo_amd._lib as L
if os.environ.get("DEVO_LIB"): L.LIB_PATH = os.path.abspath(os.environ["DEVO_LIB"])
sys.argv = [sys.argv[0], "--no-cpu-baseline"] + sys.argv[1:]
import bench
bench.main()
