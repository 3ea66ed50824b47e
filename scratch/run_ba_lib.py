import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import devo_amd._lib as L
if os.environ.get("DEVO_LIB"): L.LIB_PATH = os.path.abspath(os.environ["DEVO_LIB"])
sys.argv = [sys.argv[0], "--reps", "2"]
exec(open(os.path.join(os.path.dirname(__file__), "..", "tools", "profile_ba.py")).read())
