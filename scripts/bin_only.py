"""The binning pass of the training-shape forward alone (profiling aid)."""
import os, sys
sys.argv = [sys.argv[0]]
os.environ["ONLY_BIN"] = "1"
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fwd_vc_variants.py")).read())
