"""Developer helper: run bench.py with kernel-selection options set first.  python tools/bench_opt.py name=value ... [-- bench args]"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tpu_superresolution_amd._lib import check, lib
args = sys.argv[1:]
rest = []
if "--" in args:
    i = args.index("--"); args, rest = args[:i], args[i + 1:]
for a in args:
    k, v = a.split("=")
    check(lib().srk_set_option(k.encode(), int(v)))
sys.argv = [os.path.join(ROOT, "bench.py")] + rest
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
