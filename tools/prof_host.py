#!/usr/bin/env python
"""Host-side profile of a bench command with the autograd engine kept on the calling thread (so that cProfile sees inside the
backward functions).  `python tools/prof_host.py [bench.py flags...]` - prints the 60 most expensive functions by cumulative time."""
import cProfile, os, pstats, runpy, sys
import torch
torch.autograd.set_multithreading_enabled(False)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[1:]
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(sys.argv[0], run_name="__main__")
finally:
    pr.disable()
    st = pstats.Stats(pr, stream=sys.stderr)
    st.sort_stats("tottime").print_stats(45)
