#!/usr/bin/env python
"""Host-side (Python / dispatcher) cost of a training step: cProfile over STEPS eager steps of a benchmarks/model_steps.py
model, top entries by own time.  The eager RevGCN / ResGCN steps are host-bound (GPU busy 11 of 21 ms / 20 of 24 ms).

    python benchmarks/host_profile.py {revgcn8|resgcn28|deepergcn28} [STEPS]
"""
import cProfile
import io
import os
import pstats
import runpy
import sys

which = sys.argv[1]
steps = sys.argv[2] if len(sys.argv) > 2 else "10"
here = os.path.dirname(os.path.abspath(__file__))
sys.argv = [os.path.join(here, "model_steps.py"), which, "3"]
ns = runpy.run_path(sys.argv[0], run_name="__main__")      # builds the model, runs 3 warm steps
step, torch = ns["step"], ns["torch"]
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)    # the backward's Python runs on this thread: visible to cProfile
pr = cProfile.Profile()
pr.enable()
for _ in range(int(steps)):
    step()
torch.cuda.synchronize()
pr.disable()
out = io.StringIO()
st = pstats.Stats(pr, stream=out)
st.sort_stats("tottime").print_stats(45)
print(out.getvalue())
