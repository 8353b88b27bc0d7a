"""Development aid: host-side profile (cProfile) of the GAN iteration."""
import cProfile, pstats, os, sys, io, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--workload", "gan_step", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-extra"]
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
except SystemExit:
    pass
pr.disable()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(28)
print(st.getvalue()[:6000])
