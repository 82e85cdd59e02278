"""cProfile of the bench's host side (GPU box): python scripts/host_profile.py [bench args...]
Prints the top functions by internal time and the callers of the usual suspects (synchronous copies, device queries)."""
import cProfile, pstats, runpy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py"] + (sys.argv[1:] or ["--rays", "7000", "--steps", "200", "--warmup", "5", "--also=", "--no-cpu-baseline"])
cProfile.run("runpy.run_path('bench.py', run_name='__main__')", "/tmp/host_prof.out")
p = pstats.Stats("/tmp/host_prof.out")
p.sort_stats("tottime").print_stats(25)
for pat in ("_cuda_getDeviceCount", "method 'to' of", "method 'item'", "cuda_synchronize", "is_available"):
    p.print_callers(pat)
