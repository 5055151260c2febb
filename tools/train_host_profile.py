#!/usr/bin/env python
"""Where the HOST spends a training step (cProfile over tools/train_bench.py's loop): the device is idle for whatever the host takes
between the forward's last launch and the backward's first.   python tools/train_host_profile.py [--precision f16]"""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv = [sys.argv[0]] + sys.argv[1:] + ["--steps", "40"]
import train_bench
pr = cProfile.Profile()
pr.enable()
train_bench.main()
pr.disable()
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(35)
    print(s.getvalue()[:9000])
