import cProfile, pstats, sys, os, io
sys.path.insert(0, '/root/repo')
sys.argv = ['bench_train.py', '--steps', '3', '--warmup', '2']
import tools.bench_train as bt
pr = cProfile.Profile()
pr.enable()
bt.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
print(s.getvalue()[:7000])
