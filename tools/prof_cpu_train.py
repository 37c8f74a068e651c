import cProfile, pstats, sys, os, io
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _ROOT)
sys.argv = ['bench_train.py', '--steps', '3', '--warmup', '2']
import tools.bench_train as bt
pr = cProfile.Profile()
pr.enable()
bt.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
print(s.getvalue()[:7000])
