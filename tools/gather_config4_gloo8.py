#!/usr/bin/env python3
"""The one collective of the sampling path at BASELINE config 4's size, without compute: world size 8 over gloo, 102 400 graphs' worth of rows
(n ~ N(40, 6^2): 4.1 M atoms, 164 M bond rows, 4.2 GB of fp32), partitioned as `run_sampling_job` does, gathered to rank 0 by
`phoregen_amd.parallel.gather_predictions` (3 collectives).  Prints global-order check, collectives per rank, partition / gather seconds and
peak host memory per rank -> profiles/r06_gather_config4_gloo8.txt.   usage: python tools/gather_config4_gloo8.py [n_graphs] [world]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_parallel_gloo import run_job_gather

if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 102400
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    res = run_job_gather(n, world)
    print(f'{n} graphs, world size {world}, gloo on {os.cpu_count()} host cores')
    print('rank  in_global_order  collectives  partition_s  gather_s  peak_rss_GB  payload_GB')
    for r, ok, calls, t_part, t_gather, rss, payload in res:
        print(f'{r:4d}  {str(ok):15s}  {calls:11d}  {t_part:11.2f}  {t_gather:8.2f}  {rss / 1e9:11.2f}  {payload / 1e9:10.2f}')
