"""How many HIP streams of one process run CONCURRENTLY?  n streams each get one spinning 1-workgroup kernel (torch.cuda._sleep);
if they all run side by side the batch takes one kernel's time, if streams share hardware queues it takes a multiple.
usage: [GPU_MAX_HW_QUEUES=k] python hw_queues.py"""
import os, time, torch
torch.cuda.init()
cyc = 100_000_000
torch.cuda._sleep(1000); torch.cuda.synchronize()
t0 = time.perf_counter(); torch.cuda._sleep(cyc); torch.cuda.synchronize(); one = time.perf_counter() - t0
print(f'GPU_MAX_HW_QUEUES={os.environ.get("GPU_MAX_HW_QUEUES", "(unset)")}: one spin kernel {one * 1e3:.1f} ms')
streams = [torch.cuda.Stream() for _ in range(16)]
for n in (2, 3, 4, 5, 6, 8, 12, 16):
    for with_null in (False, True):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if with_null:
            torch.cuda._sleep(cyc)
        for s in streams[:n - (1 if with_null else 0)]:
            with torch.cuda.stream(s):
                torch.cuda._sleep(cyc)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f'  {n:2d} streams{" (null stream among them)" if with_null else "":26s}: {dt * 1e3:7.1f} ms = {dt / one:4.2f} x one kernel')
