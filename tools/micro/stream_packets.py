"""What does an event record / an event wait cost a chain of dependent kernels on one stream?
A chain of N small kernels on stream A (each ~`us` long), with between consecutive kernels
  plain        nothing
  record       an event record on A (no waiter)
  record+wait  the record, and stream B waits for it and runs a small kernel (a fork)
  wait-done    A waits for an event of stream B that completed long ago
  wait-live    B runs a short kernel, records, A waits for it (a join of work that ends at about the same time)
  fork+join    record on A, B waits + kernel + record, A waits  (one full cross-lane hop per link)
Reported: microseconds per link of the chain beyond the kernel itself.
usage: python stream_packets.py [N]"""
import ctypes, sys, time, torch
torch.cuda.init()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
A, B = torch.cuda.Stream(), torch.cuda.Stream()
SPIN = 200_000_000           # ~80 ms at 2.4 GHz
xa = torch.zeros(1 << 16, device='cuda')
xb = torch.zeros(1 << 16, device='cuda')
old = torch.cuda.Event()
with torch.cuda.stream(B):
    xb.add_(1.0)
    old.record(B)
torch.cuda.synchronize()


class RawEvent:
    """hipEventCreateWithFlags through libamdhip64 (torch.cuda.Event offers hipEventDisableTiming only)."""
    hip = ctypes.CDLL('libamdhip64.so')

    def __init__(self, flags):
        self.h = ctypes.c_void_p()
        assert self.hip.hipEventCreateWithFlags(ctypes.byref(self.h), ctypes.c_uint(flags)) == 0

    def record(self, stream):
        assert self.hip.hipEventRecord(self.h, ctypes.c_void_p(stream.cuda_stream)) == 0

    def __del__(self):
        self.hip.hipEventDestroy(self.h)


def wait_event(stream, ev):
    if isinstance(ev, RawEvent):
        assert RawEvent.hip.hipStreamWaitEvent(ctypes.c_void_p(stream.cuda_stream), ev.h, 0) == 0
    else:
        stream.wait_event(ev)


FLAGS = None                  # None: torch.cuda.Event(); else the hipEventCreateWithFlags word


def chain(kind):
    evs = [torch.cuda.Event() if FLAGS is None else RawEvent(FLAGS) for _ in range(2 * N)]
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    # the host enqueues the whole chain while a spin kernel holds stream A (and B behind it): what is timed is the device
    # working through the packets, not the host submitting them
    with torch.cuda.stream(A):
        torch.cuda._sleep(SPIN)
    t0.record(A)
    B.wait_event(t0)
    h0 = time.perf_counter()
    for i in range(N):
        with torch.cuda.stream(A):
            xa.add_(1.0)
        if kind == 'record':
            evs[i].record(A)
        elif kind == 'record+wait':
            evs[i].record(A)
            wait_event(B, evs[i])
            with torch.cuda.stream(B):
                xb.add_(1.0)
        elif kind == 'wait-done':
            A.wait_event(old)
        elif kind == 'wait-live':
            with torch.cuda.stream(B):
                xb.add_(1.0)
            evs[i].record(B)
            wait_event(A, evs[i])
        elif kind == 'fork+join':
            evs[i].record(A)
            wait_event(B, evs[i])
            with torch.cuda.stream(B):
                xb.add_(1.0)
            evs[N + i].record(B)
            wait_event(A, evs[N + i])
    t1.record(A)
    host = time.perf_counter() - h0
    torch.cuda.synchronize()
    assert host < 0.06, f'host took {host * 1e3:.0f} ms to enqueue: raise SPIN'   
    return t0.elapsed_time(t1) * 1e3 / N


for name, flags in (('torch.cuda.Event()', None), ('hipEventDisableTiming', 0x2),
                    ('hipEventDisableTiming | hipEventDisableSystemFence', 0x2 | 0x20000000),
                    ('hipEventDisableTiming | hipEventReleaseToDevice', 0x2 | 0x40000000)):
    FLAGS = flags
    print(f'== events: {name}')
    for kind in ('plain', 'record', 'record+wait', 'wait-done', 'wait-live', 'fork+join'):
        if flags is not None and kind in ('plain', 'wait-done'):
            continue
        chain(kind)
        r = sorted(chain(kind) for _ in range(5))
        print(f'{kind:12s} {r[2]:7.2f} us per link   (min {r[0]:.2f}, max {r[4]:.2f})')
print('# the plain link is kernel + dispatch; the other rows minus it = what the packets add to a dependent chain')
