"""Does an EXTERNAL event-record node of a replayed hipGraph order a stream OUTSIDE the graph on this runtime?

The overlapped data-parallel step (trainers/graph.py) relies on it: the captured step records an external event
(mvk_event_record(ev, 1, stream) = hipEventRecordWithFlags + hipEventRecordExternal) where a range of the gradient buffer is
final, and after `graph.replay()` the communication stream does mvk_stream_wait_event(comm, ev) and starts that range's
all-reduce.  The hazard: a wait issued by the host BEFORE the node has executed must still wait for THIS replay's record, not
be satisfied by the previous replay's.

Probe: graph = [A: spin ~300 us, then buf[0] = step] -> external record -> [B: spin ~300 us].  After every replay, stream 2:
wait(ev); seen[i] = buf[0] (a device-side copy).  If the wait is honoured for the current replay, seen[i] == i + 1 for every i;
a wait satisfied by the previous record reads i (or 0).  Also times the replay to show the copy did not wait for B (overlap).
"""
import ctypes as C
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multivae_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
ev = C.c_void_p()
_lib.call("mvk_event_create", C.byref(ev))
side = _lib.new_stream(dev)
N = 200
buf = torch.zeros(1, device=dev, dtype=torch.int64)
step = torch.zeros(1, device=dev, dtype=torch.int64)
seen = torch.full((N,), -1, device=dev, dtype=torch.int64)
cap = _lib.new_stream(dev)
SPIN = 600_000  # cycles of torch.cuda._sleep (~300 us)

cap.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(cap):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap):
        torch.cuda._sleep(SPIN)
        step.add_(1)
        buf.copy_(step)
        _lib.call("mvk_event_record", ev, 1, _lib.stream_ptr())
        torch.cuda._sleep(SPIN)
torch.cuda.synchronize()

t_copy_done = []
e_start = [torch.cuda.Event(enable_timing=True) for _ in range(N)]
e_side = [torch.cuda.Event(enable_timing=True) for _ in range(N)]
e_end = [torch.cuda.Event(enable_timing=True) for _ in range(N)]
main = torch.cuda.current_stream()
for i in range(N):
    e_start[i].record(main)
    g.replay()
    _lib.call("mvk_stream_wait_event", C.c_void_p(side.cuda_stream), ev)
    with torch.cuda.stream(side):
        seen[i:i + 1].copy_(buf)
        e_side[i].record(side)
    e_end[i].record(main)
    main.wait_stream(side)  # the next replay's A must not overwrite buf before the copy ran
torch.cuda.synchronize()
got = seen.cpu().tolist()
want = list(range(1, N + 1))
bad = [(i, a, b) for i, (a, b) in enumerate(zip(got, want)) if a != b]
side_ms = sorted(e_start[i].elapsed_time(e_side[i]) for i in range(N))[N // 2]
end_ms = sorted(e_start[i].elapsed_time(e_end[i]) for i in range(N))[N // 2]
print(f"external event order probe: {N - len(bad)}/{N} replays saw the current step's value behind the wait; first mismatches {bad[:5]}")
print(f"median: side copy done {side_ms * 1e3:.0f} us after the replay started, graph end {end_ms * 1e3:.0f} us "
      f"(overlap {'YES' if side_ms < 0.8 * end_ms else 'NO'}: the side stream ran before the graph's second half ended)")
_lib.call("mvk_event_destroy", ev)
sys.exit(0 if not bad else 1)
