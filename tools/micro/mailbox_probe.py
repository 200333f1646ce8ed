"""When does the host see a mid-graph mapx_publish_i32 store?  Graph = [spin A | publish | spin B]
on one stream, and with the publish forked onto a side stream; prints microseconds from replay to
visibility and to graph end."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "map-code_amd"))
import torch
from mapx import ops

dev = torch.device("cuda:0")
src = torch.tensor([7, 0], dtype=torch.int32, device=dev)
SPIN = 2_000_000     # ~1 ms


def run(fork, eager=False):
    stamp = torch.zeros(1, dtype=torch.int32, device=dev)
    box = ops.HostMailbox(4)
    side = torch.cuda.Stream()
    def body():
        torch.cuda._sleep(SPIN)
        if fork:
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                ops.publish_i32(src, 1, stamp, box)
            torch.cuda._sleep(SPIN)
            main.wait_stream(side)
        else:
            ops.publish_i32(src, 1, stamp, box)
            torch.cuda._sleep(SPIN)
    body(); torch.cuda.synchronize()
    g = None
    if not eager:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body()
    res = []
    for it in range(5):
        torch.cuda.synchronize()
        expect = int(box.np[1]) + 1
        t0 = time.perf_counter()
        g.replay() if g else body()
        t1 = time.perf_counter()
        while int(box.np[1]) != expect:
            if time.perf_counter() - t0 > 5: break
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        res.append((1e6 * (t1 - t0), 1e6 * (t2 - t0), 1e6 * (t3 - t0)))
    print(f"fork={fork} eager={eager}: launch/visible/end us:", [tuple(round(x) for x in r) for r in res[1:]])


for fork in (False, True):
    for eager in (True, False):
        run(fork, eager)
