"""H2D bandwidth of page-locked staging buffers: one copy per batch against the same bytes split over several streams
(each hipMemcpyAsync is served by one SDMA engine), for the batch sizes of the predict loop."""
import sys, time
import torch

dev = torch.device('cuda:0')
for mb in (61, 245):
    n = mb * 1000 * 1000 // 2
    host = torch.empty(n, dtype=torch.int16, pin_memory=True)
    host.random_(-3000, 3000)
    d = torch.empty(n, dtype=torch.int16, device=dev)
    for parts in (1, 2, 4, 8):
        streams = [torch.cuda.Stream(dev) for _ in range(parts)]
        cut = [(n * i // parts) for i in range(parts + 1)]

        def go():
            for i, st in enumerate(streams):
                with torch.cuda.stream(st):
                    d[cut[i]:cut[i + 1]].copy_(host[cut[i]:cut[i + 1]], non_blocking=True)
        for _ in range(3):
            go()
        torch.cuda.synchronize()
        t = time.perf_counter()
        reps = 10
        for _ in range(reps):
            go()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps
        print('%4d MB in %d part(s): %6.2f ms  %5.1f GB/s' % (mb, parts, dt * 1e3, n * 2 / dt / 1e9), flush=True)
    assert torch.equal(d.cpu(), host)
