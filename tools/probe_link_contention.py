"""Does the host side of the predict loop slow the H2D link down?  The loop's readers copy page-cache pages into page-locked
staging buffers (pread: a DRAM read + a DRAM write per byte) while the SDMA engine reads other staging buffers over PCIe.
This probe times the link alone, then again with N threads doing the readers' memory traffic (numpy copies of 240 MB blocks
into page-locked buffers), for N = 0, 4, 8, 13.   python tools/probe_link_contention.py"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench

dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
src = np.random.default_rng(0).integers(0, 255, 240 << 20, dtype=np.uint8)
print('link alone: %.2f GB/s' % bench.link_only_probe(dev))
for n in (4, 8, 13):
    stop = threading.Event()
    moved = [0] * n
    bufs = [torch.empty(240 << 20, dtype=torch.uint8, pin_memory=True) for _ in range(min(n, 4))]

    def work(k):
        dst = bufs[k % len(bufs)].numpy()
        lo = (k // len(bufs)) * (60 << 20) % (180 << 20)
        while not stop.is_set():
            np.copyto(dst[lo:lo + (60 << 20)], src[lo:lo + (60 << 20)])
            moved[k] += 60 << 20
    th = [threading.Thread(target=work, args=(k,), daemon=True) for k in range(n)]
    for t in th:
        t.start()
    time.sleep(0.3)
    m0, t0 = sum(moved), time.perf_counter()
    link = bench.link_only_probe(dev, copies=48)
    dt = time.perf_counter() - t0
    cp = (sum(moved) - m0) / dt / 1e9
    stop.set()
    for t in th:
        t.join()
    print('%2d copy threads (%.1f GB/s of host copies = %.0f k clips/s staged): link %.2f GB/s' % (n, cp, cp / 0.96 * 1e3 / 1e3, link))
