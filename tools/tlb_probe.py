"""tools/tlb_probe.py -- independent random 8-byte gathers over buffers of growing size (run on the GPU box): how much of the
random-line rate is lost to address translation once the working set is tens of GB (jump table, suffix arrays)."""
import sys, time
import torch
n = 1 << 27
for gb in (1, 4, 16, 48, 96, 160):
    elems = gb * (1 << 30) // 8
    buf = torch.empty(elems, dtype=torch.int64, device="cuda")
    buf.zero_()
    idx = torch.randint(0, elems, (n,), device="cuda")
    out = buf[idx]; torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        out = buf[idx]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print("%4d GB: %.2f G gathers/s (%.2f ms)" % (gb, n / dt / 1e9, dt * 1e3), flush=True)
    del buf, idx, out
    torch.cuda.empty_cache()
