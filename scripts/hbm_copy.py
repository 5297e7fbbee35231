"""Achievable HBM rate on the box, for context next to the stencil kernels' figures: device-to-device copy of 1 GiB
(read + write counted), a read-only reduction and a write-only fill, through PyTorch's own kernels."""
import json, time, torch
n = 1 << 28   # floats: 1 GiB
a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)
def t(fn, reps=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
gb = n * 4 / 1e9
out = {"copy_GBps (read + write)": round(2 * gb / t(lambda: b.copy_(a)), 1),
       "read_GBps (sum)": round(gb / t(lambda: a.sum()), 1),
       "write_GBps (fill)": round(gb / t(lambda: b.fill_(1.0)), 1),
       "device": torch.cuda.get_device_name(0)}
print(json.dumps(out))
