"""What the box's HBM delivers to plain streaming kernels ("confirm with a stream benchmark on the box"):
device-to-device copy, read-only reduction and fill of 4 GiB buffers through torch, median of 10.  The bench line's
roofline keeps the nominal 8 TB/s as its peak; this file says what a memory-bound kernel actually reaches."""
import time, statistics, torch
n = 1 << 30   # 4 GiB of int32
a = torch.empty(n, dtype=torch.int32, device="cuda").random_(0, 100)
b = torch.empty_like(a)
def timed(f, reps=10):
    ts = []
    for _ in range(reps + 2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return statistics.median(ts[2:])
nbytes = a.numel() * 4
t = timed(lambda: b.copy_(a)); print(f"copy  (read 4 GiB + write 4 GiB): {2 * nbytes / t / 1e12:.2f} TB/s moved, {t * 1e3:.2f} ms")
t = timed(lambda: a.sum());    print(f"read  (sum of 4 GiB):             {nbytes / t / 1e12:.2f} TB/s, {t * 1e3:.2f} ms")
t = timed(lambda: b.fill_(7)); print(f"write (fill of 4 GiB):            {nbytes / t / 1e12:.2f} TB/s, {t * 1e3:.2f} ms")
print(torch.cuda.get_device_name(0))
