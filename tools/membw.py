"""HBM bandwidth calibration on the box: torch copy / fill / sum over buffers of the sweep's size."""
import torch
n = 544 * 512 * 512
a = torch.rand(n, device="cuda"); b = torch.empty_like(a)
def t(f, reps=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
s = t(lambda: b.copy_(a)); print(f"copy  {n*4/1e6:.0f} MB -> {2*n*4/s/1e12:.2f} TB/s ({s*1e6:.1f} us)")
s = t(lambda: b.fill_(1.0)); print(f"fill  -> {n*4/s/1e12:.2f} TB/s ({s*1e6:.1f} us)")
s = t(lambda: a.sum()); print(f"sum   -> {n*4/s/1e12:.2f} TB/s ({s*1e6:.1f} us)")
c = torch.empty_like(a)
s = t(lambda: torch.add(a, b, out=c)); print(f"add   -> {3*n*4/s/1e12:.2f} TB/s ({s*1e6:.1f} us)")
