"""Achievable HBM rates on this box with plain torch kernels: fill (write only), copy (read + write), sum (read only)."""
import torch, time
dev = torch.device("cuda")
for mb in (256, 1024, 4096):
    n = mb * (1 << 20) // 8
    a = torch.empty(n, dtype=torch.float64, device=dev); b = torch.empty_like(a)
    def timed(fn, reps=20):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3
    t_fill = timed(lambda: a.fill_(1.0)); t_copy = timed(lambda: b.copy_(a)); t_sum = timed(lambda: a.sum())
    print(f"{mb} MB: fill {mb/1024/t_fill/1e3*1.0737:.2f} TB/s   copy (r+w) {2*mb/1024/t_copy/1e3*1.0737:.2f} TB/s   sum {mb/1024/t_sum/1e3*1.0737:.2f} TB/s")
