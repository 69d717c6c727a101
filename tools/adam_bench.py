"""Dev tool: dense vs active Adam on fully active tensors (cfg-2 levels) and on a mostly idle one."""
import sys, torch
sys.path.insert(0, '.')
from miso_amd import ops
dev = 'cuda:0'


def t_us(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for numel, frac in ((8 * 128 ** 3, 1.0), (8 * 64 ** 3, 1.0), (8 * 32 ** 3, 1.0), (4 * 100 * 600 * 600, 0.001)):
    p = torch.randn(numel, device=dev)
    g = torch.randn(numel, device=dev) * 1e-3
    if frac < 1:
        keep = (torch.rand(numel // 256, device=dev) < frac).repeat_interleave(256)
        g = g * keep
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    act = ops.adam_active_flags(p)
    for z in (False, True):
        gd = g.clone()
        td = t_us(lambda: ops.adam_dense_(p, gd, m, v, 3, 1e-3, zero_grad=False))
        ta = t_us(lambda: ops.adam_active_(p, gd, m, v, act, 3, 1e-3, zero_grad=False))
        print(f"numel {numel:>10} active {frac:5.3f}: dense {td:7.1f} us ({28 * numel / td / 1e6:5.2f} TB/s)  "
              f"active-chunk {ta:7.1f} us")
        break
