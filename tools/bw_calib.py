import torch
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
for mb in (300, 1200, 4000):
    n=mb*1024*1024//2
    a=torch.empty(n,dtype=torch.bfloat16,device='cuda'); b=torch.empty_like(a)
    ms=t(lambda: a.zero_()); print('fill  %5d MB: %.3f ms  %.0f GB/s write' % (mb, ms, mb*1.048576/ms))
    ms=t(lambda: b.copy_(a)); print('copy  %5d MB: %.3f ms  %.0f GB/s read + %.0f GB/s write' % (mb, ms, mb*1.048576/ms, mb*1.048576/ms))
    ms=t(lambda: a.sum()); print('read  %5d MB: %.3f ms  %.0f GB/s read' % (mb, ms, mb*1.048576/ms))
