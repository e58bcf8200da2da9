import torch, time
d=torch.device("cuda:0")
for mb in (64, 268, 536):
    x=torch.empty(mb*1024*1024//4, device=d)
    y=torch.empty_like(x)
    for name,fn in (("fill",lambda: x.fill_(1.0)),("copy",lambda: y.copy_(x))):
        fn(); torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ms=e0.elapsed_time(e1)/20
        print(f"{name} {mb} MB: {ms*1e3:.1f} us  {mb*1.048576/ms:.0f} GB/s (x2 for copy)")
