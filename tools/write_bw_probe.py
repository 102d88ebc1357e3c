import torch, time
dev=torch.device("cuda:0")
x=torch.empty((1000,149,149,64),dtype=torch.float16,device=dev)  # 2.84 GB
def t(fn,it=10):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/it
ms=t(lambda: x.fill_(1.0)); print(f"fill_ 2.84 GB: {ms:.3f} ms  {x.numel()*2/ms/1e9:.2f} TB/s")
ms=t(lambda: x.zero_()); print(f"zero_ (memset): {ms:.3f} ms  {x.numel()*2/ms/1e9:.2f} TB/s")
y=torch.empty_like(x)
ms=t(lambda: y.copy_(x)); print(f"copy 2.84 GB -> 2.84 GB: {ms:.3f} ms  {2*x.numel()*2/ms/1e9:.2f} TB/s (read+write)")
