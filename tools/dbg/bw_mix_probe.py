import torch, time
dev="cuda:0"
n=1<<30
a=torch.empty(n,dtype=torch.float32,device=dev); b=torch.empty_like(a); c=torch.empty_like(a); d=torch.empty_like(a)
def t(fn,bytes_,reps=10):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/reps
    return bytes_/ms/1e6
print("fill (write only)      GB/s", round(t(lambda: a.fill_(1.0), 4*n)))
print("copy (1r:1w)           GB/s", round(t(lambda: b.copy_(a), 8*n)))
print("sum (read only)        GB/s", round(t(lambda: a.sum(), 4*n)))
# 1 read : 2 writes  -> torch.frexp-like? use two outputs: a -> (b = a*2, c = a+1) fused? not fused in eager: do a kernel with out variants
x=torch.empty(2,n,dtype=torch.float32,device=dev)
print("1r:2w (stack of 2 ops)  GB/s", round(t(lambda: torch.stack((a,a),out=x), 12*n)))
print("2r:1w (add)            GB/s", round(t(lambda: torch.add(a,b,out=c), 12*n)))
