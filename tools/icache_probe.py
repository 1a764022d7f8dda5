import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import torch
from diffusiontexturepainting_amd import ops
from bench_ops import timeit
m,n,k=768,1280,1280
a=torch.randn(m,k,device='cuda',dtype=torch.float16); wp=ops.pack_linear(torch.randn(n,k,device='cuda')*k**-0.5)
g=torch.ones(1280,device='cuda'); b=torch.zeros(1280,device='cuda'); xg=torch.randn(3,256,1280,device='cuda',dtype=torch.float16)
q=torch.randn(3,256,3840,device='cuda',dtype=torch.float16)
def same(): 
    for _ in range(8): ops.gemm(a,wp,n,k,tile=10,splits=1)
def alt():
    for t in (10,6,2,9,5,1,8,4): ops.gemm(a,wp,n,k,tile=t,splits=1)
def single(t):
    return lambda: [ops.gemm(a,wp,n,k,tile=t,splits=1) for _ in range(8)]
def mixed():
    for t in (10,6,2,9):
        ops.gemm(a,wp,n,k,tile=t,splits=1); ops.groupnorm(xg,g,b,silu=True)
def mixed_same():
    for t in (10,10,10,10):
        ops.gemm(a,wp,n,k,tile=t,splits=1); ops.groupnorm(xg,g,b,silu=True)
# use a graph to avoid host launch overhead
def graphed(fn):
    s=torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        gph=torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph, stream=s): fn()
    return lambda: gph.replay()
for name,fn in (("same tile x8",same),("8 different tiles",alt),("gemm+gn, 4 tiles",mixed),("gemm+gn, same tile",mixed_same)):
    t=timeit(graphed(fn),iters=50)
    print(f"{name:24s} {t*1e6:8.1f} us per 8 launches", flush=True)
for t_ in (10,6,2,9,5,1,8,4):
    t=timeit(graphed(single(t_)),iters=50); print("tile",t_, f"{t*1e6/8:6.2f} us per launch")
