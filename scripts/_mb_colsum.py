import sys, torch
sys.path.insert(0,'/root/repo')
from cusrl_amd import ops
dev='cuda'
def timeit(fn, iters=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s): fn()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(20): fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters//20): g.replay()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en)*1e3/iters
for rows,H in [(24576,256),(24576,128),(24576,12),(24576,1)]:
    g=torch.randn(rows,H,device=dev); y=torch.relu(torch.randn(rows,H,device=dev))
    print(rows,H,'torch relu_bwd+sum', round(timeit(lambda: torch.ops.aten.threshold_backward(g,y,0).sum(0)),2), 'hip fused', round(timeit(lambda: ops.relu_backward_bias(g,y)),2),
          '| torch sum', round(timeit(lambda: g.sum(0)),2), 'hip colsum', round(timeit(lambda: ops.relu_backward_bias(g,None)),2))
