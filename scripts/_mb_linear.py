import torch, time
dev='cuda'
def timeit(fn, iters=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(20): fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters//20): g.replay()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en)*1e3/iters
B=24576
for (i,o) in [(48,256),(256,128),(128,12)]:
    x=torch.randn(B,i,device=dev); w=torch.randn(o,i,device=dev); b=torch.randn(o,device=dev); gy=torch.randn(B,o,device=dev)
    print(f'--- layer {i}->{o}')
    print('fwd addmm        ', timeit(lambda: torch.nn.functional.linear(x,w,b)))
    print('fwd addmm+relu   ', timeit(lambda: torch.relu(torch.nn.functional.linear(x,w,b))))
    try:
        print('fwd _addmm_act   ', timeit(lambda: torch._addmm_activation(b, x, w.t())))
        y1=torch._addmm_activation(b, x, w.t()); y2=torch.relu(torch.nn.functional.linear(x,w,b)); print('   max diff', (y1-y2).abs().max().item())
    except Exception as e: print('addmm_activation failed', e)
    print('wgrad mm         ', timeit(lambda: gy.t() @ x))
    for S in (4,8,16,32):
        print(f'wgrad bmm S={S:2d}   ', timeit(lambda: torch.bmm(gy.view(S,B//S,o).transpose(1,2), x.view(S,B//S,i))), ' +sum', timeit(lambda: torch.bmm(gy.view(S,B//S,o).transpose(1,2), x.view(S,B//S,i)).sum(0)))
    print('bias sum         ', timeit(lambda: gy.sum(0)))
    ones=torch.ones(B,device=dev)
    print('bias mv          ', timeit(lambda: torch.mv(gy.t(), ones)))
    print('dx               ', timeit(lambda: gy @ w))
    y=torch.relu(torch.nn.functional.linear(x,w,b))
    print('relu bwd         ', timeit(lambda: torch.ops.aten.threshold_backward(gy, y, 0)))
