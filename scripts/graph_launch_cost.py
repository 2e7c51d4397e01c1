"""Host cost of replaying a hipGraph as a function of its node count (diagnostic: is a many-node captured step bound by
the host-side launch of the graph or by the device?)."""
import os
import sys
import time

import torch

dev = "cuda:0"
x = torch.zeros(1024, device=dev)
print("DEBUG_CLR_GRAPH_PACKET_CAPTURE =", os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE"), flush=True)
for nodes in (20, 100, 400):
    stream, graph = torch.cuda.Stream(), torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        x.add_(1)
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=stream):
        for _ in range(nodes):
            x.add_(1)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        graph.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"nodes={nodes:4d}: host return {1e6 * (t1 - t0) / reps:8.1f} us/replay ({1e6 * (t1 - t0) / reps / nodes:5.2f} us/node), "
          f"end-to-end {1e6 * (t2 - t0) / reps:8.1f} us/replay ({1e6 * (t2 - t0) / reps / nodes:5.2f} us/node)", flush=True)
