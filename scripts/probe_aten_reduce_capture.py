#!/usr/bin/env python3
"""Do the MEMSET NODES of a replayed hipGraph execute on this stack?  (They do not, reliably: DESIGN.md section 5.)

Plain torch — `cusrl_amd` is never imported; only the "replaced" variants dlopen libcusrl_hip.so for its graph-surgery entry
point.  Two kinds of cases, each replayed N times with every replay checked on the device:

* a bare ``hipMemsetAsync`` node between two kernels (``buf += 5; memset(buf, 0); buf += 1; out = buf``): ``out`` must be 1;
* a chain of column sums ``x.sum(0)`` with small allocations between them — ATen splits such a reduction over gridDim.y
  blocks with a staging buffer and a semaphore zeroed by hipMemsetAsync (ATen/native/cuda/Reduce.cuh:1294-1301, 693-703), a
  memset node once captured — with changing inputs, compared against a float64 sum.  Variants: one graph; two graphs sharing a
  pool, alternating; eager kernels between replays; back-to-back replays without a host sync;

each as captured and with every memset node replaced by a fill-kernel node (``cusrl_graph_replace_memsets``).

    python scripts/probe_aten_reduce_capture.py [replays]      # PROBE_QUICK=1: two shapes, two modes
"""
import os
import sys

import torch

DEV = "cuda:0"
REPLAYS = int(sys.argv[1]) if len(sys.argv) > 1 else 10000


def chain(x, noise, widths):
    """Mimics the bias-gradient reductions of an MLP backward at a 1024-row minibatch."""
    outs = []
    a = x * 1.0001
    outs.append(a.sum(0))
    del a
    t = torch.empty(x.shape[1], device=DEV)
    t.copy_(outs[-1])
    outs.append(t)
    b = x + noise
    outs.append(b.sum(0))
    outs.append(outs[-1] * 2.0)
    for w in widths:
        c = (x[:, :w] * noise[:, :w]).contiguous()
        outs.append(c.sum(0))
        small = torch.zeros(16, device=DEV)  # a small block right behind the reduce: candidate for the freed semaphore
        outs.append(small + outs[-1][:16])
    outs.append((x * x).sum())  # full reduction, one output
    return outs


def reference(x, noise, widths):
    xd, nd = x.double(), noise.double()
    outs = []
    a = (x * 1.0001).double()
    outs.append(a.sum(0))
    outs.append(outs[-1])
    outs.append((x + noise).double().sum(0))
    outs.append(outs[-1] * 2.0)
    for w in widths:
        outs.append((x[:, :w] * noise[:, :w]).double().sum(0))
        outs.append(outs[-1][:16])
    outs.append((x * x).double().sum())
    return outs


_hip = None


def node_census(graph):
    """Node types of a kept hipGraph through the HIP runtime torch has loaded (0 kernel, 1 memcpy, 2 memset, ...)."""
    global _hip
    import ctypes

    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so.7")
    g = ctypes.c_void_p(graph.raw_cuda_graph())
    count = ctypes.c_size_t(0)
    rc = _hip.hipGraphGetNodes(g, None, ctypes.byref(count))
    if rc != 0:
        return {"error": rc}
    nodes = (ctypes.c_void_p * count.value)()
    _hip.hipGraphGetNodes(g, nodes, ctypes.byref(count))
    kinds = {}
    for node in nodes:
        kind = ctypes.c_int(-1)
        _hip.hipGraphNodeGetType(ctypes.c_void_p(node), ctypes.byref(kind))
        name = {0: "kernel", 1: "memcpy", 2: "memset"}.get(kind.value, f"type{kind.value}")
        kinds[name] = kinds.get(name, 0) + 1
    return kinds


_cusrl = None


def replace_memsets(graph) -> int:
    """The repo's remedy (cusrl_graph_replace_memsets in libcusrl_hip.so): memset nodes -> fill-kernel nodes, same edges."""
    global _cusrl
    import ctypes
    from pathlib import Path

    if _cusrl is None:
        _cusrl = ctypes.CDLL(str(Path(__file__).resolve().parent.parent / "cusrl_amd" / "libcusrl_hip.so"))
    replaced = ctypes.c_int64(0)
    rc = _cusrl.cusrl_graph_replace_memsets(ctypes.c_void_p(graph.raw_cuda_graph()), ctypes.byref(replaced))
    assert rc == 0, rc
    return replaced.value


def capture(fn, stream, pool=None, surgery=False):
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g, stream=stream, pool=pool):
        outs = fn()
    kinds = node_census(g)
    if surgery:
        kinds = {"before": kinds, "replaced": replace_memsets(g), "after": node_census(g)}
    g.instantiate()
    return g, outs, kinds


def run_memset_node(replays, nbytes, surgery):
    """The memset node alone, without ATen's reduction: kernel (dirty the buffer) -> hipMemsetAsync(0) -> kernel (+1) -> copy.
    Every replay must leave 1 in every word."""
    import ctypes

    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so.7")
    words = nbytes // 4
    buf = torch.zeros(words, dtype=torch.int32, device=DEV)
    out = torch.zeros(words, dtype=torch.int32, device=DEV)
    stream = torch.cuda.Stream()

    def body():
        buf.add_(5)
        rc = _hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), ctypes.c_int(0), ctypes.c_size_t(nbytes),
                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
        buf.add_(1)
        out.copy_(buf)

    with torch.cuda.stream(stream):
        body()
    torch.cuda.synchronize()
    g, _, kinds = capture(body, stream, None, surgery)
    wrong = torch.zeros((), dtype=torch.int64, device=DEV)
    first = torch.full((), replays, dtype=torch.int64, device=DEV)
    for r in range(replays):
        g.replay()
        bad = (out != 1).any()
        wrong += bad
        first = torch.where(bad & (first == replays), torch.full_like(first, r), first)
    torch.cuda.synchronize()
    print(f"memset node alone, {nbytes} bytes, surgery {surgery}: nodes {kinds}; wrong replays {int(wrong)} of {replays}, first {int(first)}; "
          f"last out head {out[:4].tolist()}", flush=True)


def errors(outs, refs, scale):
    """Per output: max |out - ref| in units of max(scale, |ref|) (the full sum of squares is ~1e5..1e7: fp32 rounding of it
    must not count as wrong)."""
    return torch.nan_to_num(torch.stack([((o.double() - ref).abs().max() / ref.abs().max().clamp_min(scale))
                                         for o, ref in zip(outs, refs)]), nan=1e30)


def run(rows, cols, widths, mode, surgery=False):
    torch.manual_seed(rows * 131 + cols)
    x = torch.randn(rows, cols, device=DEV)
    noise = torch.randn(rows, cols, device=DEV)
    fresh = torch.randn(64, rows, cols, device=DEV)
    stream = torch.cuda.Stream()
    pool = torch.cuda.graph_pool_handle()
    with torch.cuda.stream(stream):
        for _ in range(3):
            chain(x, noise, widths)
    torch.cuda.synchronize()
    g1, outs1, kinds = capture(lambda: chain(x, noise, widths), stream, pool, surgery)
    g2 = outs2 = None
    if mode == "two_graphs":
        g2, outs2, _ = capture(lambda: chain(noise, x, widths), stream, pool, surgery)
    n_out = len(outs1)
    scale = float(rows) ** 0.5 * 4
    bad = torch.zeros(n_out, dtype=torch.int64, device=DEV)     # replays in which output k was wrong
    stale = torch.zeros(n_out, dtype=torch.int64, device=DEV)   # ... and equal to the PREVIOUS replay's right answer
    worst = torch.zeros(n_out, dtype=torch.float64, device=DEV)
    first_bad = torch.full((n_out,), REPLAYS, dtype=torch.int64, device=DEV)
    prev = None
    for r in range(REPLAYS):
        x.copy_(fresh[r % 64])
        if r % 7 == 0:
            x.mul_(1.0 + 1e-3 * (r % 13))
        if mode == "eager_between":
            junk = [torch.empty(n, device=DEV).fill_(float(r)) for n in (16, 64, 128, 256)]
            (fresh[(r + 1) % 64] * 2).sum(0)
            del junk
        g1.replay()
        refs = reference(x, noise, widths)
        err = errors(outs1, refs, scale)
        wrong = err > 1e-4
        bad += wrong
        worst = torch.maximum(worst, err)
        first_bad = torch.where(wrong & (first_bad == REPLAYS), torch.full_like(first_bad, r), first_bad)
        if prev is not None:
            stale += wrong & (errors(outs1, prev, scale) < 1e-4)
        prev = refs
        if g2 is not None:
            g2.replay()
            err2 = errors(outs2, reference(noise, x, widths), scale)
            bad += err2 > 1e-4
            worst = torch.maximum(worst, err2)
        if mode != "no_sync" and r % 64 == 63:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    tag = " + memset nodes replaced by fill kernels" if surgery else ""
    print(f"rows {rows} cols {cols} mode {mode}{tag}: nodes {kinds}; wrong replays per output {bad.tolist()} (stale {stale.tolist()}) "
          f"first {first_bad.tolist()} worst {[f'{v:.1e}' for v in worst.tolist()]}", flush=True)


if __name__ == "__main__":
    print(torch.__version__, torch.version.hip, torch.cuda.get_device_name(0), "replays", REPLAYS)
    quick = os.environ.get("PROBE_QUICK") == "1"
    for nbytes in (4, 512, 65536):
        run_memset_node(REPLAYS, nbytes, False)
        run_memset_node(REPLAYS, nbytes, True)
    shapes = ((1024, 128), (24576, 256)) if quick else ((1024, 128), (1024, 256), (4096, 128), (1024, 64), (32, 64), (24576, 256))
    for rows, cols in shapes:
        for mode in ("one_graph", "two_graphs") if quick else ("one_graph", "two_graphs", "eager_between", "no_sync"):
            run(rows, cols, (64, 32, 16), mode)
            run(rows, cols, (64, 32, 16), mode, surgery=True)
