#!/usr/bin/env python3
"""PMC cases of ``cusrl_mlp2_forward`` (run under separate rocprofv3 --pmc passes by scripts/gpu_pmc_mlp_forward.sh):

  stream_16B       calibration — push of one 1 GiB leaf (a known 1 GiB streamed each way)
  act_4096         the acting launch of BASELINE config 2: 4096 rows, 48 -> 256 -> 128 -> 12, with the sampling epilogue
  pass_98304       the value / statistics pass over the whole buffer (98 304 rows, plain head output)
  pass_1m          1 048 576 rows: the matrix-core bound (weights persistent, 256 row tiles per workgroup)

``<out>/cases.json``: kernel, grid, rows, flops, algorithmic bytes and the MFMA instructions the launch must issue
(736 v_mfma_f32_16x16x4_f32 per 16-row tile: 192 + 512 + 32)."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from cusrl_amd import ops  # noqa: E402

DEV = "cuda:0"
REPEAT = 5


def main(out_dir):
    cases = {}
    big = torch.empty(1, 1 << 26, 4, device=DEV).normal_()
    storage = torch.empty_like(big)
    for _ in range(REPEAT):
        ops.buffer_push([(big[0], storage)], 0, 1 << 26)
    torch.cuda.synchronize()
    cases["stream_16B"] = {"kernel": "push_kernel", "algorithmic_bytes": 2 << 30, "launches": REPEAT}
    del big, storage
    torch.cuda.empty_cache()
    K, H1, H2, A = 48, 256, 128, 12
    g = torch.Generator().manual_seed(0)
    w = lambda *s: (torch.randn(*s, generator=g) * 0.1).to(DEV)  # noqa: E731
    layers = (w(H1, K), w(H1), w(H2, H1), w(H2), w(A, H2), w(A))
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    for order, (name, rows, sample) in enumerate((("act_4096", 4096, True), ("pass_98304", 98304, False), ("pass_1m", 1 << 20, False))):
        x = torch.randn(rows, K, device=DEV)
        eps, std = torch.randn(rows, A, device=DEV), torch.rand(A, device=DEV) + 0.5
        for _ in range(REPEAT):
            ops.mlp2_forward(x, layers, std=std, eps=eps) if sample else ops.mlp2_forward(x, layers)
        torch.cuda.synchronize()
        tiles = (rows + 15) // 16
        cases[name] = {"kernel": "mlp2_forward_kernel", "rows": rows, "grid_threads": min(tiles, cus) * 256,
                       "flops": 2 * rows * (K * H1 + H1 * H2 + H2 * A), "mfma_instructions": tiles * 736,
                       "algorithmic_bytes": rows * 4 * (K + A * (4 if sample else 1)) + 4 * sum(t.numel() for t in layers),
                       "launches": REPEAT, "order": order}  # (all three launch one workgroup per CU: told apart by dispatch order)
        del x, eps
    Path(out_dir).mkdir(parents=True, exist_ok=True)
    (Path(out_dir) / "cases.json").write_text(json.dumps(cases, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
