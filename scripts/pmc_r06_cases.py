#!/usr/bin/env python3
"""PMC cases of round 6's new kernel, run under ``rocprofv3 --pmc FETCH_SIZE`` / ``--pmc WRITE_SIZE`` / SQ counters in separate
passes (scripts/gpu_pmc_r06.sh, as MI355X_MICROARCH.md's HBM section prescribes):

  stream_16B                 calibration — push of one 1 GiB leaf (a known 1 GiB streamed each way): the factor between the read
                             counter's 64-byte tallies and the bytes of streaming 16-byte-per-lane loads
  input_layer_bwd_config2    cusrl_input_layer_bwd at BASELINE config 2's in-step size: 24 576 rows, [48] -> [256] (55 MB:
                             L2 / Infinity-Cache resident)
  input_layer_bwd_at_scale   the same launch over 3 145 728 rows (7.0 GB: beyond the caches — what reaches HBM per byte asked for)

``<out>/cases.json`` names each case's kernel, grid and algorithmic bytes; ``scripts/pmc_r06_summarize.py`` matches rows to them."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from cusrl_amd import _native, ops  # noqa: E402

DEV = "cuda:0"
REPEAT = 5


def main(out_dir):
    cases = {}
    big = torch.empty(1, 1 << 26, 4, device=DEV).normal_()
    storage = torch.empty_like(big)
    for _ in range(REPEAT):
        ops.buffer_push([(big[0], storage)], 0, 1 << 26)
    torch.cuda.synchronize()
    cases["stream_16B"] = {"kernel": "push_kernel", "algorithmic_bytes": 2 << 30, "launches": REPEAT, "index": 0}
    del big, storage
    torch.cuda.empty_cache()
    lib = _native.lib()
    for index, (name, rows) in enumerate((("input_layer_bwd_config2", 24576), ("input_layer_bwd_at_scale", 3 * (1 << 20)))):
        K, H = 48, 256
        g = torch.randn(rows, H, device=DEV)
        y = torch.relu(torch.randn(rows, H, device=DEV))
        x = torch.randn(rows, K, device=DEV)
        for _ in range(REPEAT):
            ops.input_layer_backward(g, y, x)
        torch.cuda.synchronize()
        blocks = int(lib.cusrl_input_layer_row_blocks(rows, H))
        cases[name] = {"kernel": "input_layer_bwd_kernel", "rows": rows, "grid_threads": blocks * (H // 64) * 512,
                       "algorithmic_bytes": rows * 4 * (2 * H + K), "partial_bytes": blocks * (H * K + H) * 4,
                       "launches": REPEAT, "index": index}
        del g, y, x
        torch.cuda.empty_cache()
    Path(out_dir).mkdir(parents=True, exist_ok=True)
    (Path(out_dir) / "cases.json").write_text(json.dumps(cases, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
