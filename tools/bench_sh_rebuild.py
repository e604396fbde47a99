#!/usr/bin/env python3
"""Cost of the data-parallel feature update on ONE GPU as a function of the number of ranks R: fsgs_sh_coeff_grad_adam
rebuilds the mean SH-coefficient gradient of R gathered views (factors: masked colour gradient + camera centre per rank,
16 B per Gaussian and rank) and applies Adam to features_dc / features_rest in the same launch.  The gathered buffer
is synthetic (what all_gather_into_tensor would deliver); prints one JSON line per R.  DESIGN.md §6."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fusionsense_amd._lib import load, ptr, stream_ptr


def main():
    dev = torch.device("cuda:0")
    lib = load()
    N, K = int(os.environ.get("N", 300_000)), 16
    g = torch.Generator(device=dev).manual_seed(0)
    means = torch.randn(N, 3, device=dev, generator=g)
    dc, rest = torch.randn(N, 3, device=dev, generator=g), torch.randn(N, K - 1, 3, device=dev, generator=g)
    m = [torch.zeros_like(dc), torch.zeros_like(rest)]
    v = [torch.zeros_like(dc), torch.zeros_like(rest)]
    for R in (1, 2, 4, 8):
        gathered = torch.randn(R, N + 1, 4, device=dev, generator=g)
        gathered[:, :N][torch.rand(R, N, device=dev, generator=g) < 0.3] = 0.0  # invisible in that view
        args = (R, N, K, 3, ptr(means), ptr(gathered), 1.0 / R, ptr(dc), ptr(m[0]), ptr(v[0]), 0.0025, ptr(rest), ptr(m[1]),
                ptr(v[1]), 0.000125, 1, 0.9, 0.999, 1e-15, stream_ptr(dev))
        for _ in range(5):
            assert lib.fsgs_sh_coeff_grad_adam(*args) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            lib.fsgs_sh_coeff_grad_adam(*args)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        bytes_moved = N * (48 * 4 * 6 + 12) + R * (N + 1) * 16  # params + 2 moments read and written, means, factors
        print(json.dumps({"R": R, "N": N, "us_per_launch": round(us, 1), "gathered_MB": round(R * (N + 1) * 16 / 1e6, 1),
                          "algorithmic_GBps": round(bytes_moved / us / 1e3, 1)}))


if __name__ == "__main__":
    main()
