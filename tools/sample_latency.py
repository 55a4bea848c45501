#!/usr/bin/env python3
"""Small-batch sampling latency of the full CIFAR model, eager launches vs HIP-graph replay (GPU box only)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
dpm = bench.build_model(dev).eval()
for B in (1, 4, 16, 128):
    for mode in ("0", "1"):
        os.environ["ADM_SAMPLE_GRAPH"] = mode
        dpm.sample(batch_size=B)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            dpm.sample(batch_size=B)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"B={B:4d} graph={mode}: {dt * 1e3:8.1f} ms per 10-step sample  ({B / dt:7.1f} images/s)", flush=True)
