#!/usr/bin/env python3
"""A/B of em.m_step with the HIP training step and with PyTorch autograd over several seeds: is the best training loss
after N epochs statistically the same?  (python tools/mstep_ab.py [--epochs 150 --seeds 4])"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffputer_amd import em, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=22792)
ap.add_argument("--cols", type=int, default=34)
ap.add_argument("--hid", type=int, default=1024)
ap.add_argument("--epochs", type=int, default=150)
ap.add_argument("--seeds", type=int, default=4)
a = ap.parse_args()
dev = "cuda:0"
X, mask = synth.make_table(a.rows, a.cols, seed=5, latent=3)
data = torch.from_numpy((X * (1.0 - mask)).astype(np.float32)).to(dev)
for native in (False, True):
    out = []
    for seed in range(a.seeds):
        torch.manual_seed(1000 + seed)
        gen = torch.Generator(device=dev).manual_seed(seed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, best = em.m_step(data, a.hid, max_epochs=a.epochs, generator=gen, native=native)
        torch.cuda.synchronize()
        out.append((best, time.perf_counter() - t0))
    b = np.asarray([o[0] for o in out])
    print(f"{'native  ' if native else 'autograd'}: best loss per seed {np.round(b, 5).tolist()} mean {b.mean():.5f} sd {b.std(ddof=1):.5f}; "
          f"{np.mean([o[1] for o in out]):.2f} s per M-step", flush=True)
