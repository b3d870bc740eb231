"""HBM roofline of the prep / scoring reductions: `dp_masked_mean_std` and `dp_masked_errors` through the
package's wrappers, timed with events on the launch stream, against their algorithmic bytes:

    mean/std   two passes x (4 B value + 1 B mask) per cell                       = 10 B/cell
    errors     pred 4 B + truth 4 B + mask 1 B per cell                           =  9 B/cell

    python tools/prep_bench.py [--reps 20]          -> one JSON line per shape
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from diffputer_amd import prep  # noqa: E402

SHAPES = [(1_250_000, 64), (10_000_000, 64), (4_000_000, 9), (1_000_000, 51), (2_000_000, 256), (22_792, 34)]


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for n, d in SHAPES:
        g = torch.Generator(device=dev).manual_seed(n + d)
        X = torch.randn(n, d, device=dev, generator=g) * 3.0 + 1.5
        truth = X + 0.1 * torch.randn(n, d, device=dev, generator=g)
        mask = (torch.rand(n, d, device=dev, generator=g) < 0.3).to(torch.uint8)
        ms_stats = timed(lambda: prep.masked_mean_std(X, mask), args.reps)
        ms_err = timed(lambda: prep.masked_errors(X, truth, mask), args.reps)
        cells = n * d
        print(json.dumps({"rows": n, "cols": d,
                          "mean_std": {"ms": round(ms_stats, 4), "GB_per_s": round(10 * cells / ms_stats / 1e6, 1), "frac_of_8TBps": round(10 * cells / ms_stats / 1e6 / 8000, 3)},
                          "errors": {"ms": round(ms_err, 4), "GB_per_s": round(9 * cells / ms_err / 1e6, 1), "frac_of_8TBps": round(9 * cells / ms_err / 1e6 / 8000, 3),
                                     "note": "includes the 24-byte read-back and its synchronisation"}}), flush=True)
        del X, truth, mask


if __name__ == "__main__":
    main()
