#!/usr/bin/env python3
"""GPU box: run-length sweep of the mappability track on the benchmark genome (read density 0.005): ms per pass of
pmx_cc_batch_dev, event path vs window kernels alone.  usage: python tools/gpu_edge_sweep.py  (PMX_CC_EVENTS=0 for the latter)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pymasc_amd import ffi, synth
ctx = ffi.Context(0)
dev = torch.device("cuda", 0)
S, L = 1000, 36
for on, off in ((2000, 500), (400, 100), (300, 75), (280, 70), (265, 66), (250, 62), (240, 60), (160, 40)):
    vecs = [synth.make_chromosome(ctx, dev, n, ln, S, L, 0xC0FFEE + i, mean_on=on, mean_off=off) for i, (n, ln) in enumerate(synth.HG38)]
    out = torch.zeros((len(vecs), ffi.PMX_NROWS, S + 1), dtype=torch.int64, device=dev)
    args = ([v.F.data_ptr() for v in vecs], [v.R.data_ptr() for v in vecs], [v.M.data_ptr() for v in vecs],
            [v.nbits for v in vecs], S, L, 0, [out[i].data_ptr() for i in range(len(vecs))])
    for _ in range(3):
        ctx.cc_batch_dev(*args)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(8):
        ctx.cc_batch_dev(*args)
    ctx.sync()
    ms = (time.perf_counter() - t0) / 8 * 1e3
    print(f"runs on/off {on}/{off}: ~{2 * 65536 / (on + off):.0f} edges per 64 Kbit: {ms:.3f} ms", flush=True)
    del vecs, out
    torch.cuda.empty_cache()
