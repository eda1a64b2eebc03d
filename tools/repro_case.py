#!/usr/bin/env python3
"""GPU box diagnostic: one fuzz case repeated; prints the rows that differ from the oracle per repetition."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import model as oracle
from pymasc_amd import ffi
from tests import synth
seed, S, L, clen, fd, rd, on, off, full = 1846347302, 1023, 151, 65536, 1.0, 0.0005, 30.0, 5.0, False
nbits, F, R, M = synth.make_case(seed, clen, S, L, fd, rd, True, mean_on=on, mean_off=off, full_range=full)
ref = oracle.calc_correlation(F, R, M, nbits, S, L)
rpos = np.flatnonzero(np.unpackbits(R.view(np.uint8), bitorder="little")[:nbits])
print("nbits", nbits, "R reads", rpos.size, "in tile 1:", rpos[rpos >= 65536])
import ctypes
with ffi.Context(0) as ctx:
    Lb = ffi.load_library()
    Lb.pmx_debug_poison.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32]
    def run(tag):
        out = ctx.calc_correlation(F, R, M, nbits, S, L, ffi.PMX_FLAG_FORCE_SPARSE)
        a = np.asarray(out[ffi.PMX_ROW_MSCC_RSUM]).astype(np.int64); b = np.asarray(ref["mscc_reverse_sum"]).astype(np.int64)
        dd = a - b
        e = np.diff(np.concatenate([[0], dd]))
        print(tag, "bad", int((dd != 0).sum()), "GR errors at", np.flatnonzero(e)[:10].tolist(), e[np.flatnonzero(e)[:10]].tolist())
    run("fresh")
    for pat in (0x0, 0xffffffff, 0x80808080, 0x5a5a5a5a, 0x00010001, 0xfffefffe, 0x7fffffff, 0x0000ffff, 0xffff0000):
        assert Lb.pmx_debug_poison(ctx._h, pat, 256) == 0
        run("lds pat %08x" % pat)
