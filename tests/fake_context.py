"""TEST-ONLY stand-in for ffi.Context so that the calculator's HOST logic (dedup rules, sortedness,
chromosome switching, lag re-indexing, result assembly) can be exercised on a machine without a GPU.
"Device" vectors are numpy arrays and the per-shift loop is the CPU oracle.  Never used by the product."""
import numpy as np

from oracle import model as oracle
from pymasc_amd import ffi


class FakeContext:
    def __init__(self):
        self._mem = {}
        self._next = 0x1000

    def close(self):
        self._mem.clear()

    def sync(self):
        pass

    def bits_alloc(self, nbits):
        p = self._next
        self._next += 0x1000
        self._mem[p] = np.zeros(ffi.nwords(nbits), dtype=np.uint64)
        return p

    def bits_free(self, p):
        self._mem.pop(p, None)

    def bits_clear(self, p, nbits):
        self._mem[p][:ffi.nwords(nbits)] = 0

    def bits_upload(self, p, words, nbits):
        self._mem[p][:ffi.nwords(nbits)] = words[:ffi.nwords(nbits)]

    def bits_download(self, p, nbits):
        return self._mem[p][:ffi.nwords(nbits)].copy()

    def bits_set_positions(self, p, nbits, pos):
        pos = np.asarray(pos, dtype=np.int64)
        if pos.size and (pos.min() < 0 or pos.max() >= nbits):
            raise ffi.PmxError(-1, "position out of range")
        if pos.size:
            np.bitwise_or.at(self._mem[p], pos >> 6, np.uint64(1) << (pos & 63).astype(np.uint64))

    def bits_set_regions(self, p, nbits, first, last):
        w = self._mem[p]
        for a, b in zip(np.asarray(first).tolist(), np.asarray(last).tolist()):
            if b >= a:
                oracle.lib().pmo_set_region(oracle._p(w), int(a), int(b))

    def bits_count(self, p, nbits):
        w = self._mem[p][:ffi.nwords(nbits)]
        return int(oracle.lib().pmo_count(oracle._p(np.ascontiguousarray(w)), w.size))

    def cc_dev(self, d_F, d_R, d_M, nbits, max_shift, read_len, flags, d_out):
        nw = ffi.nwords(nbits)
        F = np.ascontiguousarray(self._mem[d_F][:nw])
        R = np.ascontiguousarray(self._mem[d_R][:nw])
        M = np.ascontiguousarray(self._mem[d_M][:nw]) if d_M else None
        skip = bool(flags & ffi.PMX_FLAG_SKIP_NCC)
        ref = oracle.calc_correlation(F, R, M, nbits, max_shift, read_len, skip_ncc=skip)
        out = np.zeros((ffi.PMX_NROWS, max_shift + 1), dtype=np.uint64)
        if not skip:
            out[ffi.PMX_ROW_NCC_CCBINS] = ref["ncc_ccbins"]
            out[ffi.PMX_ROW_SCALARS, 0] = ref["ncc_forward_sum"]
            out[ffi.PMX_ROW_SCALARS, 1] = ref["ncc_reverse_sum"]
        if M is not None:
            out[ffi.PMX_ROW_MSCC_FSUM] = ref["mscc_forward_sum"]
            out[ffi.PMX_ROW_MSCC_RSUM] = ref["mscc_reverse_sum"]
            out[ffi.PMX_ROW_MSCC_CCBINS] = ref["mscc_ccbins"]
            if not flags & ffi.PMX_FLAG_SKIP_MLEN:
                out[ffi.PMX_ROW_MLEN] = ref["mappable_len_by_shift"]
        self._mem[d_out][:out.size] = out.reshape(-1)

    def mappable_len_dev(self, d_M, nbits, max_shift, flags, d_out):
        M = np.ascontiguousarray(self._mem[d_M][:ffi.nwords(nbits)])
        self._mem[d_out][:max_shift + 1] = oracle.mappable_len_readless(M, nbits, max_shift).astype(np.uint64)
