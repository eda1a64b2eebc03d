"""TEST-ONLY stand-in for ffi.Context so that the calculator's HOST logic (dedup rules, sortedness,
chromosome switching, lag re-indexing, result assembly) can be exercised on a machine without a GPU.
"Device" vectors are numpy arrays and the per-shift loop is the CPU oracle.  Never used by the product."""
import numpy as np

from oracle import model as oracle
from pymasc_amd import ffi


class _View:
    """dict-like access to the fake device memory by ANY address inside an allocation (the calculator hands out slots of
    one arena as base + offset)."""

    def __init__(self):
        self.blocks = {}

    def __setitem__(self, p, arr):
        self.blocks[p] = arr

    def pop(self, p, default=None):
        return self.blocks.pop(p, default)

    def clear(self):
        self.blocks.clear()

    def __getitem__(self, p):
        if p in self.blocks:
            return self.blocks[p]
        for base, arr in self.blocks.items():
            if base <= p < base + arr.size * 8:
                assert (p - base) % 8 == 0
                return arr[(p - base) // 8:]
        raise KeyError(p)


class FakeContext:
    def __init__(self):
        self._mem = _View()
        self._next = 0x1000

    def close(self):
        self._mem.clear()

    def sync(self):
        pass

    def bits_alloc(self, nbits):
        p = self._next
        self._next += (ffi.nwords(nbits) * 8 + 0x1fff) & ~0xfff
        self._mem[p] = np.zeros(ffi.nwords(nbits), dtype=np.uint64)
        return p

    def bits_free(self, p):
        self._mem.pop(p, None)

    def pool_alloc(self, nbits):
        p = self.bits_alloc(nbits)
        self._mem[p][:] = np.uint64(0xdeadbeefdeadbeef)     # (the real pool hands vectors out as they were left: never rely on zeros)
        return p, nbits

    def pool_free(self, p, cap):
        self.bits_free(p)

    def cc_batch_dev(self, d_F, d_R, d_M, nbits, max_shift, read_len, flags, d_out):
        for i in range(len(d_F)):
            self.cc_dev(d_F[i], d_R[i], d_M[i] if d_M else None, nbits[i], max_shift, read_len, flags, d_out[i])

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

    # ---- the stream-ordered feeders (pmx_feed_reads, pmx_bits_set_regions_async, pmx_mappable_len_batch_dev): the reference's
    # per-read rules restated read by read (mscc.pyx:351-418), state words as in include/pymasc_amd.h
    def feed_reads(self, d_F, d_R, nbits, pos, readlen, is_reverse, reads_before, d_state, whole_vectors=False):
        F, R, st = self._mem[d_F], self._mem[d_R], self._mem[d_state]
        if whole_vectors:                 # PMX_FEED_WHOLE_VECTORS: the first run writes every word of both vectors
            assert reads_before == 0
            F[:ffi.nwords(nbits)] = 0
            R[:ffi.nwords(nbits)] = 0
        pos = np.asarray(pos)
        if is_reverse is None:            # strand packed into the top bit (the width rules of ffi.Context.feed_reads)
            if pos.dtype in (np.dtype(np.uint32), np.dtype(np.uint64)):
                pos = pos.view(np.int32 if pos.dtype.itemsize == 4 else np.int64)
            elif pos.dtype not in (np.dtype(np.int32), np.dtype(np.int64)):
                raise TypeError("packed strand needs 32- or 64-bit positions")
            is_reverse = pos < 0
            pos = pos & ((1 << (8 * pos.dtype.itemsize - 1)) - 1)
        pos, rev = pos.tolist(), np.asarray(is_reverse).astype(bool).tolist()
        readlen = [int(readlen)] * len(pos) if np.ndim(readlen) == 0 else np.asarray(readlen).tolist()
        last = int(st[ffi.PMX_FEED_LAST_POS]) if reads_before else 0
        last_f = int(st[ffi.PMX_FEED_LAST_FORWARD_POS])
        fs = rs = nf = nr = 0
        for i, (p, l, r) in enumerate(zip(pos, readlen, rev)):
            if p < last and not int(st[ffi.PMX_FEED_FIRST_UNSORTED]):
                st[ffi.PMX_FEED_FIRST_UNSORTED] = ffi.PMX_FEED_ERR_BASE - (reads_before + i)
            last = p
            bit = p + l - 1 if r else p
            if bit < 0 or bit >= nbits:
                if not int(st[ffi.PMX_FEED_FIRST_OUT_OF_RANGE]):
                    st[ffi.PMX_FEED_FIRST_OUT_OF_RANGE] = ffi.PMX_FEED_ERR_BASE - (reads_before + i)
                continue
            w, m = bit >> 6, np.uint64(1) << np.uint64(bit & 63)
            if not r:
                if p == last_f:
                    continue
                last_f = p
                fs += l
                nf += 1
                F[w] |= m
            elif not int(R[w]) & int(m):
                R[w] |= m
                rs += l
                nr += 1
        st[ffi.PMX_FEED_FORWARD_LEN_SUM] += np.uint64(fs)
        st[ffi.PMX_FEED_REVERSE_LEN_SUM] += np.uint64(rs)
        st[ffi.PMX_FEED_FORWARD_KEPT] += np.uint64(nf)
        st[ffi.PMX_FEED_REVERSE_KEPT] += np.uint64(nr)
        st[ffi.PMX_FEED_LAST_POS] = max(last, 0)
        st[ffi.PMX_FEED_LAST_FORWARD_POS] = last_f
        st[ffi.PMX_FEED_READS] += np.uint64(len(pos))
        return pos, readlen, rev

    def feed_reads_delta16(self, d_F, d_R, nbits, reads, readlen, reads_before, d_state, whole_vectors=False):
        pos, rev = ffi.unpack_delta16(reads)
        return self.feed_reads(d_F, d_R, nbits, pos, readlen, rev, reads_before, d_state, whole_vectors)

    def bits_set_regions_async(self, p, nbits, first, last, first_offset=0, d_state=None, clear=False, side=False,
                               sorted_disjoint=False):
        w = self._mem[p]
        if sorted_disjoint:      # PMX_REGIONS_SORTED: the order is checked, a violation recorded (the vector is then undefined)
            a = np.asarray(first).astype(np.int64) + first_offset
            b = np.asarray(last).astype(np.int64)
            bad = (b < a)
            bad[:-1] |= a[1:] <= b[:-1]
            if bad.any():
                st = self._mem[d_state]
                st[ffi.PMX_FEED_REGIONS_UNSORTED] = max(int(st[ffi.PMX_FEED_REGIONS_UNSORTED]), ffi.PMX_FEED_ERR_BASE - int(np.flatnonzero(bad)[0]))
                w[:(nbits + 63) // 64] = np.uint64(0x5555aaaa5555aaaa)
                return first, last
            clear = True
        if clear:
            w[:(nbits + 63) // 64] = 0
        for a, b in zip(np.asarray(first).tolist(), np.asarray(last).tolist()):
            a += first_offset
            b = min(b, nbits - 1)
            if b >= a:
                oracle.lib().pmo_set_region(oracle._p(w), int(a), int(b))
        return first, last

    def mappable_len_batch_dev(self, d_M, nbits, max_shift, flags, d_out):
        for m, nb, o in zip(d_M, nbits, d_out):
            self.mappable_len_dev(m, nb, max_shift, flags, o)
