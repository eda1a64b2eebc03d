"""GPU: a fixed-seed slice of the randomized parity sweep (tools/fuzz_parity.py; the full sweep ran 63,966 cases
without a mismatch on an MI355X box in round 1): odd geometry -- tile and shift-chunk edges, max_shift vs read_len
vs chromosome length, empty and saturated vectors -- through all three kernel-path settings against the oracle."""
import numpy as np
import pytest

from oracle import model as oracle
from pymasc_amd import ffi
from tools import fuzz_parity as fz
from . import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_randomized_geometry(seed):
    rng = np.random.default_rng(seed)
    done = 0
    with ffi.Context(0) as ctx:
        while done < 60:
            S, L, clen, fd, rd, with_m, mean_on, mean_off = fz.draw_case(rng)
            if (S + 1) * (clen + S + L + 100) > 2e8:
                continue
            case_seed = int(rng.integers(0, 2**31))
            full = fz.draw_full_range(rng)
            nbits, F, R, M = synth.make_case(case_seed, clen, S, L, fd, rd, with_m, mean_on=mean_on, mean_off=mean_off,
                                             full_range=full)
            ref = oracle.calc_correlation(F, R, M, nbits, S, L)
            for flags in (0, ffi.PMX_FLAG_FORCE_DENSE, ffi.PMX_FLAG_FORCE_SPARSE):
                if flags == ffi.PMX_FLAG_FORCE_SPARSE and L > 1024:
                    continue                       # reads longer than 1024 take the dense kernels (include/pymasc_amd.h)
                out = ctx.calc_correlation(F, R, M, nbits, S, L, flags)
                tag = (f"seed={case_seed} S={S} L={L} clen={clen} fd={fd} rd={rd} m={with_m}/{mean_on}/{mean_off} "
                       f"full={full} f={flags}")
                assert fz.compare(out, ref, S, with_m, False, tag), tag
            done += 1


def test_randomized_geometry_with_every_buffer_poisoned():
    """One more seed of the same sweep with pmx_debug_poison before EVERY call: scratch buffers, slabs, flag arrays and the
    LDS of every CU hold a pattern, so a kernel that reads anything it has not written differs from the oracle here."""
    rng = np.random.default_rng(15)
    prng = np.random.default_rng(1515)
    done = 0
    with ffi.Context(0) as ctx:
        while done < 50:
            S, L, clen, fd, rd, with_m, mean_on, mean_off = fz.draw_case(rng)
            if (S + 1) * (clen + S + L + 100) > 2e8:
                continue
            case_seed = int(rng.integers(0, 2**31))
            full = fz.draw_full_range(rng)
            nbits, F, R, M = synth.make_case(case_seed, clen, S, L, fd, rd, with_m, mean_on=mean_on, mean_off=mean_off,
                                             full_range=full)
            ref = oracle.calc_correlation(F, R, M, nbits, S, L)
            for flags in (0, ffi.PMX_FLAG_FORCE_SPARSE):
                if flags == ffi.PMX_FLAG_FORCE_SPARSE and L > 1024:
                    continue
                ctx.debug_poison(int(prng.choice([0, 0xffffffff, 0x80808080, 0x7fffffff, 0x00010001, int(prng.integers(0, 2**32))])))
                out = ctx.calc_correlation(F, R, M, nbits, S, L, flags)
                tag = (f"seed={case_seed} S={S} L={L} clen={clen} fd={fd} rd={rd} m={with_m}/{mean_on}/{mean_off} "
                       f"full={full} f={flags} poisoned")
                assert fz.compare(out, ref, S, with_m, False, tag), tag
            done += 1
