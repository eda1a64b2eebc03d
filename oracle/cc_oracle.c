/*
 * cc_oracle.c -- TEST INFRASTRUCTURE ONLY (oracle + timed CPU baseline).
 *
 * A plain-C restatement of the reference's BitArray cross-correlation path:
 *   PyMaSC/core/bitarray/mscc.pyx:217-325   (_calc_correlation: the per-shift loop)
 *   PyMaSC/core/bitarray/mscc.pyx:181-215   (_fill_result: read-less mappable_len)
 *   PyMaSC/core/bitarray/bitarray.pyx:88-174 (set / count / acount / rshift / lshift / alloc_and)
 * and of the noporpoise/BitArray primitives those call (bit_array.h as declared in
 * PyMaSC/core/bitarray/bitarray.pxd:20-36; the library itself is an un-vendored,
 * unpinned git submodule that is NOT present under /root/reference, so its published
 * semantics are restated here: LSB-first bits in uint64 words, shift_right moves bit
 * i -> i-n toward index 0, shift_left moves i -> i+n without growing the array).
 *
 * Like the reference it makes one full pass over the vector per statement per shift
 * (4 passes/shift NCC, ~20 passes/shift NCC+MSCC) -- that is what the timed CPU
 * baseline is meant to represent.  Nothing in the product package may call this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Parity pin: checked in tests/test_oracle_golden.py against the reference's own
 * goldens (tests/golden/ENCFF000RMB-test_{cc,mscc,nreads}.tab, hg19_36mer-test_mappability.json)
 * and against vectors produced by the reference's compiled `successive` NCC module.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint64_t *w;
    uint64_t nbits;
    uint64_t nwords;
} obits;

static uint64_t words_for(uint64_t nbits) { return (nbits + 63) / 64; }

static int ob_init(obits *a, uint64_t nbits)
{
    a->nbits = nbits;
    a->nwords = words_for(nbits);
    a->w = (uint64_t *)calloc(a->nwords ? a->nwords : 1, sizeof(uint64_t));
    return a->w ? 0 : -1;
}

static void ob_free(obits *a) { free(a->w); a->w = NULL; }

static void ob_mask_top(obits *a)
{
    uint64_t r = a->nbits & 63;
    if (r && a->nwords) a->w[a->nwords - 1] &= (~(uint64_t)0) >> (64 - r);
}

static int ob_from_words(obits *a, const uint64_t *src, uint64_t nbits)
{
    if (ob_init(a, nbits)) return -1;
    memcpy(a->w, src, a->nwords * sizeof(uint64_t));
    ob_mask_top(a);
    return 0;
}

/* bit_array_num_bits_set (bitarray.pyx:101-107) */
static uint64_t ob_count(const obits *a)
{
    uint64_t n = 0;
    for (uint64_t i = 0; i < a->nwords; i++) n += (uint64_t)__builtin_popcountll(a->w[i]);
    return n;
}

/* bitarray.acount (bitarray.pyx:109-133): AND-popcount over min words, skipping zero words */
static uint64_t ob_acount(const obits *a, const obits *b)
{
    uint64_t n = 0;
    uint64_t m = a->nwords < b->nwords ? a->nwords : b->nwords;
    for (uint64_t i = 0; i < m; i++) {
        uint64_t x = a->w[i], y = b->w[i];
        if (x > 0 && y > 0) n += (uint64_t)__builtin_popcountll(x & y);
    }
    return n;
}

/* bit_array_and via bitarray.alloc_and (bitarray.pyx:164-174): dst sized to the larger source */
static void ob_and(obits *dst, const obits *a, const obits *b)
{
    uint64_t m = a->nwords < b->nwords ? a->nwords : b->nwords;
    for (uint64_t i = 0; i < m; i++) dst->w[i] = a->w[i] & b->w[i];
    for (uint64_t i = m; i < dst->nwords; i++) dst->w[i] = 0;
}

/* bit_array_shift_right(arr, n, fill=0) (bitarray.pyx:146-153): bit i -> i-n, zeros enter at the top */
static void ob_shift_right(obits *a, uint64_t n)
{
    if (n == 0) return;
    if (n >= a->nbits) { memset(a->w, 0, a->nwords * sizeof(uint64_t)); return; }
    uint64_t ws = n / 64, bs = n % 64;
    for (uint64_t i = 0; i < a->nwords; i++) {
        uint64_t lo = (i + ws < a->nwords) ? a->w[i + ws] : 0;
        uint64_t hi = (i + ws + 1 < a->nwords) ? a->w[i + ws + 1] : 0;
        a->w[i] = bs ? ((lo >> bs) | (hi << (64 - bs))) : lo;
    }
}

/* bit_array_shift_left(arr, 1, fill) (bitarray.pyx:155-162): bit i -> i+1, `fill` enters at index 0,
 * the array keeps its length (the top bit falls off). */
static void ob_shift_left1(obits *a, int fill)
{
    uint64_t carry = fill ? 1 : 0;
    for (uint64_t i = 0; i < a->nwords; i++) {
        uint64_t x = a->w[i];
        a->w[i] = (x << 1) | carry;
        carry = x >> 63;
    }
    ob_mask_top(a);
}

static int ob_get(const obits *a, uint64_t i) { return (int)((a->w[i >> 6] >> (i & 63)) & 1); }

/*
 * The per-chromosome loop, statement for statement (mscc.pyx:217-325).
 *   F, R, M : nwords = ceil(nbits/64) uint64 words each; M may be NULL (no mappability).
 *   nbits   : chrom_len + read_len + max_shift + 100 (mscc.pyx:134,165-167,340-341)
 *   outputs : ncc_ccbins[S+1]; mscc_fsum/rsum/ccbins[S+1]; mlen_by_d[S+1] = popcount(D_d) for EVERY d
 *             (the reference only stores some d, by lag: mscc.pyx:292-298; the Python side re-indexes).
 * R is copied first (the reference destroys it in place, mscc.pyx:316).
 */
int pmo_calc_correlation(const uint64_t *F, const uint64_t *R, const uint64_t *M,
                         uint64_t nbits, int64_t S, int64_t L, int skip_ncc,
                         uint64_t *ncc_fsum, uint64_t *ncc_rsum, uint64_t *ncc_ccbins,
                         uint64_t *mscc_fsum, uint64_t *mscc_rsum, uint64_t *mscc_ccbins,
                         uint64_t *mlen_by_d)
{
    obits f, r, m, rm, dmr, mf, mr;
    int rc = -1;
    memset(&m, 0, sizeof m); memset(&rm, 0, sizeof rm);
    memset(&dmr, 0, sizeof dmr); memset(&mf, 0, sizeof mf); memset(&mr, 0, sizeof mr);
    if (ob_from_words(&f, F, nbits)) return -1;
    if (ob_from_words(&r, R, nbits)) { ob_free(&f); return -1; }

    if (!skip_ncc) {                       /* mscc.pyx:235-239 */
        *ncc_fsum = ob_count(&f);
        *ncc_rsum = ob_count(&r);
    }

    int *buff = NULL; int64_t nbuff = 0;
    if (M) {                               /* mscc.pyx:279-282 */
        if (ob_from_words(&m, M, nbits)) goto done;
        if (ob_from_words(&rm, M, nbits)) goto done;
        if (ob_init(&dmr, nbits) || ob_init(&mf, nbits) || ob_init(&mr, nbits)) goto done;
        nbuff = L - 1 > 0 ? L - 1 : 0;
        buff = (int *)malloc(sizeof(int) * (size_t)(nbuff ? nbuff : 1));
        if (!buff) goto done;
        for (int64_t i = 0; i < nbuff; i++) buff[i] = ((uint64_t)i < nbits) ? ob_get(&rm, (uint64_t)i) : 0;
        ob_shift_right(&rm, (uint64_t)(L - 1 > 0 ? L - 1 : 0));
    }

    for (int64_t i = 0; i <= S; i++) {     /* mscc.pyx:288 */
        if (M) {
            ob_and(&dmr, &m, &rm);                         /* :291 */
            mlen_by_d[i] = ob_count(&dmr);                 /* :292-298 (every d kept) */
            ob_and(&mf, &f, &dmr);                         /* :300 */
            ob_and(&mr, &r, &dmr);                         /* :301 */
            mscc_fsum[i] = ob_count(&mf);                  /* :303 */
            mscc_rsum[i] = ob_count(&mr);                  /* :304 */
            mscc_ccbins[i] = ob_acount(&mf, &mr);          /* :305 */
            int fill = 0;                                  /* :307-310 buff.pop() or 0 */
            if (nbuff > 0) fill = buff[--nbuff];
            ob_shift_left1(&rm, fill);
        }
        if (!skip_ncc) ncc_ccbins[i] = ob_acount(&f, &r);  /* :314 */
        ob_shift_right(&r, 1);                             /* :316 */
    }
    rc = 0;
done:
    free(buff);
    ob_free(&f); ob_free(&r);
    if (m.w) ob_free(&m);
    if (rm.w) ob_free(&rm);
    if (dmr.w) ob_free(&dmr);
    if (mf.w) ob_free(&mf);
    if (mr.w) ob_free(&mr);
    return rc;
}

/* _fill_result's loop for chromosomes without reads (mscc.pyx:207-215):
 *   for i in 0..S: out[i] = M.acount(RM); RM.rshift(1, 0) */
int pmo_mappable_len(const uint64_t *M, uint64_t nbits, int64_t S, uint64_t *out)
{
    obits m, rm;
    if (ob_from_words(&m, M, nbits)) return -1;
    if (ob_from_words(&rm, M, nbits)) { ob_free(&m); return -1; }
    for (int64_t i = 0; i <= S; i++) {
        out[i] = ob_acount(&m, &rm);
        ob_shift_right(&rm, 1);
    }
    ob_free(&m); ob_free(&rm);
    return 0;
}

/* bitarray.__setitem__ / set (bitarray.pyx:72-79,88-95) on a caller-owned word array. */
void pmo_set_bit(uint64_t *w, uint64_t i) { w[i >> 6] |= (uint64_t)1 << (i & 63); }

/* set(from_, to) == bit_array_set_region(from_, to - from_ + 1): inclusive on both ends */
void pmo_set_region(uint64_t *w, uint64_t from_, uint64_t to)
{
    for (uint64_t i = from_; i <= to; ) {
        if ((i & 63) == 0 && i + 63 <= to) { w[i >> 6] = ~(uint64_t)0; i += 64; }
        else { w[i >> 6] |= (uint64_t)1 << (i & 63); i++; }
    }
}

uint64_t pmo_count(const uint64_t *w, uint64_t nwords)
{
    uint64_t n = 0;
    for (uint64_t i = 0; i < nwords; i++) n += (uint64_t)__builtin_popcountll(w[i]);
    return n;
}
