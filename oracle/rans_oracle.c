/*
 * oracle/rans_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Plain-C restatement of the entropy-coder pieces the reference reaches through
 * the third-party dependency CompressAI (pinned compressai==1.1.8 at
 * LHBDC/environment.yml:142; NOT under /root/reference, NOT installed):
 *
 *   - pmf_to_quantized_cdf      (CompressAI _CXX; reached from
 *                                EntropyBottleneck.update / GaussianConditional.update,
 *                                triggered by LHBDC/encode_B.py:34-35, decode_B.py:111-112)
 *   - RansEncoder.encode_with_indexes / RansDecoder.decode_with_indexes
 *                               (CompressAI ans; reached from
 *                                LHBDC/model/layers.py:97-98,103,108,112 and :172-187,
 *                                Flex-Rate.../b_model/layers.py:160-181,275-297)
 *
 * The algorithm is the published ryg_rans "rans64" coder (32-bit word renormalisation,
 * L = 2^31) with 16-bit probabilities and a 4-bit bypass escape for out-of-table values,
 * as described in SURVEY.md Appendix A.4/A.5.
 *
 * PARITY UNPINNED at the third-party boundary: the reference holds no golden vectors
 * for this coder and the real library cannot be run here.  What IS pinned: the
 * round-trip property, hand-computed small vectors (tests/test_oracle_rans.py) and the
 * byte-equality of the product coder (csrc/rans_host.cpp) with this file.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define PROB_BITS 16u
#define BYPASS_BITS 4u
#define BYPASS_MAX ((1u << BYPASS_BITS) - 1u) /* 15 */
#define RANS_LOW (1ull << 31)

/* ---------------------------------------------------------------------------------
 * pmf -> quantised cdf.  `pmf` holds n float32 probabilities (the last one is the tail
 * mass); `cdf` receives n+1 entries, cdf[0]=0, cdf[n]=65536, strictly increasing.
 * Returns 0, or -1 when no bin can donate frequency (degenerate pmf).
 * ------------------------------------------------------------------------------- */
int vco_pmf_to_quantized_cdf(const float *pmf, int n, uint32_t *cdf)
{
    const int len = n + 1;
    cdf[0] = 0;
    for (int i = 0; i < n; ++i) {
        /* float32 product, round-half-away-from-zero (std::round on float) */
        float scaled = pmf[i] * (float)(1u << PROB_BITS);
        cdf[i + 1] = (uint32_t)roundf(scaled);
    }
    uint32_t total = 0;
    for (int i = 0; i < len; ++i) total += cdf[i];
    if (total == 0) return -1;
    for (int i = 0; i < len; ++i)
        cdf[i] = (uint32_t)((((uint64_t)1 << PROB_BITS) * (uint64_t)cdf[i]) / total);
    for (int i = 1; i < len; ++i) cdf[i] += cdf[i - 1];
    cdf[len - 1] = 1u << PROB_BITS;

    for (int i = 0; i < len - 1; ++i) {
        if (cdf[i] != cdf[i + 1]) continue;
        /* empty bin: take one count from the smallest bin that can spare it */
        uint32_t best = 0xFFFFFFFFu;
        int donor = -1;
        for (int j = 0; j < len - 1; ++j) {
            uint32_t f = cdf[j + 1] - cdf[j];
            if (f > 1 && f < best) { best = f; donor = j; }
        }
        if (donor < 0) return -1;
        if (donor < i) {
            for (int j = donor + 1; j <= i; ++j) cdf[j]--;
        } else {
            for (int j = i + 1; j <= donor; ++j) cdf[j]++;
        }
    }
    return 0;
}

/* One queued coding step: either a table symbol (start,range at 16 bits) or a raw
 * nibble pushed with 4-bit precision. */
typedef struct { uint16_t start; uint16_t range; uint8_t raw; } step_t;

typedef struct { step_t *v; size_t n, cap; } steps_t;

static int steps_push(steps_t *s, uint16_t start, uint16_t range, uint8_t raw)
{
    if (s->n == s->cap) {
        size_t nc = s->cap ? s->cap * 2 : 1024;
        step_t *nv = (step_t *)realloc(s->v, nc * sizeof(step_t));
        if (!nv) return -1;
        s->v = nv; s->cap = nc;
    }
    s->v[s->n].start = start; s->v[s->n].range = range; s->v[s->n].raw = raw;
    s->n++;
    return 0;
}

/* ---------------------------------------------------------------------------------
 * Encode `count` symbols.  cdfs is a dense [n_tables][cdf_stride] int32 matrix,
 * cdf_sizes[t] = number of valid entries of table t, offsets[t] = value of the first
 * table bin.  Output: malloc'ed buffer in *out (caller frees with vco_free), byte
 * length as return value (<0 on error).
 * ------------------------------------------------------------------------------- */
long vco_rans_encode_with_indexes(const int32_t *symbols, const int32_t *indexes, size_t count,
                                  const int32_t *cdfs, int cdf_stride,
                                  const int32_t *cdf_sizes, const int32_t *offsets,
                                  uint8_t **out)
{
    steps_t q = {0, 0, 0};
    for (size_t i = 0; i < count; ++i) {
        const int32_t t = indexes[i];
        const int32_t *cdf = cdfs + (size_t)t * cdf_stride;
        const int32_t escape = cdf_sizes[t] - 2; /* index of the escape bin */
        int32_t v = symbols[i] - offsets[t];
        uint32_t raw = 0;
        if (v < 0) { raw = (uint32_t)(-2 * v - 1); v = escape; }
        else if (v >= escape) { raw = (uint32_t)(2 * (v - escape)); v = escape; }
        if (steps_push(&q, (uint16_t)cdf[v], (uint16_t)(cdf[v + 1] - cdf[v]), 0)) return -1;
        if (v == escape) {
            int32_t nibbles = 0;
            while ((raw >> (nibbles * BYPASS_BITS)) != 0) ++nibbles;
            int32_t left = nibbles;
            while (left >= (int32_t)BYPASS_MAX) {
                if (steps_push(&q, BYPASS_MAX, BYPASS_MAX + 1, 1)) return -1;
                left -= BYPASS_MAX;
            }
            if (steps_push(&q, (uint16_t)left, (uint16_t)(left + 1), 1)) return -1;
            for (int32_t j = 0; j < nibbles; ++j) {
                uint32_t nib = (raw >> (j * BYPASS_BITS)) & BYPASS_MAX;
                if (steps_push(&q, (uint16_t)nib, (uint16_t)(nib + 1), 1)) return -1;
            }
        }
    }

    /* rANS is LIFO: run the queue backwards, write 32-bit words from the end. */
    size_t words = q.n + 2;
    uint32_t *buf = (uint32_t *)malloc(words * sizeof(uint32_t));
    if (!buf) { free(q.v); return -1; }
    uint32_t *p = buf + words;
    uint64_t x = RANS_LOW;
    for (size_t k = q.n; k-- > 0;) {
        const step_t s = q.v[k];
        if (!s.raw) {
            uint64_t lim = ((RANS_LOW >> PROB_BITS) << 32) * (uint64_t)s.range;
            if (x >= lim) { *--p = (uint32_t)x; x >>= 32; }
            x = ((x / s.range) << PROB_BITS) + (x % s.range) + s.start;
        } else {
            uint64_t lim = ((RANS_LOW >> 16) << 32) * (uint64_t)(1u << (16 - BYPASS_BITS));
            if (x >= lim) { *--p = (uint32_t)x; x >>= 32; }
            x = (x << BYPASS_BITS) | s.start;
        }
    }
    p -= 2;
    p[0] = (uint32_t)x;
    p[1] = (uint32_t)(x >> 32);
    size_t nbytes = (size_t)((buf + words) - p) * sizeof(uint32_t);
    uint8_t *res = (uint8_t *)malloc(nbytes ? nbytes : 1);
    if (!res) { free(buf); free(q.v); return -1; }
    memcpy(res, p, nbytes);
    free(buf);
    free(q.v);
    *out = res;
    return (long)nbytes;
}

static inline uint32_t take_bits(uint64_t *px, const uint32_t **pp, uint32_t nbits)
{
    uint64_t x = *px;
    uint32_t val = (uint32_t)(x & ((1u << nbits) - 1u));
    x >>= nbits;
    if (x < RANS_LOW) { x = (x << 32) | **pp; (*pp)++; }
    *px = x;
    return val;
}

/* Decode `count` symbols from `data` (nbytes, 32-bit little-endian words). */
int vco_rans_decode_with_indexes(const uint8_t *data, size_t nbytes,
                                 const int32_t *indexes, size_t count,
                                 const int32_t *cdfs, int cdf_stride,
                                 const int32_t *cdf_sizes, const int32_t *offsets,
                                 int32_t *out)
{
    if (nbytes < 8) return -1;
    /* copy with slack so that the last renormalisation read stays in bounds */
    size_t nwords = nbytes / 4;
    uint32_t *w = (uint32_t *)calloc(nwords + 2, sizeof(uint32_t));
    if (!w) return -1;
    memcpy(w, data, nwords * 4);
    const uint32_t *p = w;
    uint64_t x = (uint64_t)p[0] | ((uint64_t)p[1] << 32);
    p += 2;
    for (size_t i = 0; i < count; ++i) {
        const int32_t t = indexes[i];
        const int32_t *cdf = cdfs + (size_t)t * cdf_stride;
        const int32_t n = cdf_sizes[t];
        const int32_t escape = n - 2;
        const uint32_t cum = (uint32_t)(x & 0xFFFFu);
        int32_t j = 0;
        while (j < n && (uint32_t)cdf[j] <= cum) ++j; /* first entry > cum */
        const int32_t s = j - 1;
        const uint32_t start = (uint32_t)cdf[s], range = (uint32_t)(cdf[s + 1] - cdf[s]);
        x = (uint64_t)range * (x >> PROB_BITS) + (x & 0xFFFFu) - start;
        if (x < RANS_LOW) { x = (x << 32) | *p; p++; }
        int32_t v = s;
        if (v == escape) {
            int32_t got = (int32_t)take_bits(&x, &p, BYPASS_BITS);
            int32_t nibbles = got;
            while (got == (int32_t)BYPASS_MAX) {
                got = (int32_t)take_bits(&x, &p, BYPASS_BITS);
                nibbles += got;
            }
            int32_t raw = 0;
            for (int32_t k = 0; k < nibbles; ++k) {
                int32_t nib = (int32_t)take_bits(&x, &p, BYPASS_BITS);
                raw |= nib << (k * BYPASS_BITS);
            }
            v = raw >> 1;
            if (raw & 1) v = -v - 1; else v += escape;
        }
        out[i] = v + offsets[t];
    }
    free(w);
    return 0;
}

/* RansDecoder.set_stream / decode_stream: the same decoder continued over several calls on one string.
 * state[0] = coder state, state[1] = index of the next unread 32-bit word; {0, 0} starts a stream. */
int vco_rans_decode_stream(const uint8_t *data, size_t nbytes, uint64_t *state,
                           const int32_t *indexes, size_t count,
                           const int32_t *cdfs, int cdf_stride,
                           const int32_t *cdf_sizes, const int32_t *offsets,
                           int32_t *out)
{
    if (nbytes < 8) return -1;
    size_t nwords = nbytes / 4;
    uint32_t *w = (uint32_t *)calloc(nwords + 2, sizeof(uint32_t));
    if (!w) return -1;
    memcpy(w, data, nwords * 4);
    const uint32_t *p = w + state[1];
    uint64_t x = state[0];
    if (state[1] == 0) {
        x = (uint64_t)p[0] | ((uint64_t)p[1] << 32);
        p += 2;
    }
    for (size_t i = 0; i < count; ++i) {
        const int32_t t = indexes[i];
        const int32_t *cdf = cdfs + (size_t)t * cdf_stride;
        const int32_t n = cdf_sizes[t];
        const int32_t escape = n - 2;
        const uint32_t cum = (uint32_t)(x & 0xFFFFu);
        int32_t j = 0;
        while (j < n && (uint32_t)cdf[j] <= cum) ++j;
        const int32_t s = j - 1;
        const uint32_t start = (uint32_t)cdf[s], range = (uint32_t)(cdf[s + 1] - cdf[s]);
        x = (uint64_t)range * (x >> PROB_BITS) + (x & 0xFFFFu) - start;
        if (x < RANS_LOW) { x = (x << 32) | *p; p++; }
        int32_t v = s;
        if (v == escape) {
            int32_t got = (int32_t)take_bits(&x, &p, BYPASS_BITS);
            int32_t nibbles = got;
            while (got == (int32_t)BYPASS_MAX) {
                got = (int32_t)take_bits(&x, &p, BYPASS_BITS);
                nibbles += got;
            }
            int32_t raw = 0;
            for (int32_t k = 0; k < nibbles; ++k) {
                int32_t nib = (int32_t)take_bits(&x, &p, BYPASS_BITS);
                raw |= nib << (k * BYPASS_BITS);
            }
            v = raw >> 1;
            if (raw & 1) v = -v - 1; else v += escape;
        }
        out[i] = v + offsets[t];
    }
    state[0] = x;
    state[1] = (uint64_t)(p - w);
    free(w);
    return 0;
}

void vco_free(void *p) { free(p); }
