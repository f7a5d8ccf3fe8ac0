// Host range coder of the product: rans64 (32-bit word renormalisation, L = 2^31), 16-bit table
// probabilities, 4-bit bypass escape -- the bitstream format of CompressAI's ans module that the
// reference reaches through EntropyModel.compress/decompress (LHBDC/model/layers.py:97-112).
//
// The state machine is inherently serial, so it runs on a host core while the GPU produces the next
// frame's symbols.  Unlike the queue-then-flush formulation, the encoder walks the symbols once,
// backwards, expanding escapes in reverse on the fly and writing 32-bit words from the end of the
// caller's buffer -- no intermediate symbol queue, no allocation.
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>
#include "vc_hip.h"

namespace {
// every table must fit its row and hold at least one real bin + the escape bin
bool tables_ok(const int32_t *cdf_sizes, int n_tables, int cdf_stride)
{
    if (n_tables < 1 || cdf_stride < 3) return false;
    for (int t = 0; t < n_tables; ++t)
        if (cdf_sizes[t] < 3 || cdf_sizes[t] > cdf_stride) return false;
    return true;
}

constexpr uint32_t kProbBits = 16;
constexpr uint32_t kBypassBits = 4;
constexpr uint32_t kBypassMax = (1u << kBypassBits) - 1u;
constexpr uint64_t kLow = 1ull << 31;

struct Encoder {
    uint64_t x = kLow;
    uint32_t *p;      // next free word (grows downwards)
    uint32_t *floor;  // lowest legal address
    bool overflow = false;

    inline void emit()
    {
        if (p == floor) { overflow = true; return; }
        *--p = static_cast<uint32_t>(x);
        x >>= 32;
    }
    inline void put(uint32_t start, uint32_t range)
    {
        const uint64_t lim = ((kLow >> kProbBits) << 32) * range;
        if (x >= lim) emit();
        x = ((x / range) << kProbBits) + (x % range) + start;
    }
    inline void put_nibble(uint32_t v)
    {
        const uint64_t lim = ((kLow >> 16) << 32) * static_cast<uint64_t>(1u << (16 - kBypassBits));
        if (x >= lim) emit();
        x = (x << kBypassBits) | v;
    }
};
}  // namespace

extern "C" size_t vc_rans_bound(size_t count) { return 4 * (2 * count + 4); }

extern "C" long long vc_rans_encode_with_indexes(const int32_t *symbols, const int32_t *indexes, size_t count,
                                                 const int32_t *cdfs, int n_tables, int cdf_stride,
                                                 const int32_t *cdf_sizes, const int32_t *offsets, uint8_t *out, size_t out_cap)
{
    if ((count && (!symbols || !indexes)) || !cdfs || !cdf_sizes || !offsets || !out || out_cap < 8) return VC_EINVAL;
    if (reinterpret_cast<uintptr_t>(out) % 4) return VC_EINVAL;
    if (!tables_ok(cdf_sizes, n_tables, cdf_stride)) return VC_EINVAL;
    const size_t cap_words = out_cap / 4;
    uint32_t *base = reinterpret_cast<uint32_t *>(out);
    Encoder enc;
    enc.p = base + cap_words;
    enc.floor = base + 2;  // keep room for the final flush
    for (size_t i = count; i-- > 0;) {
        const int32_t t = indexes[i];
        if (static_cast<uint32_t>(t) >= static_cast<uint32_t>(n_tables)) return VC_EINVAL;
        const int32_t *cdf = cdfs + static_cast<size_t>(t) * cdf_stride;
        const int32_t escape = cdf_sizes[t] - 2;
        // (64-bit: a symbol rounded from an inf / NaN latent is INT_MIN or huge, and `symbol - offset`, `-2 v - 1`,
        //  `2 (v - escape)` must not overflow before the range check below refuses it)
        const int64_t v = static_cast<int64_t>(symbols[i]) - offsets[t];
        if (v >= 0 && v < escape) {
            enc.put(static_cast<uint32_t>(cdf[v]), static_cast<uint32_t>(cdf[v + 1] - cdf[v]));
            continue;
        }
        // out-of-table value: sign/magnitude folded into `raw`, sent as nibbles after the escape bin
        const int64_t raw64 = v < 0 ? -2 * v - 1 : 2 * (v - escape);
        // (the format's own encoder counts nibbles with `raw >> (n*4)` on a 32-bit value: undefined from 2^28 on, so
        //  magnitudes that need an eighth nibble have no defined encoding -- refuse them)
        if (raw64 >> 28) return VC_EINVAL;
        const uint32_t raw = static_cast<uint32_t>(raw64);
        int32_t nibbles = 0;
        while ((raw >> (nibbles * kBypassBits)) != 0) ++nibbles;
        for (int32_t j = nibbles - 1; j >= 0; --j) enc.put_nibble((raw >> (j * kBypassBits)) & kBypassMax);
        // nibble count in base-15 "unary" chunks; forward order is 15,15,...,rest -> reverse: rest first
        enc.put_nibble(static_cast<uint32_t>(nibbles) % kBypassMax);
        for (int32_t c = nibbles / static_cast<int32_t>(kBypassMax); c > 0; --c) enc.put_nibble(kBypassMax);
        enc.put(static_cast<uint32_t>(cdf[escape]), static_cast<uint32_t>(cdf[escape + 1] - cdf[escape]));
    }
    if (enc.overflow) return VC_ENOMEM;
    enc.p -= 2;
    enc.p[0] = static_cast<uint32_t>(enc.x);
    enc.p[1] = static_cast<uint32_t>(enc.x >> 32);
    const size_t nbytes = static_cast<size_t>((base + cap_words) - enc.p) * 4;
    std::memmove(out, enc.p, nbytes);
    return static_cast<long long>(nbytes);
}

namespace {
// Decoder core, resumable: `x` / `pos` = coder state and index of the next unread 32-bit word.  A fresh stream starts
// with pos == 0 (the first two words initialise x).
int decode_core(const uint8_t *data, size_t nbytes, uint64_t &x, size_t &pos, const int32_t *indexes, size_t count,
                const int32_t *cdfs, int n_tables, int cdf_stride, const int32_t *cdf_sizes, const int32_t *offsets, int32_t *out)
{
    const size_t nwords = nbytes / 4;
    auto next_word = [&](uint32_t &w) -> bool {
        if (pos >= nwords) return false;
        std::memcpy(&w, data + 4 * pos, 4);
        ++pos;
        return true;
    };
    if (pos == 0) {
        uint32_t lo = 0, hi = 0;
        if (!next_word(lo) || !next_word(hi)) return VC_EDATA;
        x = static_cast<uint64_t>(lo) | (static_cast<uint64_t>(hi) << 32);
    }
    bool bad = false;
    auto refill = [&]() {
        if (x < kLow) {
            uint32_t w = 0;
            if (!next_word(w)) bad = true;
            x = (x << 32) | w;
        }
    };
    auto nibble = [&]() -> uint32_t {
        const uint32_t v = static_cast<uint32_t>(x & kBypassMax);
        x >>= kBypassBits;
        refill();
        return v;
    };
    // Symbol lookup: the format's reference decoder scans the table linearly.  Here every table gets a 256-entry
    // jump table over the top 8 bits of the cumulative value (first bin that can contain it), built once per call
    // (n_tables x 256 steps, negligible next to the symbols); the scan then starts at most a few bins before the
    // answer whatever the table length -- the decoder is what a frame's host time is made of (1 M symbols).
    constexpr int kLutBits = 8;
    thread_local std::vector<uint16_t> lut;
    lut.resize(static_cast<size_t>(n_tables) << kLutBits);
    for (int t = 0; t < n_tables; ++t) {
        const int32_t *cdf = cdfs + static_cast<size_t>(t) * cdf_stride;
        const int32_t n = cdf_sizes[t];
        uint16_t *row = lut.data() + (static_cast<size_t>(t) << kLutBits);
        int32_t j = 0;
        for (uint32_t b = 0; b < (1u << kLutBits); ++b) {
            const uint32_t lo = b << (16 - kLutBits);
            while (j + 1 < n - 1 && static_cast<uint32_t>(cdf[j + 1]) <= lo) ++j;   // last bin starting at or before lo
            row[b] = static_cast<uint16_t>(j);
        }
    }
    for (size_t i = 0; i < count; ++i) {
        const int32_t t = indexes[i];
        if (static_cast<uint32_t>(t) >= static_cast<uint32_t>(n_tables)) return VC_EINVAL;
        const int32_t *cdf = cdfs + static_cast<size_t>(t) * cdf_stride;
        const int32_t n = cdf_sizes[t], escape = n - 2;
        const uint32_t cum = static_cast<uint32_t>(x & 0xFFFFu);
        int32_t sidx = lut[(static_cast<size_t>(t) << kLutBits) + (cum >> (16 - kLutBits))];
        while (sidx + 1 < n - 1 && static_cast<uint32_t>(cdf[sidx + 1]) <= cum) ++sidx;
        const uint32_t start = static_cast<uint32_t>(cdf[sidx]);
        const uint32_t range = static_cast<uint32_t>(cdf[sidx + 1]) - start;
        x = static_cast<uint64_t>(range) * (x >> kProbBits) + (x & 0xFFFFu) - start;
        refill();
        int32_t v = sidx;
        if (v == escape) {
            uint32_t got = nibble(), nibbles = got;
            while (got == kBypassMax) {
                got = nibble();
                nibbles += got;
                if (bad || nibbles > 7) break;
            }
            if (nibbles > 7) return VC_EDATA;      // no encoder of this format emits more than 7 (see the encoder)
            uint32_t raw = 0;
            for (uint32_t k = 0; k < nibbles; ++k) raw |= nibble() << (k * kBypassBits);
            v = static_cast<int32_t>(raw >> 1);
            if (raw & 1u) v = -v - 1; else v += escape;
        }
        if (bad) return VC_EDATA;
        out[i] = v + offsets[t];
    }
    return VC_OK;
}
}  // namespace

extern "C" int vc_rans_decode_with_indexes(const uint8_t *data, size_t nbytes, const int32_t *indexes, size_t count,
                                           const int32_t *cdfs, int n_tables, int cdf_stride, const int32_t *cdf_sizes,
                                           const int32_t *offsets, int32_t *out)
{
    if (!data || nbytes < 8 || (nbytes % 4) || (count && (!indexes || !out)) || !cdfs || !cdf_sizes || !offsets) return VC_EINVAL;
    if (!tables_ok(cdf_sizes, n_tables, cdf_stride)) return VC_EINVAL;
    uint64_t x = 0;
    size_t pos = 0;
    return decode_core(data, nbytes, x, pos, indexes, count, cdfs, n_tables, cdf_stride, cdf_sizes, offsets, out);
}

extern "C" int vc_rans_decode_stream(const uint8_t *data, size_t nbytes, uint64_t *state, const int32_t *indexes, size_t count,
                                     const int32_t *cdfs, int n_tables, int cdf_stride, const int32_t *cdf_sizes,
                                     const int32_t *offsets, int32_t *out)
{
    if (!data || nbytes < 8 || (nbytes % 4) || !state || (count && (!indexes || !out)) || !cdfs || !cdf_sizes || !offsets) return VC_EINVAL;
    if (!tables_ok(cdf_sizes, n_tables, cdf_stride)) return VC_EINVAL;
    uint64_t x = state[0];
    size_t pos = static_cast<size_t>(state[1]);
    if (pos == 1 || pos > nbytes / 4) return VC_EINVAL;
    const int rc = decode_core(data, nbytes, x, pos, indexes, count, cdfs, n_tables, cdf_stride, cdf_sizes, offsets, out);
    state[0] = x;
    state[1] = pos;
    return rc;
}

extern "C" int vc_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf)
{
    if (!pmf || !cdf || n < 1 || precision != 16) return VC_EINVAL;
    const uint32_t one = 1u << precision;
    std::vector<uint32_t> freq(static_cast<size_t>(n) + 1, 0u);
    for (int i = 0; i < n; ++i) {
        // a probability; anything else (NaN from a broken checkpoint, negative, > 1) has no defined conversion to an integer
        if (!(pmf[i] >= 0.0f && pmf[i] <= 1.0f)) return VC_EDATA;
        freq[i + 1] = static_cast<uint32_t>(std::round(pmf[i] * static_cast<float>(one)));
    }
    const uint32_t total = std::accumulate(freq.begin(), freq.end(), 0u);
    if (total == 0) return VC_EDATA;
    for (auto &f : freq) f = static_cast<uint32_t>((static_cast<uint64_t>(one) * f) / total);
    std::partial_sum(freq.begin(), freq.end(), cdf);
    cdf[n] = one;
    for (int i = 0; i < n; ++i) {
        if (cdf[i] != cdf[i + 1]) continue;
        uint32_t best = ~0u;
        int donor = -1;
        for (int j = 0; j < n; ++j) {
            const uint32_t f = cdf[j + 1] - cdf[j];
            if (f > 1 && f < best) { best = f; donor = j; }
        }
        if (donor < 0) return VC_EDATA;
        if (donor < i) for (int j = donor + 1; j <= i; ++j) --cdf[j];
        else for (int j = i + 1; j <= donor; ++j) ++cdf[j];
    }
    return VC_OK;
}

// ABI number: bumped whenever a struct of include/vc_hip.h changes size or layout (vc_conv_desc grew in rounds 4 and 5); the Python
// binding refuses a library whose number differs from the header it was written against
extern "C" int vc_abi_version(void) { return VC_ABI_VERSION; }
extern "C" const char *vc_version(void) { return "vc_hip 0.5 (abi 5)"; }
extern "C" const char *vc_target_arch(void) { return "gfx950"; }
