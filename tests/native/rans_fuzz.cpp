// Memory-safety fuzz of the product's host range coder (csrc/rans_host.cpp), built by tests/test_rans_cpu.py with
// g++ -fsanitize=address,undefined (CPU only: the GPU pool has no sanitizer runs).  Deterministic: a 64-bit LCG drives
// (1) encode -> decode round trips over random Gaussian-like tables incl. escapes, (2) decodes of garbage, of truncated
// strings and of valid strings with flipped bits, (3) the resumable decoder on the same inputs, (4) malformed tables and
// indexes.  Every call must return VC_OK or an error code; the sanitizers turn any out-of-bounds access, misaligned
// or overflowing arithmetic into a non-zero exit.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "vc_hip.h"

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd()
{
    g_state = g_state * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(g_state >> 33);
}

int main()
{
    const int T = 24, stride = 41;
    std::vector<int32_t> cdfs(T * stride, 0), sizes(T), offs(T);
    for (int t = 0; t < T; ++t) {           // a unimodal pmf of random width -> vc_pmf_to_quantized_cdf
        const int bins = 3 + (int)(rnd() % (stride - 4));      // real bins + escape bin, cdf has bins + 1 entries
        std::vector<float> pmf(bins);
        float sum = 0.f;
        for (int i = 0; i < bins; ++i) {
            const float d = (float)(i - bins / 2) / (1.0f + (float)(rnd() % 8));
            pmf[i] = 1.0f / (1.0f + d * d) + 1e-6f;
            sum += pmf[i];
        }
        for (float &p : pmf) p /= sum;
        std::vector<uint32_t> c(bins + 1);
        if (vc_pmf_to_quantized_cdf(pmf.data(), bins, 16, c.data()) != VC_OK) { std::puts("cdf failed"); return 2; }
        for (int i = 0; i <= bins; ++i) cdfs[t * stride + i] = (int32_t)c[i];
        sizes[t] = bins + 1;
        offs[t] = -(bins / 2);
    }
    {   // table construction refuses what is not a probability vector (NaN, negative, > 1, all zero) instead of converting it
        const float nan = std::nanf(""), inf = 1.0f / 0.0f;
        const float bad[][3] = {{nan, 0.5f, 0.5f}, {-0.25f, 0.75f, 0.5f}, {inf, 0.f, 0.f}, {2.0f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {1e30f, 1e30f, 0.f}};
        uint32_t c[4];
        for (const auto &pm : bad)
            if (vc_pmf_to_quantized_cdf(pm, 3, 16, c) == VC_OK) { std::puts("bad pmf accepted"); return 7; }
        const float tiny[3] = {1.0f, 1e-9f, 0.0f};       // zero-probability bins borrow from the widest one ("steal" branch)
        if (vc_pmf_to_quantized_cdf(tiny, 3, 16, c) != VC_OK || c[0] != 0 || c[3] != 65536 || !(c[1] < c[2] && c[2] < c[3])) {
            std::puts("steal branch failed");
            return 8;
        }
    }
    {   // symbols as the GPU produces them from inf / NaN / huge latents ((int32_t)rintf(...)): the encoder must refuse what
        // has no defined encoding (|value| >= 2^27) without overflowing on the way, and still code the largest legal escape
        const int32_t extreme[] = {INT32_MIN, INT32_MIN + 1, INT32_MAX, INT32_MAX - 1, -(1 << 27) - 64, (1 << 27) + 64, 1 << 30};
        for (int32_t s : extreme)
            for (int t = 0; t < T; ++t) {
                const int32_t one_sym[1] = {s}, one_idx[1] = {t};
                std::vector<uint8_t> buf(vc_rans_bound(1));
                if (vc_rans_encode_with_indexes(one_sym, one_idx, 1, cdfs.data(), T, stride, sizes.data(), offs.data(), buf.data(),
                                                buf.size()) != VC_EINVAL) { std::puts("extreme symbol accepted"); return 9; }
            }
        const int32_t legal[] = {(1 << 27) - 64, -(1 << 27) + 64};
        for (int32_t s : legal) {
            const int32_t one_sym[1] = {s}, one_idx[1] = {0};
            int32_t back[1] = {0};
            std::vector<uint8_t> buf(vc_rans_bound(1));
            const long long len = vc_rans_encode_with_indexes(one_sym, one_idx, 1, cdfs.data(), T, stride, sizes.data(), offs.data(),
                                                              buf.data(), buf.size());
            if (len < 0 || vc_rans_decode_with_indexes(buf.data(), (size_t)len, one_idx, 1, cdfs.data(), T, stride, sizes.data(),
                                                       offs.data(), back) != VC_OK || back[0] != s) {
                std::puts("largest legal escape failed");
                return 10;
            }
        }
    }
    long ok = 0, err = 0;
    for (int round = 0; round < 200; ++round) {
        const size_t n = rnd() % 3000;
        std::vector<int32_t> sym(n), idx(n), out(n);
        for (size_t i = 0; i < n; ++i) {
            idx[i] = (int32_t)(rnd() % T);
            const int spread = (rnd() % 50 == 0) ? 100000 : sizes[idx[i]];      // some symbols far outside the table: escapes
            sym[i] = (int32_t)(rnd() % (2 * spread + 1)) - spread;
        }
        std::vector<uint8_t> buf(vc_rans_bound(n));
        const long long len = vc_rans_encode_with_indexes(sym.data(), idx.data(), n, cdfs.data(), T, stride, sizes.data(), offs.data(),
                                                          buf.data(), buf.size());
        if (len < 0) { std::printf("encode failed: %lld\n", len); return 3; }
        // exact-size copy so that the sanitizer sees any read past the string
        std::vector<uint8_t> good(buf.begin(), buf.begin() + len);
        if (vc_rans_decode_with_indexes(good.data(), good.size(), idx.data(), n, cdfs.data(), T, stride, sizes.data(), offs.data(),
                                        out.data()) != VC_OK || std::memcmp(out.data(), sym.data(), n * sizeof(int32_t)) != 0) {
            std::puts("round trip failed");
            return 4;
        }
        for (int variant = 0; variant < 6; ++variant) {
            std::vector<uint8_t> bad;
            if (variant == 0) {                       // garbage
                bad.resize(rnd() % (2 * good.size() + 8));
                for (auto &b : bad) b = (uint8_t)rnd();
            } else if (variant == 1) {                // truncated
                bad.assign(good.begin(), good.begin() + (good.empty() ? 0 : rnd() % good.size()));
            } else {                                  // bit flips
                bad = good;
                for (int f = 0; f < 1 + (int)(rnd() % 4) && !bad.empty(); ++f) bad[rnd() % bad.size()] ^= (uint8_t)(1u << (rnd() % 8));
            }
            const int rc = vc_rans_decode_with_indexes(bad.data(), bad.size(), idx.data(), n, cdfs.data(), T, stride, sizes.data(),
                                                       offs.data(), out.data());
            (rc == VC_OK ? ok : err)++;
            uint64_t st[2] = {0, 0};                  // the resumable decoder, in two calls
            const size_t half = n / 2;
            int rs = vc_rans_decode_stream(bad.data(), bad.size(), st, idx.data(), half, cdfs.data(), T, stride, sizes.data(), offs.data(),
                                           out.data());
            if (rs == VC_OK)
                rs = vc_rans_decode_stream(bad.data(), bad.size(), st, idx.data() + half, n - half, cdfs.data(), T, stride, sizes.data(),
                                           offs.data(), out.data() + half);
            (rs == VC_OK ? ok : err)++;
        }
        if (n) {                                      // malformed indexes / tables must be refused, not dereferenced
            std::vector<int32_t> bad_idx(idx);
            bad_idx[rnd() % n] = (rnd() & 1) ? T + (int32_t)(rnd() % 1000) : -1 - (int32_t)(rnd() % 1000);
            if (vc_rans_decode_with_indexes(good.data(), good.size(), bad_idx.data(), n, cdfs.data(), T, stride, sizes.data(), offs.data(),
                                            out.data()) == VC_OK) { std::puts("bad index accepted"); return 5; }
            std::vector<int32_t> bad_sizes(sizes);
            bad_sizes[rnd() % T] = stride + 1 + (int32_t)(rnd() % 100);
            if (vc_rans_decode_with_indexes(good.data(), good.size(), idx.data(), n, cdfs.data(), T, stride, bad_sizes.data(), offs.data(),
                                            out.data()) == VC_OK) { std::puts("bad table size accepted"); return 6; }
        }
    }
    std::printf("rans fuzz: %ld decodes returned symbols, %ld returned an error, no sanitizer report\n", ok, err);
    return 0;
}
