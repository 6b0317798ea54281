// Host-side entropy-coding helpers (see hesic_host.h).  Sequential by nature (rANS is a serial state
// machine); they sit on the bit-stream path, not on the throughput path.
#include "hesic_host.h"
#include <cstdlib>

#include <cmath>
#include <cstring>
#include <vector>

namespace {
constexpr int kPrecision = 16;        // probability bits of the model CDFs
constexpr int kBypassBits = 4;        // raw-bit escape width
constexpr uint32_t kBypassMax = (1u << kBypassBits) - 1;
constexpr uint64_t kLow = 1ull << 31; // normalisation interval lower bound (ryg rans64)

struct Item { uint16_t start, range; bool raw; };

// state update for a modelled symbol / for `bits` raw bits (freq = 2^(16-bits) in 16-bit terms)
inline void put(uint64_t& x, std::vector<uint32_t>& words, uint32_t start, uint32_t freq, uint32_t scale_bits) {
    const uint64_t x_max = ((kLow >> scale_bits) << 32) * freq;
    if (x >= x_max) { words.push_back((uint32_t)x); x >>= 32; }
    x = ((x / freq) << scale_bits) + (x % freq) + start;
}
inline void put_bits(uint64_t& x, std::vector<uint32_t>& words, uint32_t val, uint32_t nbits) {
    const uint64_t x_max = ((kLow >> 16) << 32) * (uint64_t)(1u << (16 - nbits));
    if (x >= x_max) { words.push_back((uint32_t)x); x >>= 32; }
    x = (x << nbits) | val;
}
}  // namespace

struct hesic_rans_encoder { std::vector<Item> q; };
struct hesic_rans_decoder { std::vector<uint32_t> words; size_t pos = 0; uint64_t x = 0; bool ready = false; };

extern "C" int hesic_pmf_to_quantized_cdf(const float* pmf, int n, int precision, uint32_t* cdf) {
    if (!pmf || !cdf || n < 1 || precision < 1 || precision > 31) return -1;
    const uint32_t one = 1u << precision;
    // 1) integer frequencies, 2) rescale so they sum to ~2^precision, 3) prefix sum, 4) repair zero-width bins
    uint32_t total = 0;
    cdf[0] = 0;
    for (int i = 0; i < n; ++i) { cdf[i + 1] = (uint32_t)std::round(pmf[i] * (float)one); total += cdf[i + 1]; }
    if (total == 0) return -1;
    uint32_t run = 0;
    for (int i = 0; i <= n; ++i) { run += (uint32_t)(((uint64_t)one * cdf[i]) / total); cdf[i] = run; }
    cdf[n] = one;
    for (int i = 0; i < n; ++i) {
        if (cdf[i] != cdf[i + 1]) continue;
        // steal one count from the narrowest bin that can spare it (width > 1), first such bin wins ties
        uint32_t best = ~0u;
        int donor = -1;
        for (int j = 0; j < n; ++j) {
            const uint32_t wdt = cdf[j + 1] - cdf[j];
            if (wdt > 1 && wdt < best) { best = wdt; donor = j; }
        }
        if (donor < 0) return -1;
        if (donor < i) for (int j = donor + 1; j <= i; ++j) cdf[j]--;
        else for (int j = i + 1; j <= donor; ++j) cdf[j]++;
    }
    return 0;
}

// Every row of a pmf matrix at once (EntropyModel.update(): one row per channel / scale level): row r codes
// pmf[r][0 .. lengths[r]) followed by its tail-mass escape bin; the quantised CDF (lengths[r] + 2 entries) lands in
// cdf_out[r][...], the rest of the row is zero.
extern "C" int hesic_pmf_rows_to_quantized_cdfs(const float* pmf, int rows, int pmf_stride, const int32_t* lengths, const float* tail_mass,
                                                int precision, int32_t* cdf_out, int cdf_stride) {
    if (!pmf || !lengths || !tail_mass || !cdf_out || rows < 1 || pmf_stride < 1) return -1;
    std::vector<float> row;
    std::vector<uint32_t> q;
    for (int r = 0; r < rows; ++r) {
        const int n = lengths[r];
        if (n < 1 || n > pmf_stride || n + 2 > cdf_stride) return -1;
        row.assign(pmf + (int64_t)r * pmf_stride, pmf + (int64_t)r * pmf_stride + n);
        row.push_back(tail_mass[r]);
        q.assign(n + 2, 0u);
        if (hesic_pmf_to_quantized_cdf(row.data(), n + 1, precision, q.data()) != 0) return -1;
        int32_t* dst = cdf_out + (int64_t)r * cdf_stride;
        for (int i = 0; i < cdf_stride; ++i) dst[i] = i < n + 2 ? (int32_t)q[i] : 0;
    }
    return 0;
}

extern "C" hesic_rans_encoder* hesic_rans_encoder_new(void) { return new hesic_rans_encoder(); }
extern "C" void hesic_rans_encoder_free(hesic_rans_encoder* e) { delete e; }

extern "C" int hesic_rans_encoder_push(hesic_rans_encoder* e, const int32_t* symbols, const int32_t* indexes, int64_t n,
                                       const int32_t* cdfs, int ncdf, int stride, const int32_t* sizes, const int32_t* offsets) {
    if (!e || !symbols || !indexes || !cdfs || !sizes || !offsets || n < 0) return -1;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t ci = indexes[i];
        if (ci < 0 || ci >= ncdf || sizes[ci] < 2 || sizes[ci] > stride) return -1;
        const int32_t* cdf = cdfs + (int64_t)ci * stride;
        const int32_t last = sizes[ci] - 2;          // index of the escape bin
        int32_t v = symbols[i] - offsets[ci];
        uint32_t raw = 0;
        if (v < 0) { raw = (uint32_t)(-2 * v - 1); v = last; }        // odd  => below the support
        else if (v >= last) { raw = (uint32_t)(2 * (v - last)); v = last; }  // even => at/above it
        e->q.push_back({(uint16_t)cdf[v], (uint16_t)(cdf[v + 1] - cdf[v]), false});
        if (v == last) {
            int32_t nib = 0;
            while ((raw >> (nib * kBypassBits)) != 0) ++nib;
            int32_t c = nib;                          // nibble count in unary-ish base-15 chunks
            while (c >= (int32_t)kBypassMax) { e->q.push_back({(uint16_t)kBypassMax, (uint16_t)(kBypassMax + 1), true}); c -= kBypassMax; }
            e->q.push_back({(uint16_t)c, (uint16_t)(c + 1), true});
            for (int32_t j = 0; j < nib; ++j) {
                const uint16_t d = (uint16_t)((raw >> (j * kBypassBits)) & kBypassMax);
                e->q.push_back({d, (uint16_t)(d + 1), true});
            }
        }
    }
    return 0;
}

extern "C" int64_t hesic_rans_encoder_flush(hesic_rans_encoder* e, uint8_t* out, int64_t cap) {
    if (!e) return -1;
    uint64_t x = kLow;
    std::vector<uint32_t> w;          // words in emission order; the stream is their reverse
    w.reserve(e->q.size() / 2 + 4);
    for (size_t i = e->q.size(); i-- > 0;) {
        const Item& s = e->q[i];
        if (s.raw) put_bits(x, w, s.start, kBypassBits);
        else put(x, w, s.start, s.range, kPrecision);
    }
    w.push_back((uint32_t)(x >> 32));
    w.push_back((uint32_t)x);
    const int64_t nbytes = (int64_t)w.size() * 4;
    if (!out || cap < nbytes) return nbytes;
    uint32_t* o = reinterpret_cast<uint32_t*>(out);
    for (size_t i = 0; i < w.size(); ++i) { const uint32_t v = w[w.size() - 1 - i]; std::memcpy(o + i, &v, 4); }
    e->q.clear();
    return nbytes;
}

extern "C" hesic_rans_decoder* hesic_rans_decoder_new(void) { return new hesic_rans_decoder(); }
extern "C" void hesic_rans_decoder_free(hesic_rans_decoder* d) { delete d; }

extern "C" int hesic_rans_decoder_set_stream(hesic_rans_decoder* d, const uint8_t* bytes, int64_t nbytes) {
    if (!d || !bytes || nbytes < 8 || nbytes % 4) return -1;
    d->words.resize((size_t)nbytes / 4);
    std::memcpy(d->words.data(), bytes, (size_t)nbytes);
    d->x = (uint64_t)d->words[0] | ((uint64_t)d->words[1] << 32);
    d->pos = 2;
    d->ready = true;
    return 0;
}

namespace {
inline uint32_t next_word(hesic_rans_decoder* d) { return d->pos < d->words.size() ? d->words[d->pos++] : 0u; }
inline uint32_t get_bits(hesic_rans_decoder* d, uint32_t nbits) {
    const uint32_t v = (uint32_t)(d->x & ((1u << nbits) - 1));
    d->x >>= nbits;
    if (d->x < kLow) d->x = (d->x << 32) | next_word(d);
    return v;
}
}  // namespace

extern "C" int hesic_rans_decoder_decode(hesic_rans_decoder* d, const int32_t* indexes, int64_t n, const int32_t* cdfs, int ncdf,
                                         int stride, const int32_t* sizes, const int32_t* offsets, int32_t* out) {
    if (!d || !d->ready || !indexes || !cdfs || !sizes || !offsets || !out || n < 0) return -1;
    const uint32_t mask = (1u << kPrecision) - 1;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t ci = indexes[i];
        if (ci < 0 || ci >= ncdf || sizes[ci] < 2 || sizes[ci] > stride) return -1;
        const int32_t* cdf = cdfs + (int64_t)ci * stride;
        const int32_t last = sizes[ci] - 2;
        const uint32_t slot = (uint32_t)(d->x & mask);
        int32_t s = 0;                                 // largest s with cdf[s] <= slot
        while (s + 1 < sizes[ci] && (uint32_t)cdf[s + 1] <= slot) ++s;
        const uint32_t start = (uint32_t)cdf[s], freq = (uint32_t)(cdf[s + 1] - cdf[s]);
        d->x = (uint64_t)freq * (d->x >> kPrecision) + slot - start;
        if (d->x < kLow) d->x = (d->x << 32) | next_word(d);
        int32_t v = s;
        if (s == last) {
            int32_t c = (int32_t)get_bits(d, kBypassBits), nib = c;
            while (c == (int32_t)kBypassMax) { c = (int32_t)get_bits(d, kBypassBits); nib += c; }
            uint32_t raw = 0;
            for (int32_t j = 0; j < nib; ++j) raw |= get_bits(d, kBypassBits) << (j * kBypassBits);
            v = (int32_t)(raw >> 1);
            v = (raw & 1) ? -v - 1 : v + last;
        }
        out[i] = v + offsets[ci];
    }
    return 0;
}


// ------------------------------------------------------------------ adaptive range coder (HSIC.compress .bin payload)
// Carry-less range coder: low is 64 bits wide, a byte is emitted while the top bytes of low and low+range agree, and an
// underflowing range is clamped to the distance to the next 2^48 boundary (the classic TOP/BOTTOM scheme).
namespace {
constexpr uint64_t RC_TOP = 1ull << 56, RC_BOT = 1ull << 48;
}
struct hesic_rc_encoder {
    uint64_t low = 0, range = ~0ull;
    std::vector<uint8_t> out;
    bool finished = false;
};
struct hesic_rc_decoder {
    uint64_t low = 0, range = ~0ull, code = 0;
    std::vector<uint8_t> in;
    size_t pos = 0;
    uint8_t next() { return pos < in.size() ? in[pos++] : 0; }
};

extern "C" hesic_rc_encoder* hesic_rc_encoder_new(void) { return new hesic_rc_encoder(); }
extern "C" void hesic_rc_encoder_free(hesic_rc_encoder* e) { delete e; }

extern "C" int hesic_rc_encoder_encode(hesic_rc_encoder* e, const int32_t* symbols, const uint32_t* cdf, int64_t n, int32_t stride) {
    if (!e || e->finished || !symbols || !cdf || n < 0 || stride < 2) return -1;
    for (int64_t i = 0; i < n; ++i) {
        const uint32_t* c = cdf + i * stride;
        const int32_t s = symbols[i];
        const uint64_t tot = c[stride - 1];
        if (s < 0 || s >= stride - 1 || c[s + 1] <= c[s] || tot == 0 || tot >= RC_BOT) return -2;
        e->range /= tot;
        e->low += (uint64_t)c[s] * e->range;
        e->range *= (uint64_t)(c[s + 1] - c[s]);
        while ((e->low ^ (e->low + e->range)) < RC_TOP || (e->range < RC_BOT && ((e->range = (0 - e->low) & (RC_BOT - 1)), true))) {
            e->out.push_back((uint8_t)(e->low >> 56));
            e->low <<= 8;
            e->range <<= 8;
        }
    }
    return 0;
}

extern "C" int64_t hesic_rc_encoder_finish(hesic_rc_encoder* e, uint8_t* out, int64_t cap) {
    if (!e) return -1;
    if (!e->finished) {
        for (int i = 0; i < 8; ++i) { e->out.push_back((uint8_t)(e->low >> 56)); e->low <<= 8; }
        e->finished = true;
    }
    const int64_t nb = (int64_t)e->out.size();
    if (out && cap >= nb) memcpy(out, e->out.data(), (size_t)nb);
    return nb;
}

extern "C" hesic_rc_decoder* hesic_rc_decoder_new(const uint8_t* bytes, int64_t nbytes) {
    if (!bytes || nbytes < 0) return nullptr;
    hesic_rc_decoder* d = new hesic_rc_decoder();
    d->in.assign(bytes, bytes + nbytes);
    for (int i = 0; i < 8; ++i) d->code = (d->code << 8) | d->next();
    return d;
}
extern "C" void hesic_rc_decoder_free(hesic_rc_decoder* d) { delete d; }

// Symbol (p, q), p outer, decoded under table row p * row_step_outer + q * row_step_inner: lets a decoder walk tables that lie in
// another order than the stream (the HESIC+ wavefront decode gets them channel-major from the device and codes pixel-major)
// without a host-side transpose.  hesic_rc_decoder_decode is the (n, 1) case.
extern "C" int hesic_rc_decoder_decode_grid(hesic_rc_decoder* d, const uint32_t* cdf, int64_t n_outer, int64_t n_inner, int64_t row_step_outer,
                                            int64_t row_step_inner, int32_t stride, int32_t* symbols_out) {
    if (!d || !cdf || !symbols_out || n_outer < 0 || n_inner < 0 || stride < 2) return -1;
    // The walk over the table rows is known in advance and the rows arrive cache-cold (a device -> host copy into pinned memory in front
    // of every call of the HESIC+ wavefront decode): the rows of the symbol two ahead are prefetched while this one is decoded (59 ->
    // ~25 ns per symbol on the GPU box's host); a table total of 2^16 -- what hesic_gmm_cdf normalises to -- turns the first division
    // into a shift.
    int64_t po = 0, qi = 0;
    auto row_of = [&](int64_t p, int64_t q) { return cdf + (p * row_step_outer + q * row_step_inner) * stride; };
    const int64_t n = n_outer * n_inner;
    const int lines = (stride * 4 + 63) / 64;
    constexpr int ahead = 4;
    for (int64_t i = 0; i < n; ++i) {
        const uint32_t* c = row_of(po, qi);
        {
            int64_t q2 = qi + ahead, p2 = po;
            while (q2 >= n_inner) { q2 -= n_inner; ++p2; }
            if (p2 < n_outer) {
                const char* nx = (const char*)row_of(p2, q2);
                for (int l = 0; l < lines; ++l) __builtin_prefetch(nx + 64 * l, 0, 1);
            }
        }
        const uint64_t tot = c[stride - 1];
        if (tot == 0 || tot >= RC_BOT) return -2;
        if (tot == 65536u) d->range >>= 16; else d->range /= tot;
        uint64_t v = (d->code - d->low) / d->range;
        if (v >= tot) v = tot - 1;
        // last table entry <= v (zero-frequency entries are skipped by taking the LAST one).  Short rows (the HESIC+ groups: ~20 - 130
        // entries) are counted branch-free -- c is non-decreasing with c[0] = 0 <= v < c[stride - 1], so the index is (number of entries
        // of c[0 .. stride-2] that are <= v) - 1, a loop the compiler vectorises -- instead of a binary search whose every step is a
        // mispredicted branch; long rows keep the search.
        int lo;
        if (stride <= 160) {
            const int32_t vi = (int32_t)v;
            int cnt = 0;
            for (int k = 0; k < stride - 1; ++k) cnt += ((int32_t)c[k] <= vi);
            lo = cnt - 1;
        } else {
            int hi = stride - 1;
            lo = 0;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (c[mid] <= v) lo = mid; else hi = mid;
            }
        }
        symbols_out[i] = lo;
        d->low += (uint64_t)c[lo] * d->range;
        d->range *= (uint64_t)(c[lo + 1] - c[lo]);
        while ((d->low ^ (d->low + d->range)) < RC_TOP || (d->range < RC_BOT && ((d->range = (0 - d->low) & (RC_BOT - 1)), true))) {
            d->code = (d->code << 8) | d->next();
            d->low <<= 8;
            d->range <<= 8;
        }
        if (++qi == n_inner) { qi = 0; ++po; }
    }
    return 0;
}

extern "C" int hesic_rc_decoder_decode(hesic_rc_decoder* d, const uint32_t* cdf, int64_t n, int32_t stride, int32_t* symbols_out) {
    return hesic_rc_decoder_decode_grid(d, cdf, n, 1, 1, 0, stride, symbols_out);
}
