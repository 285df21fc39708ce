// plsx_resample.h -- host-side (CPU) resampling index generators, draw-for-draw
// compatible with the reference's use of numpy's legacy RandomState
// (pyls/base.py:10-229, pyls/utils.py:200-224) so that a seed gives the very
// same permutation / bootstrap / split arrays.  Native because at 8 GPUs the
// device finishes 10 000 + 10 000 resamples in ~0.3 s while the Python loops
// need ~1 s (and ~30 s for the 10 000 x 100 split masks of the split-half leg):
// the serial host part would cap the strong-scaling curve.
//
// numpy algorithms restated (numpy/random/src/mt19937/mt19937.c,
// src/legacy/legacy-distributions.c, src/distributions/distributions.c; the
// build has no numpy sources -- behaviour is pinned by tests/test_resampling.py
// against numpy itself and against the reference's golden index arrays):
//   seeding        init_genrand (Knuth 1812433253) for 0 <= seed < 2^32
//   next_uint32    MT19937 with the standard tempering
//   random_sample  (a >> 5, b >> 6) -> (a * 2^26 + b) / 2^53
//   interval(max)  smallest mask 2^k - 1 >= max, redraw 32-bit values & mask until <= max
//   shuffle        for i = n-1 .. 1: swap(x[i], x[interval(i)])
//   choice(replace=True)  randint(0, n): masked rejection as interval(n - 1), one value each
//   choice(replace=False) permutation(n)[:size]
#pragma once
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <string.h>
#include <vector>

namespace plsx_rs {

struct MT {
    uint32_t key[624];
    int pos;
    void seed(uint32_t s)
    {
        for (int i = 0; i < 624; ++i) {
            key[i] = s;
            s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u;
        }
        pos = 624;
    }
    static inline uint32_t twist(uint32_t a, uint32_t b, uint32_t far_)
    {
        const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
        return far_ ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
    }
    void refill()
    {
        // three dependence-free segments (distance to the words they read >= 227), so the
        // compiler vectorises each
        for (int kk = 0; kk < 227; ++kk) key[kk] = twist(key[kk], key[kk + 1], key[kk + 397]);
        for (int kk = 227; kk < 454; ++kk) key[kk] = twist(key[kk], key[kk + 1], key[kk - 227]);
        for (int kk = 454; kk < 623; ++kk) key[kk] = twist(key[kk], key[kk + 1], key[kk - 227]);
        key[623] = twist(key[623], key[0], key[396]);
        pos = 0;
    }
    inline uint32_t next()
    {
        if (__builtin_expect(pos >= 624, 0)) refill();
        uint32_t y = key[pos++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    void skip(long n)
    {
        while (n > 0) {
            if (pos >= 624) refill();
            const long take = std::min<long>(n, 624 - pos);
            pos += (int)take;
            n -= take;
        }
    }
    double sample()
    {
        const uint32_t a = next() >> 5, b = next() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    // n draws of random_sample() into out: the tempering of 16 words (8 samples) at a time is free of the refill
    // test and vectorises (the per-sample form: two calls of next(), a branch each)
    void sample_block(double* out, int n)
    {
        constexpr int W = 16;
        int i = 0;
        while (i < n) {
            if (__builtin_expect(pos + W > 624 || n - i < W / 2, 0)) { out[i++] = sample(); continue; }
            uint32_t tv[W];
            const uint32_t* kp = key + pos;
            for (int u = 0; u < W; ++u) {
                uint32_t y = kp[u];
                y ^= (y >> 11);
                y ^= (y << 7) & 0x9d2c5680u;
                y ^= (y << 15) & 0xefc60000u;
                y ^= (y >> 18);
                tv[u] = y;
            }
            for (int u = 0; u < W / 2; ++u)
                out[i + u] = ((tv[2 * u] >> 5) * 67108864.0 + (tv[2 * u + 1] >> 6)) / 9007199254740992.0;
            pos += W;
            i += W / 2;
        }
    }
    static inline uint32_t mask_of(uint32_t max)
    {
        uint32_t mask = max;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        return mask;
    }
    uint32_t interval(uint32_t max)
    {
        if (max == 0) return 0;
        const uint32_t mask = mask_of(max);
        uint32_t v;
        while ((v = (next() & mask)) > max) {}
        return v;
    }
    void permutation(int* x, int n)
    {
        for (int i = 0; i < n; ++i) x[i] = i;
        if (n < 2) return;
        // Fisher-Yates with numpy's masked rejection, branch-free where the branch is a coin flip (a rejected
        // draw swaps x[i] with itself and does not advance: the accept test loses ~28 % of its predictions)
        // and branchy where it is not (the mask halves nine times per 500 elements: as a select it sat on
        // the loop-carried chain i -> mask -> accept -> i, 23 -> 14 ms per 10 000 x 500).  The tempering of 16
        // words runs ahead of that chain (it vectorises).
        constexpr int W = 16;
        uint32_t mask = mask_of((uint32_t)(n - 1));
        uint32_t i = (uint32_t)(n - 1);
        while (i >= 1) {
            if (__builtin_expect(pos + W > 624, 0)) {            // slow path near a refill
                const uint32_t v = next() & mask;
                if (v <= i) { const int t = x[i]; x[i] = x[v]; x[v] = t; --i; mask >>= (i <= (mask >> 1)); }
                continue;
            }
            uint32_t tv[W];
            const uint32_t* kp = key + pos;
            for (int u = 0; u < W; ++u) {
                uint32_t y = kp[u];
                y ^= (y >> 11);
                y ^= (y << 7) & 0x9d2c5680u;
                y ^= (y << 15) & 0xefc60000u;
                y ^= (y >> 18);
                tv[u] = y;
            }
            int u = 0;
            for (; u < W && i >= 1; ++u) {
                const uint32_t v = tv[u] & mask;
                const uint32_t acc = v <= i;
                const uint32_t j = acc ? v : i;
                const int t = x[i]; x[i] = x[j]; x[j] = t;
                i -= acc;
                if (__builtin_expect(i <= (mask >> 1), 0)) mask >>= 1;
            }
            pos += u;                                            // (only the words the shuffle consumed)
        }
    }
    void permutation(std::vector<int>& x, int n) { x.resize(n); permutation(x.data(), n); }
    // g draws of interval(gmax) (numpy's randint(0, gmax + 1): masked rejection, one value each), counted per
    // value: cnt[v] += 1.  Tempering 16 words ahead of the accept test, as in permutation().
    void draw_counts(uint32_t gmax, int g, int* cnt)
    {
        if (gmax == 0) { cnt[0] += g; return; }                   // interval(0) consumes nothing
        constexpr int W = 16;
        const uint32_t mask = mask_of(gmax);
        int s = 0;
        while (s < g) {
            if (__builtin_expect(pos + W > 624, 0)) {
                const uint32_t v = next() & mask;
                if (v <= gmax) { ++cnt[v]; ++s; }
                continue;
            }
            uint32_t tv[W];
            const uint32_t* kp = key + pos;
            for (int u = 0; u < W; ++u) {
                uint32_t y = kp[u];
                y ^= (y >> 11);
                y ^= (y << 7) & 0x9d2c5680u;
                y ^= (y << 15) & 0xefc60000u;
                y ^= (y >> 18);
                tv[u] = y & mask;
            }
            int u = 0;
            for (; u < W && s < g; ++u) {
                const uint32_t v = tv[u];
                if (v <= gmax) { ++cnt[v]; ++s; }
            }
            pos += u;
        }
    }
};

// Row bookkeeping (group-major, then condition, then subject; pyls/structures.py:37-44)
struct Design {
    std::vector<int> groups, g0;      // group sizes, first subject of each group
    int n_cond = 1, n_subj = 0, n_rows = 0;
    std::vector<int> rows;            // [n_cond][n_subj] row index of (condition, subject)
    Design(const int* gr, int ng, int nc) : groups(gr, gr + ng), n_cond(nc)
    {
        for (int g : groups) { g0.push_back(n_subj); n_subj += g; }
        rows.assign((size_t)n_cond * n_subj, 0);
        int row0 = 0;
        for (size_t gi = 0; gi < groups.size(); ++gi) {
            const int g = groups[gi], s0 = g0[gi];
            for (int c = 0; c < n_cond; ++c)
                for (int s = 0; s < g; ++s) rows[(size_t)c * n_subj + s0 + s] = row0 + c * g + s;
            row0 += g * n_cond;
        }
        n_rows = row0;
    }
    // (n_cond, n_subj) table -> flat per-row vector in canonical order
    template <class T, class U>
    void expand(const std::vector<T>& table, U* out) const
    {
        size_t k = 0;
        for (size_t gi = 0; gi < groups.size(); ++gi)
            for (int c = 0; c < n_cond; ++c)
                for (int s = 0; s < groups[gi]; ++s) out[k++] = (U)table[(size_t)c * n_subj + g0[gi] + s];
    }
};

// Exact duplicate detection among the rows already written (the reference compares every
// earlier column element-wise, base.py:67-69): 64-bit hash of the key bytes in an
// open-addressing table sized once for the whole run, byte comparison on a hash hit.
inline uint64_t hash_bytes(const void* p, size_t bytes)
{
    const unsigned char* c = (const unsigned char*)p;
    uint64_t h0 = 0x9e3779b97f4a7c15ull ^ bytes, h1 = 0xc2b2ae3d27d4eb4full;
    size_t i = 0;
    for (; i + 16 <= bytes; i += 16) {                 // two independent lanes: the multiplies overlap
        uint64_t w0, w1;
        memcpy(&w0, c + i, 8);
        memcpy(&w1, c + i + 8, 8);
        h0 = (h0 ^ w0) * 0xff51afd7ed558ccdull;
        h1 = (h1 ^ w1) * 0x9fb21c651e98df25ull;
        h0 ^= h0 >> 32;
        h1 ^= h1 >> 29;
    }
    for (; i < bytes; ++i) h0 = (h0 ^ c[i]) * 0x100000001b3ull;
    uint64_t h = (h0 ^ (h1 * 0x9e3779b97f4a7c15ull));
    h ^= h >> 31;
    return h;
}

struct SeenRows {
    std::vector<int32_t> slot;                        // row number + 1, 0 = empty
    std::vector<uint64_t> hashes;                     // hash of every added row
    uint64_t mask;
    const unsigned char* base;                        // first byte of row 0's key
    size_t pitch, bytes;                              // bytes between rows, key length
    SeenRows(const void* b, size_t p, size_t n, int max_rows)
        : base((const unsigned char*)b), pitch(p), bytes(n)
    {
        size_t cap = 16;
        while (cap < 2 * (size_t)std::max(max_rows, 1)) cap <<= 1;
        slot.assign(cap, 0);
        hashes.assign((size_t)std::max(max_rows, 1), 0);
        mask = cap - 1;
    }
    bool contains(const void* key, uint64_t h) const
    {
        for (uint64_t s = h & mask; slot[s]; s = (s + 1) & mask) {
            const int row = slot[s] - 1;
            if (hashes[row] == h && memcmp(base + (size_t)row * pitch, key, bytes) == 0) return true;
        }
        return false;
    }
    void add(int row, uint64_t h)
    {
        hashes[row] = h;
        uint64_t s = h & mask;
        while (slot[s]) s = (s + 1) & mask;
        slot[s] = row + 1;
    }
};

// Rows [0, *progress) of `out` are final: a consumer on another thread may ship them to
// the device while later rows are still being drawn (the duplicate test only ever looks
// backwards, and the RNG stream is consumed in the reference's order either way).
typedef int Progress;                                 // plain int in the caller's memory (C ABI)
inline void publish(Progress* p, int rows_done)
{
    if (p) __atomic_store_n(p, rows_done, __ATOMIC_RELEASE);
}

// pyls/base.py:10-79.  out: (n_perm, S) int32, one permutation per row.
inline int gen_permsamp(const Design& d, int n_perm, MT& rs, int32_t* out, Progress* progress = nullptr)
{
    const int S = d.n_rows, ns = d.n_subj, nc = d.n_cond;
    const int ng = (int)d.groups.size();
    SeenRows seen(out, (size_t)S * 4, (size_t)S * 4, n_perm);
    std::vector<int> shuffled(d.rows), perm(ns), ord(nc);
    std::vector<double> u((size_t)nc * ns);
    int warned = 0;
    for (int i = 0; i < n_perm; ++i) {
        int count = 0;
        bool dup = true;
        int32_t* row = out + (size_t)i * S;
        uint64_t h = 0;
        while (dup && count < 500) {
            ++count;
            dup = false;
            // conditions shuffled within subject: random_sample((n_cond, n_g)) per group, argsort over
            // conditions.  With ONE condition the argsort is the identity whatever was drawn: the
            // 2 n_g words of the stream are stepped over, not tempered and converted.
            for (int gi = 0; gi < ng; ++gi) {
                const int g = d.groups[gi], s0 = d.g0[gi];
                if (nc == 1) { rs.skip(2L * g); continue; }
                rs.sample_block(u.data(), nc * g);          // (row-major (n_cond, n_g), as random_sample fills it)
                for (int s = 0; s < g; ++s) {
                    for (int c = 0; c < nc; ++c) {               // insertion sort of the conditions (argsort)
                        int k = c;
                        while (k > 0 && u[(size_t)ord[k - 1] * g + s] > u[(size_t)c * g + s]) { ord[k] = ord[k - 1]; --k; }
                        ord[k] = c;
                    }
                    for (int c = 0; c < nc; ++c)
                        shuffled[(size_t)c * ns + s0 + s] = d.rows[(size_t)ord[c] * ns + s0 + s];
                }
            }
            rs.permutation(perm.data(), ns);
            if (ng > 1)
                for (int gi = 0; gi < ng; ++gi) {
                    const int a = d.g0[gi], b = a + d.groups[gi];
                    bool inside = true;
                    for (int s = a; s < b && inside; ++s) inside = perm[s] >= a && perm[s] < b;
                    if (inside) dup = true;
                }
            // shuffled[:, perm] in canonical row order (group, condition, subject)
            size_t k = 0;
            for (int gi = 0; gi < ng; ++gi)
                for (int c = 0; c < nc; ++c) {
                    const int* sh = shuffled.data() + (size_t)c * ns;
                    const int* pp = perm.data() + d.g0[gi];
                    for (int s = 0; s < d.groups[gi]; ++s) row[k++] = sh[pp[s]];
                }
            h = hash_bytes(row, (size_t)S * 4);
            if (seen.contains(row, h)) dup = true;
        }
        if (count == 500) warned = 1;
        seen.add(i, h);
        publish(progress, i + 1);
    }
    return warned;
}

// pyls/base.py:82-159.  out: (n_boot, S) int32.
inline int gen_bootsamp(const Design& d, int n_boot, MT& rs, int32_t* out, Progress* progress = nullptr)
{
    const int S = d.n_rows, ns = d.n_subj, nc = d.n_cond;
    const int ng = (int)d.groups.size();
    const int gmin = *std::min_element(d.groups.begin(), d.groups.end());
    const int min_subj = (int)std::ceil(gmin * 0.5);
    std::vector<SeenRows> seen;
    for (int gi = 0; gi < ng; ++gi)
        seen.emplace_back(out + d.g0[gi], (size_t)S * 4, (size_t)d.groups[gi] * 4, n_boot);
    std::vector<int> boot(ns + 4), cnt;
    std::vector<uint64_t> h(ng);
    int warned = 0;
    for (int i = 0; i < n_boot; ++i) {
        int count = 0;
        bool dup = true;
        int32_t* row = out + (size_t)i * S;
        while (dup && count < 500) {
            ++count;
            dup = false;
            for (int gi = 0; gi < ng; ++gi) {
                const int a = d.g0[gi], g = d.groups[gi];
                const uint32_t gmax = (uint32_t)(g - 1);
                for (;;) {
                    // draws are subject numbers of the group: counting sort (= np.sort), distinct count
                    cnt.assign(g, 0);
                    rs.draw_counts(gmax, g, cnt.data());
                    // expansion without a data-dependent inner loop (its trip count 0 / 1 / 2 / ... is a
                    // misprediction per subject): four unconditional stores, the next subject
                    // overwrites what ran past this one's count
                    int uniq = 0, k = a;
                    for (int v = 0; v < g; ++v) {
                        const int c = cnt[v], val = a + v;
                        uniq += c != 0;
                        boot[k] = val; boot[k + 1] = val; boot[k + 2] = val; boot[k + 3] = val;
                        if (__builtin_expect(c > 4, 0))
                            for (int q = 4; q < c; ++q) boot[k + q] = val;
                        k += c;
                    }
                    if (uniq >= min_subj) break;
                }
            }
            // rows[:, boot] in canonical row order
            size_t k = 0;
            for (int gi = 0; gi < ng; ++gi)
                for (int c = 0; c < nc; ++c) {
                    const int* rr = d.rows.data() + (size_t)c * ns;
                    const int* bb = boot.data() + d.g0[gi];
                    for (int s = 0; s < d.groups[gi]; ++s) row[k++] = rr[bb[s]];
                }
            // the reference compares positions [a, b) of the FLAT row vector, subject
            // numbers used as row positions (base.py:145-149)
            for (int gi = 0; gi < ng; ++gi) {
                h[gi] = hash_bytes(row + d.g0[gi], (size_t)d.groups[gi] * 4);
                if (seen[gi].contains(row + d.g0[gi], h[gi])) dup = true;
            }
        }
        if (count == 500) warned = 1;
        for (int gi = 0; gi < ng; ++gi) seen[gi].add(i, h[gi]);
        publish(progress, i + 1);
    }
    return warned;
}

// pyls/base.py:162-229.  out: (n_split, S) uint8, 1 = first half / training row.
inline int gen_splits(const Design& d, int n_split, double test_size, MT& rs, uint8_t* out)
{
    const int S = d.n_rows, ns = d.n_subj, nc = d.n_cond;
    SeenRows seen(out, (size_t)S, (size_t)S, n_split);
    std::vector<uint8_t> split(ns), table((size_t)nc * ns);
    std::vector<int> perm;
    int warned = 0;
    for (int i = 0; i < n_split; ++i) {
        int count = 0;
        bool dup = true;
        uint8_t* row = out + (size_t)i * S;
        uint64_t h = 0;
        while (dup && count < 500) {
            ++count;
            dup = false;
            std::fill(split.begin(), split.end(), 0);
            for (size_t gi = 0; gi < d.groups.size(); ++gi) {
                const int a = d.g0[gi], g = d.groups[gi];
                const uint32_t which = rs.interval(1);                    // choice([ceil, floor])
                const double want = g * (1.0 - test_size);
                const int num = (int)(which == 0 ? std::ceil(want) : std::floor(want));
                rs.permutation(perm, g);                                   // choice(replace=False)
                for (int s = 0; s < num; ++s) split[a + perm[s]] = 1;
            }
            for (int c = 0; c < nc; ++c)
                for (int s = 0; s < ns; ++s) table[(size_t)c * ns + s] = split[s];
            d.expand(table, row);
            h = hash_bytes(row, (size_t)S);
            if (seen.contains(row, h)) dup = true;
        }
        if (count == 500) warned = 1;
        seen.add(i, h);
    }
    return warned;
}

}  // namespace plsx_rs
