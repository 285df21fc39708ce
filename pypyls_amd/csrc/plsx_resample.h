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
#include <unordered_map>
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
    void refill()
    {
        const uint32_t UP = 0x80000000u, LO = 0x7fffffffu, MA = 0x9908b0dfu;
        int kk = 0;
        uint32_t y;
        for (; kk < 624 - 397; ++kk) {
            y = (key[kk] & UP) | (key[kk + 1] & LO);
            key[kk] = key[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
        }
        for (; kk < 623; ++kk) {
            y = (key[kk] & UP) | (key[kk + 1] & LO);
            key[kk] = key[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
        }
        y = (key[623] & UP) | (key[0] & LO);
        key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
        pos = 0;
    }
    uint32_t next()
    {
        if (pos >= 624) refill();
        uint32_t y = key[pos++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    double sample()
    {
        const uint32_t a = next() >> 5, b = next() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    uint32_t interval(uint32_t max)
    {
        if (max == 0) return 0;
        uint32_t mask = max;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        uint32_t v;
        while ((v = (next() & mask)) > max) {}
        return v;
    }
    // x = permutation of 0..n-1 (numpy permutation(n) / shuffle(arange(n)))
    void permutation(std::vector<int>& x, int n)
    {
        x.resize(n);
        for (int i = 0; i < n; ++i) x[i] = i;
        for (int i = n - 1; i >= 1; --i) std::swap(x[i], x[interval((uint32_t)i)]);
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

// Exact duplicate detection among the rows already written: 64-bit hash -> earlier row
// numbers, byte comparison on a hash hit (the reference compares every earlier
// column element-wise, base.py:67-69).
inline uint64_t hash_bytes(const void* p, size_t bytes)
{
    const unsigned char* c = (const unsigned char*)p;
    uint64_t h = 0x9e3779b97f4a7c15ull ^ bytes;
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) {
        uint64_t w;
        memcpy(&w, c + i, 8);
        h = (h ^ w) * 0xff51afd7ed558ccdull;
        h ^= h >> 32;
    }
    for (; i < bytes; ++i) h = (h ^ c[i]) * 0x100000001b3ull;
    return h;
}

struct SeenRows {
    std::unordered_multimap<uint64_t, int> idx;      // hash -> row number
    const unsigned char* base;                        // first byte of row 0's key
    size_t pitch, bytes;                              // bytes between rows, key length
    SeenRows(const void* b, size_t p, size_t n) : base((const unsigned char*)b), pitch(p), bytes(n) {}
    bool contains(const void* key) const
    {
        auto range = idx.equal_range(hash_bytes(key, bytes));
        for (auto it = range.first; it != range.second; ++it)
            if (memcmp(base + (size_t)it->second * pitch, key, bytes) == 0) return true;
        return false;
    }
    void add(int row) { idx.emplace(hash_bytes(base + (size_t)row * pitch, bytes), row); }
};

// pyls/base.py:10-79.  out: (n_perm, S) int32, one permutation per row.
inline int gen_permsamp(const Design& d, int n_perm, MT& rs, int32_t* out)
{
    const int S = d.n_rows, ns = d.n_subj, nc = d.n_cond;
    SeenRows seen(out, (size_t)S * 4, (size_t)S * 4);
    std::vector<int> shuffled((size_t)nc * ns), picked((size_t)nc * ns), perm, ord(nc);
    std::vector<double> u((size_t)nc * ns);
    int warned = 0;
    for (int i = 0; i < n_perm; ++i) {
        int count = 0;
        bool dup = true;
        int32_t* row = out + (size_t)i * S;
        while (dup && count < 500) {
            ++count;
            dup = false;
            // conditions shuffled within subject: random_sample((n_cond, n_g)) per group, argsort over conditions
            for (size_t gi = 0; gi < d.groups.size(); ++gi) {
                const int g = d.groups[gi], s0 = d.g0[gi];
                for (int c = 0; c < nc; ++c)
                    for (int s = 0; s < g; ++s) u[(size_t)c * g + s] = rs.sample();
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
            rs.permutation(perm, ns);
            if (d.groups.size() > 1)
                for (size_t gi = 0; gi < d.groups.size(); ++gi) {
                    const int a = d.g0[gi], b = a + d.groups[gi];
                    bool inside = true;
                    for (int s = a; s < b && inside; ++s) inside = perm[s] >= a && perm[s] < b;
                    if (inside) dup = true;
                }
            for (int c = 0; c < nc; ++c)
                for (int s = 0; s < ns; ++s) picked[(size_t)c * ns + s] = shuffled[(size_t)c * ns + perm[s]];
            d.expand(picked, row);
            if (seen.contains(row)) dup = true;
        }
        if (count == 500) warned = 1;
        seen.add(i);
    }
    return warned;
}

// pyls/base.py:82-159.  out: (n_boot, S) int32.
inline int gen_bootsamp(const Design& d, int n_boot, MT& rs, int32_t* out)
{
    const int S = d.n_rows, ns = d.n_subj, nc = d.n_cond;
    const int gmin = *std::min_element(d.groups.begin(), d.groups.end());
    const int min_subj = (int)std::ceil(gmin * 0.5);
    std::vector<SeenRows> seen;
    for (size_t gi = 0; gi < d.groups.size(); ++gi)
        seen.emplace_back(out + d.g0[gi], (size_t)S * 4, (size_t)d.groups[gi] * 4);
    std::vector<int> boot(ns), table((size_t)nc * ns), cnt;
    int warned = 0;
    for (int i = 0; i < n_boot; ++i) {
        int count = 0;
        bool dup = true;
        int32_t* row = out + (size_t)i * S;
        while (dup && count < 500) {
            ++count;
            dup = false;
            for (size_t gi = 0; gi < d.groups.size(); ++gi) {
                const int a = d.g0[gi], g = d.groups[gi];
                for (;;) {
                    // draws are subject numbers of the group: counting sort (= np.sort), distinct count
                    cnt.assign(g, 0);
                    for (int s = 0; s < g; ++s) ++cnt[rs.interval((uint32_t)(g - 1))];
                    int uniq = 0, k = a;
                    for (int v = 0; v < g; ++v) {
                        uniq += cnt[v] != 0;
                        for (int c = 0; c < cnt[v]; ++c) boot[k++] = a + v;
                    }
                    if (uniq >= min_subj) break;
                }
            }
            for (int c = 0; c < nc; ++c)
                for (int s = 0; s < ns; ++s) table[(size_t)c * ns + s] = d.rows[(size_t)c * ns + boot[s]];
            d.expand(table, row);
            // the reference compares positions [a, b) of the FLAT row vector, subject
            // numbers used as row positions (base.py:145-149)
            for (size_t gi = 0; gi < d.groups.size(); ++gi)
                if (seen[gi].contains(row + d.g0[gi])) dup = true;
        }
        if (count == 500) warned = 1;
        for (size_t gi = 0; gi < d.groups.size(); ++gi) seen[gi].add(i);
    }
    return warned;
}

// pyls/base.py:162-229.  out: (n_split, S) uint8, 1 = first half / training row.
inline int gen_splits(const Design& d, int n_split, double test_size, MT& rs, uint8_t* out)
{
    const int S = d.n_rows, ns = d.n_subj, nc = d.n_cond;
    SeenRows seen(out, (size_t)S, (size_t)S);
    std::vector<uint8_t> split(ns), table((size_t)nc * ns);
    std::vector<int> perm;
    int warned = 0;
    for (int i = 0; i < n_split; ++i) {
        int count = 0;
        bool dup = true;
        uint8_t* row = out + (size_t)i * S;
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
            if (seen.contains(row)) dup = true;
        }
        if (count == 500) warned = 1;
        seen.add(i);
    }
    return warned;
}

}  // namespace plsx_rs
