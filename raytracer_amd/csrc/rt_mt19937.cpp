// rt_mt19937.cpp -- MT19937 host utilities: seeding, block advance, GF(2) jump-ahead.
// See rt_mt19937.hpp for the state convention.  Reference behaviour being reproduced:
// `static std::mt19937 generator;` cpp/rtmathutils.h:16 (seed 5489, one global stream).
#include "rt_mt19937.hpp"

#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>
#include <stdexcept>
#include <thread>

namespace rt {
namespace {

constexpr uint32_t kUpper = 0x80000000u, kLower = 0x7fffffffu, kMatrixA = 0x9908b0dfu;
constexpr int kDeg = 19937;
constexpr int kPolyWords = 312;  // 19968 bits: holds phi (bit 19937) and any residue

inline uint32_t twist(uint32_t a, uint32_t b, uint32_t m) {
    uint32_t y = (a & kUpper) | (b & kLower);
    return m ^ (y >> 1) ^ ((y & 1u) ? kMatrixA : 0u);
}

struct Phi {
    uint64_t b[kPolyWords];
};

// Characteristic polynomial of the one-word transition: Berlekamp-Massey over GF(2) on
// bit 0 of the raw word sequence (irreducible of degree 19937, so any non-zero output
// sequence has it as minimal polynomial).
Phi compute_phi() {
    const int N = 2 * kDeg + 128;
    std::vector<uint8_t> s(N);
    MtState st;
    mt_seed(st, 5489u);
    for (int n = 0; n < N;) {
        mt_next_block(st);
        for (int k = 0; k < MT_N && n < N; ++k, ++n) s[n] = st.w[k] & 1u;
    }
    const int WB = (N + 64) / 64 + 1;
    std::vector<uint64_t> C(WB, 0), B(WB, 0), T(WB, 0), R(WB, 0);
    C[0] = B[0] = 1;
    int L = 0, m = 1;
    auto xor_shifted = [&](std::vector<uint64_t>& dst, const std::vector<uint64_t>& src, int sh) {
        const int ws = sh / 64, bs = sh % 64;
        for (int k = WB - 1 - ws; k >= 0; --k) {
            dst[k + ws] ^= src[k] << bs;
            if (bs && k + ws + 1 < WB) dst[k + ws + 1] ^= src[k] >> (64 - bs);
        }
    };
    for (int n = 0; n < N; ++n) {
        // R bit i = s[n - i]
        const int used = std::min(WB, n / 64 + 2);
        for (int k = used - 1; k > 0; --k) R[k] = (R[k] << 1) | (R[k - 1] >> 63);
        R[0] = (R[0] << 1) | s[n];
        uint64_t acc = 0;
        const int lw = L / 64 + 1;
        for (int k = 0; k < lw; ++k) acc ^= C[k] & R[k];
        const int d = __builtin_parityll(acc);
        if (!d) {
            ++m;
        } else if (2 * L <= n) {
            T = C;
            xor_shifted(C, B, m);
            L = n + 1 - L;
            B = T;
            m = 1;
        } else {
            xor_shifted(C, B, m);
            ++m;
        }
    }
    if (L != kDeg) throw std::runtime_error("MT19937 characteristic polynomial: unexpected degree");
    Phi phi;
    std::memset(phi.b, 0, sizeof(phi.b));
    for (int i = 0; i <= L; ++i)
        if ((C[i / 64] >> (i % 64)) & 1u) {
            const int e = L - i;
            phi.b[e / 64] |= uint64_t(1) << (e % 64);
        }
    return phi;
}

// phi is a constant of the generator: its 19938 coefficients are compiled in (rt_mt19937_phi.inc, written by
// profiles/tools/gen_mt_phi.py from compute_phi() above; tests/test_mt_jump.py checks the table against a fresh
// Berlekamp-Massey run).  Computing it took 9 ms (GPU box) to 30 ms of the first frame of every process.
const Phi kPhiTable = {{
#include "rt_mt19937_phi.inc"
}};

const Phi& phi() { return kPhiTable; }

inline bool poly_bit(const uint64_t* p, int i) { return (p[i / 64] >> (i % 64)) & 1u; }

// r = r * x mod phi
void mulx_mod(MtPoly& r) {
    uint64_t carry = 0;
    for (int k = 0; k < kPolyWords; ++k) {
        const uint64_t nc = r.b[k] >> 63;
        r.b[k] = (r.b[k] << 1) | carry;
        carry = nc;
    }
    if (poly_bit(r.b, kDeg)) {
        const Phi& f = phi();
        for (int k = 0; k < kPolyWords; ++k) r.b[k] ^= f.b[k];
    }
}

// r = r^2 mod phi
void sqr_mod(MtPoly& r) {
    static uint16_t spread[256];
    static std::once_flag once;
    std::call_once(once, [] {
        for (int v = 0; v < 256; ++v) {
            uint16_t o = 0;
            for (int b = 0; b < 8; ++b)
                if (v & (1 << b)) o |= uint16_t(1) << (2 * b);
            spread[v] = o;
        }
    });
    uint64_t t[2 * kPolyWords + 1];
    std::memset(t, 0, sizeof(t));
    for (int k = 0; k < kPolyWords; ++k) {
        const uint64_t v = r.b[k];
        uint64_t lo = 0, hi = 0;
        for (int byte = 0; byte < 4; ++byte) {
            lo |= uint64_t(spread[(v >> (8 * byte)) & 0xff]) << (16 * byte);
            hi |= uint64_t(spread[(v >> (32 + 8 * byte)) & 0xff]) << (16 * byte);
        }
        t[2 * k] = lo;
        t[2 * k + 1] = hi;
    }
    // Reduction mod phi, eight bits at a time from the top: red[v] is the multiple q(x) phi(x), deg q < 8, whose
    // bits kDeg .. kDeg+7 spell v, so XORing it in at the right shift clears a whole byte above the degree.
    constexpr int kRedWords = kPolyWords + 1;  // kDeg + 8 bits
    static std::vector<uint64_t> red;
    static std::once_flag red_once;
    std::call_once(red_once, [] {
        const Phi& f = phi();
        red.assign(size_t(256) * kRedWords, 0);
        for (int v = 0; v < 256; ++v) {
            uint64_t* q = red.data() + size_t(v) * kRedWords;
            for (int b = 7; b >= 0; --b) {
                const bool have = poly_bit(q, kDeg + b);
                if (have == (((v >> b) & 1) != 0)) continue;
                for (int k = 0; k < kPolyWords; ++k) {  // q ^= phi << b  (phi has kDeg + 1 bits, kPolyWords words)
                    q[k] ^= f.b[k] << b;
                    if (b) q[k + 1] ^= f.b[k] >> (64 - b);
                }
            }
        }
    });
    const int top = kDeg + ((2 * (kDeg - 1) - kDeg) / 8) * 8;  // lowest bit of the highest byte above the degree
    for (int i = top; i >= kDeg; i -= 8) {
        const int w = i / 64, o = i % 64;
        unsigned v = unsigned(t[w] >> o);
        if (o > 56) v |= unsigned(t[w + 1] << (64 - o));
        v &= 0xffu;
        if (!v) continue;
        const uint64_t* q = red.data() + size_t(v) * kRedWords;
        const int sh = i - kDeg, ws = sh / 64, bs = sh % 64;
        if (bs == 0) {
            for (int k = 0; k < kRedWords; ++k) t[k + ws] ^= q[k];
        } else {
            uint64_t carry = 0;
            for (int k = 0; k < kRedWords; ++k) {
                t[k + ws] ^= (q[k] << bs) | carry;
                carry = q[k] >> (64 - bs);
            }
            t[kRedWords + ws] ^= carry;
        }
    }
    std::memcpy(r.b, t, sizeof(r.b));
}

std::mutex g_cache_mu;
std::map<uint64_t, MtPoly> g_cache;

}  // namespace

void mt_phi_words(uint64_t out[312], bool recompute) {
    const Phi p = recompute ? compute_phi() : phi();
    std::memcpy(out, p.b, sizeof(p.b));
}

void mt_seed(MtState& s, uint32_t seed) {
    s.w[0] = seed;
    for (int i = 1; i < MT_N; ++i) s.w[i] = 1812433253u * (s.w[i - 1] ^ (s.w[i - 1] >> 30)) + uint32_t(i);
}

void mt_next_block(MtState& s) {
    uint32_t* w = s.w;
    for (int k = 0; k < MT_N - 397; ++k) w[k] = twist(w[k], w[k + 1], w[k + 397]);
    for (int k = MT_N - 397; k < MT_N - 1; ++k) w[k] = twist(w[k], w[k + 1], w[k - (MT_N - 397)]);
    w[MT_N - 1] = twist(w[MT_N - 1], w[0], w[396]);
}

constexpr size_t kMaxCachedPolys = 4096;  // 10 MB

MtPoly mt_jump_poly(uint64_t stride) {
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        auto it = g_cache.find(stride);
        if (it != g_cache.end()) return it->second;
    }
    MtPoly r;
    std::memset(r.b, 0, sizeof(r.b));
    // a cached half gives the answer with one squaring
    bool done = false;
    if (stride && (stride & 1) == 0) {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        auto it = g_cache.find(stride >> 1);
        if (it != g_cache.end()) {
            r = it->second;
            done = true;
        }
    }
    if (done) {
        sqr_mod(r);
    } else {
        r.b[0] = 1;
        int top = 63;
        while (top >= 0 && !((stride >> top) & 1)) --top;
        for (int bit = top; bit >= 0; --bit) {
            sqr_mod(r);
            if ((stride >> bit) & 1) mulx_mod(r);
        }
    }
    std::lock_guard<std::mutex> lk(g_cache_mu);
    // results are returned by value (2.5 KB), so a full cache just stops growing: callers that hold several
    // polynomials at once (mt_stream_states) can never see one overwritten by the next call
    if (g_cache.size() < kMaxCachedPolys) g_cache.emplace(stride, r);
    return r;
}

void mt_apply(const MtPoly& g, const MtState& in, MtState& out) {
    int deg = kDeg - 1;
    while (deg >= 0 && !poly_bit(g.b, deg)) --deg;
    if (deg < 0) throw std::runtime_error("mt_apply: zero polynomial");
    uint32_t s[MT_N], h[MT_N];
    std::memcpy(s, in.w, sizeof(s));
    std::memcpy(h, s, sizeof(h));  // leading coefficient
    int p = 0;                     // start of h's window inside the circular buffer
    for (int i = deg - 1; i >= 0; --i) {
        const int p1 = (p + 1 == MT_N) ? 0 : p + 1;
        const int pm = (p + 397 >= MT_N) ? p + 397 - MT_N : p + 397;
        h[p] = twist(h[p], h[p1], h[pm]);
        p = p1;
        if (poly_bit(g.b, i)) {
            const int head = MT_N - p;
            for (int j = 0; j < head; ++j) h[p + j] ^= s[j];
            for (int j = head; j < MT_N; ++j) h[j - head] ^= s[j];
        }
    }
    for (int j = 0; j < MT_N; ++j) out.w[j] = h[(p + j) % MT_N];
}

void mt_stream_states(const MtState& base, uint64_t stride, int n, int n_threads, std::vector<MtState>& out) {
    out.resize(static_cast<size_t>(std::max(n, 1)));
    out[0] = base;
    if (n <= 1) return;
    int levels = 0;
    while ((1 << levels) < n) ++levels;
    std::vector<MtPoly> polys(static_cast<size_t>(levels));
    for (int k = 0; k < levels; ++k) polys[k] = mt_jump_poly(stride << k);  // ascending: reuses squarings
    for (int k = levels - 1; k >= 0; --k) {
        const int step = 1 << k;
        std::vector<int> srcs;
        for (int r = 0; r + step < n; r += 2 * step) srcs.push_back(r);
        const int nt = std::max(1, std::min<int>(n_threads, static_cast<int>(srcs.size())));
        if (nt == 1) {
            for (int r : srcs) mt_apply(polys[k], out[r], out[r + step]);
        } else {
            std::vector<std::thread> pool;
            for (int w = 0; w < nt; ++w)
                pool.emplace_back([&, w]() {
                    for (size_t q = w; q < srcs.size(); q += nt) mt_apply(polys[k], out[srcs[q]], out[srcs[q] + step]);
                });
            for (auto& th : pool) th.join();
        }
    }
}

}  // namespace rt
