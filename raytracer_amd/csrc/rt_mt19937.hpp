// rt_mt19937.hpp -- host side of the sampler: MT19937 state handling and GF(2) jump-ahead.
//
// The reference draws every jitter offset from ONE process-global std::mt19937 (seed 5489,
// cpp/rtmathutils.h:15-19); the stream runs on across pixels, cameras and frames.  To
// generate that exact stream in parallel on the GPU we need the generator's state at many
// positions.  A state here is the window of the 624 raw (untempered) words that PRECEDE the
// position: words x[p-624 .. p-1]; the next word generated from it is x[p].
//
// Jump-ahead (Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer 2008): with A the one-word
// transition and phi its characteristic polynomial (degree 19937), A^J s = g(A) s where
// g = x^J mod phi, evaluated with Horner's rule.  phi is obtained once by Berlekamp-Massey
// on the generator's own output bits.
#pragma once
#include <cstdint>
#include <vector>

namespace rt {

constexpr int MT_N = 624;

struct MtState {
    uint32_t w[MT_N];
};

void mt_seed(MtState& s, uint32_t seed);  // std::mt19937(seed): state preceding position 0
void mt_next_block(MtState& s);           // advance by 624 words (in place)

// Bit-packed polynomial over GF(2), degree < 19937 (312 x 64 bits).
struct MtPoly {
    uint64_t b[312];
};

// The characteristic polynomial phi of the generator's one-word transition, bit-packed (bit i of word i/64 = the
// coefficient of x^i; degree 19937): the compiled-in table, or (recompute) a fresh Berlekamp-Massey run over the
// generator's output -- the two must agree (selftest).
void mt_phi_words(uint64_t out[312], bool recompute);

// g = x^stride mod phi  (cached per stride; thread-safe)
MtPoly mt_jump_poly(uint64_t stride);
// out = g(A) in   (out may alias in)
void mt_apply(const MtPoly& g, const MtState& in, MtState& out);

// States at positions base_pos + r*stride for r in [0, n), given the state at base_pos.
// Tree doubling with polynomials for stride*2^k, parallel over up to n_threads host threads.
void mt_stream_states(const MtState& base, uint64_t stride, int n, int n_threads, std::vector<MtState>& out);

}  // namespace rt
