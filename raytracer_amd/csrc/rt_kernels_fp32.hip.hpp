// rt_kernels_fp32.hip.hpp -- the fp32 FAST PATH of the render hot path (gfx950).
//
// NOT the parity path.  BASELINE.json's north_star asks for "an fp32 fast path reported separately
// with its tolerance": the same frame (same tree, same visit order, same MT19937 sample stream, same
// outputs) computed in binary32 with hardware reciprocal / rsqrt and packed fp32 multiplies.  Hit ids
// may differ from the reference for rays grazing an edge and intensities are only close; the
// measured differences are reported next to the throughput (bench.py aux.fp32_fast_path, DESIGN.md
// section 4.5) and bounded in tests/test_gpu_fp32.py.  Selected with RT_FLAG_FP32; never the default.
//
// Records (written per render call by the prep kernels from the same HBM-resident f64 scene):
//   Node32  32 B  camera-relative box rounded OUTWARD (by 2 further ulps, which covers the rounding
//                 of the slab products) + the f64 path's descriptor; same index as the f64 node, so a
//                 child pair is one 64-byte-aligned 64-byte scalar load
//   Tlas32  64 B  camera-relative box + left / count / first
//   Tri32   64 B  e0, e2, tv = o - v0, q = tv x e0, t numerator  (computed in f64, rounded once)
#pragma once
#include "rt_kernels.hip.hpp"

namespace rtk32 {

using rtk::NT;
using rtk::PACKET_STACK;

struct Node32 {
    float mn[3], mx[3];
    uint32_t desc;
    uint32_t pad;
};
static_assert(sizeof(Node32) == 32, "Node32 must be 32 bytes");

struct Tlas32 {
    float mn[3], mx[3];
    int32_t left, count, first;
    uint32_t pad[7];
};
static_assert(sizeof(Tlas32) == 64, "Tlas32 must be 64 bytes");

struct alignas(64) Tri32 {
    float e0[3], e2[3], tv[3], q[3];
    float tnum;
    float pad[3];
};
static_assert(sizeof(Tri32) == 64, "Tri32 must be 64 bytes");

// largest float <= x, then two more steps down (and the mirror image): see the header comment
__device__ __forceinline__ float f32_below(double x) {
    float f = (float)x;
    if ((double)f > x) f = nextafterf(f, -__builtin_inff());
    f = nextafterf(f, -__builtin_inff());
    return nextafterf(f, -__builtin_inff());
}
__device__ __forceinline__ float f32_above(double x) {
    float f = (float)x;
    if ((double)f < x) f = nextafterf(f, __builtin_inff());
    f = nextafterf(f, __builtin_inff());
    return nextafterf(f, __builtin_inff());
}

__global__ __launch_bounds__(256) void prep32_nodes_kernel(const rt::DevNode* __restrict__ in, Node32* __restrict__ out,
                                                           const long long n, const double ox, const double oy,
                                                           const double oz) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const rt::DevNode v = in[i];
    Node32 o;
    o.mn[0] = f32_below(v.mn[0] - ox);
    o.mn[1] = f32_below(v.mn[1] - oy);
    o.mn[2] = f32_below(v.mn[2] - oz);
    o.mx[0] = f32_above(v.mx[0] - ox);
    o.mx[1] = f32_above(v.mx[1] - oy);
    o.mx[2] = f32_above(v.mx[2] - oz);
    o.desc = v.desc;
    o.pad = 0u;
    out[i] = o;
}

__global__ __launch_bounds__(256) void prep32_tlas_kernel(const rt::DevNode* __restrict__ in, Tlas32* __restrict__ out,
                                                          const long long n, const double ox, const double oy,
                                                          const double oz) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const rt::DevNode v = in[i];
    Tlas32 o = {};
    o.mn[0] = f32_below(v.mn[0] - ox);
    o.mn[1] = f32_below(v.mn[1] - oy);
    o.mn[2] = f32_below(v.mn[2] - oz);
    o.mx[0] = f32_above(v.mx[0] - ox);
    o.mx[1] = f32_above(v.mx[1] - oy);
    o.mx[2] = f32_above(v.mx[2] - oz);
    o.left = v.left;
    o.count = v.count;
    o.first = v.first;
    out[i] = o;
}

__global__ __launch_bounds__(256) void prep32_tris_kernel(const double* __restrict__ tris, Tri32* __restrict__ out,
                                                          const long long n, const double ox, const double oy,
                                                          const double oz) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double* tp = tris + (size_t)i * 9;
    const double v0x = tp[0], v0y = tp[1], v0z = tp[2];
    const double e0x = tp[3] - v0x, e0y = tp[4] - v0y, e0z = tp[5] - v0z;
    const double e2x = tp[6] - v0x, e2y = tp[7] - v0y, e2z = tp[8] - v0z;
    const double tvx = ox - v0x, tvy = oy - v0y, tvz = oz - v0z;
    const double qx = tvy * e0z - tvz * e0y;
    const double qy = tvz * e0x - tvx * e0z;
    const double qz = tvx * e0y - tvy * e0x;
    Tri32 c = {};
    c.e0[0] = (float)e0x;
    c.e0[1] = (float)e0y;
    c.e0[2] = (float)e0z;
    c.e2[0] = (float)e2x;
    c.e2[1] = (float)e2y;
    c.e2[2] = (float)e2z;
    c.tv[0] = (float)tvx;
    c.tv[1] = (float)tvy;
    c.tv[2] = (float)tvz;
    c.q[0] = (float)qx;
    c.q[1] = (float)qy;
    c.q[2] = (float)qz;
    c.tnum = (float)((e2x * qx + e2y * qy) + e2z * qz);
    out[i] = c;
}

typedef float f2 __attribute__((ext_vector_type(2)));
typedef const RT_CONST float* cfptr;

struct Packet32Args {
    const Node32* nodes;
    const Tlas32* tlas;
    const int32_t* slot_root;
    const uint32_t* leaf_ext;
    const Tri32* tris;
    const double* colors;
    const rtk::FrameParams* fp;
    int32_t W, spp;
    int32_t row0, rows;
    int32_t tile_log2;
    int32_t tiles_x, n_tiles;
    int32_t pair64;  // spp == 64 and stack_need <= rtk::PAIR64_STACK
    int32_t interleave;  // see rtk::PacketArgs
    const double* jitter;
    double* intensity;
    uint8_t* rgb;
    int32_t* hit_prim;
    double* hit_t;
    unsigned long long* counters;
    // cuts, as in rtk::PacketArgs (made from the f64 camera-relative boxes: the node indices are the same)
    const uint32_t* cut_lvl[5];
    int32_t cut_tx[5];
    int32_t cut_eff[5];
    int32_t cut_lo;
};

struct Hit32 {
    float t, u, v;
    int32_t prim;
};

// Six products as three packed multiplies, whatever the octant: (mnx,mny)*(ix,iy), (mnz,mxx)*(iz,ix),
// (mxy,mxz)*(iy,iz).  OCT >= 0: all active lanes share the direction signs, so near/far are known
// (bit0/1/2 = ix/iy/iz negative) and the test is max3 / min3 of three registers each.
template <int OCT>
__device__ __forceinline__ unsigned long long slab32(const float b0, const float b1, const float b2, const float b3,
                                                     const float b4, const float b5, const f2 ixy, const f2 izx,
                                                     const f2 iyz) {
    const f2 a = f2{b0, b1} * ixy;  // mnx*ix, mny*iy
    const f2 c = f2{b2, b3} * izx;  // mnz*iz, mxx*ix
    const f2 e = f2{b4, b5} * iyz;  // mxy*iy, mxz*iz
    float tmin, tmax;
    if (OCT >= 0) {
        constexpr bool sx = (OCT & 1) != 0, sy = (OCT & 2) != 0, sz = (OCT & 4) != 0;
        const float nx = sx ? c.y : a.x, fx = sx ? a.x : c.y;
        const float ny = sy ? e.x : a.y, fy = sy ? a.y : e.x;
        const float nz = sz ? e.y : c.x, fz = sz ? c.x : e.y;
        tmin = __builtin_fmaxf(__builtin_fmaxf(nx, ny), nz);
        tmax = __builtin_fminf(__builtin_fminf(fx, fy), fz);
    } else {
        tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(a.x, c.y), __builtin_fminf(a.y, e.x)),
                               __builtin_fminf(c.x, e.y));
        tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(a.x, c.y), __builtin_fmaxf(a.y, e.x)),
                               __builtin_fmaxf(c.x, e.y));
    }
    // one comparison (see rtk::slab_rel): boxes whose far distance is below FLT_MIN count as missed here
    return __builtin_amdgcn_ballot_w64(__builtin_fmaxf(tmin, 0x1p-126f) <= tmax);
}

__device__ __forceinline__ float u2f(const uint32_t u) { return __builtin_bit_cast(float, u); }

// The triangles of a leaf descriptor against the lanes of `inm` (see rtk::test_leaf), in binary32.
template <bool COUNT>
__device__ __forceinline__ void test_leaf32(const Packet32Args& sc, const uint32_t leaf, const unsigned long long inm,
                                            const float dx, const float dy, const float dz, Hit32& h,
                                            rtk::WaveCounters& wc) {
    uint32_t first = leaf & rt::DESC_TRI_MASK;
    uint32_t cnt = ((leaf >> 28) & 7u) + 1u;
    if (cnt == 8u) {
        rtk::cuptr ext = (rtk::cuptr)(const void*)sc.leaf_ext;
        const uint32_t e = first;
        first = ext[2 * e];
        cnt = ext[2 * e + 1];
    }
    if (COUNT) wc.tris += (unsigned long long)cnt * (unsigned)__builtin_popcountll(inm);
    const bool in = __builtin_amdgcn_inverse_ballot_w64(inm);
    for (uint32_t k = 0; k < cnt; ++k) {
        rtk::u32x16 tr;  // the whole 64-byte record in one scalar round trip
        {
            const Tri32* trec = sc.tris + (size_t)(first + k);
            asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tr) : "s"(trec) : "memory");
        }
        const float e0x = u2f(tr[0]), e0y = u2f(tr[1]), e0z = u2f(tr[2]);
        const float e2x = u2f(tr[3]), e2y = u2f(tr[4]), e2z = u2f(tr[5]);
        const float px = dy * e2z - dz * e2y;
        const float py = dz * e2x - dx * e2z;
        const float pz = dx * e2y - dy * e2x;
        const float det = (e0x * px + e0y * py) + e0z * pz;
        if (in && det > 1e-6f) {
            const float tvx = u2f(tr[6]), tvy = u2f(tr[7]), tvz = u2f(tr[8]);
            const float qx = u2f(tr[9]), qy = u2f(tr[10]), qz = u2f(tr[11]);
            const float tnum = u2f(tr[12]);
            const float unum = (tvx * px + tvy * py) + tvz * pz;
            const float vnum = (dx * qx + dy * qy) + dz * qz;
            // sure rejects on the numerators (see rtk::test_leaf): a triangle that none of the
            // wave's rays can hit costs no reciprocal and no range checks
            if (!(__builtin_fminf(unum, vnum) < -0x1p-60f || (unum + vnum) > det * (1.0f + 0x1p-18f))) {
                const float inv = __builtin_amdgcn_rcpf(det);
                const float u = unum * inv;
                const float v = vnum * inv;
                const float t = tnum * inv;
                const bool ok = (u >= 0.0f) && (u <= 1.0f) && (v >= 0.0f) && ((u + v) <= 1.0f) && (t >= 0.0f);
                if (ok && t < h.t) {
                    h.t = t;
                    h.u = u;
                    h.v = v;
                    h.prim = (int32_t)(first + k);
                }
            }
        }
    }
}

// Same wave-packet walk as rtk::trace_packet (reference visit order, no culling), in binary32.
template <int OCT, bool COUNT>
__device__ __forceinline__ void trace_packet32(const Packet32Args& sc, const unsigned long long valid_mask,
                                               const float dx, const float dy, const float dz, const float ix,
                                               const float iy, const float iz, uint4* wstk, Hit32& h,
                                               rtk::WaveCounters& wc, const uint32_t vlist, int n_entries,
                                               const bool simple) {
    const f2 ixy = f2{ix, iy}, izx = f2{iz, ix}, iyz = f2{iy, iz};
    if (simple) {
        // CUT_SIMPLE list (rtk::cut_trim): leaves every ray reaches and trimmed leaves only -- no stack, no dispatch
        for (int e = 1; e <= n_entries; ++e) {
            uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)vlist, e);
            unsigned long long m = valid_mask;
            if (!(w & rt::DESC_LEAF)) {
                cfptr q = (cfptr)(const void*)(sc.nodes + (w & rtk::CUT_NODE_IDX));
                ++e;
                w = (uint32_t)__builtin_amdgcn_readlane((int)vlist, e);
                if (COUNT) wc.nodes += (unsigned)__builtin_popcountll(valid_mask);
                m &= slab32<OCT>(q[0], q[1], q[2], q[3], q[4], q[5], ixy, izx, iyz);
            }
            if (m != 0ull) test_leaf32<COUNT>(sc, w, m, dx, dy, dz, h, wc);
        }
        return;
    }
    int sp = 0;
    int ei = 1;  // entry list of the batch's cut, as in rtk::trace_packet (n_entries < 0: walk from the root)
    uint32_t cur = rt::DESC_TLAS;
    if (n_entries < 0) {
        n_entries = 0;
    } else {
        if (n_entries == 0) return;
        cur = (uint32_t)__builtin_amdgcn_readlane((int)vlist, 1);
    }
    unsigned long long inm = valid_mask;
    for (;;) {
        while ((cur & (rt::DESC_LEAF | rt::DESC_TLAS | rt::DESC_ROOT)) == 0u) {
            rtk::u32x16 nd;  // the child pair: 2 x 32 bytes in one scalar load at base + (cur << 5)
            const uint32_t off = cur << 5;
            asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(nd) : "s"(sc.nodes), "s"(off) : "memory");
            const uint32_t dl = nd[6], dr = nd[14];
            if (COUNT) {
                wc.nodes += 2u * (unsigned)__builtin_popcountll(inm);
                wc.steps += 2u;
            }
            const unsigned long long ml =
                slab32<OCT>(u2f(nd[0]), u2f(nd[1]), u2f(nd[2]), u2f(nd[3]), u2f(nd[4]), u2f(nd[5]), ixy, izx, iyz) & inm;
            const unsigned long long mr =
                slab32<OCT>(u2f(nd[8]), u2f(nd[9]), u2f(nd[10]), u2f(nd[11]), u2f(nd[12]), u2f(nd[13]), ixy, izx, iyz) &
                inm;
            // straight-line step with scalar selects (see rtk::trace_packet)
            uint32_t lnz = (uint32_t)ml | (uint32_t)(ml >> 32);
            uint32_t rnz = (uint32_t)mr | (uint32_t)(mr >> 32);
            uint32_t both = lnz < rnz ? lnz : rnz;
            asm volatile("" : "+s"(lnz), "+s"(rnz), "+s"(both));
            if (both != 0u) wstk[sp++] = make_uint4(dr, (uint32_t)mr, (uint32_t)(mr >> 32), 0u);
            const uint32_t nxt = (lnz != 0u) ? dl : dr;
            inm = (lnz != 0u) ? ml : mr;
            cur = ((lnz | rnz) != 0u) ? nxt : rt::DESC_NONE;
        }
        const bool alive = cur != rt::DESC_NONE;
        if (alive) {
            if (cur & rt::DESC_LEAF) {
                test_leaf32<COUNT>(sc, cur, inm, dx, dy, dz, h, wc);
            } else if ((cur & rtk::CUT_NODE_ENTRY) == rtk::CUT_NODE_ENTRY) {
                // entry of a cut: a BLAS node whose own box is tested per ray
                const uint32_t ni = cur & rtk::CUT_NODE_IDX;
                cfptr q = (cfptr)(const void*)(sc.nodes + ni);
                uint32_t desc = ((rtk::cuptr)(const void*)(sc.nodes + ni))[6];
                if (cur & rtk::CUT_TRIM) {  // the leaf's live triangles: next word of the list (rtk::cut_trim)
                    ++ei;
                    desc = (uint32_t)__builtin_amdgcn_readlane((int)vlist, ei);
                }
                if (COUNT) wc.nodes += (unsigned)__builtin_popcountll(inm);
                const unsigned long long hm = slab32<OCT>(q[0], q[1], q[2], q[3], q[4], q[5], ixy, izx, iyz) & inm;
                if (hm != 0ull) {
                    cur = desc;
                    inm = hm;
                    continue;
                }
            } else if (cur & rt::DESC_TLAS) {
                cfptr q = (cfptr)(const void*)(sc.tlas + (cur & 0x3FFFFFFFu));
                const RT_CONST int32_t* qi = (const RT_CONST int32_t*)(const void*)(sc.tlas + (cur & 0x3FFFFFFFu));
                const int32_t left = qi[6], count = qi[7], first = qi[8];
                if (COUNT) wc.nodes += (unsigned)__builtin_popcountll(inm);
                const unsigned long long hm = slab32<OCT>(q[0], q[1], q[2], q[3], q[4], q[5], ixy, izx, iyz) & inm;
                if (hm != 0ull) {
                    const uint32_t lo = (uint32_t)hm, hi = (uint32_t)(hm >> 32);
                    if (left < 0) {
                        for (int32_t k = count - 1; k >= 0; --k)
                            wstk[sp++] = make_uint4(rt::DESC_ROOT | (uint32_t)(first + k), lo, hi, 0u);
                    } else {
                        wstk[sp++] = make_uint4(rt::DESC_TLAS | (uint32_t)(left + 1), lo, hi, 0u);
                        wstk[sp++] = make_uint4(rt::DESC_TLAS | (uint32_t)left, lo, hi, 0u);
                    }
                }
            } else {
                const RT_CONST int32_t* sr = (const RT_CONST int32_t*)(const void*)sc.slot_root;
                const int32_t root = sr[cur & 0x1FFFFFFFu];
                cfptr q = (cfptr)(const void*)(sc.nodes + root);
                const uint32_t desc = ((rtk::cuptr)(const void*)(sc.nodes + root))[6];
                if (COUNT) wc.nodes += (unsigned)__builtin_popcountll(inm);
                const unsigned long long hm = slab32<OCT>(q[0], q[1], q[2], q[3], q[4], q[5], ixy, izx, iyz) & inm;
                if (hm != 0ull) {
                    cur = desc;
                    inm = hm;
                    continue;
                }
            }
        }
        if (sp == 0) {
            if (ei >= n_entries) break;
            ++ei;
            cur = (uint32_t)__builtin_amdgcn_readlane((int)vlist, ei);
            inm = valid_mask;
            continue;
        }
        --sp;
        const uint4 e = wstk[sp];
        cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.x);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.y);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.z);
        inm = ((unsigned long long)hi << 32) | lo;
    }
}

// Same tiling, ray order and in-order per-pixel sample sum as rtk::render_packet_kernel.
template <bool COUNT>
__global__ __launch_bounds__(NT, 8) void render_packet32_kernel(const Packet32Args a) {
    __shared__ uint4 s_wstk[NT / 64][PACKET_STACK];
    __shared__ float s_col[NT / 64][3][64];
    __shared__ float s_acc[NT / 64][3][64];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    int tile = blockIdx.x;
    {
        const int nb = a.n_tiles, per = nb >> 3, main = per << 3;
        if ((int)blockIdx.x < main) tile = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    }
    const int ts = 1 << a.tile_log2;
    const int npix_w = (ts * ts) >> 2;
    const int m0 = wv * npix_w;
    int tix, tiy;
    rtk::tile_coords(tile, a.n_tiles, a.tiles_x, tix, tiy);
    const int tx0 = tix << a.tile_log2;
    const int ty0 = tiy << a.tile_log2;
    const int spp = a.spp;
    const int n_rays = npix_w * spp;
    uint4* wstk = s_wstk[wv];
    float(*wcol)[64] = s_col[wv];
    float(*wacc)[64] = s_acc[wv];

    wacc[0][lane] = 0.0f;
    wacc[1][lane] = 0.0f;
    wacc[2][lane] = 0.0f;

    unsigned long long w_nodes = 0, w_tris = 0, w_steps = 0;
    uint32_t n_hits = 0;

    // The jitter pair of the NEXT batch is requested before this batch is traversed (one HBM round trip
    // off every batch's critical path: the kernel is bound by per-wave latency at 8 waves per SIMD).
    struct Slot {
        int px, py;
        bool active;
        size_t sidx;
        rtk::d2 jit;
        uint32_t vcut;  // the batch's cut: lane 0 = header, lanes 1.. = entries (0: none)
    };
    auto fetch = [&](const int base) {
        Slot q;
        const int rr = base + lane;
        const int ml = rr / spp;
        const int k = rr - ml * spp;
        const int m = rtk::wave_pixel(ml, wv, m0, a.interleave);
        q.px = tx0 + (int)rtk::compact1by1((uint32_t)m);
        q.py = ty0 + (int)rtk::compact1by1((uint32_t)m >> 1);
        q.active = (rr < n_rays) && (q.px < a.W) && (q.py < a.rows);
        q.sidx = q.active ? ((size_t)q.py * a.W + q.px) * (size_t)spp + k : 0;
        q.jit = rtk::d2{0.0, 0.0};
        // streamed once: non-temporal, so the 16 B/sample stream does not displace tree nodes in L2
        if (q.active) q.jit = __builtin_nontemporal_load(reinterpret_cast<const rtk::d2*>(a.jitter) + q.sidx);
        q.vcut = 0u;
        if (a.cut_lo >= 0) {
            const int last = min(63, n_rays - 1 - base);
            q.vcut = rtk::load_batch_cut<offsetof(Packet32Args, cut_lvl), offsetof(Packet32Args, cut_tx),
                                         offsetof(Packet32Args, cut_eff)>(
                a.cut_lo, __builtin_amdgcn_readfirstlane(m), __builtin_amdgcn_readlane(m, last),
                __builtin_amdgcn_readfirstlane(q.px), __builtin_amdgcn_readfirstlane(q.py),
                __builtin_amdgcn_readlane(q.px, last), __builtin_amdgcn_readlane(q.py, last), a.W, a.rows, lane);
        }
        return q;
    };
    Slot nxt = fetch(0);
    for (int base = 0; base < n_rays; base += 64) {
        const Slot cs = nxt;
        if (base + 64 < n_rays) nxt = fetch(base + 64);
        const int px = cs.px, py = cs.py;
        const bool active = cs.active;
        const size_t sidx = cs.sidx;
        float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
        float dx = 0.0f, dy = 0.0f, dz = -1.0f;
        if (active) {
            rtk::cdptr fp = (rtk::cdptr)(const void*)a.fp;
            const rtk::d2 jit = cs.jit;
            // the pixel sample position in f64 (sub-pixel offsets need more than 24 bits at 4096 px),
            // everything after it in f32
            const double s0 = (double)px + jit.x;
            const double s1 = (double)(py + a.row0) + jit.y;
            const float fx = (float)(((fp[3] + s0 * fp[6]) + s1 * fp[9]) - fp[0]);
            const float fy = (float)(((fp[4] + s0 * fp[7]) + s1 * fp[10]) - fp[1]);
            const float fz = (float)(((fp[5] + s0 * fp[8]) + s1 * fp[11]) - fp[2]);
            const float z = (fx * fx + fy * fy) + fz * fz;
            dx = fx;
            dy = fy;
            dz = fz;
            if (z > 0.0f) {
                const float r = __builtin_amdgcn_rsqf(z);
                dx = fx * r;
                dy = fy * r;
                dz = fz * r;
            }
        }
        const float ix = __builtin_amdgcn_rcpf(dx), iy = __builtin_amdgcn_rcpf(dy), iz = __builtin_amdgcn_rcpf(dz);
        Hit32 h;
        h.t = __builtin_inff();
        h.u = 0.0f;
        h.v = 0.0f;
        h.prim = -1;
        const unsigned long long amask = __builtin_amdgcn_ballot_w64(active);
        if (amask != 0ull) {
            rtk::WaveCounters wc = {0ull, 0ull, 0ull, 0ull};
            const unsigned long long ngx = __builtin_amdgcn_ballot_w64(ix < 0.0f) & amask;
            const unsigned long long ngy = __builtin_amdgcn_ballot_w64(iy < 0.0f) & amask;
            const unsigned long long ngz = __builtin_amdgcn_ballot_w64(iz < 0.0f) & amask;
            const bool uni = (ngx == 0ull || ngx == amask) && (ngy == 0ull || ngy == amask) && (ngz == 0ull || ngz == amask);
            // the batch's cut (made for the exact f64 test; this path's boxes are those boxes rounded outward, so a
            // ray the cut is sure about is one this path agrees on up to its own tolerance); not with infinite 1/d
            const bool wild = active && !(__builtin_fabsf(ix) < 1e30f && __builtin_fabsf(iy) < 1e30f && __builtin_fabsf(iz) < 1e30f);
            const int n_ent = __builtin_amdgcn_ballot_w64(wild) != 0ull
                                  ? -1
                                  : rtk::batch_cut_entries(cs.vcut, amask, ngx, ngy, ngz, cs.jit.x, cs.jit.y);
            // (a count >= 0 is the header's own: the batch takes the block's list as it is)
            const bool simple = n_ent >= 0 && ((uint32_t)__builtin_amdgcn_readfirstlane((int)cs.vcut) & rtk::CUT_SIMPLE) != 0u;
            if (uni) {
                const int oct = (ngx != 0ull ? 1 : 0) | (ngy != 0ull ? 2 : 0) | (ngz != 0ull ? 4 : 0);
                switch (oct) {
#define RT_OCT32_CASE(O)                                                                    \
    case O:                                                                                 \
        trace_packet32<O, COUNT>(a, amask, dx, dy, dz, ix, iy, iz, wstk, h, wc, cs.vcut, n_ent, simple); \
        break;
                    RT_OCT32_CASE(0)
                    RT_OCT32_CASE(1)
                    RT_OCT32_CASE(2)
                    RT_OCT32_CASE(3)
                    RT_OCT32_CASE(4)
                    RT_OCT32_CASE(5)
                    RT_OCT32_CASE(6)
                    RT_OCT32_CASE(7)
#undef RT_OCT32_CASE
                }
            } else {
                trace_packet32<-1, COUNT>(a, amask, dx, dy, dz, ix, iy, iz, wstk, h, wc, cs.vcut, n_ent, simple);
            }
            if (COUNT && lane == 0) {
                w_nodes += wc.nodes;
                w_tris += wc.tris;
                w_steps += wc.steps;
            }
        }
        if (active) {
            if (h.prim >= 0) {
                const double* fc = a.colors + (size_t)h.prim * 3;
                const float w = (1.0f - h.u) - h.v;
                const float f0 = (float)fc[0], f1 = (float)fc[1], f2_ = (float)fc[2];
                c0 = (h.u * f0 + h.v * f0) + w * f0;
                c1 = (h.u * f1 + h.v * f1) + w * f1;
                c2 = (h.u * f2_ + h.v * f2_) + w * f2_;
            } else {
                const float al = 0.5f * (dy + 1.0f);
                const float ia = 1.0f - al;
                c0 = ia + al * 0.5f;
                c1 = ia + al * 0.7f;
                c2 = ia + al * 1.0f;
            }
            if (COUNT) n_hits += (h.prim >= 0);
            if (a.hit_prim) __builtin_nontemporal_store(h.prim, a.hit_prim + sidx);
            if (a.hit_t) __builtin_nontemporal_store((double)h.t, a.hit_t + sidx);
        }
        if (a.pair64) {  // spp == 64: two pixels summed at a time, see rtk::render_packet_kernel
            const int second = (base >> 6) & 1;
            float* dst = second ? reinterpret_cast<float*>(wstk + rtk::PAIR64_STACK) : &wcol[0][0];
            dst[lane] = c0;
            dst[64 + lane] = c1;
            dst[128 + lane] = c2;
            __builtin_amdgcn_wave_barrier();
            if (second && lane < 6) {
                const int which = lane >= 3 ? 1 : 0;
                const int c = lane - 3 * which;
                const float* src = (which ? reinterpret_cast<const float*>(wstk + rtk::PAIR64_STACK) : &wcol[0][0]) + c * 64;
                float acc = 0.0f;
#pragma unroll
                for (int s2 = 0; s2 < 64; s2 += 8) {
                    float v8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v8[j] = src[s2 + j];
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc += v8[j];
                }
                wacc[c][(base >> 6) - 1 + which] = acc;
            }
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        wcol[0][lane] = c0;
        wcol[1][lane] = c1;
        wcol[2][lane] = c2;
        __builtin_amdgcn_wave_barrier();
        {
            const int end = min(base + 64, n_rays);
            const int m_first = base / spp;
            const int m_last = (end - 1) / spp;
            const int work = (m_last - m_first + 1) * 3;
            for (int q = lane; q < work; q += 64) {
                const int pm = m_first + q / 3;
                const int c = q - (q / 3) * 3;
                const int lo = max(pm * spp, base) - base;
                const int hi = min((pm + 1) * spp, end) - base;
                float acc = wacc[c][pm];
                const float* src = &wcol[c][lo];
                const int n = hi - lo;
                int s2 = 0;
                for (; s2 + 8 <= n; s2 += 8) {  // strictly in sample order; the 8 LDS reads are issued
                    float v8[8];                 // together, ahead of the dependent adds
#pragma unroll
                    for (int j = 0; j < 8; ++j) v8[j] = src[s2 + j];
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc += v8[j];
                }
                for (; s2 < n; ++s2) acc += src[s2];
                wacc[c][pm] = acc;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    if (lane < npix_w) {
        const int m = rtk::wave_pixel(lane, wv, m0, a.interleave);
        const int px = tx0 + (int)rtk::compact1by1((uint32_t)m);
        const int py = ty0 + (int)rtk::compact1by1((uint32_t)m >> 1);
        if (px < a.W && py < a.rows) {
            const float gray = (0.2126f * wacc[0][lane] + 0.7152f * wacc[1][lane]) + 0.0722f * wacc[2][lane];
            const float inten = gray / (float)spp;
            const int gb = min((int)(inten * 255.99f), 255);  // fp32 rounding may reach 256
            const size_t pidx = (size_t)py * a.W + px;
            a.intensity[pidx] = (double)inten;
            const uint8_t g8 = (uint8_t)gb;
            a.rgb[pidx * 3] = g8;
            a.rgb[pidx * 3 + 1] = g8;
            a.rgb[pidx * 3 + 2] = g8;
        }
    }

    if (COUNT) {
        unsigned long long v0 = w_nodes, v1 = w_tris, v2 = n_hits;
        for (int off = 32; off > 0; off >>= 1) {
            v0 += __shfl_down(v0, off, 64);
            v1 += __shfl_down(v1, off, 64);
            v2 += __shfl_down(v2, off, 64);
        }
        if (lane == 0 && a.counters) {
            atomicAdd(&a.counters[0], v0);
            atomicAdd(&a.counters[1], v1);
            atomicAdd(&a.counters[2], v2);
            atomicAdd(&a.counters[3], w_steps);
        }
    }
}

}  // namespace rtk32
