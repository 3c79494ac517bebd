// rt_kernels.hip.hpp -- gfx950 device code of the render hot path (included by rt_api.hip).
//
// Kernels
//   prep_frame_kernel   per-frame camera-relative records (everything that depends
//                      only on the camera centre, computed once per frame with the reference's operations)
//   render_packet_kernel   THE hot kernel: ray generation + wave-packet TLAS/BLAS traversal on the scalar
//                      unit + Moller-Trumbore + shading + in-order sample accumulation + luma/quantise
//                      (reference: cpp/rtrender.cpp:13-67, cpp/rtrayintersection.cpp:36-228);
//                      <COUNT, P64>: with visit counters / the one-pixel-per-batch instantiation (64 * 2^n spp
//                      and a little below)
//   cut_level_kernel   per pixel block and quadtree level: the cut at which the block's rays start their walk
//                      (conservative block-level classification of boxes and triangles; same hits, bit for bit)
//   render_kernel      the same per lane (LDS or global stack): trees deeper than the packet stack
//   trace_kernel       the same traversal for caller-supplied rays (rt_trace_rays, test seam)
//   mt_jump_kernel / mt_jump_tree_kernel / mt_jump_all_kernel   GF(2) jump-ahead of MT19937 states
//   mt_jitter_kernel   MT19937 words -> random_double()-0.5 values, one workgroup per jump-ahead
//                      stream (reference: cpp/rtmathutils.h:15-19, cpp/rtrender.cpp:53)
//   philox_jitter_kernel   counter-based fast sampler (RT_SAMPLER_PHILOX; not the reference's stream)
// Explicit __builtin_fma appears only where the result is provably the one the unfused reference sequence
// gives: in the hand-issued copies of the compiler's own division / square-root expansions, and where the
// product is exact (a power-of-two factor).
//   (the fp32 fast path lives in rt_kernels_fp32.hip.hpp)
//
// Numerics contract (parity with the reference's x86-64 -O2 build): IEEE binary64, no FMA
// contraction (this TU is compiled with -ffp-contract=off), true division and sqrt, every
// three-term sum as (a+b)+c, min/max with std::min/std::max operand order and NaN behaviour.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_bvh.hpp"

namespace rtk {

constexpr int NT = 256;  // threads per workgroup of the traversal kernels (4 waves)

struct DevScene {
    const rt::DevNode* nodes;
    const rt::DevNode* tlas;
    const int32_t* slot_root;
    const uint32_t* leaf_ext;
    const double* tris;
    const double* colors;
};

typedef double d2 __attribute__((ext_vector_type(2)));

struct Ray {
    double ox, oy, oz, dx, dy, dz, ix, iy, iz;
};

struct Hit {
    double t, u, v;
    int32_t prim;
};

struct Box {
    double mnx, mny, mnz, mxx, mxy, mxz;
};

// std::min(a,b) / std::max(a,b) as libstdc++ defines them (operand order matters for NaN)
__device__ __forceinline__ double std_min(double a, double b) { return (b < a) ? b : a; }
__device__ __forceinline__ double std_max(double a, double b) { return (a < b) ? b : a; }

__device__ __forceinline__ void load_node(const rt::DevNode* p, Box& b, int32_t& left, int32_t& count,
                                          int32_t& first, uint32_t& desc) {
    const d2* q = reinterpret_cast<const d2*>(p);
    const d2 a = q[0], c = q[1], e = q[2], m = q[3];
    b.mnx = a.x;
    b.mny = a.y;
    b.mnz = c.x;
    b.mxx = c.y;
    b.mxy = e.x;
    b.mxz = e.y;
    const unsigned long long m0 = (unsigned long long)__double_as_longlong(m.x);
    const unsigned long long m1 = (unsigned long long)__double_as_longlong(m.y);
    left = (int32_t)(uint32_t)m0;
    count = (int32_t)(uint32_t)(m0 >> 32);
    first = (int32_t)(uint32_t)m1;
    desc = (uint32_t)(m1 >> 32);
}

// intersect_AABB, cpp/rtrayintersection.cpp:124-144 (ray.t_max is always +inf).
// inverse_direction is the same value on every call, so it is computed once per ray.
__device__ __forceinline__ bool slab(const Ray& r, const Box& b) {
    const double t0 = (b.mnx - r.ox) * r.ix, t1 = (b.mxx - r.ox) * r.ix;
    const double t2 = (b.mny - r.oy) * r.iy, t3 = (b.mxy - r.oy) * r.iy;
    const double t4 = (b.mnz - r.oz) * r.iz, t5 = (b.mxz - r.oz) * r.iz;
    const double tmin = std_max(std_max(std_min(t0, t1), std_min(t2, t3)), std_min(t4, t5));
    const double tmax = std_min(std_min(std_max(t0, t1), std_max(t2, t3)), std_max(t4, t5));
    return tmin <= tmax && tmax > 0.0 && tmin < __builtin_inf();
}

// One row of intersect_bvh_triangles (cpp/rtrayintersection.cpp:36-122) followed by the
// closest-hit update.  Scanning triangles one by one with a strict `<` picks the same
// triangle as the reference's per-leaf minCoeff (first minimal index) + `closest_t < rec.t`.
__device__ __forceinline__ void tri_test(const Ray& r, const double* __restrict__ tp, int32_t prim, Hit& h) {
    const double v0x = tp[0], v0y = tp[1], v0z = tp[2];
    const double e0x = tp[3] - v0x, e0y = tp[4] - v0y, e0z = tp[5] - v0z;
    const double e2x = tp[6] - v0x, e2y = tp[7] - v0y, e2z = tp[8] - v0z;
    const double px = r.dy * e2z - r.dz * e2y;
    const double py = r.dz * e2x - r.dx * e2z;
    const double pz = r.dx * e2y - r.dy * e2x;
    const double det = (e0x * px + e0y * py) + e0z * pz;
    if (det > 1e-6) {  // back faces / parallel: every later value is masked out anyway
        const double inv = 1.0 / det;
        const double tvx = r.ox - v0x, tvy = r.oy - v0y, tvz = r.oz - v0z;
        const double u = ((tvx * px + tvy * py) + tvz * pz) * inv;
        const double qx = tvy * e0z - tvz * e0y;
        const double qy = tvz * e0x - tvx * e0z;
        const double qz = tvx * e0y - tvy * e0x;
        const double v = ((r.dx * qx + r.dy * qy) + r.dz * qz) * inv;
        const double t = ((e2x * qx + e2y * qy) + e2z * qz) * inv;
        const bool ok = (u >= 0.0) && (u <= 1.0) && (v >= 0.0) && ((u + v) <= 1.0) && (t >= 0.0) &&
                        (t <= __builtin_inf());
        if (ok && t < h.t) {
            h.t = t;
            h.u = u;
            h.v = v;
            h.prim = prim;
        }
    }
}

// TLAS -> BLAS traversal in the reference's visit order (left subtree first, TLAS leaves slot
// by slot, no closest-hit culling: the reference never shrinks t_max).  Each child pair is
// slab-tested when its parent is visited (one 128-B fetch); only passing children are kept.
// The set of nodes tested equals the set the reference pops.
// stk: this lane's stack column, entries `stride` words apart (LDS, or global for deep trees).
template <bool COUNT>
__device__ __forceinline__ void trace(const DevScene& sc, const Ray& r, uint32_t* stk, const int stride, Hit& h,
                                      uint32_t& n_nodes, uint32_t& n_tris) {
    int sp = 0;
    uint32_t cur = rt::DESC_TLAS;  // TLAS node 0
    bool have = true;
    for (;;) {
        if (!have) {
            if (sp == 0) break;
            --sp;
            cur = stk[sp * stride];
        }
        have = false;
        if (cur & rt::DESC_LEAF) {
            uint32_t first = cur & rt::DESC_TRI_MASK;
            uint32_t cnt = ((cur >> 28) & 7u) + 1u;
            if (cnt == 8u) {
                const uint32_t e = first;
                first = sc.leaf_ext[2 * e];
                cnt = sc.leaf_ext[2 * e + 1];
            }
            if (COUNT) n_tris += cnt;
            const double* tp = sc.tris + (size_t)first * 9;
            for (uint32_t k = 0; k < cnt; ++k) tri_test(r, tp + (size_t)k * 9, (int32_t)(first + k), h);
            continue;
        }
        if (cur & (rt::DESC_TLAS | rt::DESC_ROOT)) {
            Box b;
            int32_t left, count, first;
            uint32_t desc;
            if (cur & rt::DESC_TLAS) {
                load_node(sc.tlas + (cur & 0x3FFFFFFFu), b, left, count, first, desc);
                if (COUNT) n_nodes += 1;
                if (!slab(r, b)) continue;
                if (left < 0) {
                    for (int32_t k = count - 1; k >= 0; --k) stk[(sp++) * stride] = rt::DESC_ROOT | (uint32_t)(first + k);
                } else {
                    stk[(sp++) * stride] = rt::DESC_TLAS | (uint32_t)(left + 1);
                    stk[(sp++) * stride] = rt::DESC_TLAS | (uint32_t)left;
                }
            } else {
                const int32_t root = sc.slot_root[cur & 0x1FFFFFFFu];
                load_node(sc.nodes + root, b, left, count, first, desc);
                if (COUNT) n_nodes += 1;
                if (slab(r, b)) {
                    cur = desc;
                    have = true;
                }
            }
            continue;
        }
        // internal BLAS node: cur = index of its left child; the right child is adjacent
        Box bl, br;
        int32_t l0, l1, l2, r0, r1, r2;
        uint32_t dl, dr;
        load_node(sc.nodes + cur, bl, l0, l1, l2, dl);
        load_node(sc.nodes + cur + 1, br, r0, r1, r2, dr);
        if (COUNT) n_nodes += 2;
        const bool hl = slab(r, bl), hr = slab(r, br);
        if (hl) {
            if (hr) stk[(sp++) * stride] = dr;
            cur = dl;
            have = true;
        } else if (hr) {
            cur = dr;
            have = true;
        }
    }
}

__device__ __forceinline__ void make_ray(Ray& r, double ox, double oy, double oz, double dx, double dy, double dz) {
    r.ox = ox;
    r.oy = oy;
    r.oz = oz;
    r.dx = dx;
    r.dy = dy;
    r.dz = dz;
    r.ix = 1.0 / dx;
    r.iy = 1.0 / dy;
    r.iz = 1.0 / dz;
}

// return_ray_color, cpp/rtrender.cpp:13-33 (set_face_normal is dead work and is not done)
__device__ __forceinline__ void shade(const DevScene& sc, const Ray& r, const Hit& h, double& c0, double& c1,
                                      double& c2) {
    if (h.t != __builtin_inf()) {
        const double* fc = sc.colors + (size_t)h.prim * 3;
        const double w = (1.0 - h.u) - h.v;
        const double f0 = fc[0], f1 = fc[1], f2 = fc[2];
        c0 = (h.u * f0 + h.v * f0) + w * f0;
        c1 = (h.u * f1 + h.v * f1) + w * f1;
        c2 = (h.u * f2 + h.v * f2) + w * f2;
    } else {
        const double a = 0.5 * (r.dy + 1.0);
        const double ia = (1.0 - a) * 1.0;
        c0 = ia + a * 0.5;
        c1 = ia + a * 0.7;
        c2 = ia + a * 1.0;
    }
}

struct RenderArgs {
    DevScene sc;
    double cam[3], p00[3], sp[6];
    int32_t W, spp;
    int32_t row0;      // first image row of this launch (absolute)
    int32_t rows;      // rows rendered by this launch
    int32_t tile_log2; // tile side = 1 << tile_log2 (3 or 4)
    int32_t tiles_x, n_tiles;
    const double* jitter;  // (rows*W*spp) pairs, sample order (j,i,k), relative to row0
    double* intensity;     // rows*W
    uint8_t* rgb;          // rows*W*3
    int32_t* hit_prim;     // rows*W*spp or null
    double* hit_t;         // rows*W*spp or null
    unsigned long long* counters;  // [nodes, tris, hits] or null
    uint32_t* gstack;      // global stack (deep trees) or null
    int32_t gstack_cap;
};

__device__ __forceinline__ uint32_t compact1by1(uint32_t x) {
    x &= 0x55555555u;
    x = (x ^ (x >> 1)) & 0x33333333u;
    x = (x ^ (x >> 2)) & 0x0f0f0f0fu;
    x = (x ^ (x >> 4)) & 0x00ff00ffu;
    return x;
}

// One workgroup renders one square pixel tile.  Rays of the tile are ordered pixel-major in
// Morton order with the samples of a pixel innermost, so the 64 lanes of a wave hold the
// samples of one pixel (spp >= 64) or of a compact block of neighbouring pixels.
// Sample colours go through LDS and are added per pixel in sample order k = 0..spp-1 by one
// lane per (pixel, channel): the sum is bit-identical to the reference's serial loop
// (cpp/rtrender.cpp:51-60) for any GPU count or tiling.
template <int STACK, bool COUNT>
__global__ __launch_bounds__(NT) void render_kernel(const RenderArgs a) {
    __shared__ uint32_t s_stk[(STACK > 0 ? STACK : 1) * NT];
    __shared__ double s_col[3][NT];
    __shared__ double s_acc[3][NT];

    const int tid = threadIdx.x;
    // XCD-aware mapping: blocks b and b+8 share an XCD (round-robin dispatch), so give each
    // XCD a contiguous run of tiles: neighbouring tiles walk the same subtrees -> L2 hits.
    int tile = blockIdx.x;
    {
        const int nb = a.n_tiles, per = nb >> 3, main = per << 3;
        if ((int)blockIdx.x < main) tile = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    }
    const int ts = 1 << a.tile_log2;
    const int npix_tile = ts * ts;
    const int tx0 = (tile % a.tiles_x) << a.tile_log2;
    const int ty0 = (tile / a.tiles_x) << a.tile_log2;
    const int spp = a.spp;
    const int n_rays = npix_tile * spp;

    s_acc[0][tid] = 0.0;
    s_acc[1][tid] = 0.0;
    s_acc[2][tid] = 0.0;

    uint32_t* stk;
    int stride;
    if (STACK > 0) {
        stk = s_stk + tid;
        stride = NT;
    } else {
        stk = a.gstack + ((size_t)blockIdx.x * NT + tid);
        stride = (int)(gridDim.x * NT);
    }

    uint32_t n_nodes = 0, n_tris = 0, n_hits = 0;

    for (int base = 0; base < n_rays; base += NT) {
        const int rr = base + tid;
        const int m = rr / spp;
        const int k = rr - m * spp;
        const int px = tx0 + (int)compact1by1((uint32_t)m);
        const int py = ty0 + (int)compact1by1((uint32_t)m >> 1);
        const bool active = (rr < n_rays) && (px < a.W) && (py < a.rows);
        double c0 = 0.0, c1 = 0.0, c2 = 0.0;
        if (active) {
            const size_t sidx = ((size_t)py * a.W + px) * (size_t)spp + k;
            // streamed once: non-temporal, so the 16 B/sample stream does not displace tree nodes in L2
            const d2 jit = __builtin_nontemporal_load(reinterpret_cast<const d2*>(a.jitter) + sidx);
            // cpp/rtrender.cpp:53-58
            const double s0 = (double)px + jit.x;
            const double s1 = (double)(py + a.row0) + jit.y;
            const double ddx = ((a.p00[0] + s0 * a.sp[0]) + s1 * a.sp[3]) - a.cam[0];
            const double ddy = ((a.p00[1] + s0 * a.sp[1]) + s1 * a.sp[4]) - a.cam[1];
            const double ddz = ((a.p00[2] + s0 * a.sp[2]) + s1 * a.sp[5]) - a.cam[2];
            const double z = (ddx * ddx + ddy * ddy) + ddz * ddz;
            double dx = ddx, dy = ddy, dz = ddz;
            if (z > 0.0) {  // Eigen normalized(): v / sqrt(squaredNorm) when squaredNorm > 0
                const double nz = __builtin_sqrt(z);
                dx = ddx / nz;
                dy = ddy / nz;
                dz = ddz / nz;
            }
            Ray r;
            make_ray(r, a.cam[0], a.cam[1], a.cam[2], dx, dy, dz);
            Hit h;
            h.t = __builtin_inf();
            h.u = 0.0;
            h.v = 0.0;
            h.prim = -1;
            trace<COUNT>(a.sc, r, stk, stride, h, n_nodes, n_tris);
            shade(a.sc, r, h, c0, c1, c2);
            if (COUNT) n_hits += (h.prim >= 0);
            if (a.hit_prim) __builtin_nontemporal_store(h.prim, a.hit_prim + sidx);
            if (a.hit_t) __builtin_nontemporal_store(h.t, a.hit_t + sidx);
        }
        s_col[0][tid] = c0;
        s_col[1][tid] = c1;
        s_col[2][tid] = c2;
        __syncthreads();
        // in-order accumulation: one lane per (pixel, channel) touched by this batch
        {
            const int end = min(base + NT, n_rays);
            const int m_first = base / spp;
            const int m_last = (end - 1) / spp;
            const int work = (m_last - m_first + 1) * 3;
            for (int q = tid; q < work; q += NT) {
                const int pm = m_first + q / 3;
                const int c = q - (q / 3) * 3;
                const int lo = max(pm * spp, base) - base;
                const int hi = min((pm + 1) * spp, end) - base;
                double acc = s_acc[c][pm];
                for (int s = lo; s < hi; ++s) acc += s_col[c][s];
                s_acc[c][pm] = acc;
            }
        }
        __syncthreads();
    }

    if (tid < npix_tile) {
        const int px = tx0 + (int)compact1by1((uint32_t)tid);
        const int py = ty0 + (int)compact1by1((uint32_t)tid >> 1);
        if (px < a.W && py < a.rows) {
            // cpp/rtrender.cpp:61-65
            const double gray = (0.2126 * s_acc[0][tid] + 0.7152 * s_acc[1][tid]) + 0.0722 * s_acc[2][tid];
            const double inten = gray / (double)a.spp;
            const int gb = (int)(inten * 255.99);
            const size_t pidx = (size_t)py * a.W + px;
            a.intensity[pidx] = inten;
            const uint8_t g8 = (uint8_t)gb;
            a.rgb[pidx * 3] = g8;
            a.rgb[pidx * 3 + 1] = g8;
            a.rgb[pidx * 3 + 2] = g8;
        }
    }

    if (COUNT) {
        // wave-level reduction, then one atomic per wave and counter
        unsigned long long v0 = n_nodes, v1 = n_tris, v2 = n_hits;
        for (int off = 32; off > 0; off >>= 1) {
            v0 += __shfl_down(v0, off, 64);
            v1 += __shfl_down(v1, off, 64);
            v2 += __shfl_down(v2, off, 64);
        }
        if ((tid & 63) == 0 && a.counters) {
            atomicAdd(&a.counters[0], v0);
            atomicAdd(&a.counters[1], v1);
            atomicAdd(&a.counters[2], v2);
        }
    }
}


// =======================================================================================
// Packet traversal (the render path).
//
// All rays of an image share the camera centre, and the reference's visit order does not
// depend on the ray (always left child first, rtrayintersection.cpp:185-190), so a whole
// wave can walk the tree together: the walk itself (current node, stack, descriptors, loop
// control) is wave-uniform and runs on the scalar unit with scalar loads, while each lane
// only does the slab / Moller-Trumbore arithmetic for its own ray.  A lane takes part in a
// node's tests iff it passed the slab test of every ancestor -- exactly the nodes the
// reference pops for that ray -- tracked with one predicate and one pending bit per stack
// level per lane.  Results are identical to the per-lane walk; the work per node visit drops
// from ~85 to ~20 vector instructions.
//
// Because the origin is shared, everything that depends only on (origin, primitive) is
// computed once per frame by prep kernels, with the same IEEE operations the reference
// performs per ray:  node boxes relative to the camera (min-o, max-o), and per triangle
// e0, e2, tv = o-v0, q = tv x e0 and the t numerator e2.q  (rtrayintersection.cpp:64-72,92,100,105).
// =======================================================================================
struct alignas(128) CamTri {
    double e0[3];
    double e2[3];
    double tv[3];
    double q[3];
    double tnum;
    double pad[3];
};
static_assert(sizeof(CamTri) == 128, "camera-relative triangle record must be 128 bytes");

// The three prep passes of a frame in ONE launch (small frames are bound by the number of launches): blocks
// [0, nb_nodes) do the BLAS nodes, the next nb_tlas the TLAS nodes, the rest the triangles.  A node is handled by
// four neighbouring lanes, 16 bytes each (coalesced; a thread per 64-byte record reaches a quarter of the
// bandwidth): lanes 0..2 subtract the camera from two box coordinates, lane 3 copies the links.
struct PrepArgs {
    const rt::DevNode* nodes_in;
    rt::DevNode* nodes_out;
    long long n_nodes;
    const rt::DevNode* tlas_in;
    rt::DevNode* tlas_out;
    long long n_tlas;
    const double* tris;
    CamTri* tris_out;
    long long n_tris;
    double ox, oy, oz;
    unsigned nb_nodes, nb_tlas;  // blocks of 256 threads of the first two parts
};

__device__ __forceinline__ void prep_node_quarter(const rt::DevNode* __restrict__ in, rt::DevNode* __restrict__ out,
                                                  const long long n, const long long t, const double ox, const double oy,
                                                  const double oz) {
    if (t >= 4 * n) return;
    const int q = (int)(t & 3);
    d2 v = reinterpret_cast<const d2*>(in)[t];
    // record layout: mn0 mn1 | mn2 mx0 | mx1 mx2 | links
    if (q == 0) {
        v.x = v.x - ox;
        v.y = v.y - oy;
    } else if (q == 1) {
        v.x = v.x - oz;
        v.y = v.y - ox;
    } else if (q == 2) {
        v.x = v.x - oy;
        v.y = v.y - oz;
    }
    reinterpret_cast<d2*>(out)[t] = v;
}

__global__ __launch_bounds__(256) void prep_frame_kernel(const PrepArgs a) {
    const unsigned b = blockIdx.x;
    if (b < a.nb_nodes) {
        prep_node_quarter(a.nodes_in, a.nodes_out, a.n_nodes, (long long)b * 256 + threadIdx.x, a.ox, a.oy, a.oz);
    } else if (b < a.nb_nodes + a.nb_tlas) {
        prep_node_quarter(a.tlas_in, a.tlas_out, a.n_tlas, (long long)(b - a.nb_nodes) * 256 + threadIdx.x, a.ox, a.oy, a.oz);
    } else if (a.tris_out) {
        const long long i = (long long)(b - a.nb_nodes - a.nb_tlas) * 256 + threadIdx.x;
        if (i >= a.n_tris) return;
        const double* tp = a.tris + (size_t)i * 9;
        const double v0x = tp[0], v0y = tp[1], v0z = tp[2];
        CamTri c;
        c.e0[0] = tp[3] - v0x;
        c.e0[1] = tp[4] - v0y;
        c.e0[2] = tp[5] - v0z;
        c.e2[0] = tp[6] - v0x;
        c.e2[1] = tp[7] - v0y;
        c.e2[2] = tp[8] - v0z;
        c.tv[0] = a.ox - v0x;
        c.tv[1] = a.oy - v0y;
        c.tv[2] = a.oz - v0z;
        c.q[0] = c.tv[1] * c.e0[2] - c.tv[2] * c.e0[1];
        c.q[1] = c.tv[2] * c.e0[0] - c.tv[0] * c.e0[2];
        c.q[2] = c.tv[0] * c.e0[1] - c.tv[1] * c.e0[0];
        c.tnum = (c.e2[0] * c.q[0] + c.e2[1] * c.q[1]) + c.e2[2] * c.q[2];
        c.pad[0] = c.pad[1] = c.pad[2] = 0.0;
        a.tris_out[i] = c;
    }
}

struct PacketScene {
    const rt::DevNode* nodes;  // camera-relative boxes
    const rt::DevNode* tlas;   // camera-relative boxes
    const int32_t* slot_root;
    const uint32_t* leaf_ext;
    const CamTri* tris;
    const double* colors;
};

#define RT_CONST __attribute__((address_space(4)))
typedef const RT_CONST double* cdptr;
typedef const RT_CONST uint32_t* cuptr;
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ double sgpr_f64(const u32x16& v, const int i) {
    return __builtin_bit_cast(double, ((unsigned long long)v[2 * i + 1] << 32) | (unsigned long long)v[2 * i]);
}

typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
// A kernel argument read from the kernarg segment AT THE POINT OF USE (the packet kernels' PacketArgs /
// Packet32Args are the only argument, so a field sits at its offsetof).  The compiler otherwise loads every
// argument once and keeps it in SGPRs across the traversal loops, which need 60+ SGPRs themselves: the
// arguments were spilled to VGPR lanes and each v_readlane / v_writelane is a VALU issue slot per batch.
template <class T, int OFF>
__device__ __forceinline__ T* karg_ptr() {
    unsigned long long v;
    const auto k = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(k), "i"(OFF) : "memory");
    return reinterpret_cast<T*>(v);
}

struct SBox {  // wave-uniform (scalar) camera-relative box
    double mnx, mny, mnz, mxx, mxy, mxz;
};

__device__ __forceinline__ void sload_node(const rt::DevNode* p, SBox& b, int32_t& left, int32_t& count,
                                           int32_t& first, uint32_t& desc) {
    cdptr q = (cdptr)(const void*)p;
    b.mnx = q[0];
    b.mny = q[1];
    b.mnz = q[2];
    b.mxx = q[3];
    b.mxy = q[4];
    b.mxz = q[5];
    cuptr m = (cuptr)(const void*)p;
    left = (int32_t)m[12];
    count = (int32_t)m[13];
    first = (int32_t)m[14];
    desc = m[15];
}

// slab test against a camera-relative box; FAST uses v_min/v_max (identical to the std::min/
// std::max composition whenever no operand is NaN: the results are then only compared, and
// comparisons do not see the sign of a zero).  The caller selects FAST only for waves in which
// no lane has an infinite inverse-direction component (the only source of 0*inf = NaN).
// OCT >= 0 (only with FAST): every active lane of the wave has the same sign of ix, of iy and of iz
// (sx/sy/sz = that component is negative).  With
// mn <= mx per axis (every node bounds at least one triangle) and rounding monotone, the product
// with the near plane IS min(t_lo, t_hi) and the far one IS max(t_lo, t_hi), so the six inner
// min/max of intersect_AABB (rtrayintersection.cpp:136-141) are known without computing them.
// OCT = -1: mixed signs; else bit0/1/2 = ix/iy/iz negative (compile-time: a run-time select, even a
// wave-uniform one, is turned into v_cndmask on the products by the compiler).
template <bool FAST, int OCT>
__device__ __forceinline__ unsigned long long slab_rel(const SBox& b, const double ix, const double iy, const double iz) {
    if (FAST && OCT >= 0) {
        constexpr bool sx = (OCT & 1) != 0, sy = (OCT & 2) != 0, sz = (OCT & 4) != 0;
        const double nx = sx ? b.mxx : b.mnx, fx = sx ? b.mnx : b.mxx;
        const double ny = sy ? b.mxy : b.mny, fy = sy ? b.mny : b.mxy;
        const double nz = sz ? b.mxz : b.mnz, fz = sz ? b.mnz : b.mxz;
        const double tmin = __builtin_fmax(__builtin_fmax(nx * ix, ny * iy), nz * iz);
        const double tmax = __builtin_fmin(__builtin_fmin(fx * ix, fy * iy), fz * iz);
        // `tmin <= tmax && tmax > 0` as ONE comparison: tmax > 0 <=> tmax >= the smallest positive double
        // (fp64 denormals are not flushed on gfx950), so with m = max(tmin, 2^-1074):  m <= tmax  is false when
        // tmax <= 0 (m > 0), equals tmax > 0 when tmin <= 0, and equals tmin <= tmax (which implies tmax > 0)
        // when tmin > 0.  Same VALU count, one scalar AND less per box.
        return __builtin_amdgcn_ballot_w64(__builtin_fmax(tmin, 0x1p-1074) <= tmax);
    }
    const double t0 = b.mnx * ix, t1 = b.mxx * ix;
    const double t2 = b.mny * iy, t3 = b.mxy * iy;
    const double t4 = b.mnz * iz, t5 = b.mxz * iz;
    double tmin, tmax;
    if (FAST) {
        tmin = __builtin_fmax(__builtin_fmax(__builtin_fmin(t0, t1), __builtin_fmin(t2, t3)), __builtin_fmin(t4, t5));
        tmax = __builtin_fmin(__builtin_fmin(__builtin_fmax(t0, t1), __builtin_fmax(t2, t3)), __builtin_fmax(t4, t5));
    } else {
        tmin = std_max(std_max(std_min(t0, t1), std_min(t2, t3)), std_min(t4, t5));
        tmax = std_min(std_min(std_max(t0, t1), std_max(t2, t3)), std_max(t4, t5));
    }
    // one ballot per comparison: each v_cmp writes its lane mask straight into an SGPR pair and the
    // masks are combined on the scalar unit
    const unsigned long long m = __builtin_amdgcn_ballot_w64(tmin <= tmax) & __builtin_amdgcn_ballot_w64(tmax > 0.0);
    // FAST is only selected when every |inverse direction| < 1e100 and every |box coordinate - camera|
    // < 1e100 (host-checked): all six products are finite, so `tmin < inf` holds by construction.
    if (FAST) return m;
    return m & __builtin_amdgcn_ballot_w64(tmin < __builtin_inf());
}

// The set of lanes taking part in the current node is a wave-uniform 64-bit mask (SGPR pair);
// stack entries carry the mask of lanes that passed the pushed node's slab test.
struct WaveCounters {
    unsigned long long nodes, tris;  // wave-uniform sums over participating lanes
    unsigned long long steps;        // node records fetched by the wave (the packet's own work)
    unsigned long long tri_steps;    // triangle records fetched by the wave
};

#ifndef RT_QUICK_DISPATCH
#define RT_QUICK_DISPATCH 1
#endif
// Which port the one-pixel-per-batch instantiation's (wave-uniform) pixel arithmetic and cut address go to.  While
// the entry lists cost ~530 scalar-side instructions per batch against ~485 vector ones, the vector unit was the
// place (1: +0.8 % and +1 %, same box A/B); since the stack-free list loop (~310 against ~485) it is the scalar unit
// again (0: +1.1 %).
#ifndef RT_P64_VZERO
#define RT_P64_VZERO 0
#endif
#ifndef RT_P64_FRESH_ARGS
#define RT_P64_FRESH_ARGS 1
#endif
#ifndef RT_P64_VECTOR_CUT
#define RT_P64_VECTOR_CUT 0
#endif
constexpr uint32_t CUT_NODE_ENTRY = rt::DESC_TLAS | rt::DESC_ROOT;  // cut entry: BLAS node tested per ray (see "Tile cuts")
// A CUT_NODE_ENTRY with this bit names a LEAF whose triangle range is overridden by the NEXT word of the list (a leaf
// descriptor of a sub-range: cut_trim).  Cuts are only made for the packet kernels, whose node indices are < 2^26.
constexpr uint32_t CUT_TRIM = 0x10000000u;
constexpr uint32_t CUT_NODE_IDX = 0x03FFFFFFu;

// The triangles of a leaf descriptor (EXT: a count field of 8 names a range in leaf_ext) against the lanes of `inm`, in order, with the strict `<` on t of the reference (rtrayintersection.cpp:79-106).
// The records are addressed with a 32-bit scalar byte offset (the host sends scenes of >= 2^25 triangles to the
// per-lane kernel): one add and one compare per triangle on the scalar unit, which is this kernel's busiest.
template <bool FAST, bool COUNT, bool EXT>
__device__ __forceinline__ void test_leaf(const PacketScene& sc, const uint32_t leaf, const unsigned long long inm,
                                          const double dx, const double dy, const double dz, Hit& h, WaveCounters& wc) {
    uint32_t first = leaf & rt::DESC_TRI_MASK;
    uint32_t cnt = ((leaf >> 28) & 7u) + 1u;
    if (EXT && cnt == 8u) {
        cuptr ext = (cuptr)(const void*)sc.leaf_ext;
        const uint32_t e = first;
        first = ext[2 * e];
        cnt = ext[2 * e + 1];
    }
    if (COUNT) {
        wc.tris += (unsigned long long)cnt * (unsigned)__builtin_popcountll(inm);
        wc.tri_steps += cnt;
    }
    // two of the test's fp64 literals live in VGPRs (spare) instead of SGPR pairs (spilled to VGPR lanes and read back
    // with a vector instruction each, on the port this kernel is bound by)
    double k_eps = 1e-6, k_neg = -0x1p-300;
    asm("" : "+v"(k_eps));
    asm("" : "+v"(k_neg));
    uint32_t off = first << 7;  // sizeof(CamTri) == 128
    uint32_t end = off + (cnt << 7);
    asm volatile("" : "+s"(end));  // (a trip count would cost a second counter)
    // the lanes of `inm` (never empty) are the execution mask of the whole loop: each comparison's mask is then the
    // next nesting level's as it is, without an AND with `inm` per triangle
    if (!__builtin_amdgcn_inverse_ballot_w64(inm)) return;
    do {
        // the whole 128-byte record in one round trip (both 64-byte lines requested together): staged
        // loads (edges, then tv/q, then tnum) put up to three dependent scalar-cache misses on the
        // wave's critical path per triangle
        u32x16 ta, tb;
        asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx16 %1, %2, %3 offset:0x40\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(ta), "=&s"(tb)
                     : "s"(sc.tris), "s"(off)
                     : "memory");
        const double e0x = sgpr_f64(ta, 0), e0y = sgpr_f64(ta, 1), e0z = sgpr_f64(ta, 2);
        const double e2x = sgpr_f64(ta, 3), e2y = sgpr_f64(ta, 4), e2z = sgpr_f64(ta, 5);
        const double tvx = sgpr_f64(ta, 6), tvy = sgpr_f64(ta, 7), tvz = sgpr_f64(tb, 0);
        const double qx = sgpr_f64(tb, 1), qy = sgpr_f64(tb, 2), qz = sgpr_f64(tb, 3);
        const double tnum = sgpr_f64(tb, 4);
        // rtrayintersection.cpp:79-106, with the ray-independent parts precomputed
        const double px = dy * e2z - dz * e2y;
        const double py = dz * e2x - dx * e2z;
        const double pz = dx * e2y - dy * e2x;
        const double det = (e0x * px + e0y * py) + e0z * pz;
        bool need = det > k_eps;
        double unum = 0.0, vnum = 0.0;
        if (need) {
            unum = (tvx * px + tvy * py) + tvz * pz;
            vnum = (dx * qx + dy * qy) + dz * qz;
            // Sure rejects, decided on the numerators so that a triangle none of the wave's rays can
            // hit costs no division: with 1e-6 < det < 2^700 (FAST: every |coordinate - camera| <
            // 1e100 and |direction| <= 1 bound det by 2.4e201) the rounded 1/det is a normal number
            // >= 2^-700 with relative error 2^-53, so
            //   num < -2^-300        =>  num * inv is a normal negative number: `>= 0` fails;
            //   unum + vnum > det * (1 + 2^-30)  =>  had both u in [0,1] and v >= 0 held, u + v would
            //                            exceed 1 by 2^-31, far beyond the four roundings involved.
            // NaNs compare false and take the full path.  Lanes that are not sure rejects compute
            // exactly what they did before, so results are unchanged.
            if (FAST) {
                const double lo = __builtin_fmin(unum, vnum);
                need = !(lo < k_neg || (unum + vnum) > det * (1.0 + 0x1p-30));
            }
        }
        if (need) {
            const double inv = 1.0 / det;
            const double u = unum * inv;
            const double v = vnum * inv;
            const double t = tnum * inv;
            const bool ok = (u >= 0.0) && (u <= 1.0) && (v >= 0.0) && ((u + v) <= 1.0) && (t >= 0.0) &&
                            (t <= __builtin_inf());
            if (ok && t < h.t) {
                h.t = t;
                h.u = u;
                h.v = v;
                h.prim = (int32_t)(off >> 7);
            }
        }
        off += 128u;
    } while (off != end);
}

// Entry list (tile cut, see cut_level_kernel): n_entries < 0 -- walk from the TLAS root (the reference's own walk);
// otherwise lane 1 + i of `vlist` holds entry i, a descriptor at which the walk (re)starts with every valid lane
// taking part, in the reference's visit order.  The nodes above the entries are not tested: the cut guarantees
// that every ray of the tile reaches exactly the same leaves, in the same order, either way.
template <bool FAST, int OCT, bool COUNT>
__device__ __forceinline__ void trace_packet(const PacketScene& sc, const unsigned long long valid_mask,
                                          const double dx, const double dy, const double dz, const double ix,
                                          const double iy, const double iz, uint4* wstk, Hit& h, WaveCounters& wc,
                                          const uint32_t vlist, int n_entries, const bool simple = false) {
    if (simple) {
        // CUT_SIMPLE list (cut_trim): nothing but leaves every ray reaches and trimmed leaves -- no stack, no
        // dispatch: ~20 scalar instructions per entry
        for (int e = 1; e <= n_entries; ++e) {
            uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)vlist, e);
            unsigned long long m = valid_mask;
            if (!(w & rt::DESC_LEAF)) {
                SBox b;
                cdptr q = (cdptr)(const void*)(sc.nodes + (w & CUT_NODE_IDX));
                b.mnx = q[0];
                b.mny = q[1];
                b.mnz = q[2];
                b.mxx = q[3];
                b.mxy = q[4];
                b.mxz = q[5];
                ++e;
                w = (uint32_t)__builtin_amdgcn_readlane((int)vlist, e);
                if (COUNT) wc.nodes += (unsigned)__builtin_popcountll(valid_mask);
                m &= slab_rel<FAST, OCT>(b, ix, iy, iz);
            }
            if (m != 0ull) test_leaf<FAST, COUNT, false>(sc, w, m, dx, dy, dz, h, wc);
        }
        return;
    }
    int sp = 0;
    int ei = 1;
    if (n_entries < 0) {
        // the reference's own walk: the TLAS root is the one thing pending
        n_entries = 0;
        wstk[sp++] = make_uint4(rt::DESC_TLAS, (uint32_t)valid_mask, (uint32_t)(valid_mask >> 32), 0u);
    }
    // One round per pending item (stack first, else the next entry of the cut), written as a forward-only body with
    // a single back edge and ONE call site of the triangle test: loops with several exits make the compiler carry
    // guard variables, two call sites two copies of the hit -- on the scalar unit, this kernel's busiest.  A trimmed
    // leaf, most of a fine cut, costs ~25 scalar instructions this way (it was ~75).
    for (;;) {
        uint32_t cur;
        unsigned long long inm = valid_mask;  // an entry: every valid lane takes part again
        if (sp == 0) {
            if (ei > n_entries) break;
            cur = (uint32_t)__builtin_amdgcn_readlane((int)vlist, ei);
            ++ei;
        } else {
            --sp;
            const uint4 e = wstk[sp];
            cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.x);
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.y);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.z);
            inm = ((unsigned long long)hi << 32) | lo;
        }
        // ---- items whose own box is tested first ----
        if (!(cur & rt::DESC_LEAF) && (cur & CUT_NODE_ENTRY) == CUT_NODE_ENTRY) {
            // entry of a cut: a BLAS node whose own box is tested per ray (the reference tests it when it pops the
            // node); lanes that pass walk on from it -- or, CUT_TRIM, test the triangles the next word names
            SBox b;
            uint32_t desc;
            {
                cdptr q = (cdptr)(const void*)(sc.nodes + (cur & CUT_NODE_IDX));
                b.mnx = q[0];
                b.mny = q[1];
                b.mnz = q[2];
                b.mxx = q[3];
                b.mxy = q[4];
                b.mxz = q[5];
                if (cur & CUT_TRIM) {
                    desc = (uint32_t)__builtin_amdgcn_readlane((int)vlist, ei);
                    ++ei;
                } else {
                    desc = ((cuptr)(const void*)q)[15];
                }
            }
            if (COUNT) wc.nodes += (unsigned)__builtin_popcountll(inm);
            inm &= slab_rel<FAST, OCT>(b, ix, iy, iz);
            cur = inm != 0ull ? desc : rt::DESC_NONE;
        } else if (!(cur & rt::DESC_LEAF) && (cur & (rt::DESC_TLAS | rt::DESC_ROOT))) {
            SBox b;
            int32_t left, count, first;
            uint32_t desc;
            const bool tlas = (cur & rt::DESC_TLAS) != 0u;
            if (tlas) {
                sload_node(sc.tlas + (cur & 0x3FFFFFFFu), b, left, count, first, desc);
            } else {
                const RT_CONST int32_t* sr = (const RT_CONST int32_t*)(const void*)sc.slot_root;
                const int32_t root = sr[cur & 0x1FFFFFFFu];
                sload_node(sc.nodes + root, b, left, count, first, desc);  // walk into this BLAS
            }
            if (COUNT) wc.nodes += (unsigned)__builtin_popcountll(inm);
            const unsigned long long hm = slab_rel<FAST, OCT>(b, ix, iy, iz) & inm;
            if (tlas) {
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
                cur = rt::DESC_NONE;
            } else {
                cur = hm != 0ull ? desc : rt::DESC_NONE;
                inm = hm;
            }
        }
        // ---- descend through internal BLAS nodes (the hot loop of a walk from the root) ----
        // One straight-line body with a single back edge: the scalar unit issues as slowly as the fp64
        // VALU (one instruction per SIMD every 4 cycles), so SALU instructions count like VALU ones.
        // The next node / mask are scalar selects; `cur = DESC_NONE` (all bits) when no lane hit either child.
        while ((cur & (rt::DESC_LEAF | rt::DESC_TLAS | rt::DESC_ROOT)) == 0u) {
            // both children: the whole 128-byte pair as two 64-byte scalar loads at base + (cur << 6)
            // (host guarantees the node array is < 4 GB), tested unconditionally -- lanes outside `inm`
            // compute garbage that the mask discards
            SBox bl, br;
            uint32_t dl, dr;
            {
                u32x16 na, nb;
                const uint32_t off = cur << 6;
                asm volatile(
                    "s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx16 %1, %2, %3 offset:0x40\n\ts_waitcnt lgkmcnt(0)"
                    : "=&s"(na), "=&s"(nb)
                    : "s"(sc.nodes), "s"(off)
                    : "memory");
                bl.mnx = sgpr_f64(na, 0);
                bl.mny = sgpr_f64(na, 1);
                bl.mnz = sgpr_f64(na, 2);
                bl.mxx = sgpr_f64(na, 3);
                bl.mxy = sgpr_f64(na, 4);
                bl.mxz = sgpr_f64(na, 5);
                br.mnx = sgpr_f64(nb, 0);
                br.mny = sgpr_f64(nb, 1);
                br.mnz = sgpr_f64(nb, 2);
                br.mxx = sgpr_f64(nb, 3);
                br.mxy = sgpr_f64(nb, 4);
                br.mxz = sgpr_f64(nb, 5);
                dl = na[15];
                dr = nb[15];
            }
            if (COUNT) {
                wc.nodes += 2u * (unsigned)__builtin_popcountll(inm);
                wc.steps += 2u;
            }
            const unsigned long long ml = slab_rel<FAST, OCT>(bl, ix, iy, iz) & inm;
            const unsigned long long mr = slab_rel<FAST, OCT>(br, ix, iy, iz) & inm;
            // flags as opaque scalar integers (a shared i1 would travel through a VGPR)
            uint32_t lnz = (uint32_t)ml | (uint32_t)(ml >> 32);
            uint32_t rnz = (uint32_t)mr | (uint32_t)(mr >> 32);
            uint32_t both = lnz < rnz ? lnz : rnz;  // s_min_u32: non-zero iff both are
            asm volatile("" : "+s"(lnz), "+s"(rnz), "+s"(both));
            if (both != 0u) wstk[sp++] = make_uint4(dr, (uint32_t)mr, (uint32_t)(mr >> 32), 0u);
            uint32_t nxt = (lnz != 0u) ? dl : dr;
            inm = (lnz != 0u) ? ml : mr;
            cur = ((lnz | rnz) != 0u) ? nxt : rt::DESC_NONE;
        }
        // ---- a leaf, or nothing ----
        if (cur != rt::DESC_NONE) test_leaf<FAST, COUNT, true>(sc, cur, inm, dx, dy, dz, h, wc);
    }
}

// Per-frame camera block, read with scalar loads where it is needed (ray generation) instead of
// living in SGPRs across the traversal loop.
struct FrameParams {
    double cam[3], p00[3], sp[6];
};

struct PacketArgs {
    PacketScene sc;
    FrameParams fpv;  // camera block, read from the kernarg segment where it is used
    int32_t W, spp;
    int32_t row0, rows;
    int32_t tile_log2;
    int32_t tiles_x, n_tiles;
    const double* jitter;
    double* intensity;
    uint8_t* rgb;
    int32_t* hit_prim;
    double* hit_t;
    unsigned long long* counters;
    int32_t pair64;   // host: 1 + log2(spp / 64) when spp = 64 * 2^n, tiles are 8x8 and stack_need <= PAIR64_STACK; else 0
    // interleave: pixel i of wave w is Morton index 4 i + w of the tile, so the four waves of a workgroup walk a
    // 2x2 pixel block at about the same time and meet each other's node and triangle lines in the scalar data
    // cache (its 16 KB do not survive until the same wave's next pixel); 0: wave w owns the compact quadrant w
    int32_t interleave;
    int32_t fast_ok;  // host: >= 1: every |node coordinate - camera| < 1e100; 2: and every box has min <= max
    // cuts (cut_level_kernel): level l = blocks of (1 << l)^2 pixels, cut_tx[l] blocks per row, CUT_WORDS words each;
    // levels cut_lo .. tile_log2 are present (cut_lo < 0: no cuts, every batch walks from the root)
    const uint32_t* cut_lvl[5];
    int32_t cut_tx[5];
    int32_t cut_eff[5];  // level whose cuts cut_lvl[l] / cut_tx[l] are: l, or the next coarser one that was made
    int32_t cut_lo;
    // generic instantiation: floor(r / spp) for 0 <= r < spp + 128 without a division -- spp >= 64: r >= spp (0 or 1);
    // else (r * spp_magic) >> 16 with spp_magic = 65536 / spp + 1 (exact while r * spp < 65536)
    uint32_t spp_magic;
};

// blockIdx -> tile.  `tile` already gives XCD x (blocks b == x mod 8) a contiguous run of tiles.  When the
// image divides evenly (tiles_x a multiple of 16, tiles_y of 128) the run is walked in 16x16 blocks of tiles
// instead of whole tile rows: the ~256 workgroups in flight on an XCD then cover a compact 128x128-pixel
// square and share more of the tree in the scalar cache and L2 (3 % on the 4096^2 bench frame).
__device__ __forceinline__ void tile_coords(const int tile, const int n_tiles, const int tiles_x, int& tix, int& tiy) {
    tix = tile % tiles_x;
    tiy = tile / tiles_x;
    const int per = n_tiles >> 3;
    const int tiles_y = n_tiles / tiles_x;
    if ((tiles_x & 15) == 0 && (tiles_y & 127) == 0 && (int)blockIdx.x < (per << 3)) {
        const int xcd = (int)blockIdx.x & 7, local = (int)blockIdx.x >> 3;
        const int bpr = tiles_x >> 4;  // blocks per row of tiles
        const int blk = local >> 8, in = local & 255;
        tix = ((blk % bpr) << 4) + (in & 15);
        tiy = xcd * (tiles_y >> 3) + ((blk / bpr) << 4) + (in >> 4);
    }
}

// =======================================================================================
// Tile cuts.  All rays of an image start at the camera, and the rays through one square block of pixels fill a
// narrow pyramid, so most of the tree can be classified ONCE PER BLOCK instead of once per ray:
//   * inside a BLAS a child's box lies inside its parent's (the builder takes unions, rtbvh.cpp:81-94, and the
//     camera-relative subtraction and the slab products round monotonically), so in the slab test (no NaN operand)
//     tmin(parent) <= tmin(child)  and  tmax(child) <= tmax(parent)  hold exactly: a ray that passes a node's test
//     passes the tests of all its BLAS ancestors.  The triangles a ray is tested against are therefore those of the
//     leaves whose OWN box it passes -- internal BLAS tests only prune -- and any walk that reaches the same leaves
//     in the same order returns the same hit, bit for bit;
//   * TLAS nodes guard BLAS slots they need not contain (quirk 1, rtbvh.cpp:552-560), so their tests are kept
//     unless the whole block provably passes or fails them.
// A cut is a list of entries in visit order at which the walk of a block's rays starts instead of at the TLAS root:
//     descriptor (leaf / internal BLAS node)   every ray of the block passes this node: no test, walk on from it
//     CUT_NODE | index                         BLAS node tested per ray (its own box), then walked from
//     DESC_TLAS | index                        TLAS node tested per ray, as the reference does when it pops it
// cut_level_kernel builds the cuts of one quadtree level (blocks of 2^l x 2^l pixels, one thread per block) from the
// parent level's (the top level, one block, starts at the TLAS root): inherited entries are re-classified against
// the smaller pyramid and refined -- a node is replaced by those of its children that are not sure to fail -- until
// only leaves are left or the list is full (any prefix of the refinement is a valid cut).  The render kernel
// takes, per 64-ray batch, the cut of the smallest block that holds the batch's pixels: one pixel at spp >= 64.
//
// Classification of a box against a block: the rays are  dd(s) = p00 + s0 row0 + s1 row1 - cam  (un-normalised;
// the slab test is invariant under the positive scale), s in the block's sample rectangle (pixel +- 0.5, padded),
// AFFINE in s, so |dd_a| takes its extremes at the rectangle's corners.  Per axis a the component either keeps its
// sign over the block and stays away from zero (relative to |dd| and to the rounding of the sample position) --
// then T_a(s) = plane_a / |dd_a(s)| lies between plane_a * wlo_a and plane_a * whi_a, w = bounds of 1 / |dd_a| --
// or it is "mixed" (|dd_a| <= dmax_a, any sign): with the camera strictly inside the box's slab T_near_a <= 0 and
// T_far_a >= min(-mn_a, mx_a) / dmax_a, with the camera outside it a ray that gets into the slab at all does so
// at T_near_a >= (distance of the slab) / dmax_a and leaves it at T_far_a >= (distance of its far plane) / dmax_a.
// The bounds are widened by 2e-5 relative (the per-ray test rounds at ~1e-15), so
//     sure fail:  the lower bound of some T_near_a exceeds the upper bound of some T_far_b, or a far plane lies
//                 behind the camera on a sign-keeping axis;
//     sure pass:  every upper bound of T_near_a is <= every lower bound of T_far_b (a != b; T_near_a <= T_far_a
//                 holds by construction) and every T_far > 0
// for every ray of the block whose inverse direction is tame; batches with other rays (|1/d| >= 1e100, offsets
// outside +-0.5, signs that contradict the header) walk from the root.  The bounds treat the axes independently,
// which is loose for big blocks and tight for the few-pixel blocks the cuts are made for.
// =======================================================================================
constexpr int CUT_WORDS = 32;                // header + entries per block (128 bytes)
constexpr int CUT_CAP = CUT_WORDS - 1;       // entries
constexpr uint32_t CUT_VALID = 0x80000000u;  // header: bit31 valid, bit 16 CUT_SIMPLE, bits 13..8 axis kinds (2 bits each), bits 7..0 entries
constexpr uint32_t CUT_SIMPLE = 0x00010000u;  // header: every entry is a leaf (<= 7 triangles) or a CUT_TRIM pair (cut_trim)
constexpr uint32_t CUT_NODE = rt::DESC_TLAS | rt::DESC_ROOT;  // entry: BLAS node (bits 25..0) tested per ray
constexpr int CUT_THREADS = 64;  // one wave, 8 KB of LDS: fits the hole a retiring render workgroup (4 waves, 20 KB) leaves
constexpr int CUT_MAX_LEVELS = 5;  // levels the render kernel reads: blocks of 1, 2, 4, 8, 16 pixels

struct TileRays {
    int kind[3];             // 0: dd_a > 0 over the block, 1: < 0, 2: mixed
    double wlo[3], whi[3];   // bounds of 1 / |dd_a| over the block, widened by the margin (mixed: whi = +inf)
};

// 0 = every ray of the block fails the box, 1 = every ray passes, 2 = unsure.  mn/mx: camera-relative box.
__host__ __device__ inline int cut_classify(const double* mn, const double* mx, const TileRays& tr) {
    const double inf = __builtin_inf();
    double nlo[3], nhi[3], flo[3], fhi[3];
    bool pass = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (tr.kind[a] == 2) {
            const double in = __builtin_fmin(-mn[a], mx[a]);  // > 0 iff the camera is strictly inside the slab
            if (in > 0.0) {
                nlo[a] = -inf;
                nhi[a] = 0.0;
                flo[a] = in * tr.wlo[a];
            } else {
                nlo[a] = __builtin_fmax(mn[a], -mx[a]) * tr.wlo[a];   // >= 0
                nhi[a] = inf;
                flo[a] = __builtin_fmax(mx[a], -mn[a]) * tr.wlo[a];
                pass = false;
            }
            fhi[a] = inf;
        } else {
            const double nr = tr.kind[a] ? -mx[a] : mn[a];
            const double fr = tr.kind[a] ? -mn[a] : mx[a];
            if (!(fr > 0.0)) return 0;  // T_far_a <= 0 for every ray of the block
            nhi[a] = nr * (nr >= 0.0 ? tr.whi[a] : tr.wlo[a]);
            nlo[a] = nr * (nr >= 0.0 ? tr.wlo[a] : tr.whi[a]);
            flo[a] = fr * tr.wlo[a];
            fhi[a] = fr * tr.whi[a];
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            if (a == b) continue;  // T_near_a <= T_far_a holds for every ray by construction
            if (nlo[a] > fhi[b]) return 0;  // T_near_a > T_far_b for every ray of the block
            pass = pass && (nhi[a] <= flo[b]);
        }
    return pass ? 1 : 2;
}

struct CutArgs {
    const rt::DevNode* nodes;  // camera-relative
    const rt::DevNode* tlas;   // camera-relative
    const int32_t* slot_root;
    FrameParams fp;
    int32_t W, rows, row0;
    int32_t lvl;               // blocks of (1 << lvl)^2 pixels
    int32_t tiles_x, n_tiles;  // of this level
    const uint32_t* parent;    // cuts of level lvl + parent_shift (parent_tiles_x blocks per row), or null: start at the root
    int32_t parent_tiles_x;
    int32_t parent_shift;
    double mag;                // bound of the magnitudes any sample position of the image is computed from
    int32_t cap;               // entries a list of this level may hold (<= CUT_CAP)
    uint32_t* cuts;            // n_tiles * CUT_WORDS
    const CamTri* tris;        // camera-relative triangle records: the lists' leaves are trimmed (cut_trim; the finest
                               // level only -- no level may inherit from a trimmed one), or null
};

// max over the image's sample positions and components of |p00| + |s0 row0| + |s1 row1| + |cam|
__host__ __device__ inline double cut_magnitude(const FrameParams& fp, const int W, const int row0, const int rows) {
    double mag = 0.0;
    const double s0 = (double)W + 1.0, s1 = (double)(row0 + rows) + 1.0;
    for (int k = 0; k < 3; ++k) {
        const double v = __builtin_fabs(fp.p00[k]) + __builtin_fabs(s0 * fp.sp[k]) + __builtin_fabs(s1 * fp.sp[3 + k]) +
                         __builtin_fabs(fp.cam[k]);
        mag = v > mag ? v : mag;
        if (!(v == v)) mag = __builtin_inf();  // NaN camera: no cuts
    }
    return mag;
}

// The un-normalised rays through the corners of a block's sample rectangle.
__host__ __device__ inline void cut_corner_rays(const CutArgs& a, const int tix, const int tiy, double (&dd)[4][3]) {
    const int ts = 1 << a.lvl;
    // sample positions px + [-0.5, 0.5] (cpp/rtrender.cpp:53-56), a little padded; blocks of the image's last
    // column / row are cut to the image
    const int x1 = (tix * ts + ts - 1 < a.W - 1) ? tix * ts + ts - 1 : a.W - 1;
    const int y1 = (tiy * ts + ts - 1 < a.rows - 1) ? tiy * ts + ts - 1 : a.rows - 1;
    const double pad = 0.5 + 0x1p-20;
    const double s0v[2] = {(double)(tix * ts) - pad, (double)x1 + pad};
    const double s1v[2] = {(double)(tiy * ts + a.row0) - pad, (double)(y1 + a.row0) + pad};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const double s0 = s0v[c & 1], s1 = s1v[c >> 1];
#pragma unroll
        for (int k = 0; k < 3; ++k) dd[c][k] = ((a.fp.p00[k] + s0 * a.fp.sp[k]) + s1 * a.fp.sp[3 + k]) - a.fp.cam[k];
    }
}

// The sample rectangle of a block and its corner rays.  false: no cut for this block (degenerate camera).
__host__ __device__ inline bool cut_tile_rays(const CutArgs& a, const int tix, const int tiy, TileRays& tr) {
    double dd[4][3];
    const double mag = a.mag;  // (host: cut_magnitude) bounds what the per-ray sample position is computed from
    cut_corner_rays(a, tix, tiy, dd);
    double big = 0.0;
    bool finite = true;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double v = __builtin_fabs(dd[c][k]);
            big = __builtin_fmax(big, v);
            finite = finite && (v < 1e100);  // NaN / inf -> false
        }
    if (!finite || !(big > 0.0) || !(mag < 1e100)) return false;
    // a component "keeps its sign" when it does so with room: relative to the direction (no grazing rays) and to
    // the rounding of the sample position (a few ulps of `mag`)
    const double eta = 0x1p-50 * mag;
    const double floor_abs = __builtin_fmax(1e-4 * big, 1e7 * eta);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const bool ng = dd[0][k] < 0.0;
        bool uniform = true;
        double lo = __builtin_inf(), hi = 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double v = ng ? -dd[c][k] : dd[c][k];
            uniform = uniform && (v >= floor_abs);
            lo = __builtin_fmin(lo, __builtin_fabs(dd[c][k]));
            hi = __builtin_fmax(hi, __builtin_fabs(dd[c][k]));  // |dd_k(s)| is convex in s: its maximum is at a corner
        }
        tr.kind[k] = uniform ? (ng ? 1 : 0) : 2;
        // sign-keeping: |dd_k| >= 1e7 eta at the corners, the rounding of the sample position moves it by < 1e-7
        // relative; mixed: the bound of |dd_k| gets the absolute slack
        const double dmax = uniform ? hi : hi * (1.0 + 1e-5) + 1e7 * eta;
        tr.wlo[k] = (1.0 / dmax) * (1.0 - 2e-5);
        tr.whi[k] = uniform ? (1.0 / lo) * (1.0 + 2e-5) : __builtin_inf();
    }
    return true;
}

// One block's cut: a depth-first refinement of the inherited list (the top level: of the TLAS root) that emits the
// entries in visit order.  `stk`: CUT_WORDS words, element i at stk[i * stride] (the kernel interleaves the threads
// of a workgroup in LDS; the host twin passes stride 1).  Every step takes one waiting entry and, unless it is
// already known to be passed by every ray, loads ONE node record and classifies ONE box -- the same work in every
// lane of a wave, whatever the kinds of their entries.  An entry is replaced by its children only while the list
// is sure to fit (entries written + entries waiting + 2 <= cap), so the walk never runs out of room: whatever is
// not refined is emitted as it is, which is a valid cut too.
__host__ __device__ inline void cut_refine(const CutArgs& a, const int tile, uint32_t* const stk, const int stride) {
    const int tix = tile % a.tiles_x, tiy = tile / a.tiles_x;
    uint32_t* const out = a.cuts + (size_t)tile * CUT_WORDS;
    TileRays tr;
    if (!cut_tile_rays(a, tix, tiy, tr)) {
        out[0] = 0u;
        return;
    }
    int sp = 0;  // entries waiting, the next one on top
    const uint32_t* par = nullptr;
    if (a.parent) {
        par = a.parent + ((size_t)(tiy >> a.parent_shift) * a.parent_tiles_x + (tix >> a.parent_shift)) * CUT_WORDS;
        if (!(par[0] & CUT_VALID)) par = nullptr;
    }
    if (par) {
        const int pn = (int)(par[0] & 0xffu);
        for (int i = pn - 1; i >= 0; --i) stk[(sp++) * stride] = par[1 + i];
    } else {
        stk[0] = rt::DESC_TLAS;
        sp = 1;
    }
    int n = 0;  // entries written
    while (sp > 0) {
        --sp;
        const uint32_t e = stk[sp * stride];
        const uint32_t kind = e & CUT_NODE;
        const bool room = n + sp + 2 <= a.cap;
        if ((e & rt::DESC_LEAF) || kind == 0u) {
            // every ray of the block passes this node: a leaf is final, an internal node (e = index of its left
            // child) gives way to its children, whose boxes are classified when their turn comes
            if (!(e & rt::DESC_LEAF) && room) {
                stk[(sp++) * stride] = CUT_NODE | (e + 1u);
                stk[(sp++) * stride] = CUT_NODE | e;
            } else {
                out[1 + n++] = e;
            }
            continue;
        }
        // an entry with a box of its own: TLAS node, root of a BLAS slot (its test is the test of that node) or
        // BLAS node
        const bool is_tlas = kind == rt::DESC_TLAS;
        uint32_t idx = e & 0x1FFFFFFFu;
        if (kind == rt::DESC_ROOT) idx = (uint32_t)a.slot_root[idx];
        const rt::DevNode* const np = is_tlas ? a.tlas + (e & 0x3FFFFFFFu) : a.nodes + idx;
        double mn[3], mx[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            mn[k] = np->mn[k];
            mx[k] = np->mx[k];
        }
        const int32_t left = np->left, count = np->count, first = np->first;
        const uint32_t desc = np->desc;
        const int cls = cut_classify(mn, mx, tr);
        if (cls == 0) continue;  // no ray of the block passes it
        if (is_tlas) {
            // kept (tested per ray, as the reference does when it pops the node) unless every ray passes it
            if (cls == 1 && left >= 0 && room) {
                stk[(sp++) * stride] = rt::DESC_TLAS | (uint32_t)(left + 1);
                stk[(sp++) * stride] = rt::DESC_TLAS | (uint32_t)left;
            } else if (cls == 1 && left < 0 && count >= 1 && count <= 2 && room) {
                for (int q = count - 1; q >= 0; --q) stk[(sp++) * stride] = rt::DESC_ROOT | (uint32_t)(first + q);
            } else {
                out[1 + n++] = e;
            }
            continue;
        }
        if (!(desc & rt::DESC_LEAF) && room) {
            // internal: gives way to its children (its own test, sure or not, only prunes), the left one on top
            stk[(sp++) * stride] = CUT_NODE | (desc + 1u);
            stk[(sp++) * stride] = CUT_NODE | desc;
        } else {
            out[1 + n++] = (cls == 1) ? desc : (CUT_NODE | idx);
        }
    }
    out[0] = CUT_VALID | ((uint32_t)tr.kind[0] << 8) | ((uint32_t)tr.kind[1] << 10) | ((uint32_t)tr.kind[2] << 12) |
             (uint32_t)n;
}

// ---- triangles of a block's leaves that no ray of the block can hit ----
// A leaf's triangles are tested one after the other with a strict `<` on t (tri_test above): a triangle that every
// ray of the block misses changes nothing, so the list may name the leaf's remaining triangles instead of the leaf.
// With the record's values (e0, e2, tv, q: the per-ray test reads exactly these) and the true ray d = dd / |dd|,
//     det = d . (e2 x e0),   unum = d . (e2 x tv),   vnum = d . q
// in exact arithmetic, and the test fails when det <= 1e-6, unum / det < 0, vnum / det < 0 or (unum + vnum) / det > 1.
// Each numerator times |dd| is AFFINE in the sample position, so over the block's sample rectangle it takes its
// extremes at the corners; the triangle is dropped when at all four corners, by a margin,
//     L_d < 0,  or  L_u < 0,  or  L_v < 0,  or  L_u + L_v - L_d > 0          (L_x = dd . N_x).
// Margin: the per-ray values differ from the exact ones by the rounding of the sample position (|dd_fl - dd| <= eta
// = 2^-50 mag per component), of the normalisation (3 ulp) and of the products and sums (< 16 ulp of the product of
// the norms M_x); the margin taken is M_x (1e-9 max|dd| + 4 eta), five orders of magnitude more, so that the rounded
// numerator keeps its sign with |value| > 0.9e-9 M_x -- a normal number, and so is its product with 1 / det
// (1e-6 < det < 2.4e201) as long as M_x > 1e-60, which is asked for; for u + v > 1 the rounded numerators exceed
// det by 0.9e-9 S, S = M_u + M_v + M_d >= det, against < 1e-14 S of rounding in the two quotients and their sum.
// Blocks without a sign-keeping axis (|dd| may get close to zero) are left alone.  NaN / inf compare false: kept.
// Bit m of the result: triangle m of the range is kept.
__host__ __device__ inline uint32_t cut_live_triangles(const CamTri* tris, const uint32_t first, const uint32_t cnt,
                                                      const double (&dd)[4][3], const double slack) {
    uint32_t live = 0u;
    for (uint32_t m = 0; m < cnt; ++m) {
        const CamTri& t = tris[first + m];
        const double e0x = t.e0[0], e0y = t.e0[1], e0z = t.e0[2];
        const double e2x = t.e2[0], e2y = t.e2[1], e2z = t.e2[2];
        const double tvx = t.tv[0], tvy = t.tv[1], tvz = t.tv[2];
        const double qx = t.q[0], qy = t.q[1], qz = t.q[2];
        // N_d = e2 x e0, N_u = e2 x tv
        const double ndx = e2y * e0z - e2z * e0y, ndy = e2z * e0x - e2x * e0z, ndz = e2x * e0y - e2y * e0x;
        const double nux = e2y * tvz - e2z * tvy, nuy = e2z * tvx - e2x * tvz, nuz = e2x * tvy - e2y * tvx;
        const double a0 = __builtin_fabs(e0x) + __builtin_fabs(e0y) + __builtin_fabs(e0z);
        const double a2 = __builtin_fabs(e2x) + __builtin_fabs(e2y) + __builtin_fabs(e2z);
        const double at = __builtin_fabs(tvx) + __builtin_fabs(tvy) + __builtin_fabs(tvz);
        const double m_d = a0 * a2, m_u = at * a2, m_v = __builtin_fabs(qx) + __builtin_fabs(qy) + __builtin_fabs(qz);
        const double m_s = (m_d + m_u) + m_v;
        bool d_neg = true, u_neg = m_u > 1e-60, v_neg = m_v > 1e-60, w_pos = m_s > 1e-60;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double ld = (dd[c][0] * ndx + dd[c][1] * ndy) + dd[c][2] * ndz;
            const double lu = (dd[c][0] * nux + dd[c][1] * nuy) + dd[c][2] * nuz;
            const double lv = (dd[c][0] * qx + dd[c][1] * qy) + dd[c][2] * qz;
            d_neg = d_neg && (ld < -(m_d * slack));
            u_neg = u_neg && (lu < -(m_u * slack));
            v_neg = v_neg && (lv < -(m_v * slack));
            w_pos = w_pos && (((lu + lv) - ld) > m_s * slack);
        }
        if (!(d_neg || u_neg || v_neg || w_pos)) live |= 1u << m;
    }
    return live;
}

// Rewrites a finished list (cut_refine) with its leaves trimmed.  Leaves of <= 7 triangles only (others stay):
//   sure leaf (descriptor)     -> the runs of its live triangles as leaf descriptors of their own, in order, while
//                                 the list has room, else the one range from the first to the last live triangle;
//   leaf tested per ray        -> CUT_NODE | CUT_TRIM | index, then the descriptor of that one range;
//   no live triangle           -> the entry goes (with its box test).
// Never longer than CUT_CAP: an entry takes more than one word only while every entry after it still has one.
// `stk` (CUT_WORDS words, `stride` apart) holds the old list meanwhile.
__host__ __device__ inline void cut_trim(const CutArgs& a, const int tile, uint32_t* const stk, const int stride) {
    uint32_t* const out = a.cuts + (size_t)tile * CUT_WORDS;
    const uint32_t hdr = out[0];
    if (!(hdr & CUT_VALID)) return;
    const int n = (int)(hdr & 0xffu);
    const int kinds = (int)((hdr >> 8) & 0x3fu);
    if ((kinds & 3) == 2 && ((kinds >> 2) & 3) == 2 && ((kinds >> 4) & 3) == 2) return;  // no sign-keeping axis
    const int tix = tile % a.tiles_x, tiy = tile / a.tiles_x;
    double dd[4][3];
    cut_corner_rays(a, tix, tiy, dd);
    double ddmax = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        ddmax = __builtin_fmax(ddmax, (__builtin_fabs(dd[c][0]) + __builtin_fabs(dd[c][1])) + __builtin_fabs(dd[c][2]));
    const double slack = 1e-9 * ddmax + 4.0 * (0x1p-50 * a.mag);
    if (!(slack < 1e100)) return;
    for (int i = 0; i < n; ++i) stk[i * stride] = out[1 + i];
    int w = 0;
    bool simple = true;
    for (int i = 0; i < n; ++i) {
        const uint32_t e = stk[i * stride];
        const int rest = n - 1 - i;  // every later entry can do with one word
        uint32_t leaf = 0u;
        bool per_ray = false;
        if (e & rt::DESC_LEAF) {
            leaf = e;
        } else if ((e & CUT_NODE) == CUT_NODE) {
            const uint32_t d = a.nodes[e & CUT_NODE_IDX].desc;
            if (d & rt::DESC_LEAF) {
                leaf = d;
                per_ray = true;
            }
        }
        const uint32_t cnt = ((leaf >> 28) & 7u) + 1u;
        if (leaf == 0u || cnt == 8u) {  // not a leaf, or one with its range in leaf_ext: as it is
            out[1 + w++] = e;
            simple = false;
            continue;
        }
        const uint32_t first = leaf & rt::DESC_TRI_MASK;
        const uint32_t live = cut_live_triangles(a.tris, first, cnt, dd, slack);
        if (live == 0u) continue;
        int lo = 0, hi = (int)cnt - 1;
        while (!((live >> lo) & 1u)) ++lo;
        while (!((live >> hi) & 1u)) --hi;
        const uint32_t range = rt::DESC_LEAF | ((uint32_t)(hi - lo) << 28) | (first + (uint32_t)lo);
        if (per_ray) {
            // (a leaf with every triangle live is written as a pair too: the list may then be CUT_SIMPLE)
            if (w + 2 + rest <= CUT_CAP) {
                out[1 + w++] = e | CUT_TRIM;
                out[1 + w++] = range;
            } else {
                out[1 + w++] = e;
                simple = false;
            }
            continue;
        }
        int runs = 0;
        for (int m = lo; m <= hi; ++m) runs += ((live >> m) & 1u) && !(m > lo && ((live >> (m - 1)) & 1u));
        if (runs > 1 && w + runs + rest <= CUT_CAP) {
            int m = lo;
            while (m <= hi) {
                if (!((live >> m) & 1u)) {
                    ++m;
                    continue;
                }
                int m1 = m;
                while (m1 + 1 <= hi && ((live >> (m1 + 1)) & 1u)) ++m1;
                out[1 + w++] = rt::DESC_LEAF | ((uint32_t)(m1 - m) << 28) | (first + (uint32_t)m);
                m = m1 + 1;
            }
        } else {
            out[1 + w++] = range;
        }
    }
    out[0] = (hdr & ~0xffu) | (uint32_t)w | (simple ? CUT_SIMPLE : 0u);
}

// One thread per block.  A wave takes an 8 x 8 patch of blocks (16 parents, neighbouring parts of the tree: its
// loads hit each other's lines in the vector L1), a workgroup a row of four patches.
__global__ __launch_bounds__(CUT_THREADS) void cut_level_kernel(const CutArgs a) {
    __shared__ uint32_t s_stk[CUT_WORDS * CUT_THREADS];  // [entry][thread]: conflict-free
    const int patches_x = (a.tiles_x + 7) >> 3;
    const int patch = blockIdx.x * (CUT_THREADS / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int tix = ((patch % patches_x) << 3) + (lane & 7);
    const int tiy = ((patch / patches_x) << 3) + (lane >> 3);
    const int tiles_y = a.n_tiles / a.tiles_x;
    if (tix >= a.tiles_x || tiy >= tiles_y) return;
    cut_refine(a, tiy * a.tiles_x + tix, s_stk + threadIdx.x, CUT_THREADS);
    if (a.tris) cut_trim(a, tiy * a.tiles_x + tix, s_stk + threadIdx.x, CUT_THREADS);
}

// =======================================================================================
// One level straight from the TLAS root with SEVERAL LANES PER BLOCK (cut_mode 1: frames of 2 M .. 12 M rays, e.g.
// the reference's own 400 x 225 x 50).  cut_level_kernel gives a block one thread that walks depth first: ~100
// dependent steps, and an image of 5 700 blocks is 90 waves on a GPU of 1 024 SIMDs -- 0.2 ms of latency for a
// traversal of 0.1 ms.  Here a block's list is refined breadth first, one lane per entry: in every round each waiting
// entry is classified (one node record, one box) and replaced IN PLACE -- by nothing (no ray of the block passes
// it), by itself (final), or by its two children (waiting), the latter while the list has room, granted left to
// right -- so the list stays in visit order and a round costs one step whatever the list's length: as many rounds
// as the tree is deep.  Any list produced this way is a valid cut (every step is one of cut_refine's); which nodes
// get refined when room runs out differs from the depth-first walk, so the host twin (cut_single_serial) follows
// the same rounds.  The leaves are then trimmed like cut_trim does (one lane per entry tests its triangles, the list
// is assembled in order by one lane).
// =======================================================================================
constexpr int CUT1_LANES = 32;  // lanes per block >= entries per list + 1
static_assert(CUT1_LANES > CUT_CAP && 64 % CUT1_LANES == 0, "one lane per entry");

struct Cut1Step {
    int k;           // words this entry turns into: 0, 1 or 2 (two: only while the list has room, else `alone`)
    uint32_t o[2];   // the words
    bool fin[2];     // final (true) or waiting to be classified in the next round
    uint32_t alone;  // k == 2 without room: this one word instead, final
};

// What one WAITING entry turns into (the cases of cut_refine's loop body, one to one).
__host__ __device__ inline Cut1Step cut1_step(const CutArgs& a, const TileRays& tr, const uint32_t e) {
    Cut1Step s;
    s.k = 1;
    s.o[0] = e;
    s.o[1] = 0u;
    s.fin[0] = s.fin[1] = true;
    s.alone = e;
    if (e & rt::DESC_LEAF) return s;  // (a leaf every ray passes: final)
    const uint32_t kind = e & CUT_NODE;
    if (kind == 0u) {  // an internal node every ray passes (e = index of its left child): its children, to be classified
        s.k = 2;
        s.o[0] = CUT_NODE | e;
        s.o[1] = CUT_NODE | (e + 1u);
        s.fin[0] = s.fin[1] = false;
        return s;
    }
    const bool is_tlas = kind == rt::DESC_TLAS;
    uint32_t idx = e & 0x1FFFFFFFu;
    if (kind == rt::DESC_ROOT) idx = (uint32_t)a.slot_root[idx];
    const rt::DevNode* const np = is_tlas ? a.tlas + (e & 0x3FFFFFFFu) : a.nodes + idx;
    double mn[3], mx[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        mn[k] = np->mn[k];
        mx[k] = np->mx[k];
    }
    const int32_t left = np->left, count = np->count, first = np->first;
    const uint32_t desc = np->desc;
    const int cls = cut_classify(mn, mx, tr);
    if (cls == 0) {
        s.k = 0;
        return s;
    }
    if (is_tlas) {
        // kept (tested per ray, as the reference does when it pops the node) unless every ray passes it
        if (cls == 1 && left >= 0) {
            s.k = 2;
            s.o[0] = rt::DESC_TLAS | (uint32_t)left;
            s.o[1] = rt::DESC_TLAS | (uint32_t)(left + 1);
            s.fin[0] = s.fin[1] = false;
        } else if (cls == 1 && left < 0 && count >= 1 && count <= 2) {
            s.k = count;
            for (int q = 0; q < count; ++q) {
                s.o[q] = rt::DESC_ROOT | (uint32_t)(first + q);
                s.fin[q] = false;
            }
        }
        return s;
    }
    s.alone = (cls == 1) ? desc : (CUT_NODE | idx);
    if (!(desc & rt::DESC_LEAF)) {  // internal: gives way to its children (its own test, sure or not, only prunes)
        s.k = 2;
        s.o[0] = CUT_NODE | desc;
        s.o[1] = CUT_NODE | (desc + 1u);
        s.fin[0] = s.fin[1] = false;
    } else {
        s.o[0] = s.alone;
    }
    return s;
}

// What cut_trim needs to know about one entry of a finished list: its leaf (0: not a trimmable leaf), whether the
// leaf's box is tested per ray, and which of its triangles some ray of the block may hit.
struct Cut1Leaf {
    uint32_t leaf, live;
    bool per_ray;
};
__host__ __device__ inline Cut1Leaf cut1_leaf(const CutArgs& a, const uint32_t e, const double (&dd)[4][3], const double slack) {
    Cut1Leaf r;
    r.leaf = 0u;
    r.live = 0u;
    r.per_ray = false;
    if (e & rt::DESC_LEAF) {
        r.leaf = e;
    } else if ((e & CUT_NODE) == CUT_NODE) {
        const uint32_t d = a.nodes[e & CUT_NODE_IDX].desc;
        if (d & rt::DESC_LEAF) {
            r.leaf = d;
            r.per_ray = true;
        }
    }
    const uint32_t cnt = ((r.leaf >> 28) & 7u) + 1u;
    if (r.leaf == 0u || cnt == 8u) {  // not a leaf, or one with its range in leaf_ext: stays as it is
        r.leaf = 0u;
        return r;
    }
    r.live = cut_live_triangles(a.tris, r.leaf & rt::DESC_TRI_MASK, cnt, dd, slack);
    return r;
}

// The trimmed list, in order, from the entries and what cut1_leaf found (element i at x[i * stride]); cut_trim's
// rules word for word.  Returns the number of words written to out[1..]; *simple as cut_trim sets CUT_SIMPLE.
__host__ __device__ inline int cut1_assemble(const int n, const uint32_t* ent, const uint32_t* leafs, const uint32_t* lives,
                                             const uint32_t* per_ray, const int stride, uint32_t* const out, bool* simple_out) {
    int w = 0;
    bool simple = true;
    for (int i = 0; i < n; ++i) {
        const uint32_t e = ent[i * stride], leaf = leafs[i * stride], live = lives[i * stride];
        const int rest = n - 1 - i;  // every later entry can do with one word
        if (leaf == 0u) {
            out[1 + w++] = e;
            simple = false;
            continue;
        }
        if (live == 0u) continue;
        const uint32_t cnt = ((leaf >> 28) & 7u) + 1u;
        const uint32_t first = leaf & rt::DESC_TRI_MASK;
        int lo = 0, hi = (int)cnt - 1;
        while (!((live >> lo) & 1u)) ++lo;
        while (!((live >> hi) & 1u)) --hi;
        const uint32_t range = rt::DESC_LEAF | ((uint32_t)(hi - lo) << 28) | (first + (uint32_t)lo);
        if (per_ray[i * stride]) {
            if (w + 2 + rest <= CUT_CAP) {
                out[1 + w++] = e | CUT_TRIM;
                out[1 + w++] = range;
            } else {
                out[1 + w++] = e;
                simple = false;
            }
            continue;
        }
        int runs = 0;
        for (int m = lo; m <= hi; ++m) runs += ((live >> m) & 1u) && !(m > lo && ((live >> (m - 1)) & 1u));
        if (runs > 1 && w + runs + rest <= CUT_CAP) {
            int m = lo;
            while (m <= hi) {
                if (!((live >> m) & 1u)) {
                    ++m;
                    continue;
                }
                int m1 = m;
                while (m1 + 1 <= hi && ((live >> (m1 + 1)) & 1u)) ++m1;
                out[1 + w++] = rt::DESC_LEAF | ((uint32_t)(m1 - m) << 28) | (first + (uint32_t)m);
                m = m1 + 1;
            }
        } else {
            out[1 + w++] = range;
        }
    }
    *simple_out = simple;
    return w;
}

// Whether a finished list of this block may be trimmed, and with which corner rays and slack (cut_trim's preamble).
__host__ __device__ inline bool cut1_trim_setup(const CutArgs& a, const int tile, const TileRays& tr, double (&dd)[4][3],
                                                double& slack) {
    if (!a.tris) return false;
    if (tr.kind[0] == 2 && tr.kind[1] == 2 && tr.kind[2] == 2) return false;  // no sign-keeping axis
    cut_corner_rays(a, tile % a.tiles_x, tile / a.tiles_x, dd);
    double ddmax = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        ddmax = __builtin_fmax(ddmax, (__builtin_fabs(dd[c][0]) + __builtin_fabs(dd[c][1])) + __builtin_fabs(dd[c][2]));
    slack = 1e-9 * ddmax + 4.0 * (0x1p-50 * a.mag);
    return slack < 1e100;
}

// Host twin of cut_single_kernel for one block: the same rounds, one entry after the other.
__host__ __device__ inline void cut_single_serial(const CutArgs& a, const int tile) {
    uint32_t* const out = a.cuts + (size_t)tile * CUT_WORDS;
    TileRays tr;
    if (!cut_tile_rays(a, tile % a.tiles_x, tile / a.tiles_x, tr)) {
        out[0] = 0u;
        return;
    }
    uint32_t ent[2][CUT_WORDS];
    bool fin[2][CUT_WORDS];
    int cur = 0, n = 1;
    ent[0][0] = rt::DESC_TLAS;
    fin[0][0] = false;
    for (;;) {
        bool waiting = false;
        for (int i = 0; i < n; ++i) waiting = waiting || !fin[cur][i];
        if (!waiting) break;
        Cut1Step st[CUT_WORDS];
        int total1 = 0;
        for (int i = 0; i < n; ++i) {
            if (fin[cur][i]) {
                st[i].k = 1;
                st[i].o[0] = ent[cur][i];
                st[i].fin[0] = true;
            } else {
                st[i] = cut1_step(a, tr, ent[cur][i]);
            }
            total1 += st[i].k >= 1;
        }
        int extra = a.cap - total1, m = 0;
        for (int i = 0; i < n; ++i) {
            if (st[i].k == 2) {
                if (extra > 0) {
                    --extra;
                } else {
                    st[i].k = 1;
                    st[i].o[0] = st[i].alone;
                    st[i].fin[0] = true;
                }
            }
            for (int q = 0; q < st[i].k; ++q) {
                ent[cur ^ 1][m] = st[i].o[q];
                fin[cur ^ 1][m] = st[i].fin[q];
                ++m;
            }
        }
        n = m;
        cur ^= 1;
    }
    uint32_t hdr = CUT_VALID | ((uint32_t)tr.kind[0] << 8) | ((uint32_t)tr.kind[1] << 10) | ((uint32_t)tr.kind[2] << 12);
    double dd[4][3], slack = 0.0;
    if (cut1_trim_setup(a, tile, tr, dd, slack)) {
        uint32_t leafs[CUT_WORDS], lives[CUT_WORDS], per_ray[CUT_WORDS];
        for (int i = 0; i < n; ++i) {
            const Cut1Leaf r = cut1_leaf(a, ent[cur][i], dd, slack);
            leafs[i] = r.leaf;
            lives[i] = r.live;
            per_ray[i] = r.per_ray ? 1u : 0u;
        }
        bool simple = false;
        const int w = cut1_assemble(n, ent[cur], leafs, lives, per_ray, 1, out, &simple);
        out[0] = hdr | (uint32_t)w | (simple ? CUT_SIMPLE : 0u);
    } else {
        for (int i = 0; i < n; ++i) out[1 + i] = ent[cur][i];
        out[0] = hdr | (uint32_t)n;
    }
}

// 64 / CUT1_LANES blocks per wave, one lane per entry of a block's list.
__global__ __launch_bounds__(64) void cut_single_kernel(const CutArgs a) {
    constexpr int G = 64 / CUT1_LANES;
    __shared__ uint32_t s_ent[2][G][CUT_WORDS];
    __shared__ uint32_t s_fin[2][G][CUT_WORDS];
    __shared__ uint32_t s_leaf[G][CUT_WORDS], s_live[G][CUT_WORDS], s_pr[G][CUT_WORDS];
    const int lane = threadIdx.x & 63;
    const int g = lane / CUT1_LANES, l = lane % CUT1_LANES;
    const int tile_raw = (int)blockIdx.x * G + g;
    const bool have = tile_raw < a.n_tiles;
    const int tile = have ? tile_raw : a.n_tiles - 1;  // (lanes without a block follow the last one and write nothing)
    uint32_t* const out = a.cuts + (size_t)tile * CUT_WORDS;
    TileRays tr;
    const bool ok = cut_tile_rays(a, tile % a.tiles_x, tile / a.tiles_x, tr);
    const unsigned long long gbits = (CUT1_LANES == 64) ? ~0ull : (((1ull << CUT1_LANES) - 1ull) << (g * CUT1_LANES));
    const unsigned long long below = gbits & ((1ull << lane) - 1ull);
    int cur = 0, n = ok ? 1 : 0;
    if (l == 0) {
        s_ent[0][g][0] = rt::DESC_TLAS;
        s_fin[0][g][0] = 0u;
    }
    __syncthreads();
    for (;;) {
        const bool mine = l < n;
        const uint32_t e = mine ? s_ent[cur][g][l] : 0u;
        const bool f = mine ? (s_fin[cur][g][l] != 0u) : true;
        if (__builtin_amdgcn_ballot_w64(mine && !f) == 0ull) break;
        Cut1Step st;
        st.k = mine ? 1 : 0;
        st.o[0] = e;
        st.o[1] = 0u;
        st.fin[0] = st.fin[1] = true;
        st.alone = e;
        if (mine && !f) st = cut1_step(a, tr, e);
        const unsigned long long b1 = __builtin_amdgcn_ballot_w64(st.k >= 1) & gbits;
        const unsigned long long b2 = __builtin_amdgcn_ballot_w64(st.k == 2) & gbits;
        const int extra = a.cap - __builtin_popcountll(b1);
        const bool granted = st.k == 2 && __builtin_popcountll(b2 & below) < extra;
        if (st.k == 2 && !granted) {
            st.k = 1;
            st.o[0] = st.alone;
            st.fin[0] = true;
        }
        const unsigned long long bg = __builtin_amdgcn_ballot_w64(granted) & gbits;
        const int off = __builtin_popcountll(b1 & below) + __builtin_popcountll(bg & below);
        for (int q = 0; q < st.k; ++q) {
            s_ent[cur ^ 1][g][off + q] = st.o[q];
            s_fin[cur ^ 1][g][off + q] = st.fin[q] ? 1u : 0u;
        }
        n = __builtin_popcountll(b1) + __builtin_popcountll(bg);
        cur ^= 1;
        __syncthreads();
    }
    const uint32_t hdr = CUT_VALID | ((uint32_t)tr.kind[0] << 8) | ((uint32_t)tr.kind[1] << 10) | ((uint32_t)tr.kind[2] << 12);
    double dd[4][3], slack = 0.0;
    const bool trim = ok && cut1_trim_setup(a, tile, tr, dd, slack);  // (the same for every lane of a block)
    if (trim && l < n) {
        const Cut1Leaf r = cut1_leaf(a, s_ent[cur][g][l], dd, slack);
        s_leaf[g][l] = r.leaf;
        s_live[g][l] = r.live;
        s_pr[g][l] = r.per_ray ? 1u : 0u;
    }
    __syncthreads();
    if (!have) return;
    if (!ok) {
        if (l == 0) out[0] = 0u;
    } else if (trim) {
        if (l == 0) {
            bool simple = false;
            const int w = cut1_assemble(n, s_ent[cur][g], s_leaf[g], s_live[g], s_pr[g], 1, out, &simple);
            out[0] = hdr | (uint32_t)w | (simple ? CUT_SIMPLE : 0u);
        }
    } else {
        if (l < n) out[1 + l] = s_ent[cur][g][l];
        if (l == 0) out[0] = hdr | (uint32_t)n;
    }
}

// ---- the render kernels' side of the cuts (shared by the f64 and the fp32 packet kernels) ----
// The cut of the smallest aligned pixel block that holds a batch's pixels: m_first / m_last are the Morton indices
// (within the tile) of its first and last ray's pixel -- an aligned block is an interval of Morton indices -- and
// (pxf, pyf), (pxl, pyl) those pixels.  Lane i < CUT_WORDS returns word i of the block's record (lane 0 = header),
// 0 when a pixel lies outside the image.  The level tables sit in the kernel arguments at the given offsets and are
// read from the kernarg segment here (see karg_ptr).
template <int OFF_LVL, int OFF_TX, int OFF_EFF>
__device__ __forceinline__ uint32_t load_batch_cut(const int cut_lo, const int m_first, const int m_last, const int pxf,
                                                   const int pyf, const int pxl, const int pyl, const int W,
                                                   const int rows, const int lane) {
    uint32_t v = 0u;
    const uint32_t x = (uint32_t)(m_first ^ m_last);
    int lvl = x ? ((31 - __builtin_clz(x)) >> 1) + 1 : 0;
    lvl = max(lvl, cut_lo);
    if (pxf < W && pyf < rows && pxl < W && pyl < rows) {
        unsigned long long base_ptr;
        uint32_t ctx, eff;
        const auto kseg = __builtin_amdgcn_kernarg_segment_ptr();
        const uint32_t o1 = (uint32_t)OFF_LVL + 8u * (uint32_t)lvl;
        const uint32_t o2 = (uint32_t)OFF_TX + 4u * (uint32_t)lvl;
        const uint32_t o3 = (uint32_t)OFF_EFF + 4u * (uint32_t)lvl;
        asm volatile(
            "s_load_dwordx2 %0, %3, %4\n\ts_load_dword %1, %3, %5\n\ts_load_dword %2, %3, %6\n\ts_waitcnt lgkmcnt(0)"
            : "=&s"(base_ptr), "=&s"(ctx), "=&s"(eff)
            : "s"(kseg), "s"(o1), "s"(o2), "s"(o3)
            : "memory");
        const uint32_t* const cp = reinterpret_cast<const uint32_t*>(base_ptr) +
                                   ((size_t)(pyf >> eff) * ctx + (size_t)(pxf >> eff)) * CUT_WORDS;
        if (lane < CUT_WORDS) v = cp[lane];
    }
    return v;
}

// Number of entries of the batch's cut, or -1: walk from the root.  The cut holds for rays through its block with
// offsets in [-0.5, 0.5] whose direction signs are the ones it was computed for (caller-supplied offsets may be
// anything); the caller adds that no lane is NaN-prone.
__device__ __forceinline__ int batch_cut_entries(const uint32_t vcut, const unsigned long long amask,
                                                 const unsigned long long ngx, const unsigned long long ngy,
                                                 const unsigned long long ngz, const double jx, const double jy) {
    const uint32_t cut_hdr = (uint32_t)__builtin_amdgcn_readfirstlane((int)vcut);
    if (!(cut_hdr & CUT_VALID)) return -1;
    const uint32_t kx = (cut_hdr >> 8) & 3u, ky = (cut_hdr >> 10) & 3u, kz = (cut_hdr >> 12) & 3u;
    const bool sx_ok = kx == 2u || (kx == 1u ? ngx == amask : ngx == 0ull);
    const bool sy_ok = ky == 2u || (ky == 1u ? ngy == amask : ngy == 0ull);
    const bool sz_ok = kz == 2u || (kz == 1u ? ngz == amask : ngz == 0ull);
    const bool in_range = __builtin_fabs(jx) <= 0.5 && __builtin_fabs(jy) <= 0.5;
    if (sx_ok && sy_ok && sz_ok && (__builtin_amdgcn_ballot_w64(!in_range) & amask) == 0ull) return (int)(cut_hdr & 0xffu);
    return -1;
}

// ---- the compiler's fp64 division, for operands that need no scaling (see render_packet_kernel) ----
__device__ __forceinline__ uint32_t hi_word(const double x) {
    return (uint32_t)(__builtin_bit_cast(unsigned long long, x) >> 32);
}
__device__ __forceinline__ uint32_t hi_abs(const double x) {
    return (uint32_t)(__builtin_bit_cast(unsigned long long, x) >> 32) & 0x7fffffffu;
}
__device__ __forceinline__ double refined_rcp(const double d) {  // v_rcp_f64 + two Newton steps
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
    return r;
}
__device__ __forceinline__ double div_by(const double n, const double d, const double r) {  // n / d, r = refined_rcp(d)
    const double q = n * r;
    return __builtin_fma(__builtin_fma(-d, q, n), r, q);
}
// The compiler's fp64 sqrt for 2^-767 <= x < inf (below that it pre-scales by 2^256; zero and infinity are
// passed through): v_rsq_f64 and the same Goldschmidt steps, without the scaling and class selects.
__device__ __forceinline__ double sqrt_noscale(const double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
    g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
    return g;
}
__device__ __forceinline__ double rcp_exact(const double d) {  // 1.0 / d
    const double r = refined_rcp(d);
    return __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
}

// Morton index (within the tile) of pixel i of wave w.  g = 0: wave w owns the compact quadrant starting at m0;
// g = 2^k: the waves take turns every g pixels, so together they walk 4 g consecutive Morton indices at a time
// (g = 1: a 2x2 block, one pixel each) and meet each other's lines in the scalar data cache.
__device__ __forceinline__ int wave_pixel(const int i, const int w, const int m0, const int g) {
    if (g == 0) return m0 + i;
    return ((i & ~(g - 1)) << 2) + w * g + (i & (g - 1));
}

constexpr int PAIR64_STACK = 32;   // stack entries left to the traversal in pair64 mode (see render_packet_kernel)
constexpr int PACKET_STACK = 128;  // entries of the per-wave LDS stack ({descriptor, lane mask}, 16 B each)
static_assert((PACKET_STACK - PAIR64_STACK) * 16 >= 3 * 64 * 8, "the second pixel's colours must fit above the pair64 stack");

// One workgroup = one square pixel tile, one wave = one quarter of the tile's pixels (every fourth
// pixel in Morton order, see PacketArgs::interleave; or a compact quadrant).  A wave walks its own pixels' samples 64 at a time and owns its part
// of the accumulation, so waves never wait for each other: there is no workgroup barrier.
// Rays are ordered pixel-major with the samples of a pixel innermost: at spp >= 64 the 64 lanes
// of a wave are samples of ONE pixel, otherwise a compact block of neighbouring pixels.
// Sample colours go through the wave's LDS slice and are added per pixel in sample order
// k = 0..spp-1 by one lane per (pixel, channel): bit-identical to the reference's serial sum
// (cpp/rtrender.cpp:51-60) whatever the tiling or GPU count.
// P64: the host launches this instantiation when spp is 64, 128, 256, ... (a power of two), tiles are 8x8 and the
// tree needs <= PAIR64_STACK stack entries (PacketArgs::pair64 = 1 + log2(spp / 64)).  A batch then lies in one pixel: the pixel's coordinates, its bounds
// test and the base of its sample index are wave-uniform (scalar unit), divisions by spp disappear, and the
// per-pixel sums are done two pixels at a time.
template <bool COUNT, bool P64>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4, 8), amdgpu_num_vgpr(64), amdgpu_num_sgpr(94))) void render_packet_kernel(const PacketArgs a) {
    __shared__ uint4 s_wstk[NT / 64][PACKET_STACK];
    __shared__ double s_col[NT / 64][3][64];
    __shared__ double s_acc[NT / 64][3][64];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform, and the compiler is told so
    int tile = blockIdx.x;
    {
        const int nb = a.n_tiles, per = nb >> 3, main = per << 3;
        if ((int)blockIdx.x < main) tile = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    }
    const int ts = 1 << a.tile_log2;
    const int npix_w = (ts * ts) >> 2;  // pixels per wave: 16 (8x8 tile) or 64 (16x16 tile)
    const int m0 = wv * npix_w;         // first Morton index of this wave's quadrant (!interleave)
    int tix, tiy;
    tile_coords(tile, a.n_tiles, a.tiles_x, tix, tiy);
    const int tx0 = tix << a.tile_log2;
    const int ty0 = tiy << a.tile_log2;
    const int lg64 = P64 ? a.pair64 - 1 : 0;  // P64: spp = 64 << lg64
    const int spp = P64 ? (64 << lg64) : a.spp;
    const int n_rays = npix_w * spp;
    uint4* wstk = s_wstk[wv];
    double(*wcol)[64] = s_col[wv];
    double(*wacc)[64] = s_acc[wv];

    wacc[0][lane] = 0.0;
    wacc[1][lane] = 0.0;
    wacc[2][lane] = 0.0;

    unsigned long long w_nodes = 0, w_tris = 0, w_steps = 0, w_tsteps = 0;  // lane 0 of each wave carries the wave's sums
    uint32_t n_hits = 0;

    // The jitter pair of the NEXT batch is requested before this batch is traversed (one HBM round trip
    // off every batch's critical path: the kernel is bound by per-wave latency at 8 waves per SIMD).
    struct Slot {
        int px, py;
        bool active;
        size_t sidx;
        d2 jit;
        uint32_t vcut;  // the batch's cut: lane 0 = header, lanes 1.. = entries (0: none)
    };
    int vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
    // generic instantiation: the ray index base + lane is split into (pixel, sample) without a per-lane division:
    // qb = base / spp and rb = base % spp are wave-uniform and advance by (64 / spp, 64 % spp) per batch
    const uint32_t magic = a.spp_magic;
    auto div_spp = [&](const int r) -> int {  // floor(r / spp), 0 <= r < spp + 128
        return spp >= 64 ? (r >= spp ? 1 : 0) : (int)(((uint32_t)r * magic) >> 16);
    };
    const int q64 = P64 ? 0 : 64 / spp, r64 = P64 ? 0 : 64 - q64 * spp;
    auto fetch = [&](const int base, const int qb, const int rb) {
        Slot q;
        const int rr = base + lane;
        // P64: wave-uniform (RT_P64_VZERO: carried in a VGPR all the same, by adding an opaque zero, so that the
        // Morton decode, the bounds test and the sample index below are vector instructions; see the macro)
        // P64: the launch constants this arithmetic needs are read from the kernel arguments HERE, every batch (scalar
        // loads + a few SALU instructions): kept across the traversal they were spilled to VGPR lanes, and each
        // v_readlane is an issue slot of the vector port, which is the busier one (RT_P64_FRESH_ARGS, same box A/B)
        int lgq = lg64, ilv = a.interleave;
        if (P64 && RT_P64_FRESH_ARGS) {
            const auto kseg = __builtin_amdgcn_kernarg_segment_ptr();
            int p64v;
            asm volatile("s_load_dword %0, %2, %3\n\ts_load_dword %1, %2, %4\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(p64v), "=&s"(ilv)
                         : "s"(kseg), "i"(offsetof(PacketArgs, pair64)), "i"(offsetof(PacketArgs, interleave))
                         : "memory");
            lgq = p64v - 1;
        }
        const int sppq = P64 ? (64 << lgq) : spp;
        const int dq = P64 ? 0 : div_spp(rb + lane);
        const int ml = P64 ? (base >> (6 + lgq)) + (RT_P64_VZERO ? vzero : 0) : qb + dq;
        const int k = P64 ? (base & (sppq - 1)) + lane : rb + lane - dq * spp;
        const int m = wave_pixel(ml, wv, m0, ilv);
        q.px = tx0 + (int)compact1by1((uint32_t)m);
        q.py = ty0 + (int)compact1by1((uint32_t)m >> 1);
        // P64 with 32 < a.spp < 64 (host: pair64 = 1): one pixel per batch all the same, lanes a.spp.. idle -- their
        // colours stay 0.0, which the in-order sums below add at the end without changing a bit (the sums start at
        // +0.0, so they are never -0.0)
        q.active = (rr < n_rays) && (q.px < a.W) && (q.py < a.rows) && (!P64 || k < a.spp);
        q.sidx = q.active ? ((size_t)q.py * a.W + q.px) * (size_t)(P64 ? a.spp : spp) + k : 0;
        q.jit = d2{0.0, 0.0};
        // streamed once: non-temporal, so the 16 B/sample stream does not displace tree nodes in L2
        if (q.active)
            q.jit = __builtin_nontemporal_load(karg_ptr<const d2, offsetof(PacketArgs, jitter)>() + q.sidx);
        q.vcut = 0u;
        if (P64 && RT_P64_VECTOR_CUT && a.cut_lo >= 0) {
            // one pixel per batch: the cut is the finest level's, always; its address is per-lane arithmetic on the
            // (uniform) pixel coordinates that already sit in VGPRs -- the scalar unit only fetches the level's table
            unsigned long long base_ptr;
            uint32_t ctx;
            const auto kseg = __builtin_amdgcn_kernarg_segment_ptr();
            const uint32_t o1 = (uint32_t)offsetof(PacketArgs, cut_lvl) + 8u * (uint32_t)a.cut_lo;
            const uint32_t o2 = (uint32_t)offsetof(PacketArgs, cut_tx) + 4u * (uint32_t)a.cut_lo;
            asm volatile("s_load_dwordx2 %0, %2, %3\n\ts_load_dword %1, %2, %4\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(base_ptr), "=&s"(ctx)
                         : "s"(kseg), "s"(o1), "s"(o2)
                         : "memory");
            // (cut_eff[cut_lo] == cut_lo: the finest level is always made)
            const uint32_t blk = (uint32_t)(q.py >> a.cut_lo) * ctx + (uint32_t)(q.px >> a.cut_lo);
            const uint32_t* const cp = reinterpret_cast<const uint32_t*>(base_ptr) + (size_t)blk * CUT_WORDS;
            if (lane < CUT_WORDS && q.px < a.W && q.py < a.rows) q.vcut = cp[lane];
        } else if (P64 && RT_P64_FRESH_ARGS) {
            // one pixel per batch: the cut is the finest level's (cut_eff[cut_lo] == cut_lo: it is always made)
            const auto kseg = __builtin_amdgcn_kernarg_segment_ptr();
            int lo;
            asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(lo) : "s"(kseg), "i"(offsetof(PacketArgs, cut_lo)) : "memory");
            const int pxs = __builtin_amdgcn_readfirstlane(q.px), pys = __builtin_amdgcn_readfirstlane(q.py);
            if (lo >= 0 && pxs < a.W && pys < a.rows) {
                unsigned long long base_ptr;
                uint32_t ctx;
                const uint32_t o1 = (uint32_t)offsetof(PacketArgs, cut_lvl) + 8u * (uint32_t)lo;
                const uint32_t o2 = (uint32_t)offsetof(PacketArgs, cut_tx) + 4u * (uint32_t)lo;
                asm volatile("s_load_dwordx2 %0, %2, %3\n\ts_load_dword %1, %2, %4\n\ts_waitcnt lgkmcnt(0)"
                             : "=&s"(base_ptr), "=&s"(ctx)
                             : "s"(kseg), "s"(o1), "s"(o2)
                             : "memory");
                const uint32_t* const cp = reinterpret_cast<const uint32_t*>(base_ptr) +
                                           ((size_t)(pys >> lo) * ctx + (size_t)(pxs >> lo)) * CUT_WORDS;
                if (__builtin_amdgcn_inverse_ballot_w64(0xFFFFFFFFull)) q.vcut = cp[lane];  // lanes 0 .. CUT_WORDS - 1
            }
        } else if (a.cut_lo >= 0) {
            const int last = min(63, n_rays - 1 - base);
            const int mf = __builtin_amdgcn_readfirstlane(m), ml2 = P64 ? mf : __builtin_amdgcn_readlane(m, last);
            const int pxf = __builtin_amdgcn_readfirstlane(q.px), pyf = __builtin_amdgcn_readfirstlane(q.py);
            const int pxl = P64 ? pxf : __builtin_amdgcn_readlane(q.px, last);
            const int pyl = P64 ? pyf : __builtin_amdgcn_readlane(q.py, last);
            q.vcut = load_batch_cut<offsetof(PacketArgs, cut_lvl), offsetof(PacketArgs, cut_tx), offsetof(PacketArgs, cut_eff)>(
                a.cut_lo, mf, ml2, pxf, pyf, pxl, pyl, a.W, a.rows, lane);
        }
        return q;
    };
    int qn = 0, rn = 0;  // (base + 64) / spp and % spp of the batch being fetched
    Slot nxt = fetch(0, 0, 0);
    for (int base = 0; base < n_rays; base += 64) {
        const Slot cs = nxt;
        const int qc = qn, rc = rn;  // base / spp, base % spp
        if (!P64) {
            qn += q64;
            rn += r64;
            if (rn >= spp) {
                rn -= spp;
                ++qn;
            }
        }
        if (base + 64 < n_rays) nxt = fetch(base + 64, qn, rn);
        const int px = cs.px, py = cs.py;
        const bool active = cs.active;
        const size_t sidx = cs.sidx;
        double c0 = 0.0, c1 = 0.0, c2 = 0.0;
        double dx = 0.0, dy = 0.0, dz = -1.0;
        double ddx = 1.0, ddy = 1.0, ddz = 1.0;
        {
            // the camera block (12 doubles) is fetched here, every batch, with scalar loads pinned to this spot:
            // hoisted out of the loop it would hold 24 SGPRs across the traversal
            u32x16 fa;  // cam[0..2] p00[0..2] sp[0..1]
            u32x8 fb;   // sp[2..5]
            // (the block travels in the kernel arguments themselves: no per-frame copy to the device)
            const auto kseg = __builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx8 %1, %2, %4\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(fa), "=&s"(fb)
                         : "s"(kseg), "i"(offsetof(PacketArgs, fpv)), "i"(offsetof(PacketArgs, fpv) + 0x40)
                         : "memory");
            const d2 jit = cs.jit;
            // cpp/rtrender.cpp:53-58
            const double s0 = (double)px + jit.x;
            const double s1 = (double)(py + a.row0) + jit.y;
            const double sp2 = __builtin_bit_cast(double, ((unsigned long long)fb[1] << 32) | fb[0]);
            const double sp3 = __builtin_bit_cast(double, ((unsigned long long)fb[3] << 32) | fb[2]);
            const double sp4 = __builtin_bit_cast(double, ((unsigned long long)fb[5] << 32) | fb[4]);
            const double sp5 = __builtin_bit_cast(double, ((unsigned long long)fb[7] << 32) | fb[6]);
            const double tx = ((sgpr_f64(fa, 3) + s0 * sgpr_f64(fa, 6)) + s1 * sp3) - sgpr_f64(fa, 0);
            const double ty = ((sgpr_f64(fa, 4) + s0 * sgpr_f64(fa, 7)) + s1 * sp4) - sgpr_f64(fa, 1);
            const double tz = ((sgpr_f64(fa, 5) + s0 * sp2) + s1 * sp5) - sgpr_f64(fa, 2);
            if (active) {
                ddx = tx;
                ddy = ty;
                ddz = tz;
            }
        }
        double ix, iy, iz;
        // The six divisions of a ray (three by the norm, three reciprocals) are 84 of the batch's issue slots as
        // the compiler expands them (v_div_scale x2, v_rcp, 4 fma, mul, fma, v_div_fmas, v_div_fixup each).  When
        // every component of every active lane has a binary exponent within +-300, v_div_scale scales nothing,
        // v_div_fmas is a plain fma and v_div_fixup passes its operand through, so the expansion reduces to the
        // instructions below -- the same ones, with the denominator's part (reciprocal + two Newton steps)
        // shared by the three quotients, and `1.0 * r` not multiplied.  Bit-identical; any other wave takes the
        // compiler's divisions.
        const uint32_t ex = hi_abs(ddx), ey = hi_abs(ddy), ez = hi_abs(ddz);
        const uint32_t emin = min(min(ex, ey), ez), emax = max(max(ex, ey), ez);
        const bool tame_div = emin >= ((1023u - 300u) << 20) && emax < ((1023u + 300u) << 20);
        if (__builtin_amdgcn_ballot_w64(!tame_div) == 0ull) {
            const double z = (ddx * ddx + ddy * ddy) + ddz * ddz;  // > 0: Eigen normalized() divides
            const double nz = sqrt_noscale(z);  // 2^-599 < z < 2^602
            const double rn = refined_rcp(nz);
            const double qx = div_by(ddx, nz, rn), qy = div_by(ddy, nz, rn), qz = div_by(ddz, nz, rn);
            ix = rcp_exact(qx);
            iy = rcp_exact(qy);
            iz = rcp_exact(qz);
            dx = qx;  // inactive lanes (components 1, 1, 1) are masked out of everything below
            dy = qy;
            dz = qz;
        } else {
            if (active) {
                const double z = (ddx * ddx + ddy * ddy) + ddz * ddz;
                dx = ddx;
                dy = ddy;
                dz = ddz;
                if (z > 0.0) {  // Eigen normalized(): v / sqrt(squaredNorm) when squaredNorm > 0
                    const double nz = __builtin_sqrt(z);
                    dx = ddx / nz;
                    dy = ddy / nz;
                    dz = ddz / nz;
                }
            }
            ix = 1.0 / dx;
            iy = 1.0 / dy;
            iz = 1.0 / dz;
        }
        Hit h;
        h.t = __builtin_inf();
        h.u = 0.0;
        h.v = 0.0;
        h.prim = -1;
        // FAST slab path needs every product (box - camera) * inverse direction finite and non-NaN
        const bool tame = (__builtin_fabs(ix) < 1e100) && (__builtin_fabs(iy) < 1e100) && (__builtin_fabs(iz) < 1e100);
        const bool risky = active && !(tame && a.fast_ok);
        const unsigned long long amask = __builtin_amdgcn_ballot_w64(active);
        if (amask != 0ull) {
            WaveCounters wc = {0ull, 0ull, 0ull, 0ull};
            // The common case first, decided with ONE ballot: the batch's cut is valid and promises a sign for every
            // axis (no "mixed" axis) -- then the octant is the header's, and all that is left to check per lane is that
            // the ray is one the cut was made for: tame, offsets within +-0.5, signs as promised (the sign bit is the
            // sign: a tame 1/d is neither a zero nor a NaN).  Per-lane flags and a vector compare instead of three
            // ballots and a dozen scalar compares: the kernel is bound by scalar issue.
            bool quick = false;
            int oct = 0, n_ent = -1;
            bool uni = false;
            const uint32_t hdr0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)cs.vcut);
            if (RT_QUICK_DISPATCH && (hdr0 & CUT_VALID) && ((hdr0 >> 8) & 0x2au) == 0u && a.fast_ok == 2) {
                const uint32_t want = ((hdr0 >> 8) & 1u) | ((hdr0 >> 9) & 2u) | ((hdr0 >> 10) & 4u);  // kind 1 = negative
                const uint32_t sb = (hi_word(ix) >> 31) | ((hi_word(iy) >> 31) << 1) | ((hi_word(iz) >> 31) << 2);
                const bool fits = tame && sb == want && __builtin_fabs(cs.jit.x) <= 0.5 && __builtin_fabs(cs.jit.y) <= 0.5;
                if ((__builtin_amdgcn_ballot_w64(!fits) & amask) == 0ull) {
                    quick = true;
                    uni = true;
                    oct = (int)want;
                    n_ent = (int)(hdr0 & 0xffu);
                }
            }
            bool any_risky = false;
            if (!quick) {
                // per axis: do all active lanes look the same way?  (inactive lanes are masked out of every result)
                const unsigned long long ngx = __builtin_amdgcn_ballot_w64(ix < 0.0) & amask;
                const unsigned long long ngy = __builtin_amdgcn_ballot_w64(iy < 0.0) & amask;
                const unsigned long long ngz = __builtin_amdgcn_ballot_w64(iz < 0.0) & amask;
                uni = (ngx == 0ull || ngx == amask) && (ngy == 0ull || ngy == amask) && (ngz == 0ull || ngz == amask) &&
                      a.fast_ok == 2;
                n_ent = batch_cut_entries(cs.vcut, amask, ngx, ngy, ngz, cs.jit.x, cs.jit.y);
                oct = (ngx != 0ull ? 1 : 0) | (ngy != 0ull ? 2 : 0) | (ngz != 0ull ? 4 : 0);
                any_risky = __builtin_amdgcn_ballot_w64(risky) != 0ull;
            }
            // (a count >= 0 is the header's own: the batch takes the block's list as it is)
            const bool simple = n_ent >= 0 && (hdr0 & CUT_SIMPLE) != 0u && a.fast_ok == 2;
            if (any_risky) {
                trace_packet<false, -1, COUNT>(a.sc, amask, dx, dy, dz, ix, iy, iz, wstk, h, wc, 0u, -1);
            } else if (uni) {
                switch (oct) {
#define RT_OCT_CASE(O)                                                                                       \
    case O:                                                                                                  \
        trace_packet<true, O, COUNT>(a.sc, amask, dx, dy, dz, ix, iy, iz, wstk, h, wc, cs.vcut, n_ent,      \
                                     simple);                                                               \
        break;
                    RT_OCT_CASE(0)
                    RT_OCT_CASE(1)
                    RT_OCT_CASE(2)
                    RT_OCT_CASE(3)
                    RT_OCT_CASE(4)
                    RT_OCT_CASE(5)
                    RT_OCT_CASE(6)
                    RT_OCT_CASE(7)
#undef RT_OCT_CASE
                }
            } else {
                // mixed signs: the general slab test (min/max per axis), every product finite
                trace_packet<true, -1, COUNT>(a.sc, amask, dx, dy, dz, ix, iy, iz, wstk, h, wc, cs.vcut,
                                              a.fast_ok == 2 ? n_ent : -1, simple);
            }
            if (COUNT && lane == 0) {
                w_nodes += wc.nodes;
                w_tris += wc.tris;
                w_steps += wc.steps;
                w_tsteps += wc.tri_steps;
            }
        }
        if (active) {
            // return_ray_color, cpp/rtrender.cpp:21-32
            if (h.t != __builtin_inf()) {
                const double* fc = karg_ptr<const double, offsetof(PacketArgs, sc) + offsetof(PacketScene, colors)>() + (size_t)h.prim * 3;
                const double w = (1.0 - h.u) - h.v;
                const double f0 = fc[0], f1 = fc[1], f2 = fc[2];
                c0 = (h.u * f0 + h.v * f0) + w * f0;
                c1 = (h.u * f1 + h.v * f1) + w * f1;
                c2 = (h.u * f2 + h.v * f2) + w * f2;
            } else {
                const double al = 0.5 * (dy + 1.0);
                const double ia = (1.0 - al) * 1.0;
                c0 = ia + al * 0.5;
                c1 = ia + al * 0.7;
                c2 = ia + al * 1.0;
            }
            if (COUNT) n_hits += (h.prim >= 0);
            int32_t* const hit_prim = karg_ptr<int32_t, offsetof(PacketArgs, hit_prim)>();
            double* const hit_t = karg_ptr<double, offsetof(PacketArgs, hit_t)>();
            if (hit_prim) __builtin_nontemporal_store(h.prim, hit_prim + sidx);
            if (hit_t) __builtin_nontemporal_store(h.t, hit_t + sidx);
        }
        if (P64) {
            // spp = 64 << lg64: a batch lies within one pixel.  The colours of two batches are kept (the second in
            // the upper three quarters of the wave's stack, idle between traversals; the host selects this
            // instantiation only when the tree needs <= 32 stack entries).  spp == 64: the two batches are two
            // pixels and six lanes add 64 values each, in sample order, starting from 0.0 like the generic path:
            // half the dependent adds per pixel.  spp >= 128: both belong to the same pixel and three lanes
            // continue that pixel's running sums with 128 values.  No index arithmetic either way.
            const int second = (base >> 6) & 1;
            double* const parked = reinterpret_cast<double*>(wstk + PAIR64_STACK);
            double* dst = second ? parked : &wcol[0][0];
            dst[lane] = c0;
            dst[64 + lane] = c1;
            dst[128 + lane] = c2;
            __builtin_amdgcn_wave_barrier();
            if (second && lane < (lg64 == 0 ? 6 : 3)) {
                const int which = lane >= 3 ? 1 : 0;
                const int c = lane - 3 * which;
                const int pm = lg64 == 0 ? (base >> 6) - 1 + which : (base >> (6 + lg64));
                double acc = lg64 == 0 ? 0.0 : wacc[c][pm];
                const double* src = (which ? parked : &wcol[0][0]) + c * 64;
                for (int half = 0; half < (lg64 == 0 ? 1 : 2); ++half) {
#pragma unroll
                    for (int s2 = 0; s2 < 64; s2 += 8) {
                        double v8[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) v8[j] = src[s2 + j];
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc += v8[j];
                    }
                    src = parked + c * 64;
                }
                wacc[c][pm] = acc;
            }
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        wcol[0][lane] = c0;
        wcol[1][lane] = c1;
        wcol[2][lane] = c2;
        // one wave, in-order LDS: only the compiler needs the fence
        __builtin_amdgcn_wave_barrier();
        {
            const int end = min(base + 64, n_rays);
            const int m_first = qc;                                    // base / spp
            const int m_last = qc + div_spp(rc + (end - 1 - base));   // (end - 1) / spp
            const int work = (m_last - m_first + 1) * 3;
            for (int q = lane; q < work; q += 64) {
                const int pm = m_first + q / 3;
                const int c = q - (q / 3) * 3;
                const int lo = max(pm * spp, base) - base;
                const int hi = min((pm + 1) * spp, end) - base;
                double acc = wacc[c][pm];
                const double* src = &wcol[c][lo];
                const int n = hi - lo;
                int s2 = 0;
                for (; s2 + 8 <= n; s2 += 8) {  // strictly in sample order; the 8 LDS reads are issued
                    double v8[8];                 // together, ahead of the dependent adds
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
        const int m = wave_pixel(lane, wv, m0, a.interleave);
        const int px = tx0 + (int)compact1by1((uint32_t)m);
        const int py = ty0 + (int)compact1by1((uint32_t)m >> 1);
        if (px < a.W && py < a.rows) {
            // cpp/rtrender.cpp:61-65
            const double gray = (0.2126 * wacc[0][lane] + 0.7152 * wacc[1][lane]) + 0.0722 * wacc[2][lane];
            const double inten = gray / (double)a.spp;
            const int gb = (int)(inten * 255.99);
            const size_t pidx = (size_t)py * a.W + px;
            karg_ptr<double, offsetof(PacketArgs, intensity)>()[pidx] = inten;
            const uint8_t g8 = (uint8_t)gb;
            uint8_t* const rgb = karg_ptr<uint8_t, offsetof(PacketArgs, rgb)>();
            rgb[pidx * 3] = g8;
            rgb[pidx * 3 + 1] = g8;
            rgb[pidx * 3 + 2] = g8;
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
            atomicAdd(&a.counters[4], w_tsteps);
        }
    }
}

struct TraceArgs {
    DevScene sc;
    long long n;
    const double* origins;
    const double* dirs;
    double* t;
    double* bary;
    double* color;
    int32_t* prim;
    uint32_t* gstack;
};

template <int STACK>
__global__ __launch_bounds__(NT) void trace_kernel(const TraceArgs a) {
    __shared__ uint32_t s_stk[(STACK > 0 ? STACK : 1) * NT];
    const int tid = threadIdx.x;
    const long long i = (long long)blockIdx.x * NT + tid;
    if (i >= a.n) return;
    uint32_t* stk;
    int stride;
    if (STACK > 0) {
        stk = s_stk + tid;
        stride = NT;
    } else {
        stk = a.gstack + ((size_t)blockIdx.x * NT + tid);
        stride = (int)(gridDim.x * NT);
    }
    Ray r;
    make_ray(r, a.origins[3 * i], a.origins[3 * i + 1], a.origins[3 * i + 2], a.dirs[3 * i], a.dirs[3 * i + 1],
             a.dirs[3 * i + 2]);
    Hit h;
    h.t = __builtin_inf();
    h.u = 0.0;
    h.v = 0.0;
    h.prim = -1;
    uint32_t nn = 0, nt = 0;
    trace<false>(a.sc, r, stk, stride, h, nn, nt);
    double c0, c1, c2;
    shade(a.sc, r, h, c0, c1, c2);
    if (a.t) a.t[i] = h.t;
    if (a.bary) {
        // HitRecord::barycentric_coordinates starts as zeros and is only written on a hit
        const bool hit = h.prim >= 0;
        a.bary[3 * i] = hit ? h.u : 0.0;
        a.bary[3 * i + 1] = hit ? h.v : 0.0;
        a.bary[3 * i + 2] = hit ? (1.0 - h.u) - h.v : 0.0;
    }
    if (a.color) {
        a.color[3 * i] = c0;
        a.color[3 * i + 1] = c1;
        a.color[3 * i + 2] = c2;
    }
    if (a.prim) a.prim[i] = h.prim;
}

// ---------------------------------------------------------------------------------------
// MT19937 jitter.  One workgroup per stream (see mt_jitter_kernel); words are tempered and converted
// pairwise exactly like libstdc++'s generate_canonical<double,53>: (a + b*2^32) / 2^64, clamped below 1.
// Output value is random_double() - 0.5 (cpp/rtrender.cpp:53).
// ---------------------------------------------------------------------------------------
// gfx950's three-input bit operation (truth table in the immediate: a = 0xF0, b = 0xCC, c = 0xAA) folds the
// mask-and-xor pairs of the generator into one VALU instruction each: the sampler runs beside the traversal
// kernel, which is bound by VALU issue, so every instruction saved here is given back to it.
__device__ __forceinline__ uint32_t xor_and(const uint32_t a, const uint32_t b, const uint32_t c) {  // a ^ (b & c)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x78);
}
__device__ __forceinline__ uint32_t bit_select(const uint32_t a, const uint32_t b, const uint32_t m) {
    return __builtin_amdgcn_bitop3_b32(a, b, m, 0xE4);  // (a & m) | (b & ~m)
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y = xor_and(y, y << 7, 0x9d2c5680u);
    y = xor_and(y, y << 15, 0xefc60000u);
    y ^= (y >> 18);
    return y;
}
// next raw word from x[k], x[k+1], x[k+397]
__device__ __forceinline__ uint32_t mt_twist(const uint32_t xa, const uint32_t xb, const uint32_t xm) {
    const uint32_t y = bit_select(xa, xb, 0x80000000u);
    const uint32_t odd = (uint32_t)((int32_t)(xb << 31) >> 31);  // all ones when the low bit of xb (= of y) is set
    return xor_and(xm ^ (y >> 1), odd, 0x9908b0dfu);
}

constexpr int JITTER_THREADS = 256;  // one workgroup (4 waves) per stream

// One WORKGROUP per stream: the 624-word state lives in LDS and is regenerated in place in three chunks of
// <= 227 words (word k needs k, k+1 and k+397 of the old block or k-227 of the new one, so 227 words are
// independent at a time), reads before writes with a barrier in between; then 312 doubles are tempered,
// converted exactly like libstdc++'s generate_canonical<double,53> and stored.  A stream is a chain of
// dependent regenerations, so its speed is latency: four waves per stream roughly halve the time per block
// against one wave walking 64 words at a time, which halves the number of streams (and jump-aheads) a frame
// needs to keep the sampler hidden behind the traversal.
__global__ __launch_bounds__(JITTER_THREADS) void mt_jitter_kernel(const uint32_t* __restrict__ states,
                                                                   double* __restrict__ out,
                                                                   const unsigned long long doubles_per_stream,
                                                                   const unsigned long long total_doubles) {
    __shared__ uint32_t x[624];
    const int tid = threadIdx.x;
    const unsigned long long stream = blockIdx.x;
    for (int k = tid; k < 624; k += JITTER_THREADS) x[k] = states[stream * 624 + k];
    unsigned long long pos = stream * doubles_per_stream;
    unsigned long long end = pos + doubles_per_stream;
    if (end > total_doubles) end = total_doubles;
    __syncthreads();
    for (; pos < end; pos += 312) {
#pragma unroll
        for (int c0 = 0; c0 < 624; c0 += 227) {
            const int k = c0 + tid;
            const bool on = tid < 227 && k < 624;
            uint32_t nw = 0;
            if (on) {
                const uint32_t xa = x[k];
                const uint32_t xb = x[(k + 1 == 624) ? 0 : k + 1];  // k = 623 reads the NEW x[0] (written by chunk 0)
                const uint32_t xm = x[(k + 397 >= 624) ? k + 397 - 624 : k + 397];
                nw = mt_twist(xa, xb, xm);
            }
            __syncthreads();
            if (on) x[k] = nw;
            __syncthreads();
        }
#pragma unroll
        for (int q = tid; q < 312; q += JITTER_THREADS) {
            if (pos + q < end) {
                const double wa = (double)mt_temper(x[2 * q]);
                const double wb = (double)mt_temper(x[2 * q + 1]);
                // wb * 2^32 and sum * 2^-64 are exact (powers of two, no under/overflow), so each explicit fma below
                // rounds once, exactly where `wa + wb * 2^32` and `sum / 2^64 - 0.5` round: same bits, two VALU less
                const double sum = __builtin_fma(wb, 4294967296.0, wa);
                // `if (ret >= 1.0) ret = nextafter(1.0, 0.0)` then `- 0.5`: ret - 0.5 is exact for ret in [0.5, 1],
                // so the clamp is a min with 0.5 - 2^-53 after the subtraction (one v_min instead of cmp + 2 selects)
                const double r = __builtin_fmin(__builtin_fma(sum, 0x1p-64, -0.5), 0x1.ffffffffffffep-2);
                __builtin_nontemporal_store(r, out + pos + q);  // written once, read once by the next frame
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// MT19937 jump-ahead on the GPU:  dst = g(A) src  for a polynomial g over GF(2) (312 x 64 bit,
// degree < 19937), A the generator's one-word transition.  With x[0..623] = src (the window of
// raw words preceding the position) and x[624..] the raw words the generator produces from it,
// A^i src is the window x[i .. i+623], so
//        dst[j] = XOR over the set bits i of g of x[i + j],        j = 0..623.
// Eight 256-thread workgroups per jump, each owning 2496 consecutive coefficients of g: a part
// regenerates the words it needs in LDS 227 at a time (word n needs n-624, n-623 and n-227: all
// older than the chunk), then each lane walks the set bits of its slice of g (a wave-uniform,
// scalar loop) and XORs x[i+j] -- consecutive lanes read consecutive words, conflict-free -- and
// the parts combine with atomic XORs.  Small workgroups (15.5 KB LDS) so they are dispatched
// between the workgroups of a running traversal kernel.  The host only supplies the polynomials
// (cached per stride).  Bits 0..30 of dst[0] are not part of the generator's
// state and come out unspecified, exactly as in the host implementation.
// ---------------------------------------------------------------------------------------
constexpr int JUMP_THREADS = 256;  // same footprint as a traversal workgroup: slots in between them
constexpr int JUMP_PARTS = 8;      // workgroups per jump; each takes 39 of g's 312 words (2496 coefficients)
constexpr int JUMP_SPAN = (312 / JUMP_PARTS) * 64;                  // 2496 = 4 generator blocks
constexpr int JUMP_WORDS = 624 + JUMP_SPAN;                         // window a part reads: x[i+j]
constexpr int JUMP_LDS_WORDS = JUMP_WORDS + 3 * JUMP_THREADS;       // + zero words (15.5 KB in all)
static_assert(JUMP_SPAN % 624 == 0, "a part's first coefficient must sit on a generator block boundary");

// Workgroup `part` of JUMP_PARTS XORs its share of  dst[j] ^= x[i + j]  (i over the set bits of its
// slice of g) into dst with atomics; dst must be zero beforehand.  The part first advances the source
// state by part*JUMP_SPAN words (whole 624-word blocks, regenerated in place), then extends it
// linearly by JUMP_SPAN words, so only a 3120-word window lives in LDS.
__device__ __forceinline__ void mt_jump_block(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                              const unsigned long long* __restrict__ poly, const int part,
                                              uint32_t* x) {
    const int tid = threadIdx.x;
    for (int k = tid; k < 624; k += JUMP_THREADS) x[k] = src[k];
    x[JUMP_WORDS + tid] = 0u;
    x[JUMP_WORDS + JUMP_THREADS + tid] = 0u;
    x[JUMP_WORDS + 2 * JUMP_THREADS + tid] = 0u;
    __syncthreads();
    // advance by whole blocks, in place: three chunks of <= 227 words, reads before writes
    for (int blk = 0; blk < part * (JUMP_SPAN / 624); ++blk) {
        for (int c0 = 0; c0 < 624; c0 += 227) {
            const int k = c0 + tid;
            uint32_t nw = 0;
            const bool on = tid < 227 && k < 624;
            if (on) {
                const uint32_t xa = x[k];
                const uint32_t xb = x[(k + 1 == 624) ? 0 : k + 1];
                const uint32_t xm = x[(k + 397 >= 624) ? k + 397 - 624 : k + 397];
                nw = mt_twist(xa, xb, xm);
            }
            __syncthreads();
            if (on) x[k] = nw;
            __syncthreads();
        }
    }
    // linear extension: word n needs n-624, n-623 and n-227, all older than its chunk
    for (int base = 624; base < JUMP_WORDS; base += 227) {
        const int n = base + tid;
        if (tid < 227 && n < JUMP_WORDS) {
            const uint32_t xa = x[n - 624], xb = x[n - 623], xm = x[n - 227];
            x[n] = mt_twist(xa, xb, xm);
        }
        __syncthreads();
    }
    const RT_CONST unsigned long long* g =
        (const RT_CONST unsigned long long*)(const void*)(poly + part * (312 / JUMP_PARTS));
    // lane handles j = tid, tid + 256 and (tid < 112) tid + 512; the third read of the other lanes
    // stays inside the zero-padded buffer and is discarded
    uint32_t a0 = 0, a1 = 0, a2 = 0;
    const uint32_t* xj = x + tid;
    for (int w = 0; w < 312 / JUMP_PARTS; ++w) {
        unsigned long long bits = g[w];  // scalar load: the loop over set bits is wave-uniform
        const uint32_t* xw = xj + 64 * w;
        while (bits) {
            // two set bits per trip: six LDS reads in flight
            const int b0 = __builtin_ctzll(bits);
            bits &= bits - 1ull;
            const bool more = bits != 0ull;
            const int b1 = more ? __builtin_ctzll(bits) : 0;
            bits &= bits - 1ull;
            const uint32_t* p0 = xw + b0;
            const uint32_t* p1 = more ? xw + b1 : x + JUMP_WORDS + tid;  // zeros
            const uint32_t u0 = p0[0], u1 = p0[JUMP_THREADS], u2 = p0[2 * JUMP_THREADS];
            const uint32_t v0 = p1[0], v1 = p1[JUMP_THREADS], v2 = p1[2 * JUMP_THREADS];
            a0 ^= u0 ^ v0;
            a1 ^= u1 ^ v1;
            a2 ^= u2 ^ v2;
        }
    }
    atomicXor(&dst[tid], a0);
    atomicXor(&dst[tid + JUMP_THREADS], a1);
    if (tid + 2 * JUMP_THREADS < 624) atomicXor(&dst[tid + 2 * JUMP_THREADS], a2);
}

// single jump: dst ^= g(A) src, dst zeroed by the caller and distinct from src.  grid (1, JUMP_PARTS)
__global__ __launch_bounds__(JUMP_THREADS) void mt_jump_kernel(const uint32_t* src, uint32_t* dst,
                                                               const unsigned long long* poly) {
    extern __shared__ uint32_t jump_x[];
    mt_jump_block(src, dst, poly, blockIdx.y, jump_x);
}

// one level of the doubling tree over an array of stream states (all but states[0] zeroed beforehand):
// blocks (q, part) compute states[q*2*step + step] = g(A) states[q*2*step]   (g = x^(stride*step));
// blocks whose target lies beyond n do nothing.  grid (ceil(n / 2step), JUMP_PARTS)
__global__ __launch_bounds__(JUMP_THREADS) void mt_jump_tree_kernel(uint32_t* states, const unsigned long long* poly,
                                                                    const int step, const int n) {
    extern __shared__ uint32_t jump_x[];
    const long long srci = (long long)blockIdx.x * 2 * step;
    const long long dsti = srci + step;
    if (dsti >= n) return;
    mt_jump_block(states + srci * 624, states + dsti * 624, poly, blockIdx.y, jump_x);
}

// Counter-based FAST sampler (RT_SAMPLER_PHILOX; not the reference's stream, reported separately):
// sample s of the global stream gets Philox4x32-10(counter = {s, 0, 0}, key = {seed, 0}) (Salmon et al.,
// SC'11); the four words become the (ox, oy) pair with the same two-words-per-double conversion as
// generate_canonical.  No state, no jump-ahead: any band / frame / GPU computes its samples directly.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

__global__ __launch_bounds__(256) void philox_jitter_kernel(double* __restrict__ out, const unsigned long long n_samples,
                                                            const unsigned long long sample_base, const uint32_t key0,
                                                            const uint32_t key1) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_samples) return;
    const unsigned long long sidx = sample_base + i;
    uint32_t w[4];
    philox4x32_10((uint32_t)sidx, (uint32_t)(sidx >> 32), 0u, 0u, key0, key1, w);
    d2 r;
    {
        double v = ((double)w[0] + (double)w[1] * 4294967296.0) / 18446744073709551616.0;
        if (v >= 1.0) v = 0x1.fffffffffffffp-1;
        r.x = v - 0.5;
    }
    {
        double v = ((double)w[2] + (double)w[3] * 4294967296.0) / 18446744073709551616.0;
        if (v >= 1.0) v = 0x1.fffffffffffffp-1;
        r.y = v - 0.5;
    }
    __builtin_nontemporal_store(r, reinterpret_cast<d2*>(out) + i);
}

// every stream state advanced by the same distance: dst[i] = g(A) src[i], grid (n, JUMP_PARTS); dst zeroed
// beforehand.  A frame loop with a constant distance between frames needs only this one launch per frame
// instead of the doubling tree.
__global__ __launch_bounds__(JUMP_THREADS) void mt_jump_all_kernel(const uint32_t* src, uint32_t* dst,
                                                                   const unsigned long long* poly) {
    extern __shared__ uint32_t jump_x[];
    const long long i = blockIdx.x;
    mt_jump_block(src + i * 624, dst + i * 624, poly, blockIdx.y, jump_x);
}

}  // namespace rtk
