/*
 * oracle/rt_oracle.c -- TEST INFRASTRUCTURE ONLY. NOT PART OF THE PRODUCT PATH.
 *
 * Plain-C, single-threaded, scalar CPU restatement of the render hot path of
 * AnalogArnold/rayTracer (reference mounted at /root/reference, never copied):
 *   host BVH build   cpp/rtbvh.cpp:36-685, cpp/rtbvh.h:33-167
 *   traversal        cpp/rtrayintersection.cpp:124-228
 *   Moller-Trumbore  cpp/rtrayintersection.cpp:20-122
 *   shading / image  cpp/rtrender.cpp:13-82
 *   RNG              cpp/rtmathutils.h:15-19 (libstdc++ mt19937 + generate_canonical<double,53>)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker.  The product (raytracer_amd/) never
 * links, imports or calls it.
 *
 * PARITY STATUS: the reference's hot-path .cpp files include <nanobind/...>
 * (cpp/rtbvh.h:11-14, cpp/rtrender.h:9-12) and nanobind is absent from this
 * image, so the reference is unbuildable here and no stand-in headers are
 * written; the reference holds no test or golden vector for this path.  What
 * pins this file instead (DESIGN.md section 6 has the table):
 *   - orc_rng_* against the reference's own header cpp/rtmathutils.h compiled
 *     from the mount (oracle/eigen_probe.cpp -> oracle/_ref/, golden vectors in
 *     tests/golden/) and against MT19937's published known answer;
 *   - the floating-point evaluation order of every Eigen expression on the path
 *     (normalized(), rowwise().sum(), cwise products, minCoeff, 1/array) against
 *     the reference's vendored Eigen headers evaluated on the same expression
 *     shapes (oracle/eigen_probe.cpp), tests/test_oracle_pins.py;
 *   - the camera golden values recorded in SURVEY.md section 8(c);
 *   - the builder's branches, the TLAS slot quirk and the walk's tie rules against
 *     hand-derivable fixtures (oracle/derive_tree_fixtures.py, a line-by-line third
 *     statement evaluated in binary64 and in exact rationals; derivation in
 *     tests/golden/tree_fixtures_derivation.txt), tests/test_tree_fixtures.py;
 *     the same statement in binary64 alone on irregular meshes of a few thousand
 *     triangles (decisions on roundings and ties), node for node;
 *   - the walk's hit ids against orc_brute_force below -- every triangle tested
 *     with the pinned Moller-Trumbore row, no tree -- tests/test_treeless.py;
 *   - the survey's visit statistics of the compiled reference (43.9 / 4.39 / 0.906).
 * Still unpinned by anything reference-derived: builder decisions that hinge on a
 * rounding (three statements of one reading of the source agree there), and whole
 * images against a file of the reference (it ships none that can be regenerated).
 *
 * All arithmetic is IEEE binary64, round-to-nearest-even, no FMA contraction
 * (compile with -ffp-contract=off, never -ffast-math), comparisons false on NaN.
 * Every three-term sum is (a+b)+c as in Eigen's redux for 3 columns.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------ */
/* a10  RNG: static std::mt19937 generator (seed 5489) +                     */
/*      std::uniform_real_distribution<double>(0,1)   rtmathutils.h:15-19    */
/* ------------------------------------------------------------------------ */
typedef struct {
    uint32_t mt[624];
    int idx;
} orc_rng;

static void rng_seed(orc_rng *r, uint32_t seed) {
    r->mt[0] = seed;
    for (int i = 1; i < 624; ++i)
        r->mt[i] = 1812433253u * (r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) + (uint32_t)i;
    r->idx = 624;
}

static void rng_refill(orc_rng *r) {
    uint32_t *mt = r->mt;
    for (int k = 0; k < 624; ++k) {
        uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
        mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    r->idx = 0;
}

static uint32_t rng_u32(orc_rng *r) {
    if (r->idx >= 624) rng_refill(r);
    uint32_t y = r->mt[r->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* libstdc++ generate_canonical<double,53>(mt19937): k = 2 draws,
 * sum = a + b*2^32 (one rounding), ret = sum / 2^64, clamp to nextafter(1,0). */
static double rng_double(orc_rng *r) {
    double a = (double)rng_u32(r);
    double b = (double)rng_u32(r);
    double sum = a + b * 4294967296.0;
    double ret = sum / 18446744073709551616.0;
    if (ret >= 1.0) ret = nextafter(1.0, 0.0);
    return ret;
}

ORC_API orc_rng *orc_rng_new(uint32_t seed) {
    orc_rng *r = (orc_rng *)malloc(sizeof(orc_rng));
    if (r) rng_seed(r, seed);
    return r;
}
ORC_API void orc_rng_free(orc_rng *r) { free(r); }
ORC_API void orc_rng_seed(orc_rng *r, uint32_t seed) { rng_seed(r, seed); }
ORC_API uint32_t orc_rng_u32(orc_rng *r) { return rng_u32(r); }
ORC_API double orc_rng_double(orc_rng *r) { return rng_double(r); }
ORC_API void orc_rng_discard(orc_rng *r, uint64_t n) {
    while (n--) (void)rng_u32(r);
}
ORC_API void orc_rng_fill_u32(orc_rng *r, uint32_t *out, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) out[i] = rng_u32(r);
}
ORC_API void orc_rng_fill_double(orc_rng *r, double *out, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) out[i] = rng_double(r);
}
/* raw state injection: lets the full-size checks start the oracle's generator at a far stream
 * position (the state comes from a jump that tests/test_mt_jump.py validates against discard()) */
ORC_API void orc_rng_set_state(orc_rng *r, const uint32_t *mt624) {
    memcpy(r->mt, mt624, sizeof(r->mt));
    r->idx = 624; /* all 624 words consumed: the next draw regenerates the block */
}
/* raw state access: lets tests compare against the product's jump-ahead */
ORC_API void orc_rng_get_state(const orc_rng *r, uint32_t *mt624, int *idx) {
    memcpy(mt624, r->mt, sizeof(r->mt));
    *idx = r->idx;
}

/* ------------------------------------------------------------------------ */
/* AABB  rtbvh.h:33-81  (std::min / std::max semantics spelled out)          */
/* ------------------------------------------------------------------------ */
typedef struct {
    double mn[3], mx[3];
} aabb;

static inline double std_min(double a, double b) { return (b < a) ? b : a; }
static inline double std_max(double a, double b) { return (a < b) ? b : a; }

static void aabb_init(aabb *b) {
    b->mn[0] = b->mn[1] = b->mn[2] = INFINITY;
    b->mx[0] = b->mx[1] = b->mx[2] = -INFINITY;
}
static void aabb_add_point(aabb *b, const double *p) { /* rtbvh.h:55-61 */
    for (int i = 0; i < 3; ++i) {
        b->mn[i] = std_min(b->mn[i], p[i]);
        b->mx[i] = std_max(b->mx[i], p[i]);
    }
}
static void aabb_add_aabb(aabb *b, const aabb *o) { /* rtbvh.h:63-68 */
    for (int i = 0; i < 3; ++i) {
        b->mn[i] = std_min(b->mn[i], o->mn[i]);
        b->mx[i] = std_max(b->mx[i], o->mx[i]);
    }
}
static double aabb_extent(const aabb *b, int axis) { /* rtbvh.h:70-74 */
    double r = b->mx[axis] - b->mn[axis];
    if (r < 0) return 0.0;
    return r;
}
static double aabb_area(const aabb *b) { /* rtbvh.h:75-80 */
    double height = aabb_extent(b, 2);
    double width = aabb_extent(b, 1);
    double depth = aabb_extent(b, 0);
    return 2 * (height * width + width * depth + height * depth);
}

/* ------------------------------------------------------------------------ */
/* Tree containers  rtbvh.h:97-144                                           */
/* ------------------------------------------------------------------------ */
typedef struct {
    aabb box;
    size_t element_count; /* != 0 -> leaf */
    int left;             /* -1 -> leaf; right child = left + 1 */
    int min_idx;          /* node_minimum_element_index[i], rtbvh.cpp:322,369-370 */
    double *coords;       /* leaf-owned copy: count*9 (v0 v1 v2), rtbvh.cpp:526-530 */
    double *colors;       /* leaf-owned copy: count*3, rtbvh.cpp:533-536 */
} blas_node;

typedef struct {
    blas_node *nodes;
    int n_nodes, cap_nodes;
    aabb box;
    int root;
    int *elem_idx; /* mesh_element_indices after the build (sorted order) */
    int n_elem;
    int mesh; /* original mesh index (for hit ids; the reference keeps none) */
} blas;

typedef struct {
    aabb box;
    int blas_count, left, min_blas_idx;
} tlas_node;

typedef struct {
    blas *blases; /* tlas.blases: *stored* (slot) order, rtbvh.cpp:544-561 */
    int n_blases;
    tlas_node *nodes;
    int n_nodes, cap_nodes;
    int *blas_idx; /* scene_blas_indices after the TLAS build */
    long midpoint_msgs; /* count of "SAH splitting failed" lines, rtbvh.cpp:128 */
} orc_scene;

typedef struct {
    size_t element_count;
    int node_idx;
    int min_idx;
} build_task; /* rtbvh.h:90-94 */

/* ------------------------------------------------------------------------ */
/* Splitter  rtbvh.cpp:81-299 (shared by BLAS and TLAS builds)               */
/* ------------------------------------------------------------------------ */
static aabb node_aabb(const aabb *el_aabb, const int *idx, int min_idx, int count) {
    aabb b;
    aabb_init(&b);
    for (int i = min_idx; i < min_idx + count; ++i) aabb_add_aabb(&b, &el_aabb[idx[i]]);
    return b;
}

/* rtbvh.cpp:108-120 */
static double sah_cost_full(unsigned nl, unsigned nr, const aabb *lb, const aabb *rb, const aabb *pb) {
    double cost_traversal = 1.0;
    double cost_intersection = 2.0;
    double area_parent = aabb_area(pb);
    double area_left = aabb_area(lb);
    double area_right = aabb_area(rb);
    return cost_traversal +
           cost_intersection * (area_left / area_parent * nl + area_right / area_parent * nr);
}

#define NUM_BINS 8

/* rtbvh.cpp:132-247 */
static int binned_sah_split(size_t count, int min_idx, const double (*cent)[3], const aabb *el_aabb,
                            const int *idx, unsigned *out_axis, double *out_pos, long *msgs) {
    if (count <= 2) return 0;
    unsigned max_idx = (unsigned)min_idx + (unsigned)count;

    aabb cb;
    aabb_init(&cb);
    for (unsigned i = (unsigned)min_idx; i < max_idx; ++i) aabb_add_point(&cb, cent[idx[i]]);

    int best_axis = 0;
    double axis_extent = aabb_extent(&cb, 0);
    for (int i = 1; i < 3; ++i) {
        double e = aabb_extent(&cb, i);
        if (e > axis_extent) {
            best_axis = i;
            axis_extent = e;
        }
    }
    *out_axis = (unsigned)best_axis;

    if (axis_extent == 0) { /* rtbvh.cpp:169-173, midpoint_split :122-130 */
        ++*msgs;
        *out_pos = cb.mn[best_axis] + axis_extent * 0.5;
        return 1;
    }

    aabb bin_box[NUM_BINS];
    int bin_cnt[NUM_BINS];
    for (int i = 0; i < NUM_BINS; ++i) {
        aabb_init(&bin_box[i]);
        bin_cnt[i] = 0;
    }
    const double inverse_extent = 1.0 / axis_extent;
    for (unsigned i = (unsigned)min_idx; i < max_idx; ++i) {
        int e = idx[i];
        double t = (cent[e][best_axis] - cb.mn[best_axis]) * inverse_extent;
        int bin_id = (int)(t * NUM_BINS);
        if (bin_id == NUM_BINS) bin_id = NUM_BINS - 1;
        bin_cnt[bin_id]++;
        aabb_add_aabb(&bin_box[bin_id], &el_aabb[e]);
    }

    unsigned left_count[NUM_BINS], right_count[NUM_BINS];
    aabb left_bounds[NUM_BINS], right_bounds[NUM_BINS];
    aabb run;
    unsigned run_cnt = 0;
    aabb_init(&run);
    for (int i = 0; i < NUM_BINS; ++i) {
        if (bin_cnt[i] > 0) aabb_add_aabb(&run, &bin_box[i]);
        run_cnt += (unsigned)bin_cnt[i];
        left_bounds[i] = run;
        left_count[i] = run_cnt;
    }
    aabb_init(&run);
    run_cnt = 0;
    for (int i = NUM_BINS - 1; i >= 0; --i) {
        if (bin_cnt[i] > 0) aabb_add_aabb(&run, &bin_box[i]);
        run_cnt += (unsigned)bin_cnt[i];
        right_bounds[i] = run;
        right_count[i] = run_cnt;
    }

    double best_cost = INFINITY;
    int best_bin = -1;
    aabb parent = node_aabb(el_aabb, idx, min_idx, (int)count);
    for (int i = 0; i < NUM_BINS - 1; ++i) {
        unsigned nl = left_count[i], nr = right_count[i + 1];
        if (nl == 0 || nr == 0) continue;
        double cost = sah_cost_full(nl, nr, &left_bounds[i], &right_bounds[i + 1], &parent);
        if (cost < best_cost) {
            best_cost = cost;
            best_bin = i;
        }
    }
    if (best_bin == -1) { /* rtbvh.cpp:236-240 */
        ++*msgs;
        *out_pos = cb.mn[best_axis] + axis_extent * 0.5;
        return 1;
    }
    double bin_width = axis_extent / NUM_BINS;
    *out_pos = cb.mn[best_axis] + bin_width * (best_bin + 1);
    return 1;
}

/* rtbvh.cpp:249-299: two-pointer unstable partition; permutes idx even when it fails */
static int split_node(size_t count, int min_idx, const double (*cent)[3], const aabb *el_aabb, int *idx,
                      int *out_left_min, size_t *out_left_count, long *msgs) {
    unsigned axis = 0;
    double pos = 0.0;
    if (!binned_sah_split(count, min_idx, cent, el_aabb, idx, &axis, &pos, msgs)) return 0;
    unsigned begin = (unsigned)min_idx, end = begin + (unsigned)count, mid = begin;
    while (mid < end) {
        int e = idx[mid];
        if (cent[e][axis] < pos) {
            ++mid;
        } else {
            --end;
            int tmp = idx[mid];
            idx[mid] = idx[end];
            idx[end] = tmp;
        }
    }
    size_t left_count = mid - begin;
    size_t right_count = count - left_count;
    *out_left_count = left_count;
    *out_left_min = (int)begin;
    if (left_count == 0 || right_count == 0) return 0;
    return 1;
}

/* ------------------------------------------------------------------------ */
/* BLAS build  rtbvh.cpp:301-390, data copy :478-542                         */
/* ------------------------------------------------------------------------ */
static void blas_push(blas *b, const aabb *box, size_t count, int left, int min_idx) {
    if (b->n_nodes == b->cap_nodes) {
        b->cap_nodes = b->cap_nodes ? b->cap_nodes * 2 : 64;
        b->nodes = (blas_node *)realloc(b->nodes, (size_t)b->cap_nodes * sizeof(blas_node));
    }
    blas_node *n = &b->nodes[b->n_nodes++];
    n->box = *box;
    n->element_count = count;
    n->left = left;
    n->min_idx = min_idx;
    n->coords = NULL;
    n->colors = NULL;
}

static void build_blas(blas *b, const double (*cent)[3], const aabb *el_aabb, int *idx, size_t n_elem,
                       long *msgs) {
    const int MAX_LEAF = 4; /* rtbvh.cpp:308 */
    aabb root_box = node_aabb(el_aabb, idx, 0, (int)n_elem);
    blas_push(b, &root_box, n_elem, -1, 0);
    b->root = 0;

    size_t cap = 64, sp = 0;
    build_task *stack = (build_task *)malloc(cap * sizeof(build_task));
    stack[sp++] = (build_task){n_elem, 0, 0};
    while (sp) {
        build_task task = stack[--sp];
        int node_idx = task.node_idx, min_idx = task.min_idx, count = (int)task.element_count;
        if (count <= MAX_LEAF) {
            blas_node *n = &b->nodes[node_idx];
            n->element_count = (size_t)count;
            n->box = node_aabb(el_aabb, idx, min_idx, count);
            n->left = -1;
            continue;
        }
        int left_min = min_idx;
        size_t left_count = 0;
        if (!split_node(task.element_count, min_idx, cent, el_aabb, idx, &left_min, &left_count, msgs)) {
            blas_node *n = &b->nodes[node_idx];
            n->element_count = (size_t)count;
            n->box = node_aabb(el_aabb, idx, min_idx, count);
            n->left = -1;
            continue;
        }
        int left_child = b->n_nodes;
        size_t right_count = (size_t)count - left_count;
        int right_min = left_min + (int)left_count;
        aabb lb = node_aabb(el_aabb, idx, left_min, (int)left_count);
        blas_push(b, &lb, left_count, -1, left_min);
        aabb rb = node_aabb(el_aabb, idx, right_min, (int)right_count);
        blas_push(b, &rb, right_count, -1, right_min);
        b->nodes[node_idx].left = left_child;
        b->nodes[node_idx].element_count = 0;
        if (sp + 2 > cap) {
            cap *= 2;
            stack = (build_task *)realloc(stack, cap * sizeof(build_task));
        }
        stack[sp++] = (build_task){right_count, left_child + 1, right_min};
        stack[sp++] = (build_task){left_count, left_child, left_min};
    }
    free(stack);
}

static void copy_data_to_blas(blas *b, const double *coords_t, const double *colors_t) {
    for (int i = 0; i < b->n_nodes; ++i) {
        blas_node *n = &b->nodes[i];
        size_t cnt = n->element_count;
        if (!cnt) continue;
        n->coords = (double *)malloc(cnt * 9 * sizeof(double));
        n->colors = (double *)malloc(cnt * 3 * sizeof(double));
        for (size_t k = 0; k < cnt; ++k) {
            int orig = b->elem_idx[n->min_idx + (int)k];
            memcpy(n->coords + k * 9, coords_t + (size_t)orig * 9, 9 * sizeof(double));
            memcpy(n->colors + k * 3, colors_t + (size_t)orig * 3, 3 * sizeof(double));
        }
    }
}

/* ------------------------------------------------------------------------ */
/* TLAS build  rtbvh.cpp:392-476, BLAS placement :544-561                    */
/* ------------------------------------------------------------------------ */
static void tlas_push(orc_scene *s, const aabb *box, int count, int left, int min_blas_idx) {
    if (s->n_nodes == s->cap_nodes) {
        s->cap_nodes = s->cap_nodes ? s->cap_nodes * 2 : 16;
        s->nodes = (tlas_node *)realloc(s->nodes, (size_t)s->cap_nodes * sizeof(tlas_node));
    }
    tlas_node *n = &s->nodes[s->n_nodes++];
    n->box = *box;
    n->blas_count = count;
    n->left = left;
    n->min_blas_idx = min_blas_idx;
}

static void build_tlas(orc_scene *s, const double (*cent)[3], const aabb *boxes, int *idx, size_t n_mesh) {
    const int MAX_LEAF = 2; /* rtbvh.cpp:398 */
    aabb root_box = node_aabb(boxes, idx, 0, (int)n_mesh);
    tlas_push(s, &root_box, (int)n_mesh, -1, 0);

    size_t cap = 64, sp = 0;
    build_task *stack = (build_task *)malloc(cap * sizeof(build_task));
    stack[sp++] = (build_task){n_mesh, 0, 0};
    while (sp) {
        build_task task = stack[--sp];
        int node_idx = task.node_idx, min_idx = task.min_idx, count = (int)task.element_count;
        if (count <= MAX_LEAF) {
            tlas_node *n = &s->nodes[node_idx];
            n->min_blas_idx = min_idx;
            n->blas_count = count;
            n->box = node_aabb(boxes, idx, min_idx, count);
            n->left = -1;
            continue;
        }
        int left_min = min_idx;
        size_t left_count = 0;
        if (!split_node(task.element_count, min_idx, cent, boxes, idx, &left_min, &left_count,
                        &s->midpoint_msgs)) {
            tlas_node *n = &s->nodes[node_idx];
            n->min_blas_idx = min_idx;
            n->blas_count = count;
            n->box = node_aabb(boxes, idx, min_idx, count);
            n->left = -1;
            continue;
        }
        int left_child = s->n_nodes;
        size_t right_count = (size_t)count - left_count;
        int right_min = left_min + (int)left_count;
        aabb lb = node_aabb(boxes, idx, left_min, (int)left_count);
        tlas_push(s, &lb, (int)left_count, -1, left_min);
        aabb rb = node_aabb(boxes, idx, right_min, (int)right_count);
        tlas_push(s, &rb, (int)right_count, -1, right_min);
        s->nodes[node_idx].left = left_child;
        s->nodes[node_idx].blas_count = 0;
        if (sp + 2 > cap) {
            cap *= 2;
            stack = (build_task *)realloc(stack, cap * sizeof(build_task));
        }
        stack[sp++] = (build_task){right_count, left_child + 1, right_min};
        stack[sp++] = (build_task){left_count, left_child, left_min};
    }
    free(stack);
}

/* ------------------------------------------------------------------------ */
/* build_acceleration_structures  rtbvh.cpp:590-685                          */
/*   coords[m]: (T,E_m,3,3) C-contiguous f64, colors[m]: (T,E_m,3)           */
/* ------------------------------------------------------------------------ */
ORC_API void orc_scene_free(orc_scene *s);

ORC_API orc_scene *orc_build(int n_meshes, const double *const *coords, const double *const *colors,
                             const int64_t *elem_counts, int timestep) {
    orc_scene *s = (orc_scene *)calloc(1, sizeof(orc_scene));
    blas *scene_blases = (blas *)calloc((size_t)n_meshes, sizeof(blas));
    double(*mesh_cent)[3] = (double(*)[3])malloc((size_t)n_meshes * sizeof(double[3]));
    aabb *mesh_box = (aabb *)malloc((size_t)n_meshes * sizeof(aabb));

    for (int m = 0; m < n_meshes; ++m) {
        size_t E = (size_t)elem_counts[m];
        const double *ct = coords[m] + (size_t)timestep * E * 9;  /* 64-bit stride (quirk 5) */
        const double *kt = colors[m] + (size_t)timestep * E * 3;
        double(*cent)[3] = (double(*)[3])malloc((E ? E : 1) * sizeof(double[3]));
        aabb *el = (aabb *)malloc((E ? E : 1) * sizeof(aabb));
        aabb_init(&mesh_box[m]);
        for (size_t e = 0; e < E; ++e) { /* process_element_data_tri3, rtbvh.cpp:36-79 */
            const double *tc = ct + e * 9;
            for (int i = 0; i < 3; ++i) cent[e][i] = (tc[i] + tc[i + 3] + tc[i + 6]) / 3.0;
            aabb_init(&el[e]);
            for (int node = 0; node < 3; ++node) aabb_add_point(&el[e], tc + node * 3);
            aabb_add_aabb(&mesh_box[m], &el[e]);
        }
        for (int i = 0; i < 3; ++i) mesh_cent[m][i] = (mesh_box[m].mn[i] + mesh_box[m].mx[i]) / 2.0;

        blas *b = &scene_blases[m];
        b->mesh = m;
        b->n_elem = (int)E;
        b->elem_idx = (int *)malloc((E ? E : 1) * sizeof(int));
        for (size_t e = 0; e < E; ++e) b->elem_idx[e] = (int)e;
        build_blas(b, (const double(*)[3])cent, el, b->elem_idx, E, &s->midpoint_msgs);
        copy_data_to_blas(b, ct, kt);
        b->box = mesh_box[m];
        free(cent);
        free(el);
    }

    s->blas_idx = (int *)malloc((size_t)(n_meshes ? n_meshes : 1) * sizeof(int));
    for (int m = 0; m < n_meshes; ++m) s->blas_idx[m] = m;
    build_tlas(s, (const double(*)[3])mesh_cent, mesh_box, s->blas_idx, (size_t)n_meshes);

    /* copy_data_to_TLAS: append in TLAS-*node-index* order (quirk 1) */
    s->blases = (blas *)calloc((size_t)(n_meshes ? n_meshes : 1), sizeof(blas));
    s->n_blases = 0;
    for (int i = 0; i < s->n_nodes; ++i) {
        tlas_node *n = &s->nodes[i];
        for (int j = n->min_blas_idx; j < n->min_blas_idx + n->blas_count; ++j)
            s->blases[s->n_blases++] = scene_blases[s->blas_idx[j]]; /* shallow move: each mesh once */
    }
    free(scene_blases);
    free(mesh_cent);
    free(mesh_box);
    return s;
}

ORC_API void orc_scene_free(orc_scene *s) {
    if (!s) return;
    for (int k = 0; k < s->n_blases; ++k) {
        blas *b = &s->blases[k];
        for (int i = 0; i < b->n_nodes; ++i) {
            free(b->nodes[i].coords);
            free(b->nodes[i].colors);
        }
        free(b->nodes);
        free(b->elem_idx);
    }
    free(s->blases);
    free(s->nodes);
    free(s->blas_idx);
    free(s);
}

/* ---- structure export (tests compare the product's flattened tree with this) ---- */
ORC_API void orc_scene_counts(const orc_scene *s, int *n_tlas_nodes, int *n_slots, long *midpoint_msgs) {
    *n_tlas_nodes = s->n_nodes;
    *n_slots = s->n_blases;
    *midpoint_msgs = s->midpoint_msgs;
}
ORC_API void orc_tlas_nodes(const orc_scene *s, double *boxes6, int *blas_count, int *left, int *min_blas_idx) {
    for (int i = 0; i < s->n_nodes; ++i) {
        memcpy(boxes6 + 6 * i, s->nodes[i].box.mn, 3 * sizeof(double));
        memcpy(boxes6 + 6 * i + 3, s->nodes[i].box.mx, 3 * sizeof(double));
        blas_count[i] = s->nodes[i].blas_count;
        left[i] = s->nodes[i].left;
        min_blas_idx[i] = s->nodes[i].min_blas_idx;
    }
}
ORC_API void orc_slot_info(const orc_scene *s, int slot, int *mesh, int *n_nodes, int *n_elem) {
    *mesh = s->blases[slot].mesh;
    *n_nodes = s->blases[slot].n_nodes;
    *n_elem = s->blases[slot].n_elem;
}
ORC_API void orc_slot_nodes(const orc_scene *s, int slot, double *boxes6, int *left, int64_t *count,
                            int *min_idx) {
    const blas *b = &s->blases[slot];
    for (int i = 0; i < b->n_nodes; ++i) {
        memcpy(boxes6 + 6 * i, b->nodes[i].box.mn, 3 * sizeof(double));
        memcpy(boxes6 + 6 * i + 3, b->nodes[i].box.mx, 3 * sizeof(double));
        left[i] = b->nodes[i].left;
        count[i] = (int64_t)b->nodes[i].element_count;
        min_idx[i] = b->nodes[i].min_idx;
    }
}
ORC_API void orc_slot_indices(const orc_scene *s, int slot, int *idx) {
    memcpy(idx, s->blases[slot].elem_idx, (size_t)s->blases[slot].n_elem * sizeof(int));
}

/* ------------------------------------------------------------------------ */
/* Hot path                                                                  */
/* ------------------------------------------------------------------------ */
typedef struct {
    double t;
    double bary[3];
    double fc[3];
    int slot, node, row; /* hit id pieces (the reference records none) */
} hit_record;

typedef struct {
    uint64_t nodes; /* TLAS + BLAS nodes popped (slab-tested) */
    uint64_t tris;  /* leaf triangles tested */
} visit_counters;

/* a6  intersect_AABB  rtrayintersection.cpp:124-144; ray.t_max is always +inf */
static int slab(const double o[3], const double d[3], const aabb *b) {
    double ta[6];
    double inv[3] = {1 / d[0], 1 / d[1], 1 / d[2]};
    for (int i = 0; i < 3; ++i) {
        ta[2 * i] = (b->mn[i] - o[i]) * inv[i];
        ta[2 * i + 1] = (b->mx[i] - o[i]) * inv[i];
    }
    double t_min = std_max(std_max(std_min(ta[0], ta[1]), std_min(ta[2], ta[3])), std_min(ta[4], ta[5]));
    double t_max = std_min(std_min(std_max(ta[0], ta[1]), std_max(ta[2], ta[3])), std_max(ta[4], ta[5]));
    return t_min <= t_max && t_max > 0.0 && t_min < INFINITY;
}

/* a7  one row of intersect_bvh_triangles  rtrayintersection.cpp:36-122
 * (rows are independent; the early returns are pure shortcuts) */
static double moller_trumbore(const double o[3], const double d[3], const double *tri, double bary[3]) {
    double e0[3], e2[3], p[3], tv[3], q[3];
    for (int j = 0; j < 3; ++j) {
        e0[j] = tri[3 + j] - tri[j];
        e2[j] = tri[6 + j] - tri[j];
    }
    p[0] = d[1] * e2[2] - d[2] * e2[1];
    p[1] = d[2] * e2[0] - d[0] * e2[2];
    p[2] = d[0] * e2[1] - d[1] * e2[0];
    double det = (e0[0] * p[0] + e0[1] * p[1]) + e0[2] * p[2];
    int ok = (det > 1e-6) && (det > 0);
    double inv = 1.0 / det;
    for (int j = 0; j < 3; ++j) tv[j] = o[j] - tri[j];
    double u = ((tv[0] * p[0] + tv[1] * p[1]) + tv[2] * p[2]) * inv;
    ok = ok && (u >= 0) && (u <= 1);
    q[0] = tv[1] * e0[2] - tv[2] * e0[1];
    q[1] = tv[2] * e0[0] - tv[0] * e0[2];
    q[2] = tv[0] * e0[1] - tv[1] * e0[0];
    double v = ((d[0] * q[0] + d[1] * q[1]) + d[2] * q[2]) * inv;
    ok = ok && (v >= 0) && ((u + v) <= 1);
    double t = ((e2[0] * q[0] + e2[1] * q[1]) + e2[2] * q[2]) * inv;
    ok = ok && (t >= 0.0) && (t <= INFINITY);
    bary[0] = u;
    bary[1] = v;
    bary[2] = 1.0 - u - v;
    return ok ? t : INFINITY;
}

/* a5  intersect_BLAS  rtrayintersection.cpp:146-195 */
static void trace_blas(const double o[3], const double d[3], const blas *b, int slot, hit_record *rec,
                       visit_counters *vc, int **stk, size_t *cap) {
    size_t sp = 0;
    (*stk)[sp++] = b->root;
    while (sp) {
        int ni = (*stk)[--sp];
        const blas_node *n = &b->nodes[ni];
        vc->nodes++;
        if (!slab(o, d, &n->box)) continue;
        if (n->left == -1) {
            /* leaf: a7 over all rows, a8 first minimal t (strict < scan), then strict < vs record */
            size_t cnt = n->element_count;
            vc->tris += cnt;
            double best_t = 0.0, best_b[3] = {0, 0, 0};
            size_t best_row = 0;
            for (size_t r = 0; r < cnt; ++r) {
                double bb[3];
                double t = moller_trumbore(o, d, n->coords + r * 9, bb);
                if (r == 0 || t < best_t) {
                    best_t = t;
                    best_row = r;
                    memcpy(best_b, bb, sizeof(bb));
                }
            }
            if (cnt && best_t < rec->t) {
                rec->t = best_t;
                memcpy(rec->bary, best_b, sizeof(best_b));
                memcpy(rec->fc, n->colors + best_row * 3, 3 * sizeof(double));
                rec->slot = slot;
                rec->node = ni;
                rec->row = (int)best_row;
            }
        } else {
            if (sp + 2 > *cap) {
                *cap *= 2;
                *stk = (int *)realloc(*stk, *cap * sizeof(int));
            }
            int left = n->left, right = left + 1;
            if (right != 0) (*stk)[sp++] = right;
            if (left != -1) (*stk)[sp++] = left;
        }
    }
}

/* a4  intersect_TLAS  rtrayintersection.cpp:197-228.  Leaf indexes tlas.blases by
 * sorted-range position although it was filled in node-index order (quirk 1). */
static void trace_tlas(const double o[3], const double d[3], const orc_scene *s, hit_record *rec,
                       visit_counters *vc, int **stk, size_t *cap) {
    int tstk[256];
    int tsp = 0;
    tstk[tsp++] = 0;
    while (tsp) {
        const tlas_node *n = &s->nodes[tstk[--tsp]];
        vc->nodes++;
        if (!slab(o, d, &n->box)) continue;
        if (n->left == -1) {
            for (int i = n->min_blas_idx; i < n->min_blas_idx + n->blas_count; ++i)
                trace_blas(o, d, &s->blases[i], i, rec, vc, stk, cap);
        } else {
            int left = n->left, right = left + 1;
            if (tsp + 2 > 256) abort();
            if (right != 0) tstk[tsp++] = right;
            if (left != -1) tstk[tsp++] = left;
        }
    }
}

/* a3  return_ray_color  rtrender.cpp:13-33 (set_face_normal is dead work) */
static void ray_color(const double o[3], const double d[3], const orc_scene *s, double col[3], hit_record *rec,
                      visit_counters *vc, int **stk, size_t *cap) {
    rec->t = INFINITY;
    rec->bary[0] = rec->bary[1] = rec->bary[2] = 0.0;
    rec->fc[0] = rec->fc[1] = rec->fc[2] = 0.0;
    rec->slot = rec->node = rec->row = -1;
    trace_tlas(o, d, s, rec, vc, stk, cap);
    if (rec->t != INFINITY) {
        for (int c = 0; c < 3; ++c)
            col[c] = (rec->bary[0] * rec->fc[c] + rec->bary[1] * rec->fc[c]) + rec->bary[2] * rec->fc[c];
        return;
    }
    double a = 0.5 * (d[1] + 1.0);
    static const double blue[3] = {0.5, 0.7, 1.0};
    for (int c = 0; c < 3; ++c) col[c] = (1.0 - a) * 1.0 + a * blue[c];
}

static void hit_ids(const orc_scene *s, const hit_record *rec, int *mesh, int *elem, int *sorted_pos) {
    if (rec->slot < 0) {
        *mesh = *elem = *sorted_pos = -1;
        return;
    }
    const blas *b = &s->blases[rec->slot];
    int pos = b->nodes[rec->node].min_idx + rec->row;
    *mesh = b->mesh;
    *elem = b->elem_idx[pos];
    *sorted_pos = pos;
}

/* Arbitrary rays: origins/dirs are (n,3); dirs are used as given (no normalisation). */
ORC_API void orc_trace(const orc_scene *s, int64_t n, const double *origins, const double *dirs, double *out_t,
                       double *out_bary, double *out_color, int *out_slot, int *out_mesh, int *out_elem,
                       int *out_sorted_pos, uint64_t *out_nodes, uint64_t *out_tris) {
    size_t cap = 256;
    int *stk = (int *)malloc(cap * sizeof(int));
    visit_counters vc = {0, 0};
    for (int64_t i = 0; i < n; ++i) {
        hit_record rec;
        double col[3];
        ray_color(origins + 3 * i, dirs + 3 * i, s, col, &rec, &vc, &stk, &cap);
        if (out_t) out_t[i] = rec.t;
        if (out_bary) memcpy(out_bary + 3 * i, rec.bary, sizeof(rec.bary));
        if (out_color) memcpy(out_color + 3 * i, col, sizeof(col));
        int mesh, elem, pos;
        hit_ids(s, &rec, &mesh, &elem, &pos);
        if (out_slot) out_slot[i] = rec.slot;
        if (out_mesh) out_mesh[i] = mesh;
        if (out_elem) out_elem[i] = elem;
        if (out_sorted_pos) out_sorted_pos[i] = pos;
    }
    if (out_nodes) *out_nodes = vc.nodes;
    if (out_tris) *out_tris = vc.tris;
    free(stk);
}

/* a1/a2  render_ppm_image  rtrender.cpp:35-82 without the file write.
 * Renders rows [row0,row1) of an H x W image; the caller positions `rng`
 * (rows before row0 consume row0*W*spp*4 words).  If `jitter` is non-NULL it
 * supplies (ox,oy) = random_double()-0.5 pairs per sample in row-major
 * (j,i,k) order *relative to row0* and `rng` is not touched.
 * Outputs are indexed relative to row0. */
ORC_API void orc_render_rows(const orc_scene *s, const double cam[3], const double p00[3], const double sp[6],
                             int H, int W, int spp, int row0, int row1, orc_rng *rng, const double *jitter,
                             uint8_t *rgb, double *intensity, int *hit_mesh, int *hit_elem, double *hit_t,
                             uint64_t *out_nodes, uint64_t *out_tris, uint64_t *out_hits) {
    (void)H;
    size_t cap = 256;
    int *stk = (int *)malloc(cap * sizeof(int));
    visit_counters vc = {0, 0};
    uint64_t hits = 0;
    size_t pix = 0, smp = 0;
    for (int j = row0; j < row1; ++j) {
        for (int i = 0; i < W; ++i, ++pix) {
            double pc[3] = {0.0, 0.0, 0.0};
            for (int k = 0; k < spp; ++k, ++smp) {
                double off0, off1;
                if (jitter) {
                    off0 = jitter[2 * smp];
                    off1 = jitter[2 * smp + 1];
                } else {
                    off0 = rng_double(rng) - 0.5;
                    off1 = rng_double(rng) - 0.5;
                }
                double s0 = i + off0, s1 = j + off1;
                double dd[3], d[3];
                for (int c = 0; c < 3; ++c) {
                    double ps = (p00[c] + s0 * sp[c]) + s1 * sp[3 + c];
                    dd[c] = ps - cam[c];
                }
                /* Eigen normalized(): z = squaredNorm; z > 0 ? v / sqrt(z) : v  (Dot.h:92-102) */
                double z = (dd[0] * dd[0] + dd[1] * dd[1]) + dd[2] * dd[2];
                if (z > 0.0) {
                    double nz = sqrt(z);
                    for (int c = 0; c < 3; ++c) d[c] = dd[c] / nz;
                } else {
                    for (int c = 0; c < 3; ++c) d[c] = dd[c];
                }
                hit_record rec;
                double col[3];
                ray_color(cam, d, s, col, &rec, &vc, &stk, &cap);
                for (int c = 0; c < 3; ++c) pc[c] += col[c];
                if (rec.slot >= 0) ++hits;
                if (hit_mesh || hit_elem) {
                    int mesh, elem, pos;
                    hit_ids(s, &rec, &mesh, &elem, &pos);
                    if (hit_mesh) hit_mesh[smp] = mesh;
                    if (hit_elem) hit_elem[smp] = elem;
                }
                if (hit_t) hit_t[smp] = rec.t;
            }
            double gray = 0.2126 * pc[0] + 0.7152 * pc[1] + 0.0722 * pc[2];
            double inten = gray / spp;
            int gray_byte = (int)(inten * 255.99);
            if (intensity) intensity[pix] = inten;
            if (rgb) {
                rgb[3 * pix] = (uint8_t)gray_byte;
                rgb[3 * pix + 1] = (uint8_t)gray_byte;
                rgb[3 * pix + 2] = (uint8_t)gray_byte;
            }
        }
    }
    if (out_nodes) *out_nodes = vc.nodes;
    if (out_tris) *out_tris = vc.tris;
    if (out_hits) *out_hits = hits;
    free(stk);
}

/* ---- primitives exported for the pinning tests (tests/test_oracle_pins.py) ---- */
ORC_API void orc_normalized(const double v[3], double out[3]) {
    double z = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
    if (z > 0.0) {
        double nz = sqrt(z);
        for (int c = 0; c < 3; ++c) out[c] = v[c] / nz;
    } else {
        for (int c = 0; c < 3; ++c) out[c] = v[c];
    }
}
ORC_API void orc_pixel_sample(const double p00[3], const double sp[6], int i, int j, double off0, double off1,
                              double out[3]) {
    double s0 = i + off0, s1 = j + off1;
    for (int c = 0; c < 3; ++c) out[c] = (p00[c] + s0 * sp[c]) + s1 * sp[3 + c];
}
ORC_API int orc_slab(const double o[3], const double d[3], const double mn[3], const double mx[3]) {
    aabb b;
    memcpy(b.mn, mn, sizeof(b.mn));
    memcpy(b.mx, mx, sizeof(b.mx));
    return slab(o, d, &b);
}
ORC_API double orc_moller_trumbore(const double o[3], const double d[3], const double tri[9], double bary[3]) {
    return moller_trumbore(o, d, tri, bary);
}
ORC_API void orc_shade_hit(const double bary[3], const double fc[3], double out[3]) {
    for (int c = 0; c < 3; ++c) out[c] = (bary[0] * fc[c] + bary[1] * fc[c]) + bary[2] * fc[c];
}
ORC_API void orc_shade_sky(double dir_y, double out[3]) {
    double a = 0.5 * (dir_y + 1.0);
    static const double blue[3] = {0.5, 0.7, 1.0};
    for (int c = 0; c < 3; ++c) out[c] = (1.0 - a) * 1.0 + a * blue[c];
}
ORC_API int orc_quantise(const double pc[3], int spp, double *intensity) {
    double gray = 0.2126 * pc[0] + 0.7152 * pc[1] + 0.0722 * pc[2];
    double inten = gray / spp;
    if (intensity) *intensity = inten;
    return (int)(uint8_t)(int)(inten * 255.99);
}

/* ------------------------------------------------------------------------ */
/* Treeless checker (round 3).  Built ONLY from the probe-pinned primitives: */
/* the ray of a sample (orc_pixel_sample + orc_normalized, pinned against    */
/* the reference's Eigen expressions) and one row of intersect_bvh_triangles */
/* (moller_trumbore, pinned likewise).  No tree, no builder, no walk: for    */
/* every ray the minimum t over ALL triangles in the order given, the first  */
/* index that attains it (Eigen minCoeff's strict-< scan,                    */
/* rtrayintersection.cpp:173) and how many attain it.  A hit id that equals  */
/* this argmin does not depend on either BVH builder or on either traversal. */
/* ------------------------------------------------------------------------ */
#include <pthread.h>

/* t of moller_trumbore() with the invalid cases returned early.  Same operations in the same order on the
 * values that reach the result; pinned against moller_trumbore() itself by tests/test_oracle_pins.py. */
static inline double mt_t_edges(const double o[3], const double d[3], const double *v0, const double *e0,
                                const double *e2) {
    double p[3], tv[3], q[3];
    p[0] = d[1] * e2[2] - d[2] * e2[1];
    p[1] = d[2] * e2[0] - d[0] * e2[2];
    p[2] = d[0] * e2[1] - d[1] * e2[0];
    double det = (e0[0] * p[0] + e0[1] * p[1]) + e0[2] * p[2];
    if (!(det > 1e-6)) return INFINITY;
    double inv = 1.0 / det;
    for (int j = 0; j < 3; ++j) tv[j] = o[j] - v0[j];
    double u = ((tv[0] * p[0] + tv[1] * p[1]) + tv[2] * p[2]) * inv;
    if (!((u >= 0) && (u <= 1))) return INFINITY;
    q[0] = tv[1] * e0[2] - tv[2] * e0[1];
    q[1] = tv[2] * e0[0] - tv[0] * e0[2];
    q[2] = tv[0] * e0[1] - tv[1] * e0[0];
    double v = ((d[0] * q[0] + d[1] * q[1]) + d[2] * q[2]) * inv;
    if (!((v >= 0) && ((u + v) <= 1))) return INFINITY;
    double t = ((e2[0] * q[0] + e2[1] * q[1]) + e2[2] * q[2]) * inv;
    if (!((t >= 0.0) && (t <= INFINITY))) return INFINITY;
    return t;
}
/* the two edge vectors depend on the triangle alone (rtrayintersection.cpp:64-72): the brute-force loop
 * computes them once per triangle -- the same two subtractions, so the same bits -- instead of once per ray */
static inline void mt_edges(const double *tri, double *rec /* v0, e0, e2 */) {
    for (int j = 0; j < 3; ++j) {
        rec[j] = tri[j];
        rec[3 + j] = tri[3 + j] - tri[j];
        rec[6 + j] = tri[6 + j] - tri[j];
    }
}
static inline double mt_t_only(const double o[3], const double d[3], const double *tri) {
    double rec[9];
    mt_edges(tri, rec);
    return mt_t_edges(o, d, rec, rec + 3, rec + 6);
}
ORC_API double orc_mt_t_only(const double o[3], const double d[3], const double tri[9]) {
    return mt_t_only(o, d, tri);
}

typedef struct {
    int64_t r0, r1, n_tris;
    const double *origins;
    int64_t origin_stride; /* 0: one origin for all rays, 3: one per ray */
    const double *dirs, *tris;
    double *out_t;
    int32_t *out_first, *out_nmin;
} brute_job;

static void *brute_worker(void *arg) {
    brute_job *jb = (brute_job *)arg;
    for (int64_t r = jb->r0; r < jb->r1; ++r) {
        const double *o = jb->origins + r * jb->origin_stride;
        const double *d = jb->dirs + 3 * r;
        double best = INFINITY;
        int32_t first = -1, nmin = 0;
        for (int64_t k = 0; k < jb->n_tris; ++k) {
            const double *rec = jb->tris + 9 * k; /* v0, e0, e2 */
            double t = mt_t_edges(o, d, rec, rec + 3, rec + 6);
            if (t < best) {
                best = t;
                first = (int32_t)k;
                nmin = 1;
            } else if (t == best && t != INFINITY) {
                ++nmin;
            }
        }
        jb->out_t[r] = best;
        jb->out_first[r] = first;
        if (jb->out_nmin) jb->out_nmin[r] = nmin;
    }
    return NULL;
}

/* origins: 3 doubles (origin_stride 0) or n_rays*3 (origin_stride 3); tris: n_tris*9 in the caller's order.
 * out_first = -1 and out_t = +inf for a ray no triangle accepts. */
ORC_API void orc_brute_force(int64_t n_rays, const double *origins, int origin_stride, const double *dirs,
                             int64_t n_tris, const double *tris, double *out_t, int32_t *out_first,
                             int32_t *out_nmin, int n_threads) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 64) n_threads = 64;
    pthread_t th[64];
    brute_job jobs[64];
    double *recs = (double *)malloc((size_t)(n_tris ? n_tris : 1) * 9 * sizeof(double));
    for (int64_t k = 0; k < n_tris; ++k) mt_edges(tris + 9 * k, recs + 9 * k);
    for (int i = 0; i < n_threads; ++i) {
        jobs[i] = (brute_job){n_rays * i / n_threads, n_rays * (i + 1) / n_threads, n_tris, origins,
                              origin_stride, dirs, recs, out_t, out_first, out_nmin};
        if (i + 1 < n_threads) pthread_create(&th[i], NULL, brute_worker, &jobs[i]);
    }
    brute_worker(&jobs[n_threads - 1]);
    for (int i = 0; i + 1 < n_threads; ++i) pthread_join(th[i], NULL);
    free(recs);
}

/* The rays render_ppm_image shoots (rtrender.cpp:53-58) for rows [row0,row1), sample order (j,i,k), from the
 * offsets (ox,oy) = random_double()-0.5 given in `jitter` (relative to row0): the two pinned primitives, nothing
 * else. */
ORC_API void orc_ray_dirs(const double cam[3], const double p00[3], const double sp[6], int W, int spp, int row0,
                          int row1, const double *jitter, double *out_dirs) {
    size_t smp = 0;
    for (int j = row0; j < row1; ++j)
        for (int i = 0; i < W; ++i)
            for (int k = 0; k < spp; ++k, ++smp) {
                double ps[3], dd[3];
                orc_pixel_sample(p00, sp, i, j, jitter[2 * smp], jitter[2 * smp + 1], ps);
                for (int c = 0; c < 3; ++c) dd[c] = ps[c] - cam[c];
                orc_normalized(dd, out_dirs + 3 * smp);
            }
}
