// rt_bvh.cpp -- host BVH build (binned SAH, reference-identical decisions) and flattening.
//
// Spec followed, decision by decision (reference cpp/rtbvh.cpp, cpp/rtbvh.h):
//   triangle centroid ((a+b)+c)/3.0 per axis                      rtbvh.h:153-160
//   triangle / mesh AABBs with std::min/std::max semantics         rtbvh.h:43-68, rtbvh.cpp:36-79
//   mesh centroid (min+max)/2.0                                    rtbvh.h:162-167
//   DFS build stack, left first, children allocated as a pair      rtbvh.cpp:301-390 / 392-476
//   leaf if count <= 4 (BLAS) / <= 2 (TLAS) or the split fails     rtbvh.cpp:308,340,398,424
//   split: longest centroid axis (first max, strict >), 8 bins,
//          cost 1 + 2*(Al/Ap*nl + Ar/Ap*nr), first strictly lower  rtbvh.cpp:132-247
//   midpoint fallback when extent == 0 or no valid bin boundary    rtbvh.cpp:122-130,169-173,236-240
//   two-pointer unstable partition, centroid < pos goes left       rtbvh.cpp:249-299
//   leaves own their triangles in sorted-index order               rtbvh.cpp:478-542
//   tlas.blases filled in TLAS node-index order (quirk 1)          rtbvh.cpp:544-561
// Host arithmetic is IEEE binary64 without FMA contraction (built with -ffp-contract=off;
// the x86-64 baseline has no FMA instructions).  Index arithmetic is 64-bit (the reference's
// `int` strides overflow past 2^31 doubles).
#include "rt_bvh.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>

namespace rt {
namespace {

constexpr double kInf = std::numeric_limits<double>::infinity();
constexpr int kBins = 8;

struct Box {
    double mn[3], mx[3];
    void reset() {
        mn[0] = mn[1] = mn[2] = kInf;
        mx[0] = mx[1] = mx[2] = -kInf;
    }
    // std::min(a,b) = (b<a)?b:a ; std::max(a,b) = (a<b)?b:a  (NaN / signed-zero behaviour kept)
    void add_point(const double* p) {
        for (int i = 0; i < 3; ++i) {
            mn[i] = (p[i] < mn[i]) ? p[i] : mn[i];
            mx[i] = (mx[i] < p[i]) ? p[i] : mx[i];
        }
    }
    void add_box(const Box& o) {
        for (int i = 0; i < 3; ++i) {
            mn[i] = (o.mn[i] < mn[i]) ? o.mn[i] : mn[i];
            mx[i] = (mx[i] < o.mx[i]) ? o.mx[i] : mx[i];
        }
    }
    double extent(int axis) const {
        double r = mx[axis] - mn[axis];
        if (r < 0) return 0.0;
        return r;
    }
    double area() const {
        double h = extent(2), w = extent(1), d = extent(0);
        return 2 * (h * w + w * d + h * d);
    }
};

struct Centroid {
    double c[3];
};

// Elements (triangles of a mesh, or meshes of the scene) seen by the splitter.
struct Elements {
    const Centroid* cent;
    const Box* box;
};

Box range_box(const Elements& el, const int32_t* idx, int64_t first, int64_t count) {
    Box b;
    b.reset();
    for (int64_t i = first; i < first + count; ++i) b.add_box(el.box[idx[i]]);
    return b;
}

// Decides axis and position; returns false only for count <= 2 (rtbvh.cpp:140).
bool choose_split(const Elements& el, const int32_t* idx, int64_t first, int64_t count, const Box& parent,
                  int& axis_out, double& pos_out, int64_t& fallbacks) {
    if (count <= 2) return false;
    Box cb;
    cb.reset();
    for (int64_t i = first; i < first + count; ++i) cb.add_point(el.cent[idx[i]].c);

    int axis = 0;
    double ext = cb.extent(0);
    for (int a = 1; a < 3; ++a) {
        double e = cb.extent(a);
        if (e > ext) {
            axis = a;
            ext = e;
        }
    }
    axis_out = axis;
    if (ext == 0) {
        ++fallbacks;
        pos_out = cb.mn[axis] + ext * 0.5;
        return true;
    }

    Box bin_box[kBins];
    uint32_t bin_cnt[kBins];
    for (int b = 0; b < kBins; ++b) {
        bin_box[b].reset();
        bin_cnt[b] = 0;
    }
    const double inv_ext = 1.0 / ext;
    const double cmin = cb.mn[axis];
    for (int64_t i = first; i < first + count; ++i) {
        const int32_t e = idx[i];
        double t = (el.cent[e].c[axis] - cmin) * inv_ext;
        int b = static_cast<int>(t * kBins);
        if (b == kBins) b = kBins - 1;
        bin_cnt[b]++;
        bin_box[b].add_box(el.box[e]);
    }

    uint32_t lcnt[kBins], rcnt[kBins];
    Box lbox[kBins], rbox[kBins];
    Box run;
    run.reset();
    uint32_t n = 0;
    for (int b = 0; b < kBins; ++b) {
        if (bin_cnt[b] > 0) run.add_box(bin_box[b]);
        n += bin_cnt[b];
        lbox[b] = run;
        lcnt[b] = n;
    }
    run.reset();
    n = 0;
    for (int b = kBins - 1; b >= 0; --b) {
        if (bin_cnt[b] > 0) run.add_box(bin_box[b]);
        n += bin_cnt[b];
        rbox[b] = run;
        rcnt[b] = n;
    }

    double best = kInf;
    int best_bin = -1;
    const double ap = parent.area();
    for (int b = 0; b < kBins - 1; ++b) {
        const uint32_t nl = lcnt[b], nr = rcnt[b + 1];
        if (nl == 0 || nr == 0) continue;
        const double al = lbox[b].area(), ar = rbox[b + 1].area();
        const double cost = 1.0 + 2.0 * (al / ap * nl + ar / ap * nr);
        if (cost < best) {
            best = cost;
            best_bin = b;
        }
    }
    if (best_bin == -1) {
        ++fallbacks;
        pos_out = cmin + ext * 0.5;
        return true;
    }
    const double width = ext / kBins;
    pos_out = cmin + width * (best_bin + 1);
    return true;
}

// Returns the left count, or -1 when the node must become a leaf.  The permutation of idx
// happens even when the split is then rejected, exactly as in the reference.
int64_t partition(const Elements& el, int32_t* idx, int64_t first, int64_t count, const Box& parent,
                  int64_t& fallbacks) {
    int axis = 0;
    double pos = 0.0;
    if (!choose_split(el, idx, first, count, parent, axis, pos, fallbacks)) return -1;
    int64_t mid = first, end = first + count;
    while (mid < end) {
        if (el.cent[idx[mid]].c[axis] < pos) {
            ++mid;
        } else {
            --end;
            std::swap(idx[mid], idx[end]);
        }
    }
    const int64_t nl = mid - first;
    if (nl == 0 || nl == count) return -1;
    return nl;
}

struct Task {
    int64_t count;
    int32_t node;
    int64_t first;
};

void set_box(HostNode& n, const Box& b) {
    std::memcpy(n.mn, b.mn, sizeof(n.mn));
    std::memcpy(n.mx, b.mx, sizeof(n.mx));
}
Box get_box(const HostNode& n) {
    Box b;
    std::memcpy(b.mn, n.mn, sizeof(n.mn));
    std::memcpy(b.mx, n.mx, sizeof(n.mx));
    return b;
}

// Shared stack builder (BLAS: max_leaf 4, TLAS: max_leaf 2).
void build_tree(std::vector<HostNode>& nodes, const Elements& el, int32_t* idx, int64_t n_elem, int max_leaf,
                int64_t& fallbacks) {
    nodes.clear();
    nodes.reserve(static_cast<size_t>(n_elem) * 2 / std::max(1, max_leaf - 1) + 8);
    HostNode root{};
    set_box(root, range_box(el, idx, 0, n_elem));
    root.count = n_elem;
    root.left = -1;
    root.first = 0;
    nodes.push_back(root);

    std::vector<Task> stack;
    stack.push_back({n_elem, 0, 0});
    while (!stack.empty()) {
        const Task t = stack.back();
        stack.pop_back();
        if (t.count <= max_leaf) continue;  // node already carries count / box / first / left = -1
        const Box parent = get_box(nodes[t.node]);
        const int64_t nl = partition(el, idx, t.first, t.count, parent, fallbacks);
        if (nl < 0) continue;  // stays a leaf with t.count elements (possibly > max_leaf)
        if (nodes.size() + 2 > static_cast<size_t>(std::numeric_limits<int32_t>::max()))
            throw std::runtime_error("BVH has more than 2^31 nodes");
        const int32_t left = static_cast<int32_t>(nodes.size());
        const int64_t nr = t.count - nl;
        HostNode l{}, r{};
        set_box(l, range_box(el, idx, t.first, nl));
        l.count = nl;
        l.left = -1;
        l.first = static_cast<int32_t>(t.first);
        set_box(r, range_box(el, idx, t.first + nl, nr));
        r.count = nr;
        r.left = -1;
        r.first = static_cast<int32_t>(t.first + nl);
        nodes.push_back(l);
        nodes.push_back(r);
        nodes[t.node].left = left;
        nodes[t.node].count = 0;
        stack.push_back({nr, left + 1, t.first + nl});
        stack.push_back({nl, left, t.first});
    }
}

void build_one_blas(HostBlas& b, int32_t mesh, const double* coords_t, int64_t E, Centroid& mesh_cent) {
    std::vector<Centroid> cent(static_cast<size_t>(E));
    std::vector<Box> boxes(static_cast<size_t>(E));
    Box mesh_box;
    mesh_box.reset();
    for (int64_t e = 0; e < E; ++e) {
        const double* tc = coords_t + e * 9;
        for (int i = 0; i < 3; ++i) cent[e].c[i] = (tc[i] + tc[i + 3] + tc[i + 6]) / 3.0;
        boxes[e].reset();
        for (int v = 0; v < 3; ++v) boxes[e].add_point(tc + 3 * v);
        mesh_box.add_box(boxes[e]);
    }
    for (int i = 0; i < 3; ++i) mesh_cent.c[i] = (mesh_box.mn[i] + mesh_box.mx[i]) / 2.0;
    std::memcpy(b.mn, mesh_box.mn, sizeof(b.mn));
    std::memcpy(b.mx, mesh_box.mx, sizeof(b.mx));
    b.mesh = mesh;
    b.sorted_to_elem.resize(static_cast<size_t>(E));
    for (int64_t e = 0; e < E; ++e) b.sorted_to_elem[e] = static_cast<int32_t>(e);
    Elements el{cent.data(), boxes.data()};
    build_tree(b.nodes, el, b.sorted_to_elem.data(), E, 4, b.midpoint_fallbacks);
}

uint32_t leaf_desc(int64_t first, int64_t count, std::vector<uint32_t>& ext) {
    if (count >= 1 && count <= 7)
        return DESC_LEAF | (static_cast<uint32_t>(count - 1) << 28) | static_cast<uint32_t>(first);
    const uint32_t slot = static_cast<uint32_t>(ext.size() / 2);
    ext.push_back(static_cast<uint32_t>(first));
    ext.push_back(static_cast<uint32_t>(count));
    return DESC_LEAF | (7u << 28) | slot;
}

}  // namespace

Scene* build_scene(int n_meshes, const double* const* coords, const double* const* colors,
                   const int64_t* elem_counts, int timestep, int n_threads) {
    const auto t_start = std::chrono::steady_clock::now();
    if (n_meshes < 1) throw std::runtime_error("scene needs at least one mesh");
    if (timestep < 0) throw std::runtime_error("negative timestep");
    int64_t total_tris = 0;
    for (int m = 0; m < n_meshes; ++m) {
        if (!coords[m] || !colors[m]) throw std::runtime_error("null mesh array");
        // The reference asserts (Eigen minCoeff on an empty array) when a ray reaches an empty mesh.
        if (elem_counts[m] < 1) throw std::runtime_error("mesh " + std::to_string(m) + " has no elements");
        if (elem_counts[m] > std::numeric_limits<int32_t>::max() / 2)
            throw std::runtime_error("mesh " + std::to_string(m) + " has too many elements");
        total_tris += elem_counts[m];
    }
    if (total_tris > MAX_TRIS) throw std::runtime_error("scene exceeds 2^28-1 triangles");

    std::unique_ptr<Scene> sc(new Scene());
    std::vector<HostBlas> blases(static_cast<size_t>(n_meshes));
    std::vector<Centroid> mesh_cent(static_cast<size_t>(n_meshes));

    // ---- BLAS per mesh (independent; built concurrently, results do not depend on threads) ----
    auto work = [&](int m) {
        const int64_t E = elem_counts[m];
        build_one_blas(blases[m], m, coords[m] + static_cast<int64_t>(timestep) * E * 9, E, mesh_cent[m]);
    };
    int nt = n_threads <= 0 ? std::min(n_meshes, 16) : std::min(n_threads, n_meshes);
    if (nt <= 1) {
        for (int m = 0; m < n_meshes; ++m) work(m);
    } else {
        std::vector<std::thread> pool;
        std::vector<std::string> errors(static_cast<size_t>(nt));
        for (int w = 0; w < nt; ++w)
            pool.emplace_back([&, w]() {
                try {
                    for (int m = w; m < n_meshes; m += nt) work(m);
                } catch (const std::exception& e) {
                    errors[w] = e.what();
                }
            });
        for (auto& th : pool) th.join();
        for (auto& e : errors)
            if (!e.empty()) throw std::runtime_error(e);
    }

    // ---- TLAS over meshes ----
    std::vector<Box> mesh_boxes(static_cast<size_t>(n_meshes));
    for (int m = 0; m < n_meshes; ++m) {
        std::memcpy(mesh_boxes[m].mn, blases[m].mn, sizeof(blases[m].mn));
        std::memcpy(mesh_boxes[m].mx, blases[m].mx, sizeof(blases[m].mx));
        sc->midpoint_fallbacks += blases[m].midpoint_fallbacks;
    }
    std::vector<int32_t> blas_idx(static_cast<size_t>(n_meshes));
    for (int m = 0; m < n_meshes; ++m) blas_idx[m] = m;
    Elements mel{mesh_cent.data(), mesh_boxes.data()};
    build_tree(sc->tlas, mel, blas_idx.data(), n_meshes, 2, sc->midpoint_fallbacks);

    // tlas.blases: appended leaf by leaf in TLAS *node-index* order (quirk 1: intersect_TLAS
    // then indexes this vector by sorted-range position; reproduced, not fixed).
    sc->slots.reserve(static_cast<size_t>(n_meshes));
    for (size_t i = 0; i < sc->tlas.size(); ++i) {
        const HostNode& n = sc->tlas[i];
        for (int64_t j = n.first; j < n.first + n.count; ++j) sc->slots.push_back(std::move(blases[blas_idx[j]]));
    }
    if (static_cast<int>(sc->slots.size()) != n_meshes) throw std::runtime_error("internal: slot count mismatch");

    // ---- flatten ----
    const int n_slots = n_meshes;
    sc->slot_tri_base.resize(n_slots);
    sc->slot_root.resize(n_slots);
    int64_t tri_base = 0, node_base = 0;
    std::vector<int64_t> slot_node_base(n_slots);
    for (int s = 0; s < n_slots; ++s) {
        sc->slot_tri_base[s] = tri_base;
        tri_base += static_cast<int64_t>(sc->slots[s].sorted_to_elem.size());
        slot_node_base[s] = node_base;  // even; node i lives at base + 1 + i => child pairs are even-aligned
        int64_t span = static_cast<int64_t>(sc->slots[s].nodes.size()) + 1;
        node_base += (span + 1) & ~int64_t(1);
        sc->n_blas_nodes += static_cast<int64_t>(sc->slots[s].nodes.size());
    }
    if (node_base > MAX_DEV_NODES) throw std::runtime_error("scene exceeds 2^29 BVH nodes");
    sc->dev_nodes.assign(static_cast<size_t>(node_base), DevNode{});
    sc->tris.resize(static_cast<size_t>(total_tris) * 9);
    sc->colors.resize(static_cast<size_t>(total_tris) * 3);
    sc->prim_mesh.resize(static_cast<size_t>(total_tris));
    sc->prim_elem.resize(static_cast<size_t>(total_tris));

    std::vector<int32_t> slot_need(n_slots, 0);
    for (int s = 0; s < n_slots; ++s) {
        const HostBlas& b = sc->slots[s];
        const int64_t E = static_cast<int64_t>(b.sorted_to_elem.size());
        const double* ct = coords[b.mesh] + static_cast<int64_t>(timestep) * E * 9;
        const double* kt = colors[b.mesh] + static_cast<int64_t>(timestep) * E * 3;
        const int64_t tb = sc->slot_tri_base[s];
        for (int64_t p = 0; p < E; ++p) {
            const int64_t e = b.sorted_to_elem[p];
            std::memcpy(&sc->tris[(tb + p) * 9], ct + e * 9, 9 * sizeof(double));
            std::memcpy(&sc->colors[(tb + p) * 3], kt + e * 3, 3 * sizeof(double));
            sc->prim_mesh[tb + p] = b.mesh;
            sc->prim_elem[tb + p] = static_cast<int32_t>(e);
        }
        const int64_t nb = slot_node_base[s] + 1;
        sc->slot_root[s] = static_cast<int32_t>(nb);
        const int64_t nn = static_cast<int64_t>(b.nodes.size());
        // children have larger indices than their parent: one reverse sweep gives the stack need
        std::vector<int32_t> need(static_cast<size_t>(nn), 0);
        for (int64_t i = nn - 1; i >= 0; --i) {
            const HostNode& h = b.nodes[i];
            DevNode& d = sc->dev_nodes[nb + i];
            std::memcpy(d.mn, h.mn, sizeof(d.mn));
            std::memcpy(d.mx, h.mx, sizeof(d.mx));
            if (h.left < 0) {
                d.left = -1;
                d.count = static_cast<int32_t>(h.count);
                d.first = static_cast<int32_t>(tb + h.first);
                d.desc = leaf_desc(tb + h.first, h.count, sc->leaf_ext);
                sc->max_leaf_tris = std::max<int32_t>(sc->max_leaf_tris, static_cast<int32_t>(h.count));
                need[i] = 0;
            } else {
                d.left = static_cast<int32_t>(nb + h.left);
                d.count = 0;
                d.first = -1;
                d.desc = static_cast<uint32_t>(nb + h.left);
                need[i] = std::max(need[h.left] + 1, need[h.left + 1]);
            }
        }
        slot_need[s] = need[0];
    }

    // TLAS device nodes + stack need (entries above the base while a popped node is processed)
    const int64_t ntl = static_cast<int64_t>(sc->tlas.size());
    sc->dev_tlas.assign(static_cast<size_t>(ntl), DevNode{});
    std::vector<int32_t> tneed(static_cast<size_t>(ntl), 0);
    for (int64_t i = ntl - 1; i >= 0; --i) {
        const HostNode& h = sc->tlas[i];
        DevNode& d = sc->dev_tlas[i];
        std::memcpy(d.mn, h.mn, sizeof(d.mn));
        std::memcpy(d.mx, h.mx, sizeof(d.mx));
        d.left = h.left;
        d.count = static_cast<int32_t>(h.count);
        d.first = h.first;
        d.desc = DESC_TLAS | static_cast<uint32_t>(i);
        if (h.left < 0) {
            int32_t nmax = static_cast<int32_t>(h.count);
            for (int64_t k = 0; k < h.count; ++k)
                nmax = std::max<int32_t>(nmax, static_cast<int32_t>(h.count - 1 - k) + slot_need[h.first + k]);
            tneed[i] = nmax;
        } else {
            tneed[i] = std::max({2, tneed[h.left] + 1, tneed[h.left + 1]});
        }
    }
    sc->stack_need = std::max(1, tneed[0]) + 1;
    {
        double m = 0.0;
        for (double v : sc->tris) {
            const double a = std::fabs(v);
            if (!(a <= m)) m = (a == a) ? a : kInf;  // NaN -> inf
        }
        sc->max_abs_coord = m;
    }

    sc->build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    return sc.release();
}

}  // namespace rt
