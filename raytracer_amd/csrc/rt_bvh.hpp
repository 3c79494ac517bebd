// rt_bvh.hpp -- host acceleration-structure build + flattening for the GPU traversal kernels.
//
// The tree must be the reference's tree, node for node (cpp/rtbvh.cpp:36-561), so that GPU
// hits can be checked against it.  The layout is ours: one flat 64-byte node array, one flat
// triangle array in leaf order, one colour array, a slot table.  See DESIGN.md "Data layout".
#pragma once
#include <cstddef>
#include <cstdint>
#include <new>
#include <utility>
#include <vector>

namespace rt {

// Host threads of every host-side pool of this library (see rt_bvh.cpp; RT_HOST_THREADS, else affinity cores /
// LOCAL_WORLD_SIZE, 1..16; never OMP_NUM_THREADS).
int host_threads();

// The pool serves the OLDEST build first (every build_scene call draws a ticket; rt_bvh.cpp, Pool).  A caller that
// starts builds on threads of its own draws the tickets in the order it will wait for the builds, and each thread adopts
// its ticket for the length of the build.
int64_t draw_build_ticket();
struct AdoptBuildTicket {
    explicit AdoptBuildTicket(int64_t t);
    ~AdoptBuildTicket();
    AdoptBuildTicket(const AdoptBuildTicket&) = delete;
    AdoptBuildTicket& operator=(const AdoptBuildTicket&) = delete;
    int64_t outer;
};

// ---- host tree in the reference's numbering (kept for introspection / parity tests) ----
struct HostNode {
    double mn[3], mx[3];
    int64_t count;  // != 0 -> leaf (BLAS: triangles, TLAS: BLAS slots)
    int32_t left;   // -1 -> leaf; right = left + 1
    int32_t first;  // BLAS: node_minimum_element_index; TLAS: min_blas_idx
};

// Blocks of the big host arrays (element records, index arrays, tree nodes, the flattened scene) are RECYCLED: a job
// builds a scene of the same shape for every timestep, and a fresh block per build is first-touch page faults on
// every megabyte of it (the kernel zeroes each page: at 8 x 32,768 triangles a fifth of a build's CPU time).  Freed
// blocks of >= 64 KB go to per-size lists (sizes rounded up to eighths of a power of two) and are handed out again;
// the cache holds at most RT_HOST_POOL_BYTES (default 4 GiB, 0: off) and gives the rest back to the C++ heap.
void* flat_alloc(size_t bytes);
void flat_free(void* p, size_t bytes) noexcept;

// std::vector allocator that default-initialises (leaves trivially constructible elements untouched)
// on resize: the big flat arrays are written exactly once, by several threads, right after they are
// sized; zero-filling them first would be a serial pass over a gigabyte.
template <class T>
struct NoInitAlloc {
    using value_type = T;
    NoInitAlloc() = default;
    template <class U>
    NoInitAlloc(const NoInitAlloc<U>&) {}
    T* allocate(size_t n) { return static_cast<T*>(flat_alloc(n * sizeof(T))); }
    void deallocate(T* p, size_t n) { flat_free(p, n * sizeof(T)); }
    template <class U, class... A>
    void construct(U* p, A&&... a) {
        if constexpr (sizeof...(A) == 0)
            ::new (static_cast<void*>(p)) U;
        else
            ::new (static_cast<void*>(p)) U(std::forward<A>(a)...);
    }
    template <class U>
    bool operator==(const NoInitAlloc<U>&) const { return true; }
    template <class U>
    bool operator!=(const NoInitAlloc<U>&) const { return false; }
};
template <class T>
using FlatVec = std::vector<T, NoInitAlloc<T>>;

struct HostBlas {
    FlatVec<HostNode> nodes;
    FlatVec<int32_t> sorted_to_elem;  // mesh_element_indices after the build
    int32_t mesh = -1;
    double mn[3], mx[3];
    int64_t midpoint_fallbacks = 0;
};

// ---- device records ----
// 64-byte node.  Children of an internal node are adjacent (left, left+1) and the pair is
// 128-byte aligned: one contiguous 128-B fetch tests both.
struct alignas(64) DevNode {
    double mn[3];
    double mx[3];
    int32_t left;   // absolute index of the left child in the device node array, -1 for a leaf
    int32_t count;  // leaf: triangles (BLAS) / slots (TLAS); internal: 0
    int32_t first;  // leaf: absolute index of its first triangle (BLAS) / first slot (TLAS)
    uint32_t desc;  // packed traversal descriptor, see below
};
static_assert(sizeof(DevNode) == 64, "node record must be 64 bytes");

// Traversal stack / descriptor encoding (32 bit):
//   bit31 = 1 : BLAS leaf.  bits 30..28 = c, bits 27..0 = f.
//               c < 7  : triangles [f, f+c+1)
//               c == 7 : f indexes leaf_ext[] = {first, count} (leaves larger than 7 triangles)
//   bit31 = 0, bit30 = 1 : TLAS node, bits 29..0 = TLAS node index (slab-tested when popped)
//   bit31 = 0, bit30 = 0, bit29 = 1 : root of BLAS slot (bits 28..0), slab-tested when popped
//   otherwise : internal BLAS node, value = absolute index of its left child (< 2^29)
constexpr uint32_t DESC_LEAF = 0x80000000u;
constexpr uint32_t DESC_TLAS = 0x40000000u;
constexpr uint32_t DESC_ROOT = 0x20000000u;
constexpr uint32_t DESC_TRI_MASK = 0x0FFFFFFFu;
constexpr uint32_t DESC_NONE = 0xFFFFFFFFu;  // never a real descriptor (< 2^28 - 1 triangles / leaf_ext slots): "no node"
constexpr int64_t MAX_DEV_NODES = (1ll << 29) - 2;
constexpr int64_t MAX_TRIS = (1ll << 28) - 1;

struct Scene {
    // host tree
    FlatVec<HostNode> tlas;           // reference numbering
    std::vector<HostBlas> slots;      // tlas.blases: *stored* order (rtbvh.cpp:544-561, quirk 1)
    std::vector<int64_t> slot_tri_base;
    // device arrays
    FlatVec<DevNode> dev_nodes;       // BLAS nodes of all slots; slot s root at slot_root[s]
    std::vector<DevNode> dev_tlas;    // TLAS nodes (left = TLAS index, first = min_blas_idx)
    std::vector<int32_t> slot_root;   // device node index of each slot's root
    std::vector<uint32_t> leaf_ext;   // {first, count} pairs for leaves with > 7 triangles
    FlatVec<double> tris;             // n_tris * 9, leaf (sorted) order, slot after slot
    FlatVec<double> colors;           // n_tris * 3, same order
    FlatVec<int32_t> prim_mesh, prim_elem;      // prim id -> original ids
    int32_t stack_need = 1;
    double max_abs_coord = 0.0;  // max |vertex coordinate|; +inf if any coordinate is NaN/inf
    int32_t max_leaf_tris = 0;
    int64_t midpoint_fallbacks = 0;
    int64_t n_blas_nodes = 0;
    double build_seconds = 0.0;
};

// Throws std::runtime_error on invalid input / encoding limits.
// test seam (rt_partition_selftest): mismatches between the serial two-pointer partition and its parallel closed form
int64_t partition_selftest(int64_t cases, uint64_t seed, int max_n);
Scene* build_scene(int n_meshes, const double* const* coords, const double* const* colors,
                   const int64_t* elem_counts, int timestep, int n_threads, bool serial_trees = false);

}  // namespace rt
