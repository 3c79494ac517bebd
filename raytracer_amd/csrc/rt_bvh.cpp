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

#include <emmintrin.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>

namespace rt {

// ---- recycled blocks of the big host arrays (see rt_bvh.hpp) ----
namespace {
struct BlockPool {
    std::mutex mu;
    std::map<size_t, std::vector<void*>> free_lists;  // class size -> blocks
    size_t cached = 0;
    size_t cap = [] {
        const char* e = std::getenv("RT_HOST_POOL_BYTES");
        return e ? static_cast<size_t>(std::max(0.0, std::atof(e))) : (size_t(4) << 30);
    }();
    ~BlockPool() {
        for (auto& kv : free_lists)
            for (void* p : kv.second) ::operator delete(p);
    }
};
BlockPool& block_pool() {
    static BlockPool* p = new BlockPool();  // (leaked on purpose: vectors with static storage may outlive a static pool)
    return *p;
}
constexpr size_t kPoolMin = size_t(64) << 10;
size_t pool_class(size_t bytes) {  // round up to an eighth of a power of two
    size_t p2 = kPoolMin;
    while (p2 < bytes) p2 <<= 1;
    const size_t step = p2 >> 4;  // sixteenth of the upper power = eighth of the lower one
    return (bytes + step - 1) / step * step;
}
}  // namespace

void* flat_alloc(size_t bytes) {
    if (bytes < kPoolMin) return ::operator new(bytes);
    const size_t cls = pool_class(bytes);
    BlockPool& bp = block_pool();
    if (bp.cap) {
        std::lock_guard<std::mutex> lk(bp.mu);
        auto it = bp.free_lists.find(cls);
        if (it != bp.free_lists.end() && !it->second.empty()) {
            void* p = it->second.back();
            it->second.pop_back();
            bp.cached -= cls;
            return p;
        }
    }
    return ::operator new(cls);
}

void flat_free(void* p, size_t bytes) noexcept {
    if (!p) return;
    if (bytes < kPoolMin) {
        ::operator delete(p);
        return;
    }
    const size_t cls = pool_class(bytes);
    BlockPool& bp = block_pool();
    {
        std::lock_guard<std::mutex> lk(bp.mu);
        if (bp.cached + cls <= bp.cap) {
            try {
                bp.free_lists[cls].push_back(p);
                bp.cached += cls;
                return;
            } catch (...) {
            }
        }
    }
    ::operator delete(p);
}

namespace {

constexpr double kInf = std::numeric_limits<double>::infinity();
constexpr int kBins = 8;

struct Box {
    double mn[3], mx[3];
    void reset() {
        mn[0] = mn[1] = mn[2] = kInf;
        mx[0] = mx[1] = mx[2] = -kInf;
    }
    // std::min(a,b) = (b<a)?b:a ; std::max(a,b) = (a<b)?b:a  (NaN / signed-zero behaviour kept).  Two components at a
    // time with SSE2 (x86-64 baseline): MINPD(p, m) is (p < m) ? p : m and MAXPD(p, m) is (p > m) ? p : m lane by
    // lane -- the second operand whenever the comparison is false, NaNs and equal zeros included -- i.e. exactly the
    // two expressions above with p the incoming value and m the running extreme.
    // mn[2] and mx[0] are neighbours in memory: the middle pair takes its low half from the MINPD and its high half
    // from the MAXPD of the same two operands (three 16-byte updates per box instead of two and two scalar ones).
    void add_point(const double* p) {
        const __m128d p01 = _mm_loadu_pd(p), p12 = _mm_loadu_pd(p + 1);
        const __m128d p20 = _mm_shuffle_pd(p12, p01, 1);  // (p[2], p[0])
        const __m128d m20 = _mm_loadu_pd(mn + 2);
        _mm_storeu_pd(mn, _mm_min_pd(p01, _mm_loadu_pd(mn)));
        _mm_storeu_pd(mn + 2, _mm_move_sd(_mm_max_pd(p20, m20), _mm_min_pd(p20, m20)));
        _mm_storeu_pd(mx + 1, _mm_max_pd(p12, _mm_loadu_pd(mx + 1)));
    }
    void add_box(const Box& o) {
        const __m128d o20 = _mm_loadu_pd(o.mn + 2), m20 = _mm_loadu_pd(mn + 2);
        _mm_storeu_pd(mn, _mm_min_pd(_mm_loadu_pd(o.mn), _mm_loadu_pd(mn)));
        _mm_storeu_pd(mn + 2, _mm_move_sd(_mm_max_pd(o20, m20), _mm_min_pd(o20, m20)));
        _mm_storeu_pd(mx + 1, _mm_max_pd(_mm_loadu_pd(o.mx + 1), _mm_loadu_pd(mx + 1)));
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

static_assert(sizeof(Box) == 6 * sizeof(double), "Box: mn[3] and mx[3] back to back");

struct Centroid {
    double c[3];
};

// Elements (triangles of a mesh, or meshes of the scene) seen by the splitter.  `rec` (BLAS builds): the elements'
// boxes and centroids kept IN idx ORDER -- rec[i] belongs to element idx[i]; the partition pass permutes both --
// so that the scans of a range read memory front to back instead of chasing idx into arrays in element order (the
// values and the order they are combined in are the same: the tree does not change).  Null: indexed access.
struct Rec {
    Box box;
    Centroid c;
};
static_assert(sizeof(Rec) == 9 * sizeof(double), "Rec: box and centroid back to back");
struct Elements {
    const Centroid* cent;
    const Box* box;
    Rec* rec = nullptr;
};
inline const Box& elem_box(const Elements& el, const int32_t* idx, int64_t i) {
    return el.rec ? el.rec[i].box : el.box[idx[i]];
}
inline const double* elem_cent(const Elements& el, const int32_t* idx, int64_t i) {
    return el.rec ? el.rec[i].c.c : el.cent[idx[i]].c;
}

// Binning with the current bin held in registers: consecutive elements of a range mostly fall into the same bin (a
// mesh's elements are stored in some spatial order, and the partition keeps runs of it), and updating a bin in memory
// for each of them is a chain of store-to-load forwards -- ten cycles an element.  The same minimum / maximum
// operations in the same order, the bin written back when the run ends: the same bits.
template <bool CENT>
void bin_records_runs(const Rec* r, const Rec* const re, int axis, double cmin, double inv_ext, Box* box, Box* cent,
                      uint32_t* cnt) {
    int cur = -1;
    uint32_t n = 0;
    __m128d b01 = _mm_setzero_pd(), b20 = b01, b12 = b01, c01 = b01, c20 = b01, c12 = b01;
    for (; r != re; ++r) {
        const double t = (r->c.c[axis] - cmin) * inv_ext;
        int b = static_cast<int>(t * kBins);
        b = b == kBins ? kBins - 1 : b;
        if (b != cur) {
            if (cur >= 0) {
                _mm_storeu_pd(box[cur].mn, b01);
                _mm_storeu_pd(box[cur].mn + 2, b20);
                _mm_storeu_pd(box[cur].mx + 1, b12);
                if (CENT) {
                    _mm_storeu_pd(cent[cur].mn, c01);
                    _mm_storeu_pd(cent[cur].mn + 2, c20);
                    _mm_storeu_pd(cent[cur].mx + 1, c12);
                }
                cnt[cur] += n;
            }
            cur = b;
            n = 0;
            b01 = _mm_loadu_pd(box[b].mn);
            b20 = _mm_loadu_pd(box[b].mn + 2);
            b12 = _mm_loadu_pd(box[b].mx + 1);
            if (CENT) {
                c01 = _mm_loadu_pd(cent[b].mn);
                c20 = _mm_loadu_pd(cent[b].mn + 2);
                c12 = _mm_loadu_pd(cent[b].mx + 1);
            }
        }
        ++n;
        const __m128d e01 = _mm_loadu_pd(r->box.mn), e20 = _mm_loadu_pd(r->box.mn + 2), e12 = _mm_loadu_pd(r->box.mx + 1);
        b01 = _mm_min_pd(e01, b01);
        b20 = _mm_move_sd(_mm_max_pd(e20, b20), _mm_min_pd(e20, b20));
        b12 = _mm_max_pd(e12, b12);
        if (CENT) {
            const __m128d p01 = _mm_loadu_pd(r->c.c), p12 = _mm_loadu_pd(r->c.c + 1);
            const __m128d p20 = _mm_shuffle_pd(p12, p01, 1);  // (c[2], c[0])
            c01 = _mm_min_pd(p01, c01);
            c20 = _mm_move_sd(_mm_max_pd(p20, c20), _mm_min_pd(p20, c20));
            c12 = _mm_max_pd(p12, c12);
        }
    }
    if (cur >= 0) {
        _mm_storeu_pd(box[cur].mn, b01);
        _mm_storeu_pd(box[cur].mn + 2, b20);
        _mm_storeu_pd(box[cur].mx + 1, b12);
        if (CENT) {
            _mm_storeu_pd(cent[cur].mn, c01);
            _mm_storeu_pd(cent[cur].mn + 2, c20);
            _mm_storeu_pd(cent[cur].mx + 1, c12);
        }
        cnt[cur] += n;
    }
}
// Ranges of at least this many elements look at their first 32 bin indices: runs of four and more on average take
// bin_records_runs, anything else (a soup in random order: one mostly-taken branch and a write-back per element would
// cost 5 % more than the plain loop) the plain loop.  Either way the same bins.
const int64_t g_runs_min = std::getenv("RT_BIN_RUNS_MIN") ? std::max(32ll, std::atoll(std::getenv("RT_BIN_RUNS_MIN"))) : 128;
inline bool bins_come_in_runs(const Rec* r, int axis, double cmin, double inv_ext) {
    int changes = 0, prev = -1;
    for (int i = 0; i < 32; ++i) {
        const double t = (r[i].c.c[axis] - cmin) * inv_ext;
        int b = static_cast<int>(t * kBins);
        b = b == kBins ? kBins - 1 : b;
        changes += b != prev;
        prev = b;
    }
    return changes <= 8;
}

// ---------------------------------------------------------------------------------------
// Host parallelism.  A small persistent pool; parallel_for's caller takes part in its own job and
// only ever waits for chunks that some thread is already running, so nested and concurrent calls
// (one build per mesh, several timesteps in flight) cannot deadlock.  Every parallel pass below
// merges its per-chunk partial results IN CHUNK ORDER with the same `(p < m) ? p : m` rule the
// serial loop uses, so the outcome is bit-identical to the serial pass (first occurrence wins
// among equal values, which also fixes the sign of a zero) whatever the thread count.
// ---------------------------------------------------------------------------------------
}  // namespace
// One helper for every host-side pool (BVH build, host jump states).  RT_HOST_THREADS wins; otherwise the
// cores this process may run on (sched_getaffinity) divided among the ranks of this node (LOCAL_WORLD_SIZE,
// set by torchrun), 1..16.  OMP_NUM_THREADS is deliberately NOT consulted: torchrun exports OMP_NUM_THREADS=1
// to multi-rank jobs, and a rank that built its trees on one thread would be host-bound (hard part 3).
static int host_threads_uncached() {
    if (const char* e = std::getenv("RT_HOST_THREADS")) {
        const int n = std::atoi(e);
        if (n > 0) return std::min(n, 64);
    }
    int cores = 0;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) cores = CPU_COUNT(&set);
    if (cores <= 0) cores = static_cast<int>(std::thread::hardware_concurrency());
    int ranks = 1;
    if (const char* e = std::getenv("LOCAL_WORLD_SIZE")) ranks = std::max(1, std::atoi(e));
    return std::max(1, std::min(cores / ranks, 16));
}
int host_threads() {  // (read once: the pool is sized by the first call)
    static const int n = host_threads_uncached();
    return n;
}
namespace {

// One process per GPU (torchrun): LOCAL_WORLD_SIZE ranks share this node's cores, and each rank's build pool has
// cores / ranks threads (host_threads).  Left to the scheduler, the pools of eight ranks wander over the same cores;
// so rank LOCAL_RANK's workers are bound to the LOCAL_RANK-th of LOCAL_WORLD_SIZE equal spans of the CPUs this process
// may run on (the first `threads` CPUs of the span).  Only when the ranks evidently share one CPU set: more than one
// rank, LOCAL_RANK given, and room for every rank's threads.  RT_NO_PIN=1 leaves the threads unbound.  Empty: no pinning.
std::vector<int> rank_cpu_slice(int threads) {
    std::vector<int> out;
    const char* lw = std::getenv("LOCAL_WORLD_SIZE");
    const char* lr = std::getenv("LOCAL_RANK");
    if (!lw || !lr || std::getenv("RT_NO_PIN")) return out;
    const int ranks = std::atoi(lw), rank = std::atoi(lr);
    if (ranks < 2 || rank < 0 || rank >= ranks || threads < 1) return out;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) != 0) return out;
    std::vector<int> allowed;
    for (int c = 0; c < CPU_SETSIZE; ++c)
        if (CPU_ISSET(c, &set)) allowed.push_back(c);
    const int span = static_cast<int>(allowed.size()) / ranks;
    if (span < threads) return out;
    for (int i = 0; i < threads; ++i) out.push_back(allowed[static_cast<size_t>(rank * span + i)]);
    return out;
}

// Oldest build first: every build_scene call draws a ticket, the jobs it (and the tasks of its jobs) hand to the pool
// carry it, and the workers take the job with the lowest ticket.  A frame driver keeps several timesteps' builds in
// flight and waits for the OLDEST one: its last passes (TLAS, flatten) must not queue behind the tree builds of the
// timesteps after it.  -1: no build in progress on this thread (a job then gets a ticket of its own: plain FIFO).
thread_local int64_t tl_ticket = -1;
std::atomic<int64_t> g_next_ticket{0};
struct TicketScope {
    int64_t outer;
    TicketScope() : outer(tl_ticket) {
        if (tl_ticket < 0) tl_ticket = g_next_ticket.fetch_add(1);
    }
    ~TicketScope() { tl_ticket = outer; }
};

class Pool {
public:
    static Pool& get() {
        static Pool p(host_threads() - 1);
        return p;
    }
    int width() const { return static_cast<int>(workers_.size()) + 1; }
    void parallel_for(int64_t n, const std::function<void(int64_t)>& fn) {
        if (n <= 0) return;
        if (n == 1 || workers_.empty()) {
            for (int64_t i = 0; i < n; ++i) fn(i);
            return;
        }
        auto job = std::make_shared<Job>();
        job->n = n;
        job->fn = &fn;
        job->ticket = (tl_ticket >= 0 && !fifo_) ? tl_ticket : g_next_ticket.fetch_add(1);
        const int64_t helpers = std::min<int64_t>(n - 1, static_cast<int64_t>(workers_.size()));
        {
            std::lock_guard<std::mutex> lk(mu_);
            auto at = queue_.end();  // behind every entry of the same or an older build
            while (at != queue_.begin() && (*(at - 1))->ticket > job->ticket) --at;
            queue_.insert(at, static_cast<size_t>(helpers), job);
        }
        // (a worker that finds the queue empty goes back to sleep: waking more of them than there are entries would
        // only cost the wake-ups)
        if (helpers >= static_cast<int64_t>(workers_.size()))
            cv_.notify_all();
        else
            for (int64_t h = 0; h < helpers; ++h) cv_.notify_one();
        run(*job);
        std::unique_lock<std::mutex> lk(job->mu);
        job->cv.wait(lk, [&] { return job->done.load() == job->n; });
        if (!job->error.empty()) throw std::runtime_error(job->error);
    }

private:
    struct Job {
        int64_t n = 0;
        int64_t ticket = 0;
        const std::function<void(int64_t)>* fn = nullptr;  // valid until done == n (the caller waits)
        std::atomic<int64_t> next{0}, done{0};
        std::mutex mu;
        std::condition_variable cv;
        std::string error;
    };
    static void run(Job& j) {
        const int64_t outer = tl_ticket;
        tl_ticket = j.ticket;  // the jobs this job's tasks hand out belong to the same build
        for (;;) {
            const int64_t i = j.next.fetch_add(1);
            if (i >= j.n) break;
            try {
                (*j.fn)(i);
            } catch (const std::exception& e) {
                std::lock_guard<std::mutex> lk(j.mu);
                j.error = e.what();
            }
            if (j.done.fetch_add(1) + 1 == j.n) {
                std::lock_guard<std::mutex> lk(j.mu);
                j.cv.notify_all();
            }
        }
        tl_ticket = outer;
    }
    explicit Pool(int n) {
        const std::vector<int> cpus = rank_cpu_slice(n + 1);
        for (int i = 0; i < n; ++i)
            workers_.emplace_back([this, cpus] {
                if (!cpus.empty()) {  // several ranks on this node: this rank's workers keep to its own cores
                    cpu_set_t set;
                    CPU_ZERO(&set);
                    for (int c : cpus) CPU_SET(c, &set);
                    (void)sched_setaffinity(0, sizeof(set), &set);
                }
                for (;;) {
                    std::shared_ptr<Job> job;
                    {
                        std::unique_lock<std::mutex> lk(mu_);
                        cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
                        if (stop_ && queue_.empty()) return;
                        job = queue_.front();
                        queue_.pop_front();
                    }
                    if (job->next.load() < job->n) run(*job);
                }
            });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    const bool fifo_ = std::getenv("RT_POOL_FIFO") != nullptr;  // (comparison knob: jobs in arrival order)
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::shared_ptr<Job>> queue_;
    bool stop_ = false;
};

}  // namespace
int64_t draw_build_ticket() { return g_next_ticket.fetch_add(1); }
AdoptBuildTicket::AdoptBuildTicket(int64_t t) : outer(tl_ticket) { tl_ticket = t; }
AdoptBuildTicket::~AdoptBuildTicket() { tl_ticket = outer; }
namespace {

// RT_BUILD_TIMING=1: phase times of every large tree on stderr (tuning aid)
const bool g_build_timing = std::getenv("RT_BUILD_TIMING") != nullptr;
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// build_scene(..., serial_trees = true): every mesh's tree is built by ONE thread (the reference's own loop, the least
// CPU time per tree).  The whole-job driver asks for it when the meshes of the timesteps it keeps in flight already
// outnumber the cores: parallelism then comes from the meshes, and the 40 % of CPU time the task-split build spends on
// hand-overs goes into more timesteps per second instead (8 x 32,768 triangles: 72 -> 51 ms of CPU time per timestep).
thread_local bool tl_serial_trees = false;
// trees over at least this many elements build their subtrees concurrently (RT_TREE_PAR_MIN: tuning knob, read once)
const int64_t kTreeParMin = std::getenv("RT_TREE_PAR_MIN") ? std::atoll(std::getenv("RT_TREE_PAR_MIN")) : (1 << 13);
constexpr int64_t kParMin = 1 << 16;    // ranges at least this long are scanned by several threads
constexpr int64_t kParChunk = 1 << 14;  // elements per chunk of such a scan

// fn(chunk, lo, hi) over [first, first+count) in chunks; serial (one chunk) for short ranges
template <class F>
void for_chunks(int64_t first, int64_t count, int64_t& n_chunks, const F& fn) {
    if (count < kParMin) {
        n_chunks = 1;
        fn(0, first, first + count);
        return;
    }
    n_chunks = (count + kParChunk - 1) / kParChunk;
    const int64_t nc = n_chunks;
    Pool::get().parallel_for(nc, [&](int64_t c) {
        const int64_t lo = first + c * kParChunk;
        fn(c, lo, std::min(first + count, lo + kParChunk));
    });
}

Box range_box(const Elements& el, const int32_t* idx, int64_t first, int64_t count) {
    if (count < kParMin) {
        Box b;
        b.reset();
        for (int64_t i = first; i < first + count; ++i) b.add_box(elem_box(el, idx, i));
        return b;
    }
    const int64_t max_chunks = (count + kParChunk - 1) / kParChunk;
    std::vector<Box> part(static_cast<size_t>(max_chunks));
    int64_t nc = 0;
    for_chunks(first, count, nc, [&](int64_t c, int64_t lo, int64_t hi) {
        Box b;
        b.reset();
        for (int64_t i = lo; i < hi; ++i) b.add_box(elem_box(el, idx, i));
        part[c] = b;
    });
    Box b;
    b.reset();
    for (int64_t c = 0; c < nc; ++c) b.add_box(part[c]);
    return b;
}

// Boxes of the two sides of a partition, gathered during the partition pass itself instead of by
// separate scans.  A union is a set of exact min/max, so the order of accumulation can only show in the
// SIGN of a zero extreme (+0 and -0 compare equal; the serial scans keep the first one met): a box
// with any component equal to zero is therefore not trusted (`ok` false) and recomputed by the
// ordered scan, which keeps the tree bit-identical to the serial reference build.
struct Sides {
    Box box[2], cent[2];
    bool box_ok[2] = {false, false}, cent_ok[2] = {false, false};
};

bool no_zero_component(const Box& b) {
    for (int i = 0; i < 3; ++i)
        if (b.mn[i] == 0.0 || b.mx[i] == 0.0) return false;
    return true;
}

// Decides axis and position; returns false only for count <= 2 (rtbvh.cpp:140).
// `sides` (ranges below kParMin): the binning pass also keeps every bin's centroid bounds, and when the chosen
// boundary separates the elements exactly as the partition's own test `centroid < position` will -- every centroid
// of the bins up to it is below the position, none of the others is (checked on the bins' bounds; the two can differ
// only by a rounding of the bin index) -- the two sides' boxes and centroid bounds are the unions of the bins' and
// *sides_ready is set: the partition pass then only compares and swaps.  (Unions of exact min/max: the same values
// as the reference's scan over the partitioned range, except for the sign of a zero -- see Sides.)
bool choose_split(const Elements& el, const int32_t* idx, int64_t first, int64_t count, const Box& parent,
                  const Box* cb_known, int& axis_out, double& pos_out, int64_t& fallbacks, Sides* sides = nullptr,
                  bool* sides_ready = nullptr) {
    if (sides_ready) *sides_ready = false;
    if (count <= 2) return false;
    Box cb;
    cb.reset();
    if (cb_known) {
        cb = *cb_known;  // centroid bounds of this range, accumulated while the parent was partitioned
    } else if (count < kParMin) {
        for (int64_t i = first; i < first + count; ++i) cb.add_point(elem_cent(el, idx, i));
    } else {
        const int64_t max_chunks = (count + kParChunk - 1) / kParChunk;
        std::vector<Box> part(static_cast<size_t>(max_chunks));
        int64_t nc = 0;
        for_chunks(first, count, nc, [&](int64_t c, int64_t lo, int64_t hi) {
            Box b;
            b.reset();
            for (int64_t i = lo; i < hi; ++i) b.add_point(elem_cent(el, idx, i));
            part[c] = b;
        });
        for (int64_t c = 0; c < nc; ++c) cb.add_box(part[c]);
    }

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

    Box bin_box[kBins], bin_cent[kBins];
    uint32_t bin_cnt[kBins];
    for (int b = 0; b < kBins; ++b) {
        bin_box[b].reset();
        bin_cnt[b] = 0;
    }
    const double inv_ext = 1.0 / ext;
    const double cmin = cb.mn[axis];
    const bool with_cent = sides != nullptr && count < kParMin;
    if (with_cent) {
        for (int b = 0; b < kBins; ++b) bin_cent[b].reset();
        if (el.rec && count >= g_runs_min && bins_come_in_runs(el.rec + first, axis, cmin, inv_ext)) {
            bin_records_runs<true>(el.rec + first, el.rec + first + count, axis, cmin, inv_ext, bin_box, bin_cent, bin_cnt);
        } else if (const Rec* const rec = el.rec) {  // (the same loop on the records themselves: no per-element dispatch)
            for (const Rec *r = rec + first, *const re = r + count; r != re; ++r) {
                const double t = (r->c.c[axis] - cmin) * inv_ext;
                int b = static_cast<int>(t * kBins);
                b = b == kBins ? kBins - 1 : b;
                bin_cnt[b]++;
                bin_box[b].add_box(r->box);
                bin_cent[b].add_point(r->c.c);
            }
        } else
        for (int64_t i = first; i < first + count; ++i) {
            const double* cc = elem_cent(el, idx, i);
            double t = (cc[axis] - cmin) * inv_ext;
            int b = static_cast<int>(t * kBins);
            if (b == kBins) b = kBins - 1;
            bin_cnt[b]++;
            bin_box[b].add_box(elem_box(el, idx, i));
            bin_cent[b].add_point(cc);
        }
    } else if (count < kParMin) {  // straight into the bins, nothing on the heap
        for (int64_t i = first; i < first + count; ++i) {
            double t = (elem_cent(el, idx, i)[axis] - cmin) * inv_ext;
            int b = static_cast<int>(t * kBins);
            if (b == kBins) b = kBins - 1;
            bin_cnt[b]++;
            bin_box[b].add_box(elem_box(el, idx, i));
        }
    } else {
        struct Bins {
            Box box[kBins];
            uint32_t cnt[kBins];
        };
        const int64_t max_chunks = (count + kParChunk - 1) / kParChunk;
        std::vector<Bins> part(static_cast<size_t>(max_chunks));
        int64_t nc = 0;
        for_chunks(first, count, nc, [&](int64_t c, int64_t lo, int64_t hi) {
            Bins& p = part[c];
            for (int b = 0; b < kBins; ++b) {
                p.box[b].reset();
                p.cnt[b] = 0;
            }
            if (el.rec && hi - lo >= g_runs_min && bins_come_in_runs(el.rec + lo, axis, cmin, inv_ext)) {
                bin_records_runs<false>(el.rec + lo, el.rec + hi, axis, cmin, inv_ext, p.box, nullptr, p.cnt);
                return;
            }
            for (int64_t i = lo; i < hi; ++i) {
                double t = (elem_cent(el, idx, i)[axis] - cmin) * inv_ext;
                int b = static_cast<int>(t * kBins);
                if (b == kBins) b = kBins - 1;
                p.cnt[b]++;
                p.box[b].add_box(elem_box(el, idx, i));
            }
        });
        for (int64_t c = 0; c < nc; ++c)
            for (int b = 0; b < kBins; ++b) {
                if (part[c].cnt[b] == 0) continue;
                bin_cnt[b] += part[c].cnt[b];
                bin_box[b].add_box(part[c].box[b]);
            }
    }

    // suffix (right-side) areas and counts first, then one forward sweep with the running left box:
    // the same unions in the same order as prefix / suffix arrays of boxes, without the copies
    double rarea[kBins];
    uint32_t rcnt[kBins];
    Box run;
    run.reset();
    uint32_t n = 0;
    for (int b = kBins - 1; b >= 0; --b) {
        if (bin_cnt[b] > 0) run.add_box(bin_box[b]);
        n += bin_cnt[b];
        rarea[b] = run.area();
        rcnt[b] = n;
    }
    double best = kInf;
    int best_bin = -1;
    const double ap = parent.area();
    run.reset();
    n = 0;
    for (int b = 0; b < kBins - 1; ++b) {
        // (a boundary behind an empty bin separates the same two sets as the one before it: the same cost, which
        // is never STRICTLY lower -- not evaluated)
        if (bin_cnt[b] == 0) continue;
        run.add_box(bin_box[b]);
        n += bin_cnt[b];
        const uint32_t nl = n, nr = rcnt[b + 1];
        if (nl == 0 || nr == 0) continue;
        const double al = run.area(), ar = rarea[b + 1];
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
    if (with_cent) {
        Box lb, rb, lc, rc;
        lb.reset();
        rb.reset();
        lc.reset();
        rc.reset();
        for (int b = 0; b <= best_bin; ++b) {
            if (bin_cnt[b] == 0) continue;
            lb.add_box(bin_box[b]);
            lc.add_box(bin_cent[b]);
        }
        for (int b = best_bin + 1; b < kBins; ++b) {
            if (bin_cnt[b] == 0) continue;
            rb.add_box(bin_box[b]);
            rc.add_box(bin_cent[b]);
        }
        // both sides are non-empty here (the boundary was valid); the partition's test is `centroid < pos_out`
        if (lc.mx[axis] < pos_out && !(rc.mn[axis] < pos_out)) {
            sides->box[0] = lb;
            sides->box[1] = rb;
            sides->cent[0] = lc;
            sides->cent[1] = rc;
            sides->box_ok[0] = no_zero_component(lb);
            sides->box_ok[1] = no_zero_component(rb);
            sides->cent_ok[0] = no_zero_component(lc);
            sides->cent_ok[1] = no_zero_component(rc);
            *sides_ready = true;
        }
    }
    return true;
}

// ---- the reference's two-pointer partition (rtbvh.cpp:271-285) as a data-parallel pass: SAME permutation ----
// The loop `while (mid < end) { if (left(mid)) ++mid; else swap(idx[mid], idx[--end]); }` is sequential, but the
// permutation it produces has a closed form.  Positions (mid, end) always hold their original elements (swaps write
// only positions mid and end), so with nl = number of left-class elements, S = first + nl, E = first + count:
//   * a left-class element originally below S is examined in place and never moves;
//   * every right-class element below S ("hole" k = 1..A, ascending) is thrown to the top of the unexamined range and
//     the loop then pulls elements down from the top until a left-class one arrives: hole k receives the k-th
//     left-class element of [S, E) counted from the END (positions q_1 > q_2 > ... > q_A; A holes, A such elements);
//   * a right-class element pulled on the way lands one position below where it was (r -> r - 1), and the element
//     thrown out of hole k lands just above that run: at q_{k-1} - 1 (q_0 = E);
//   * what is left when mid reaches S is a run [S, q_A) of right-class elements (q_A = E when A = 0): the loop
//     rotates it -- the element at S goes to q_A - 1, the others move down by one.
// So every element's destination follows from two prefix sums (right-class below S, left-class from the end), the
// passes below are chunk-parallel, and the result equals the serial loop's element for element -- which matters:
// leaves keep their triangles in this order (rtbvh.cpp:515-536) and ties in t go to the first (rtrayintersection.cpp
// :173-177).  rt_partition_selftest compares the two on random class patterns; tests/test_bvh_build.py on whole trees.
// `sides` (with `el`): the boxes and centroid bounds of the two sides are gathered in the classification pass, chunk
// by chunk, and merged in chunk order (see Sides: trusted only without zero components).
template <class IsLeft, class Run>
int64_t partition_closed_form(int32_t* idx, Rec* rec, int64_t first, int64_t count, int64_t chunk, const IsLeft& is_left,
                              const Run& run, const Elements* el = nullptr, Sides* sides = nullptr) {
    const int64_t E = first + count, nc = (count + chunk - 1) / chunk;
    std::vector<uint8_t> flag(static_cast<size_t>(count));  // 1 = left-class, by ORIGINAL position
    std::vector<int64_t> cl(static_cast<size_t>(nc)), pa(static_cast<size_t>(nc) + 1, 0), sb(static_cast<size_t>(nc) + 1, 0);
    struct Four {
        Box b[4];  // left box, right box, left centroid bounds, right centroid bounds
    };
    std::vector<Four> part(sides ? static_cast<size_t>(nc) : 0);
    run(nc, [&](int64_t c) {
        const int64_t lo = first + c * chunk, hi = std::min(E, lo + chunk);
        int64_t n = 0;
        if (sides) {
            Four f;
            for (Box& b : f.b) b.reset();
            for (int64_t i = lo; i < hi; ++i) {
                const int l = is_left(i) ? 1 : 0;
                flag[i - first] = static_cast<uint8_t>(l);
                n += l;
                f.b[1 - l].add_box(elem_box(*el, idx, i));
                f.b[3 - l].add_point(elem_cent(*el, idx, i));
            }
            part[c] = f;
        } else {
            for (int64_t i = lo; i < hi; ++i) n += (flag[i - first] = is_left(i) ? 1 : 0);
        }
        cl[c] = n;
    });
    if (sides) {
        Four f;
        for (Box& b : f.b) b.reset();
        for (int64_t c = 0; c < nc; ++c)
            for (int k = 0; k < 4; ++k) f.b[k].add_box(part[c].b[k]);
        sides->box[0] = f.b[0];
        sides->box[1] = f.b[1];
        sides->cent[0] = f.b[2];
        sides->cent[1] = f.b[3];
        sides->box_ok[0] = no_zero_component(f.b[0]);
        sides->box_ok[1] = no_zero_component(f.b[1]);
        sides->cent_ok[0] = no_zero_component(f.b[2]);
        sides->cent_ok[1] = no_zero_component(f.b[3]);
    }
    int64_t nl = 0;
    for (int64_t c = 0; c < nc; ++c) nl += cl[c];
    const int64_t S = first + nl;
    // pa[c] = right-class elements below S in chunks before c; sb[c] = left-class elements at or above S in chunks after c
    for (int64_t c = 0; c < nc; ++c) {
        const int64_t lo = first + c * chunk, hi = std::min(E, lo + chunk);
        int64_t a;
        if (hi <= S) {
            a = (hi - lo) - cl[c];
        } else if (lo >= S) {
            a = 0;
        } else {
            a = 0;
            for (int64_t i = lo; i < S; ++i) a += 1 - flag[i - first];
        }
        pa[c + 1] = pa[c] + a;
    }
    for (int64_t c = nc - 1; c >= 0; --c) {
        const int64_t lo = first + c * chunk, hi = std::min(E, lo + chunk);
        int64_t b;
        if (lo >= S) {
            b = cl[c];
        } else if (hi <= S) {
            b = 0;
        } else {
            b = 0;
            for (int64_t i = S; i < hi; ++i) b += flag[i - first];
        }
        sb[c] = sb[c + 1] + b;  // sb[c] here = left-class at or above S in chunks c.. ; chunk c's own start at sb[c + 1]
    }
    const int64_t A = pa[nc];
    if (A != sb[0]) throw std::runtime_error("internal: partition hole / filler counts differ");
    if (E == S) return nl;  // everything is left-class: the loop moves nothing
    // In place.  Pass 1: the elements thrown out of the holes are saved (x), and q is filled.  Pass 2: every chunk
    // of [S, E) sweeps its positions ASCENDING and moves old[r] to its destination -- r - 1 for a right-class element
    // (already read by this chunk, or the last position of the chunk below: that one write is deferred to pass 3),
    // hole k for the left-class element at q_k, whose place-below q_k - 1 then takes x_{k+1}.  The element at S, if
    // right-class, and x_1 go to the tops of their runs, which may lie in other chunks: deferred too.
    std::vector<int64_t> hole(static_cast<size_t>(A)), q(static_cast<size_t>(A));
    std::vector<int32_t> x_idx(static_cast<size_t>(A));
    std::vector<Rec> x_rec(rec ? static_cast<size_t>(A) : 0);
    run(nc, [&](int64_t c) {
        const int64_t lo = first + c * chunk, hi = std::min(E, lo + chunk);
        int64_t j = pa[c];
        for (int64_t i = lo; i < std::min(hi, S); ++i)
            if (!flag[i - first]) {
                hole[j] = i;
                x_idx[j] = idx[i];
                if (rec) x_rec[j] = rec[i];
                ++j;
            }
        j = sb[c + 1];
        for (int64_t i = hi - 1; i >= std::max(lo, S); --i)
            if (flag[i - first]) q[j++] = i;
    });
    struct Late {  // one deferred write per chunk (+ the two run tops)
        int64_t dst = -1;
        int32_t idx = 0;
        Rec rec;
    };
    std::vector<Late> late(static_cast<size_t>(nc) + 2);
    auto put = [&](int64_t dst, int32_t v, const Rec* r) {
        idx[dst] = v;
        if (rec) rec[dst] = *r;
    };
    run(nc, [&](int64_t c) {
        const int64_t lo = std::max(first + c * chunk, S), hi = std::min(E, first + (c + 1) * chunk);
        if (lo >= hi) return;
        // rank (1-based, from the end) of the next left-class element at or above lo: those above this chunk + this chunk's
        int64_t k = sb[c + 1];
        for (int64_t i = lo; i < hi; ++i) k += flag[i - first];
        for (int64_t r = lo; r < hi; ++r) {
            const int32_t v = idx[r];
            const Rec* rr = rec ? &rec[r] : nullptr;
            if (flag[r - first]) {  // r == q_k: to hole k; x_{k+1} lands just below it (if there is a hole k + 1)
                const int64_t kk = k--;  // q_kk == r
                const int64_t h = hole[kk - 1];
                // the hole is below S: no chunk of this pass reads it (its old content was saved in pass 1)
                put(h, v, rr);
                if (kk < A) {
                    const int64_t dst = r - 1;  // q_kk - 1 takes x_{kk+1}
                    if (dst >= lo) {
                        put(dst, x_idx[kk], rec ? &x_rec[kk] : nullptr);
                    } else {
                        Late& L = late[c];
                        L.dst = dst;
                        L.idx = x_idx[kk];
                        if (rec) L.rec = x_rec[kk];
                    }
                }
            } else if (r == S) {  // the run [S, q_A) rotates: its first element goes to the top of the run
                Late& L = late[nc];
                L.dst = (A == 0 ? E : q[A - 1]) - 1;
                L.idx = v;
                if (rec) L.rec = *rr;
            } else if (r - 1 >= lo) {
                put(r - 1, v, rr);
            } else {
                Late& L = late[c];
                L.dst = r - 1;
                L.idx = v;
                if (rec) L.rec = *rr;
            }
        }
    });
    if (A > 0) {  // x_1 goes to the very top
        Late& L = late[nc + 1];
        L.dst = E - 1;
        L.idx = x_idx[0];
        if (rec) L.rec = x_rec[0];
    }
    for (const Late& L : late)
        if (L.dst >= 0) put(L.dst, L.idx, &L.rec);
    return nl;
}

// Returns the left count, or -1 when the node must become a leaf.  The permutation of idx
// happens even when the split is then rejected, exactly as in the reference.
int64_t partition(const Elements& el, int32_t* idx, int64_t first, int64_t count, const Box& parent,
                  const Box* cb_known, int64_t& fallbacks, Sides* sides) {
    int axis = 0;
    double pos = 0.0;
    static const bool fold = std::getenv("RT_NO_FOLD") == nullptr;
    bool sides_ready = false;
    if (!choose_split(el, idx, first, count, parent, cb_known, axis, pos, fallbacks, fold ? sides : nullptr, &sides_ready))
        return -1;
    int64_t mid = first, end = first + count;
    static const bool par_partition = std::getenv("RT_SERIAL_PARTITION") == nullptr;
    if (par_partition && count >= kParMin && Pool::get().width() > 1) {
        const int64_t nl = partition_closed_form(
            idx, el.rec, first, count, kParChunk, [&](int64_t i) { return elem_cent(el, idx, i)[axis] < pos; },
            [](int64_t n, const std::function<void(int64_t)>& fn) { Pool::get().parallel_for(n, fn); }, fold ? &el : nullptr,
            fold ? sides : nullptr);
        if (nl == 0 || nl == count) return -1;
        return nl;
    }
    Rec* const rec = el.rec;
    if (fold && sides && count < kParMin && !sides_ready) {
        Box lb, rb, lc, rc;
        lb.reset();
        rb.reset();
        lc.reset();
        rc.reset();
        while (mid < end) {
            const double* cc = elem_cent(el, idx, mid);
            const Box& bx = elem_box(el, idx, mid);
            if (cc[axis] < pos) {
                lb.add_box(bx);
                lc.add_point(cc);
                ++mid;
            } else {
                rb.add_box(bx);
                rc.add_point(cc);
                --end;
                std::swap(idx[mid], idx[end]);
                if (el.rec) std::swap(el.rec[mid], el.rec[end]);
            }
        }
        sides->box[0] = lb;
        sides->box[1] = rb;
        sides->cent[0] = lc;
        sides->cent[1] = rc;
        sides->box_ok[0] = no_zero_component(lb);
        sides->box_ok[1] = no_zero_component(rb);
        sides->cent_ok[0] = no_zero_component(lc);
        sides->cent_ok[1] = no_zero_component(rc);
    } else if (rec) {  // (the loop below on the records themselves)
        while (mid < end) {
            if (rec[mid].c.c[axis] < pos) {
                ++mid;
            } else {
                --end;
                std::swap(idx[mid], idx[end]);
                std::swap(rec[mid], rec[end]);
            }
        }
    } else {
        while (mid < end) {
            if (elem_cent(el, idx, mid)[axis] < pos) {
                ++mid;
            } else {
                --end;
                std::swap(idx[mid], idx[end]);
            }
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
    bool cent_ok = false;  // `cent` holds the centroid bounds of the range
    Box cent;
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

// Serial stack builder (BLAS: max_leaf 4, TLAS: max_leaf 2): the reference's loop.  nodes[0] must be
// the root (box, count, first set, left = -1); descendants are appended in the reference's order.
// `defer_below` > 0: a node with max_leaf < count <= defer_below is not expanded but recorded in
// `deferred` together with the index its first descendant would have received.
struct Deferred {
    int32_t node;        // provisional index of the subtree root
    int64_t first, count;
    int64_t insert_pos;  // nodes.size() when the root was popped
};

void expand(FlatVec<HostNode>& nodes, const Elements& el, int32_t* idx, int max_leaf, int64_t& fallbacks,
            int64_t defer_below, std::vector<Deferred>* deferred) {
    std::vector<Task> stack;
    {
        Task t0{};
        t0.count = nodes[0].count;
        t0.node = 0;
        t0.first = nodes[0].first;
        stack.push_back(t0);
    }
    while (!stack.empty()) {
        const Task t = stack.back();
        stack.pop_back();
        if (t.count <= max_leaf) continue;  // node already carries count / box / first / left = -1
        if (deferred && t.count <= defer_below) {
            deferred->push_back({t.node, t.first, t.count, static_cast<int64_t>(nodes.size())});
            continue;
        }
        const Box parent = get_box(nodes[t.node]);
        Sides sides;
        const int64_t nl = partition(el, idx, t.first, t.count, parent, t.cent_ok ? &t.cent : nullptr, fallbacks, &sides);
        if (nl < 0) continue;  // stays a leaf with t.count elements (possibly > max_leaf)
        if (nodes.size() + 2 > static_cast<size_t>(std::numeric_limits<int32_t>::max()))
            throw std::runtime_error("BVH has more than 2^31 nodes");
        const int32_t left = static_cast<int32_t>(nodes.size());
        const int64_t nr = t.count - nl;
        HostNode l{}, r{};
        set_box(l, sides.box_ok[0] ? sides.box[0] : range_box(el, idx, t.first, nl));
        l.count = nl;
        l.left = -1;
        l.first = static_cast<int32_t>(t.first);
        set_box(r, sides.box_ok[1] ? sides.box[1] : range_box(el, idx, t.first + nl, nr));
        r.count = nr;
        r.left = -1;
        r.first = static_cast<int32_t>(t.first + nl);
        nodes.push_back(l);
        nodes.push_back(r);
        nodes[t.node].left = left;
        nodes[t.node].count = 0;
        Task tr{}, tl{};
        tr.count = nr;
        tr.node = left + 1;
        tr.first = t.first + nl;
        tr.cent_ok = sides.cent_ok[1];
        tr.cent = sides.cent[1];
        tl.count = nl;
        tl.node = left;
        tl.first = t.first;
        tl.cent_ok = sides.cent_ok[0];
        tl.cent = sides.cent[0];
        stack.push_back(tr);
        stack.push_back(tl);
    }
}

// Tree over idx[0, n_elem).  Large inputs: the top of the tree is expanded by one thread (its long
// scans are themselves chunk-parallel), subtrees of <= n/64 elements are built concurrently on
// disjoint ranges of idx, each with LOCAL node numbers, and spliced in at the positions the serial
// loop would have given them: the loop numbers a node's descendants contiguously from the moment
// the node is popped, so a subtree's block starts at its insert_pos plus the sizes of the blocks
// spliced in before it.  The result is identical to the serial build, node for node.
void build_tree(FlatVec<HostNode>& nodes, const Elements& el, int32_t* idx, int64_t n_elem, int max_leaf,
                int64_t& fallbacks) {
    nodes.clear();
    HostNode root{};
    set_box(root, range_box(el, idx, 0, n_elem));
    root.count = n_elem;
    root.left = -1;
    root.first = 0;
    // (from 8 K elements on: the multi-frame driver rebuilds meshes of a few 10 K triangles per timestep, and a
    // build that only runs one thread per mesh leaves half of a 16-core host idle with 8 meshes)
    if (n_elem < kTreeParMin || Pool::get().width() == 1 || tl_serial_trees) {
        nodes.reserve(static_cast<size_t>(n_elem) * 2 / std::max(1, max_leaf - 1) + 8);
        nodes.push_back(root);
        expand(nodes, el, idx, max_leaf, fallbacks, 0, nullptr);
        return;
    }
    // The top of the tree: the splits of different nodes are independent of each other (disjoint ranges of idx),
    // only the node NUMBERS depend on the order the reference's loop pops them in.  So the ranges are partitioned
    // by recursive tasks (the two children of a node concurrently) into a pointer tree, and the loop is then
    // replayed over that tree, serially and without touching an element, to hand out the numbers.
    const double t_tree0 = now_s();
    std::vector<HostNode> top;
    top.push_back(root);
    std::vector<Deferred> deferred;
    const int64_t defer_below = std::max<int64_t>(n_elem >= 2 * kParMin ? 4096 : 256, n_elem / 64);
    {
        struct TopNode {
            HostNode h{};
            bool deferred = false;
            std::unique_ptr<TopNode> l, r;
        };
        std::atomic<int64_t> fb{0};
        std::function<void(TopNode*, bool, const Box&)> grow = [&](TopNode* t, bool cent_ok, const Box& cent) {
            const int64_t count = t->h.count, first = t->h.first;
            if (count <= max_leaf) return;
            if (count <= defer_below) {
                t->deferred = true;
                return;
            }
            const Box parent = get_box(t->h);
            Sides sides;
            int64_t f = 0;
            const int64_t nl = partition(el, idx, first, count, parent, cent_ok ? &cent : nullptr, f, &sides);
            fb += f;
            if (nl < 0) return;  // stays a leaf with `count` elements (possibly > max_leaf)
            const int64_t nr = count - nl;
            t->l.reset(new TopNode());
            t->r.reset(new TopNode());
            set_box(t->l->h, sides.box_ok[0] ? sides.box[0] : range_box(el, idx, first, nl));
            t->l->h.count = nl;
            t->l->h.left = -1;
            t->l->h.first = static_cast<int32_t>(first);
            set_box(t->r->h, sides.box_ok[1] ? sides.box[1] : range_box(el, idx, first + nl, nr));
            t->r->h.count = nr;
            t->r->h.left = -1;
            t->r->h.first = static_cast<int32_t>(first + nl);
            auto child = [&](int64_t k) {
                if (k == 0)
                    grow(t->l.get(), sides.cent_ok[0], sides.cent[0]);
                else
                    grow(t->r.get(), sides.cent_ok[1], sides.cent[1]);
            };
            if (std::min(nl, nr) > defer_below) {
                Pool::get().parallel_for(2, child);
            } else {
                child(0);
                child(1);
            }
        };
        TopNode troot;
        troot.h = root;
        grow(&troot, false, Box{});
        fallbacks += fb.load();
        // replay of the reference's loop (see expand): pop, number the children, push right then left
        std::vector<std::pair<const TopNode*, int32_t>> stack;
        stack.push_back({&troot, 0});
        while (!stack.empty()) {
            const TopNode* t = stack.back().first;
            const int32_t node = stack.back().second;
            stack.pop_back();
            if (t->h.count <= max_leaf) continue;
            if (t->deferred) {
                deferred.push_back({node, t->h.first, t->h.count, static_cast<int64_t>(top.size())});
                continue;
            }
            if (!t->l) continue;
            const int32_t left = static_cast<int32_t>(top.size());
            top.push_back(t->l->h);
            top.push_back(t->r->h);
            top[node].left = left;
            top[node].count = 0;
            stack.push_back({t->r.get(), left + 1});
            stack.push_back({t->l.get(), left});
        }
    }

    const double t_tree1 = now_s();
    const int64_t nd = static_cast<int64_t>(deferred.size());
    std::vector<FlatVec<HostNode>> sub(static_cast<size_t>(nd));
    std::vector<int64_t> sub_fallbacks(static_cast<size_t>(nd), 0);
    Pool::get().parallel_for(nd, [&](int64_t i) {
        const Deferred& d = deferred[i];
        FlatVec<HostNode>& v = sub[i];
        v.reserve(static_cast<size_t>(d.count) * 2 / std::max(1, max_leaf - 1) + 8);
        v.push_back(top[d.node]);  // local root (index 0); descendants get local indices 1, 2, ...
        expand(v, el, idx, max_leaf, sub_fallbacks[i], 0, nullptr);
    });

    // splice
    const double t_tree2 = now_s();
    std::vector<int64_t> block_start(static_cast<size_t>(nd));
    int64_t extra = 0;
    for (int64_t i = 0; i < nd; ++i) {
        block_start[i] = deferred[i].insert_pos + extra;
        extra += static_cast<int64_t>(sub[i].size()) - 1;
        fallbacks += sub_fallbacks[i];
    }
    const int64_t total = static_cast<int64_t>(top.size()) + extra;
    if (total > static_cast<int64_t>(std::numeric_limits<int32_t>::max()))
        throw std::runtime_error("BVH has more than 2^31 nodes");
    nodes.resize(static_cast<size_t>(total));
    // provisional index p -> p + sizes of the blocks inserted at or before p
    std::vector<int64_t> shift_at(top.size() + 1, 0);
    {
        int64_t d = 0, acc = 0;
        for (size_t p = 0; p <= top.size(); ++p) {
            while (d < nd && deferred[d].insert_pos <= static_cast<int64_t>(p)) {
                acc += static_cast<int64_t>(sub[d].size()) - 1;
                ++d;
            }
            shift_at[p] = acc;
        }
    }
    for (size_t p = 0; p < top.size(); ++p) {
        HostNode n = top[p];
        if (n.left >= 0) n.left = static_cast<int32_t>(n.left + shift_at[n.left]);
        nodes[p + shift_at[p]] = n;
    }
    Pool::get().parallel_for(nd, [&](int64_t i) {
        const FlatVec<HostNode>& v = sub[i];
        const int64_t base = block_start[i] - 1;  // local index L >= 1  ->  base + L
        HostNode& r = nodes[deferred[i].node + shift_at[deferred[i].node]];
        r = v[0];
        if (r.left >= 0) r.left = static_cast<int32_t>(base + r.left);
        for (size_t L = 1; L < v.size(); ++L) {
            HostNode n = v[L];
            if (n.left >= 0) n.left = static_cast<int32_t>(base + n.left);
            nodes[base + static_cast<int64_t>(L)] = n;
        }
    });
    if (g_build_timing && n_elem >= kParMin)
        std::fprintf(stderr, "  build_tree %lld elements: top (%zu nodes, %lld deferred) %.1f ms, subtrees %.1f ms, splice %.1f ms\n",
                     (long long)n_elem, top.size(), (long long)nd, (t_tree1 - t_tree0) * 1e3, (t_tree2 - t_tree1) * 1e3,
                     (now_s() - t_tree2) * 1e3);
}

void build_one_blas(HostBlas& b, int32_t mesh, const double* coords_t, int64_t E, Centroid& mesh_cent) {
    const double t_b0 = now_s();
    FlatVec<Rec> rec(static_cast<size_t>(E));  // (not zero-filled first) boxes + centroids, permuted together with the index array
    Box mesh_box;
    mesh_box.reset();
    std::atomic<bool> non_finite{false};
    {
        const int64_t max_chunks = E < kParMin ? 1 : (E + kParChunk - 1) / kParChunk;
        std::vector<Box> part(static_cast<size_t>(max_chunks));
        int64_t nc = 0;
        for_chunks(0, E, nc, [&](int64_t c, int64_t lo, int64_t hi) {
            // rtbvh.cpp:48-78: centroid (v0 + v1 + v2) / 3.0 per component (:20-27), box = the three vertices added to an
            // empty box in order, mesh box += box: Box::add_point / add_box spelt out on registers (the same MINPD /
            // MAXPD with the incoming value first, x and y as a pair, z alone; the divisions stay divisions) -- through
            // the records in memory the loop was three times as long
            const __m128d inf2 = _mm_set1_pd(kInf), ninf2 = _mm_set1_pd(-kInf), three2 = _mm_set1_pd(3.0);
            __m128d mb_mn01 = inf2, mb_mx01 = ninf2, mb_mn2 = inf2, mb_mx2 = ninf2, bad = _mm_setzero_pd();
            for (int64_t e = lo; e < hi; ++e) {
                const double* tc = coords_t + e * 9;
                const __m128d a01 = _mm_loadu_pd(tc), b01 = _mm_loadu_pd(tc + 3), c01 = _mm_loadu_pd(tc + 6);
                const __m128d a2 = _mm_load_sd(tc + 2), b2 = _mm_load_sd(tc + 5), c2 = _mm_load_sd(tc + 8);
                Rec& r = rec[e];
                const __m128d cen01 = _mm_div_pd(_mm_add_pd(_mm_add_pd(a01, b01), c01), three2);
                const __m128d cen2 = _mm_div_sd(_mm_add_sd(_mm_add_sd(a2, b2), c2), three2);
                _mm_storeu_pd(r.c.c, cen01);
                _mm_store_sd(r.c.c + 2, cen2);
                // (x - x is a NaN exactly when x is a NaN or an infinity)
                const __m128d z01 = _mm_sub_pd(cen01, cen01), z2 = _mm_sub_sd(cen2, cen2);
                bad = _mm_or_pd(bad, _mm_or_pd(_mm_cmpunord_pd(z01, z01), _mm_cmpunord_sd(z2, z2)));
                const __m128d mn01 = _mm_min_pd(c01, _mm_min_pd(b01, _mm_min_pd(a01, inf2)));
                const __m128d mx01 = _mm_max_pd(c01, _mm_max_pd(b01, _mm_max_pd(a01, ninf2)));
                const __m128d mn2 = _mm_min_sd(c2, _mm_min_sd(b2, _mm_min_sd(a2, inf2)));
                const __m128d mx2 = _mm_max_sd(c2, _mm_max_sd(b2, _mm_max_sd(a2, ninf2)));
                _mm_storeu_pd(r.box.mn, mn01);
                _mm_store_sd(r.box.mn + 2, mn2);
                _mm_storeu_pd(r.box.mx, mx01);
                _mm_store_sd(r.box.mx + 2, mx2);
                mb_mn01 = _mm_min_pd(mn01, mb_mn01);
                mb_mx01 = _mm_max_pd(mx01, mb_mx01);
                mb_mn2 = _mm_min_sd(mn2, mb_mn2);
                mb_mx2 = _mm_max_sd(mx2, mb_mx2);
            }
            Box mb;
            _mm_storeu_pd(mb.mn, mb_mn01);
            _mm_store_sd(mb.mn + 2, mb_mn2);
            _mm_storeu_pd(mb.mx, mb_mx01);
            _mm_store_sd(mb.mx + 2, mb_mx2);
            part[c] = mb;
            if (_mm_movemask_pd(bad)) non_finite.store(true, std::memory_order_relaxed);
        });
        for (int64_t c = 0; c < nc; ++c) mesh_box.add_box(part[c]);
    }
    // A centroid that is a NaN or an infinity (such vertex coordinates, or a sum that overflows) turns the reference's
    // bin index static_cast<int>(t * NUM_BINS) (rtbvh.cpp:184) into undefined behaviour -- an out-of-range write on
    // x86-64 -- as soon as the mesh is split at all (more than 4 triangles): refused here instead.
    if (non_finite.load(std::memory_order_relaxed) && E > 4)
        throw std::runtime_error("mesh " + std::to_string(mesh) +
                                 ": a triangle's centroid is not finite (NaN, infinite or overflowing vertex coordinates)");
    for (int i = 0; i < 3; ++i) mesh_cent.c[i] = (mesh_box.mn[i] + mesh_box.mx[i]) / 2.0;
    std::memcpy(b.mn, mesh_box.mn, sizeof(b.mn));
    std::memcpy(b.mx, mesh_box.mx, sizeof(b.mx));
    b.mesh = mesh;
    b.sorted_to_elem.resize(static_cast<size_t>(E));
    for (int64_t e = 0; e < E; ++e) b.sorted_to_elem[e] = static_cast<int32_t>(e);
    Elements el{nullptr, nullptr, rec.data()};
    const double t_b1 = now_s();
    build_tree(b.nodes, el, b.sorted_to_elem.data(), E, 4, b.midpoint_fallbacks);
    if (g_build_timing && E >= kParMin)
        std::fprintf(stderr, "  mesh %d: element records %.1f ms, tree %.1f ms\n", mesh, (t_b1 - t_b0) * 1e3, (now_s() - t_b1) * 1e3);
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

// Test seam: `cases` random class patterns of up to max_n elements (xorshift from `seed`), each partitioned by the
// reference's serial loop and by partition_closed_form with a random chunk length (so chunk borders fall everywhere,
// including on S): returns the number of cases whose index permutation, moved records or left count differ.
int64_t partition_selftest(int64_t cases, uint64_t seed, int max_n) {
    uint64_t x = seed ? seed : 0x9e3779b97f4a7c15ull;
    auto rnd = [&]() {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        return x;
    };
    int64_t bad = 0;
    std::vector<int32_t> a, b;
    std::vector<Rec> ra, rb;
    std::vector<uint8_t> left;
    const auto serial_run = [](int64_t n, const std::function<void(int64_t)>& fn) {
        for (int64_t i = 0; i < n; ++i) fn(i);
    };
    for (int64_t t = 0; t < cases; ++t) {
        const int64_t n = 1 + static_cast<int64_t>(rnd() % static_cast<uint64_t>(std::max(1, max_n)));
        const int64_t first = static_cast<int64_t>(rnd() % 5);
        const uint64_t bias = rnd() % 11;  // 0: all right-class ... 10: all left-class
        a.assign(static_cast<size_t>(first + n), -1);
        ra.assign(static_cast<size_t>(first + n), Rec{});
        left.assign(static_cast<size_t>(first + n), 0);
        for (int64_t i = first; i < first + n; ++i) {
            a[i] = static_cast<int32_t>(i - first);
            left[i] = (rnd() % 10) < bias;
            ra[i].c.c[0] = left[i] ? -1.0 : 1.0;  // the class travels with the record
            ra[i].c.c[1] = static_cast<double>(i - first);
        }
        b = a;
        rb = ra;
        int64_t mid = first, end = first + n;  // the reference's loop, rtbvh.cpp:271-285
        while (mid < end) {
            if (ra[mid].c.c[0] < 0.0) {
                ++mid;
            } else {
                --end;
                std::swap(a[mid], a[end]);
                std::swap(ra[mid], ra[end]);
            }
        }
        const int64_t chunk = 1 + static_cast<int64_t>(rnd() % 9);
        const int64_t nl = partition_closed_form(b.data(), rb.data(), first, n, chunk,
                                                 [&](int64_t i) { return rb[i].c.c[0] < 0.0; }, serial_run);
        bool ok = nl == mid - first && a == b;
        for (int64_t i = first; ok && i < first + n; ++i)
            ok = ra[i].c.c[1] == rb[i].c.c[1] && ra[i].c.c[0] == rb[i].c.c[0];
        bad += !ok;
    }
    return bad;
}

Scene* build_scene(int n_meshes, const double* const* coords, const double* const* colors,
                   const int64_t* elem_counts, int timestep, int n_threads, bool serial_trees) {
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

    const TicketScope ticket;  // this build's place in the pool's queue (oldest build first)
    std::unique_ptr<Scene> sc(new Scene());
    std::vector<HostBlas> blases(static_cast<size_t>(n_meshes));
    std::vector<Centroid> mesh_cent(static_cast<size_t>(n_meshes));

    // ---- BLAS per mesh (independent; built concurrently, results do not depend on threads) ----
    auto work = [&](int m) {
        const bool outer = tl_serial_trees;  // (thread-local: this mesh's worker, for the length of this tree)
        tl_serial_trees = serial_trees;
        const int64_t E = elem_counts[m];
        try {
            build_one_blas(blases[m], m, coords[m] + static_cast<int64_t>(timestep) * E * 9, E, mesh_cent[m]);
        } catch (...) {
            tl_serial_trees = outer;
            throw;
        }
        tl_serial_trees = outer;
    };
    int nt = n_threads <= 0 ? std::min(n_meshes, host_threads()) : std::min(n_threads, n_meshes);
    if (nt <= 1) {
        for (int m = 0; m < n_meshes; ++m) work(m);
    } else if (n_threads <= 0) {
        // the process's own pool: the meshes of the timesteps a frame driver keeps in flight queue up behind each other
        // (the oldest build is served first and finishes first) instead of time-sharing the cores as 8 threads per build
        Pool::get().parallel_for(n_meshes, [&](int64_t m) { work(static_cast<int>(m)); });
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

    const double t_blas = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    // ---- TLAS over meshes ----
    std::vector<Box> mesh_boxes(static_cast<size_t>(n_meshes));
    for (int m = 0; m < n_meshes; ++m) {
        std::memcpy(mesh_boxes[m].mn, blases[m].mn, sizeof(blases[m].mn));
        std::memcpy(mesh_boxes[m].mx, blases[m].mx, sizeof(blases[m].mx));
        sc->midpoint_fallbacks += blases[m].midpoint_fallbacks;
    }
    std::vector<int32_t> blas_idx(static_cast<size_t>(n_meshes));
    for (int m = 0; m < n_meshes; ++m) blas_idx[m] = m;
    if (n_meshes > 2)  // (the TLAS is split from three meshes on: the same bin index, rtbvh.cpp:184)
        for (int m = 0; m < n_meshes; ++m)
            for (int i = 0; i < 3; ++i)
                if (!std::isfinite(mesh_cent[m].c[i]))
                    throw std::runtime_error("mesh " + std::to_string(m) + ": the centre of its bounding box is not finite");
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
    // (190 MB at 5 M triangles: not filled first -- every node record is written by the sweep below, only the pad
    // records in front of and behind a slot's nodes are cleared here)
    sc->dev_nodes.resize(static_cast<size_t>(node_base));
    for (int s = 0; s < n_slots; ++s) {
        const int64_t lo = slot_node_base[s], nn = static_cast<int64_t>(sc->slots[s].nodes.size());
        const int64_t hi = s + 1 < n_slots ? slot_node_base[s + 1] : node_base;
        sc->dev_nodes[lo] = DevNode{};
        for (int64_t i = lo + 1 + nn; i < hi; ++i) sc->dev_nodes[i] = DevNode{};
    }
    sc->tris.resize(static_cast<size_t>(total_tris) * 9);
    sc->colors.resize(static_cast<size_t>(total_tris) * 3);
    sc->prim_mesh.resize(static_cast<size_t>(total_tris));
    sc->prim_elem.resize(static_cast<size_t>(total_tris));

    std::vector<int32_t> slot_need(n_slots, 0), slot_max_leaf(n_slots, 0);
    std::vector<double> slot_max_abs(static_cast<size_t>(n_slots), 0.0);
    std::vector<std::vector<int64_t>> big_leaves(n_slots);  // leaves that need a leaf_ext entry, in sweep order
    Pool::get().parallel_for(n_slots, [&](int64_t s) {
        const HostBlas& b = sc->slots[s];
        const int64_t E = static_cast<int64_t>(b.sorted_to_elem.size());
        const double* ct = coords[b.mesh] + static_cast<int64_t>(timestep) * E * 9;
        const double* kt = colors[b.mesh] + static_cast<int64_t>(timestep) * E * 3;
        const int64_t tb = sc->slot_tri_base[s];
        {
            // (the largest |coordinate| of the scene is gathered on the way: NaN counts as infinity)
            std::vector<double> part(static_cast<size_t>(E < kParMin ? 1 : (E + kParChunk - 1) / kParChunk), 0.0);
            int64_t nc = 0;
            for_chunks(0, E, nc, [&](int64_t c, int64_t lo, int64_t hi) {
                const __m128d absmask = _mm_castsi128_pd(_mm_set1_epi64x(0x7fffffffffffffffll));
                __m128d vmax = _mm_setzero_pd(), vnan = _mm_setzero_pd();
                for (int64_t p = lo; p < hi; ++p) {
                    const int64_t e = b.sorted_to_elem[p];
                    double* const dst = &sc->tris[(tb + p) * 9];
                    std::memcpy(dst, ct + e * 9, 9 * sizeof(double));
                    std::memcpy(&sc->colors[(tb + p) * 3], kt + e * 3, 3 * sizeof(double));
                    sc->prim_mesh[tb + p] = b.mesh;
                    sc->prim_elem[tb + p] = static_cast<int32_t>(e);
                    // |x| pairwise; MAXPD keeps the running maximum when the incoming value is a NaN, so NaNs are
                    // noted apart (x != x) and turn the result into infinity at the end
                    const __m128d x01 = _mm_loadu_pd(dst), x23 = _mm_loadu_pd(dst + 2), x45 = _mm_loadu_pd(dst + 4),
                                  x67 = _mm_loadu_pd(dst + 6), x88 = _mm_load1_pd(dst + 8);
                    vmax = _mm_max_pd(_mm_max_pd(_mm_and_pd(x01, absmask), _mm_and_pd(x23, absmask)),
                                      _mm_max_pd(_mm_max_pd(_mm_and_pd(x45, absmask), _mm_and_pd(x67, absmask)),
                                                 _mm_max_pd(_mm_and_pd(x88, absmask), vmax)));
                    vnan = _mm_or_pd(vnan, _mm_or_pd(_mm_or_pd(_mm_cmpunord_pd(x01, x23), _mm_cmpunord_pd(x45, x67)),
                                                     _mm_cmpunord_pd(x88, x88)));
                }
                double m2[2];
                _mm_storeu_pd(m2, vmax);
                part[static_cast<size_t>(c)] = _mm_movemask_pd(vnan) ? kInf : std::max(m2[0], m2[1]);
            });
            double m = 0.0;
            for (int64_t c = 0; c < nc; ++c)
                if (!(part[static_cast<size_t>(c)] <= m)) m = part[static_cast<size_t>(c)];
            slot_max_abs[static_cast<size_t>(s)] = m;
        }
        const int64_t nb = slot_node_base[s] + 1;
        sc->slot_root[s] = static_cast<int32_t>(nb);
        const int64_t nn = static_cast<int64_t>(b.nodes.size());
        // node records: independent of each other, written by several threads
        {
            int64_t ncn = 0;
            for_chunks(0, nn, ncn, [&](int64_t, int64_t lo, int64_t hi) {
                for (int64_t i = lo; i < hi; ++i) {
                    const HostNode& h = b.nodes[i];
                    DevNode& d = sc->dev_nodes[nb + i];
                    std::memcpy(d.mn, h.mn, sizeof(d.mn));
                    std::memcpy(d.mx, h.mx, sizeof(d.mx));
                    if (h.left < 0) {
                        d.left = -1;
                        d.count = static_cast<int32_t>(h.count);
                        d.first = static_cast<int32_t>(tb + h.first);
                        d.desc = (h.count >= 1 && h.count <= 7)
                                     ? DESC_LEAF | (static_cast<uint32_t>(h.count - 1) << 28) | static_cast<uint32_t>(tb + h.first)
                                     : 0u;  // leaf_ext slots are handed out below, in a fixed order
                    } else {
                        d.left = static_cast<int32_t>(nb + h.left);
                        d.count = 0;
                        d.first = -1;
                        d.desc = static_cast<uint32_t>(nb + h.left);
                    }
                }
            });
        }
        // children have larger indices than their parent: one reverse sweep over the host nodes gives the stack need
        std::vector<int32_t> need(static_cast<size_t>(nn), 0);
        for (int64_t i = nn - 1; i >= 0; --i) {
            const HostNode& h = b.nodes[i];
            if (h.left < 0) {
                if (!(h.count >= 1 && h.count <= 7)) big_leaves[s].push_back(i);
                slot_max_leaf[s] = std::max<int32_t>(slot_max_leaf[s], static_cast<int32_t>(h.count));
            } else {
                need[i] = std::max(need[h.left] + 1, need[h.left + 1]);
            }
        }
        slot_need[s] = need[0];
    });
    for (int s = 0; s < n_slots; ++s) {
        sc->max_leaf_tris = std::max(sc->max_leaf_tris, slot_max_leaf[s]);
        const int64_t nb = slot_node_base[s] + 1, tb = sc->slot_tri_base[s];
        for (int64_t i : big_leaves[s]) {
            const HostNode& h = sc->slots[s].nodes[i];
            sc->dev_nodes[nb + i].desc = leaf_desc(tb + h.first, h.count, sc->leaf_ext);
        }
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
    for (int s = 0; s < n_slots; ++s)
        if (!(slot_max_abs[static_cast<size_t>(s)] <= sc->max_abs_coord)) sc->max_abs_coord = slot_max_abs[static_cast<size_t>(s)];

    sc->build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    if (std::getenv("RT_BUILD_TIMING"))
        std::fprintf(stderr, "rt_scene_build: %lld triangles, BLAS builds %.1f ms, TLAS + flatten %.1f ms\n",
                     (long long)total_tris, t_blas * 1e3, (sc->build_seconds - t_blas) * 1e3);
    return sc.release();
}

}  // namespace rt
