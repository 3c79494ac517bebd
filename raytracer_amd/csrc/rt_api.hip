// rt_api.hip -- C ABI of librtgpu.so (see include/rtgpu.h): contexts, uploads, launches,
// the MT19937 stream bookkeeping and the whole-job driver.  Host + device in one TU so the
// kernels can be launched with <<<>>>; compiled for gfx950 only, -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rtgpu.h"
#include "rt_bvh.hpp"
#include "rt_kernels.hip.hpp"
#include "rt_kernels_fp32.hip.hpp"
#include "rt_mt19937.hpp"

struct rt_scene {
    std::unique_ptr<rt::Scene> s;
};

// Scratch of rt_render_static (the entry point that allocates nothing): the camera-relative node / TLAS /
// triangle records the prep kernels write for the traversal kernel, and the visit counters.  One copy per GPU,
// held until the process ends.  In host code this name is only a handle for hipGetSymbolAddress: the device
// address is resolved once per device (static_scratch_ptr) and that pointer is what kernels, memsets and
// copies get.
static __device__ __align__(256) unsigned char scratch[RT_STATIC_SCRATCH_BYTES];

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(RT_E_NO_DEVICE, std::string(#expr) + " failed: " + hipGetErrorString(e_)); \
    } while (0)


// ---- options -----------------------------------------------------------------------------------------------
// Every RT_* tuning / comparison / test variable of this file is read from the environment ONCE
// -- by the first rt_create, or by the first host-only call that needs one -- and again only on rt_reload_options():
// no render, upload or driver call reaches getenv.  opt("RT_X") returns the value taken then, or null when unset,
// so each site keeps the meaning it had when it asked the environment itself.
const char* const kOptionNames[] = {"RT_ADAPT_MIN_WORDS", "RT_ADAPT_RATIO", "RT_BUILD_AHEAD", "RT_BUILD_TIMING", "RT_COUNT_WITH_CUTS", "RT_CUT1_CAP", "RT_CUT1_MAX_TRIS_PER_BLOCK", "RT_CUT1_MIN_RAYS", "RT_CUT1_SERIAL", "RT_CUT_LO", "RT_CUT_MIN_RAYS", "RT_CUT_STEP", "RT_CUT_TOP", "RT_CUT_TOPCAP", "RT_DRIVER_TIMING", "RT_FORCE_PERLANE", "RT_FRAMES_AHEAD", "RT_INTERLEAVE", "RT_JITTER_SLOT_BYTES", "RT_JITTER_STREAMS", "RT_NO_ADAPT_STREAMS", "RT_NO_CUT", "RT_NO_FRAME_GROUPS", "RT_NO_INTERLEAVE", "RT_NO_LOOKAHEAD", "RT_NO_PAIR64", "RT_NO_PIPELINE", "RT_NO_PREP_OVERLAP", "RT_NO_SERIAL_TREES", "RT_NO_SMALL_TILES", "RT_NO_STATE_REUSE", "RT_NO_TRIM", "RT_PAD64_FROM", "RT_RNG_PRIO", "RT_STAGE_STREAMS", "RT_TILE_LOG2", "RT_UPLOAD_PAR_MIN"};
constexpr int kNumOptions = (int)(sizeof(kOptionNames) / sizeof(kOptionNames[0]));
struct OptionTable {
    std::mutex mu;
    bool loaded = false;
    bool set[kNumOptions] = {};
    std::string val[kNumOptions];
    void load() {  // (callers hold mu)
        for (int i = 0; i < kNumOptions; ++i) {
            const char* e = std::getenv(kOptionNames[i]);
            set[i] = e != nullptr;
            val[i] = e ? e : "";
        }
        loaded = true;
    }
} g_options;

// Not safe against a concurrent rt_reload_options (documented: reload while no call is in flight).
const char* opt(const char* name) {
    if (!g_options.loaded) {
        std::lock_guard<std::mutex> lk(g_options.mu);
        if (!g_options.loaded) g_options.load();
    }
    for (int i = 0; i < kNumOptions; ++i)
        if (std::strcmp(kOptionNames[i], name) == 0) return g_options.set[i] ? g_options.val[i].c_str() : nullptr;
    return nullptr;  // (not a registered option: a typo in this file)
}

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;  // elements
    int ensure(size_t n) {
        if (n <= cap) return RT_OK;
        // A buffer that grows a second time is likely to grow again (the trees of a deforming mesh gain a few nodes
        // per timestep): an eighth of headroom, because hipFree waits for the whole device -- in a pipelined frame
        // loop that is one frame time per call, every call.  The first allocation is exact.
        if (p) {
            (void)hipFree(p);
            n += n / 8;
        }
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(n, 1) * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            return fail(RT_E_NOMEM, std::string("hipMalloc failed: ") + hipGetErrorString(e));
        }
        cap = n;
        return RT_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// Levels of the cut hierarchy of one launch (cut_level_kernel): level l = blocks of (1 << l)^2 pixels, from the
// single block of level `top` down to level `lo`, the finest one a 64-ray batch can ask for.
struct CutPlan {
    int lo = 0, top = 0;
    bool present[32] = {};  // levels that are made (top, lo and, with RT_CUT_STEP=1, every level in between)
    int tiles_x[32] = {};
    int64_t tiles[32] = {};
    int64_t offset_words[32] = {};
    int64_t total_words = 0;
    bool single = false;  // cut_mode 1: one level, the render tile's
};

// How a frame's walk starts (RT_CUT_MIN_RAYS, RT_CUT1_MIN_RAYS, RT_CUT1_MAX_TRIS_PER_BLOCK: tuning knobs):
//   2  the whole hierarchy of cuts down to the finest level a batch can ask for, leaves trimmed: frames of >= 12 M
//      rays (its kernels give a block one thread that walks depth first: they need many blocks to fill the GPU);
//   1  ONE level, the render tile's blocks, straight from the root, made breadth first with one lane per list entry
//      (cut_single_kernel): frames of 1 M .. 12 M rays -- the reference's own 400 x 225 x 50 -- of scenes with at
//      most 8 triangles per block.  A list of <= 31 entries pays when it reaches the leaves; over a block that
//      holds hundreds of triangles it is 31 internal nodes, each a per-ray test with its dispatch, and loses
//      3-10 % against the root walk (profiles/r03f/cut_policy_sweep.txt);
//   0  every batch walks from the root.
double cut_min_rays() {
    const char* mr = opt("RT_CUT_MIN_RAYS");
    return mr ? std::atof(mr) : 12e6;
}
double cut1_min_rays() {
    const char* mr = opt("RT_CUT1_MIN_RAYS");
    return mr ? std::atof(mr) : 1e6;
}
int cut_mode(int W, int rows, int spp, int tile_log2, int64_t n_tris) {
    const double rays = double(W) * double(rows) * double(spp);
    if (rays >= cut_min_rays()) return 2;
    if (rays < cut1_min_rays()) return 0;
    const int ts = 1 << tile_log2;
    const double blocks = double((W + ts - 1) / ts) * double((rows + ts - 1) / ts);
    const char* mt = opt("RT_CUT1_MAX_TRIS_PER_BLOCK");
    const double per_block = mt ? std::atof(mt) : 8.0;
    return double(n_tris) <= per_block * blocks ? 1 : 0;
}

// Entries a list of level l may hold: coarse levels only hand a starting point down (their blocks are refined
// again, four times over, by their children), so they stop at half the room.
int cut_cap(int l, const CutPlan& cp) {
    if (cp.single) {
        const int cap1 = opt("RT_CUT1_CAP") ? std::atoi(opt("RT_CUT1_CAP")) : (int)rtk::CUT_CAP;  // (tuning knob)
        return std::max(1, std::min(cap1, (int)rtk::CUT_CAP));
    }
    const int top_cap = opt("RT_CUT_TOPCAP") ? std::atoi(opt("RT_CUT_TOPCAP")) : 15;  // (tuning knob)
    return (l > cp.lo + 1) ? std::max(1, std::min(top_cap, (int)rtk::CUT_CAP)) : (int)rtk::CUT_CAP;
}

CutPlan plan_cuts(int W, int rows, int spp, int tile_log2, int64_t n_tris) {
    CutPlan cp;
    // a batch holds the rays of ceil(64 / spp) pixels or more: the smallest aligned block it can lie in
    const int pix = (64 + spp - 1) / spp;
    while ((1 << (2 * cp.lo)) < pix) ++cp.lo;
    // ... but not below 2 x 2 pixels: measured on the 4096^2 x 64 spp frame, per-pixel cuts take 0.7 ms off the
    // render kernel and cost 3 ms to make
    cp.lo = std::max(cp.lo, 1);
    if (const char* e = opt("RT_CUT_LO")) cp.lo = std::max(0, std::atoi(e));  // (tuning knob)
    cp.lo = std::min(cp.lo, tile_log2);
    // The top level's blocks start at the TLAS root, one thread each: a long chain of dependent steps, but the
    // same for any number of blocks, so the hierarchy starts where there are enough blocks to fill the GPU with
    // that chain (8 x 8 pixels or the render tile), not at the single block that holds the image.
    cp.top = std::max(tile_log2, 3);
    // (tuning knob; never below the render tile's level: a batch looks its cut up at levels lo..tile_log2 and takes
    // the next coarser level that was made, which must exist)
    if (const char* e = opt("RT_CUT_TOP")) cp.top = std::max({cp.lo, tile_log2, std::atoi(e)});
    if (cut_mode(W, rows, spp, tile_log2, n_tris) == 1) {
        cp.single = true;
        cp.lo = cp.top = tile_log2;
    }
    int whole = 0;
    while ((1 << whole) < std::max(W, rows)) ++whole;
    cp.top = std::max(cp.lo, std::min(cp.top, std::max(whole, tile_log2)));
    // every level: a batch that straddles two blocks of the finest level takes the next coarser level's cut, and the
    // further up that is, the looser (2048^2 x 50 spp, two pixels per batch: 6.21 -> 5.87 ms per frame against
    // every second level; one pixel per batch at 64 spp: no difference)
    const int step = opt("RT_CUT_STEP") ? std::max(1, std::atoi(opt("RT_CUT_STEP"))) : 1;  // (tuning knob)
    for (int l = cp.top; l >= cp.lo; --l) {
        // every `step`-th level below the top, and the finest one
        if (l != cp.lo && (cp.top - l) % step != 0) continue;
        cp.present[l] = true;
        const int ts = 1 << l;
        cp.tiles_x[l] = (W + ts - 1) / ts;
        cp.tiles[l] = int64_t(cp.tiles_x[l]) * ((rows + ts - 1) / ts);
        cp.offset_words[l] = cp.total_words;
        cp.total_words += cp.tiles[l] * rtk::CUT_WORDS;
    }
    return cp;
}

}  // namespace

struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;      // traversal kernels
    hipStream_t rng_stream = nullptr;  // jitter kernel (overlaps the previous band's traversal)
    hipEvent_t ev[6] = {};             // 0/1 render start/stop, 2/3 jitter start/stop, 4 jitter done, 5 render done
    // scene
    bool have_scene = false;
    DevBuf<rt::DevNode> nodes, tlas;
    DevBuf<int32_t> slot_root;
    DevBuf<uint32_t> leaf_ext;
    DevBuf<double> tris, colors;
    // Second set of scene buffers: rt_upload_scene fills it on copy_stream while kernels launched
    // earlier still read the current set, then the two sets trade places.
    struct BackScene {
        DevBuf<rt::DevNode> nodes, tlas;
        DevBuf<int32_t> slot_root;
        DevBuf<uint32_t> leaf_ext;
        DevBuf<double> tris, colors;
    } back;
    hipStream_t copy_stream = nullptr;
    hipEvent_t front_last_use = nullptr, back_last_use = nullptr;  // last render launched on each set
    bool front_used = false, back_used = false;
    // rt_render_scene's helper threads stage the tree they just built into one of these sets themselves
    // (own stream, off the calling thread); the calling thread then only swaps pointers (activate_staged).
    struct StagedSet {
        BackScene b;
        hipStream_t stream = nullptr;
        bool owns_stream = false;
        hipEvent_t last_use = nullptr;  // after a swap: last render on the buffers this set now holds
        bool used = false;
        int state = 0;  // 0 free, 1 being filled, 2 staged
        size_t n_nodes = 0, n_tlas = 0;
        int64_t n_tris = 0;
        int32_t stack_need = 1;
        double max_abs_coord = 0.0;
    };
    std::vector<std::unique_ptr<StagedSet>> staged;
    std::mutex staged_mu;
    // Camera-relative copies for the packet kernels and the cuts made from them, rebuilt per render by the prep and
    // cut kernels.  Two sets used alternately: the prep + cut kernels of a launch run on prep_stream into the set
    // the PREVIOUS launch is not reading, so they overlap that launch's traversal kernel (they are chains of short,
    // latency-bound launches; the traversal is VALU-bound).  `ready`: the set's prep + cut kernels (prep_stream);
    // `read_seq`: the launch whose traversal kernel read it last (done_event_for finds the event that covers it).
    struct CamSet {
        DevBuf<rt::DevNode> nodes, tlas;
        DevBuf<rtk::CamTri> tris;
        DevBuf<uint32_t> cuts;  // per-block entry lists (cut_level_kernel)
        hipEvent_t ready = nullptr;
        uint64_t read_seq = 0;  // the launch whose traversal kernel read the set last
        bool read_pending = false;
    } cam[2];
    int cam_idx = 0;                    // set of the last launch
    hipStream_t prep_stream = nullptr;
    CutPlan cut_plan;             // levels of the last rt_render's cuts in cam[cam_idx].cuts
    bool cut_plan_valid = false;  // false: the last render made none
    // fp32 fast path records (RT_FLAG_FP32), allocated on first use
    DevBuf<rtk32::Node32> nodes32;
    DevBuf<rtk32::Tlas32> tlas32;
    DevBuf<rtk32::Tri32> tris32;
    // per-frame camera block, two copies used alternately (a pipelined launch must not overwrite the
    // host or device copy the previous launch may still be reading)
    DevBuf<rtk::FrameParams> frame_params_buf[2];
    rtk::FrameParams h_frame_params_buf[2];
    int fp_idx = 0;
    // rt_render_scene's pipelined launches: frame q+1 is enqueued before frame q is read back, so the
    // frame buffers alternate (intensity/rgb always hold the LAST launch's output, prev_* the one before)
    DevBuf<double> prev_intensity, prev2_intensity;  // the frames of the previous launch and of the one before it
    DevBuf<uint8_t> prev_rgb, prev2_rgb;
    hipEvent_t out_done = nullptr, prev_out_done = nullptr, prev2_out_done = nullptr;  // traversal kernel of the last / previous / one before
    uint64_t launch_seq = 0;                                    // launches of render_impl so far
    uint64_t out_seq = 0, prev_out_seq = 0, prev2_out_seq = 0;  // the launch each of the three events was recorded behind (0: never)
    size_t n_nodes = 0, n_tlas = 0;
    double max_abs_coord = 0.0;  // largest |coordinate| of the uploaded scene (NaN/inf -> inf)
    int32_t stack_need = 1;
    int64_t n_tris = 0;
    // sampler
    uint32_t seed = 5489u;
    uint64_t pos = 0;            // position of the global stream, in 32-bit outputs
    uint64_t state_pos = 0;      // position d_state belongs to (<= pos: a slot of several frames is consumed frame by
                                 // frame without a jump per frame; sync_state() brings d_state up to pos when needed)
    DevBuf<uint32_t> d_tmp;      // scratch state for in-place jumps
    DevBuf<uint32_t> d_state;    // generator state preceding `pos` (624 raw words), device resident;
                                 // every sampler operation is ordered on rng_stream
    rt::MtState h_seed_state;    // host copy of the seeded state, kept alive for the async upload
    std::map<uint64_t, unsigned long long*> d_polys;  // jump polynomials x^stride mod phi, uploaded once
    // Jitter slots: the pairs for one launch each.  While the traversal kernel of call n reads
    // slot A, the jitter kernel for the predicted call n+1 fills slot B on rng_stream.
    struct JitterSlot {
        DevBuf<double> buf;
        DevBuf<uint32_t> d_states;  // one state per stream
        DevBuf<uint32_t> d_end;     // state preceding pos + rows*W*spp*4
        bool valid = false;
        uint64_t pos = 0;  // stream position the slot starts at
        int rows = 0, W = 0, spp = 0;
        int frames = 1;    // consecutive launches of that shape the slot holds the pairs of (small frames: several)
        hipEvent_t done = nullptr, t0 = nullptr, t1 = nullptr;
        uint64_t read_seq = 0;  // the launch whose traversal kernel read buf last (refills wait for it)
        bool read_pending = false;
        double host_ms = 0.0;
        // what d_states holds (survives `valid = false`): n states, `stride` words apart, from states_pos
        bool states_ok = false;
        uint64_t states_pos = 0, states_stride = 0;
        int states_n = 0;
    } slot[2];
    int last_slot = 0;
    // Streams of the look-ahead fills of very large frames (>= 2^30 words), learnt from the last launch of the same
    // shape: as few as keep the sampler (jumps + jitter kernel, timed while it overlapped that launch's traversal)
    // within 0.8 of the traversal kernel's time; 0: not learnt yet (the default plan, 1024 streams).
    int adapt_n = 0;
    int adapt_rows = 0, adapt_w = 0, adapt_spp = 0;
    // a new count needs new jump polynomials (~20 ms of host work): a helper thread computes them into the host
    // cache while frames go on with the old count, which gives way once they are there
    std::future<void> adapt_job;
    int adapt_pending_n = 0, adapt_pending_rows = 0, adapt_pending_w = 0, adapt_pending_spp = 0;
    bool have_last_start = false;
    uint64_t last_start = 0;
    DevBuf<uint32_t> d_states;  // rt_rng_jitter only
    DevBuf<double> jitter;      // rt_rng_jitter only
    // outputs of the last render
    DevBuf<double> intensity, hit_t;
    DevBuf<uint8_t> rgb;
    DevBuf<int32_t> hit_prim;
    DevBuf<unsigned long long> counters;
    // a frame whose jitter slot would exceed the budget is rendered band by band (render_frame): whole-frame outputs
    DevBuf<double> frame_intensity, frame_hit_t;
    DevBuf<uint8_t> frame_rgb;
    DevBuf<int32_t> frame_hit_prim;
    unsigned long long* stats_counters = nullptr;  // where the last launch's visit counters are (or null)
    unsigned char* static_scratch = nullptr;       // this GPU's copy of the file-scope `scratch` (rt_render_static)
    hipEvent_t static_done = nullptr;              // recorded behind every rt_render_static launch (created in rt_create)
    bool skip_timing_events = false;               // this launch: no ev[0] / ev[1] around the traversal kernel (render_impl)
    DevBuf<uint32_t> gstack;
    int32_t out_rows = 0, out_w = 0, out_spp = 0;
    uint32_t out_flags = 0;
    bool timing_pending = false, jitter_timed = false;
    bool kernel_ms_known = false;       // stats.render_kernel_ms of the last launch has been read from its events
    double kernel_ms_total = 0.0;       // over all launches since rt_create whose events were read
    uint64_t kernel_launches_timed = 0;
    rt_stats stats{};
    // scratch for rt_trace_rays
    DevBuf<double> tr_o, tr_d, tr_t, tr_b, tr_c;
    DevBuf<int32_t> tr_p;
};

namespace {

rtk::DevScene dev_scene(const rt_ctx* c) {
    rtk::DevScene d;
    d.nodes = c->nodes.p;
    d.tlas = c->tlas.p;
    d.slot_root = c->slot_root.p;
    d.leaf_ext = c->leaf_ext.p;
    d.tris = c->tris.p;
    d.colors = c->colors.p;
    return d;
}

// Kernel time of the last launch from its HIP events (the launch must have finished), once: kept in
// stats.render_kernel_ms and added to the context's running total.
int resolve_kernel_ms(rt_ctx* c) {
    if (!c->timing_pending || c->kernel_ms_known) return RT_OK;
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    c->stats.render_kernel_ms = ms;
    c->kernel_ms_total += ms;
    c->kernel_launches_timed += 1;
    c->kernel_ms_known = true;
    return RT_OK;
}

// Smallest LDS stack variant that holds the tree's need; 0 = global-memory stack.
int pick_stack(int need) {
    if (need <= 16) return 16;
    if (need <= 24) return 24;
    if (need <= 32) return 32;
    if (need <= 48) return 48;
    return 0;
}

template <bool COUNT>
void launch_render(int stack, dim3 grid, hipStream_t st, const rtk::RenderArgs& a) {
    switch (stack) {
        case 16: hipLaunchKernelGGL((rtk::render_kernel<16, COUNT>), grid, dim3(rtk::NT), 0, st, a); break;
        case 24: hipLaunchKernelGGL((rtk::render_kernel<24, COUNT>), grid, dim3(rtk::NT), 0, st, a); break;
        case 32: hipLaunchKernelGGL((rtk::render_kernel<32, COUNT>), grid, dim3(rtk::NT), 0, st, a); break;
        case 48: hipLaunchKernelGGL((rtk::render_kernel<48, COUNT>), grid, dim3(rtk::NT), 0, st, a); break;
        default: hipLaunchKernelGGL((rtk::render_kernel<0, COUNT>), grid, dim3(rtk::NT), 0, st, a); break;
    }
}

bool force_per_lane() {
    const char* e = opt("RT_FORCE_PERLANE");
    return e && e[0] == '1';
}

void launch_trace(int stack, dim3 grid, hipStream_t st, const rtk::TraceArgs& a) {
    switch (stack) {
        case 16: hipLaunchKernelGGL((rtk::trace_kernel<16>), grid, dim3(rtk::NT), 0, st, a); break;
        case 24: hipLaunchKernelGGL((rtk::trace_kernel<24>), grid, dim3(rtk::NT), 0, st, a); break;
        case 32: hipLaunchKernelGGL((rtk::trace_kernel<32>), grid, dim3(rtk::NT), 0, st, a); break;
        case 48: hipLaunchKernelGGL((rtk::trace_kernel<48>), grid, dim3(rtk::NT), 0, st, a); break;
        default: hipLaunchKernelGGL((rtk::trace_kernel<0>), grid, dim3(rtk::NT), 0, st, a); break;
    }
}

// Streams of one jitter launch: stream g produces the words [g*stride, (g+1)*stride) of the launch.
struct StreamPlan {
    int n = 1;
    uint64_t stride = 0;       // words
    uint64_t total_words = 0;  // rows*W*spp*4 (the last stream stops there)
};

// Frames from this many words on learn the stream count of their look-ahead fills (adapt_streams).
uint64_t adapt_min_words() {
    const char* e = opt("RT_ADAPT_MIN_WORDS");  // (test knob: the suites' frames are far smaller)
    return e ? (uint64_t)std::max(0.0, std::atof(e)) : (uint64_t(1) << 30);
}

StreamPlan plan_streams(int rows, int W, int spp, int frames = 1, int hidden_n = 0) {
    StreamPlan p;
    // One workgroup per stream.  Measured: a stream regenerates 624 words in ~1.2 us (a chain of dependent
    // LDS round trips), i.e. ~1.9 ns/word, and one jump costs ~1.2 us of the whole GPU.  The traversal kernel
    // needs ~0.015 ns per word, so ~250 streams keep the jitter kernel shorter than half the traversal it
    // hides behind, whatever the frame size; more streams only add jump work.  Up to 256 streams (1024 for
    // very large frames, below), none shorter than 2^16 words (2^13 for the first 64); strides are whole
    // regenerations, not whole rows.
    const uint64_t words = uint64_t(rows) * uint64_t(W) * uint64_t(spp) * 4u * uint64_t(frames);
    uint64_t target = std::min<uint64_t>(256, std::max<uint64_t>(1, words >> 16));
    // small frames are bound by the latency of one wave walking its stream: allow up to 64 shorter streams
    target = std::max<uint64_t>(target, std::min<uint64_t>(64, words >> 13));
    // very large frames (>= 2^30 words): up to 1024 streams.  In a frame loop this changes nothing (the sampler
    // is hidden behind the traversal either way, and its instruction count is the same), but the FIRST frame of
    // a job waits for its sampler: 4096^2 x 64 spp, 14.8 ms with 256 streams, 5.8 ms with 1024.
    target = std::max<uint64_t>(target, std::min<uint64_t>(1024, words >> 22));
    // ... and a look-ahead fill of such a frame, which nobody waits for, takes the number the context has learnt
    // (adapt_streams below): the sampler's workgroups hold wave slots the traversal kernel would use, so it should
    // run on as few as still finish well before the traversal it hides behind
    if (hidden_n > 0 && words >= adapt_min_words()) target = (uint64_t)hidden_n;
    if (const char* e = opt("RT_JITTER_STREAMS")) target = (uint64_t)std::max(1, std::atoi(e));
    uint64_t stride = (words + target - 1) / target;
    stride = (stride + 623) / 624 * 624;
    p.n = (int)((words + stride - 1) / stride);
    p.stride = stride;
    p.total_words = words;
    return p;
}

constexpr size_t kJumpLds = size_t(rtk::JUMP_LDS_WORDS) * sizeof(uint32_t);

// Device copy of the jump polynomial x^stride mod phi (computed on the host once per stride).
int dev_poly(rt_ctx* c, uint64_t stride, const unsigned long long** out) {
    auto it = c->d_polys.find(stride);
    if (it == c->d_polys.end()) {
        if (c->d_polys.size() >= 1024) {  // bound the device-side cache: drop it once nothing uses it any more
            HIP_TRY(hipStreamSynchronize(c->rng_stream));
            for (auto& kv : c->d_polys) (void)hipFree(kv.second);
            c->d_polys.clear();
        }
        const rt::MtPoly& g = rt::mt_jump_poly(stride);  // may throw
        unsigned long long* d = nullptr;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), sizeof(g.b)));
        hipError_t e = hipMemcpy(d, g.b, sizeof(g.b), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(d);
            return fail(RT_E_NO_DEVICE, std::string("hipMemcpy(poly) failed: ") + hipGetErrorString(e));
        }
        it = c->d_polys.emplace(stride, d).first;
    }
    *out = it->second;
    return RT_OK;
}

// dst = state `stride` words after src (dst may alias src), on rng_stream.  The jump kernel
// accumulates into a zeroed buffer with atomics, so an aliased jump goes through c->d_tmp.
int dev_jump(rt_ctx* c, const uint32_t* src, uint32_t* dst, uint64_t stride) {
    if (stride == 0) {
        if (src != dst)
            HIP_TRY(hipMemcpyAsync(dst, src, sizeof(rt::MtState), hipMemcpyDeviceToDevice, c->rng_stream));
        return RT_OK;
    }
    const unsigned long long* poly = nullptr;
    int rc = dev_poly(c, stride, &poly);
    if (rc) return rc;
    uint32_t* out = dst;
    if (src == dst) {
        if ((rc = c->d_tmp.ensure(rt::MT_N))) return rc;
        out = c->d_tmp.p;
    }
    HIP_TRY(hipMemsetAsync(out, 0, sizeof(rt::MtState), c->rng_stream));
    hipLaunchKernelGGL(rtk::mt_jump_kernel, dim3(1, rtk::JUMP_PARTS), dim3(rtk::JUMP_THREADS), kJumpLds, c->rng_stream,
                       src, out, poly);
    HIP_TRY(hipGetLastError());
    if (out != dst) HIP_TRY(hipMemcpyAsync(dst, out, sizeof(rt::MtState), hipMemcpyDeviceToDevice, c->rng_stream));
    return RT_OK;
}

// states[r] = state at base + r*stride for r in [0,n), states[0] given: doubling tree, one launch per level
int dev_stream_states(rt_ctx* c, uint32_t* states, int n, uint64_t stride) {
    if (n > 1) HIP_TRY(hipMemsetAsync(states + rt::MT_N, 0, size_t(n - 1) * sizeof(rt::MtState), c->rng_stream));
    int levels = 0;
    while ((1 << levels) < n) ++levels;
    // host polynomials in ascending order: each is one squaring of the one before (the tree uses them top down)
    for (int k = 0; k < levels; ++k)
        if (c->d_polys.find(stride << k) == c->d_polys.end()) (void)rt::mt_jump_poly(stride << k);  // may throw
    for (int k = levels - 1; k >= 0; --k) {
        const int step = 1 << k;
        const unsigned long long* poly = nullptr;
        int rc = dev_poly(c, stride << k, &poly);
        if (rc) return rc;
        const int blocks = (n + 2 * step - 1) / (2 * step);
        hipLaunchKernelGGL(rtk::mt_jump_tree_kernel, dim3(blocks, rtk::JUMP_PARTS), dim3(rtk::JUMP_THREADS), kJumpLds,
                           c->rng_stream, states, poly, step, n);
        HIP_TRY(hipGetLastError());
    }
    return RT_OK;
}

// stream states + jitter kernel for one launch, all on rng_stream; d_states[0] must hold the base state
int launch_jitter(rt_ctx* c, uint32_t* d_states, const StreamPlan& pl, double* d_out) {
    int rc = dev_stream_states(c, d_states, pl.n, pl.stride);
    if (rc) return rc;
    hipLaunchKernelGGL(rtk::mt_jitter_kernel, dim3(pl.n), dim3(rtk::JITTER_THREADS), 0, c->rng_stream, d_states, d_out,
                       (unsigned long long)(pl.stride / 2), (unsigned long long)(pl.total_words / 2));
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

// d_state := the generator state at c->pos (it may lag behind while a slot of several frames is consumed).
int sync_state(rt_ctx* c) {
    if (c->state_pos == c->pos) return RT_OK;
    int rc = dev_jump(c, c->d_state.p, c->d_state.p, c->pos - c->state_pos);  // state_pos <= pos always
    if (rc) return rc;
    c->state_pos = c->pos;
    return RT_OK;
}

// The event behind which launch `seq` of this context is known to have finished: the oldest of the three rotating
// out_done events that was recorded behind that launch or a later one (same stream: a later launch's end implies it).
hipEvent_t done_event_for(rt_ctx* c, uint64_t seq) {
    if (c->prev2_out_seq >= seq) return c->prev2_out_done;
    if (c->prev_out_seq >= seq) return c->prev_out_done;
    return c->out_done;
}

// The current scene set is about to be swapped out: its event (which travels with the buffers) is recorded now, behind
// everything launched on it so far, instead of behind every launch.
int retire_front(rt_ctx* c) {
    if (c->front_used) HIP_TRY(hipEventRecord(c->front_last_use, c->stream));
    return RT_OK;
}

// Generate the jitter of `frames` consecutive launches (rows x W x spp samples each, the first one starting `ahead`
// words after the context's current position) into slot `k`, entirely on rng_stream.
int fill_slot(rt_ctx* c, int k, uint64_t ahead, int rows, int W, int spp, int frames = 1, int hidden_n = 0) {
    rt_ctx::JitterSlot& sl = c->slot[k];
    sl.valid = false;
    const size_t n_smp = size_t(rows) * size_t(W) * size_t(spp) * size_t(frames);
    int rc = sl.buf.ensure(n_smp * 2);
    if (rc) return rc;
    const StreamPlan pl = plan_streams(rows, W, spp, frames, hidden_n);
    if ((rc = sl.d_states.ensure(size_t(pl.n) * rt::MT_N))) return rc;
    if ((rc = sl.d_end.ensure(rt::MT_N))) return rc;
    const double t0 = now_ms();
    std::future<void> end_poly;  // (joined by its destructor on every early return)
    // a traversal kernel launched earlier may still be reading this slot's pairs
    if (sl.read_pending) HIP_TRY(hipStreamWaitEvent(c->rng_stream, done_event_for(c, sl.read_seq), 0));
    HIP_TRY(hipEventRecord(sl.t0, c->rng_stream));
    // Steady state of a frame loop: the other slot holds the stream states of a launch of the same
    // shape D words earlier -> every state advances by D in ONE launch (instead of one jump to the
    // start plus a doubling tree of log2(n) dependent launches).
    const rt_ctx::JitterSlot& ot = c->slot[1 - k];
    const uint64_t new_pos = c->pos + ahead;
    sl.states_ok = false;
    if (ot.states_ok && ot.states_n == pl.n && ot.states_stride == pl.stride && new_pos > ot.states_pos &&
        !opt("RT_NO_STATE_REUSE")) {  // (tuning knob)
        const unsigned long long* poly = nullptr;
        if ((rc = dev_poly(c, new_pos - ot.states_pos, &poly))) return rc;
        HIP_TRY(hipMemsetAsync(sl.d_states.p, 0, size_t(pl.n) * sizeof(rt::MtState), c->rng_stream));
        hipLaunchKernelGGL(rtk::mt_jump_all_kernel, dim3(pl.n, rtk::JUMP_PARTS), dim3(rtk::JUMP_THREADS), kJumpLds,
                           c->rng_stream, ot.d_states.p, sl.d_states.p, poly);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(rtk::mt_jitter_kernel, dim3(pl.n), dim3(rtk::JITTER_THREADS), 0, c->rng_stream, sl.d_states.p, sl.buf.p,
                           (unsigned long long)(pl.stride / 2), (unsigned long long)(pl.total_words / 2));
        HIP_TRY(hipGetLastError());
    } else {
        // first fill of a shape: the polynomial of the slot's length (needed last, for the end state) is computed by a
        // helper thread while this thread computes the doubling tree's -- two chains of ~26 squarings mod phi each,
        // ~9 ms apiece: the first frame of a new shape waits ~10 ms for the host instead of ~19
        if (c->d_polys.find(pl.total_words) == c->d_polys.end())
            end_poly = std::async(std::launch::async, [tw = pl.total_words]() {
                try {
                    (void)rt::mt_jump_poly(tw);
                } catch (const std::exception&) {  // (the jump below reports it)
                }
            });
        if ((rc = dev_jump(c, c->d_state.p, sl.d_states.p, ahead + (c->pos - c->state_pos)))) return rc;
        if ((rc = launch_jitter(c, sl.d_states.p, pl, sl.buf.p))) return rc;
    }
    sl.states_ok = true;
    sl.states_pos = new_pos;
    sl.states_stride = pl.stride;
    sl.states_n = pl.n;
    HIP_TRY(hipEventRecord(sl.t1, c->rng_stream));
    HIP_TRY(hipEventRecord(sl.done, c->rng_stream));
    // the state at the slot's end is only needed when the slot's last frame has been handed out (a copy on this
    // stream): its jump goes BEHIND the pairs, off the chain the traversal waits for
    if (end_poly.valid()) end_poly.wait();
    if ((rc = dev_jump(c, sl.d_states.p, sl.d_end.p, pl.total_words))) return rc;
    sl.host_ms = now_ms() - t0;
    sl.pos = c->pos + ahead;
    sl.rows = rows;
    sl.W = W;
    sl.spp = spp;
    sl.frames = frames;
    sl.valid = true;
    return RT_OK;
}

// host arrays of a built scene -> one set of device buffers, on `st`; waits for the copies (the host
// vectors are pageable and their owner may free them right away)
int copy_scene_to(rt_ctx::BackScene& bk, const rt::Scene& s, hipStream_t st) {
    int rc;
    if ((rc = bk.nodes.ensure(s.dev_nodes.size()))) return rc;
    if ((rc = bk.tlas.ensure(s.dev_tlas.size()))) return rc;
    if ((rc = bk.slot_root.ensure(s.slot_root.size()))) return rc;
    if ((rc = bk.leaf_ext.ensure(std::max<size_t>(s.leaf_ext.size(), 2)))) return rc;
    if ((rc = bk.tris.ensure(s.tris.size()))) return rc;
    if ((rc = bk.colors.ensure(s.colors.size()))) return rc;
    HIP_TRY(hipMemcpyAsync(bk.tlas.p, s.dev_tlas.data(), s.dev_tlas.size() * sizeof(rt::DevNode), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(bk.slot_root.p, s.slot_root.data(), s.slot_root.size() * sizeof(int32_t), hipMemcpyHostToDevice,
                           st));
    if (!s.leaf_ext.empty())
        HIP_TRY(hipMemcpyAsync(bk.leaf_ext.p, s.leaf_ext.data(), s.leaf_ext.size() * sizeof(uint32_t), hipMemcpyHostToDevice,
                               st));
    // The three big arrays.  A copy from pageable memory is staged through the runtime's own pinned buffers by the
    // thread that calls it -- one thread moves ~20 GiB/s on the GPU box, four ~31 (profiles/tools/h2d_threads.py) -- so
    // a large scene's bytes are cut into four spans, three of them copied by helper threads on streams of their own.
    struct Span {
        char* dst;
        const char* src;
        size_t bytes;
    };
    const Span big[3] = {{reinterpret_cast<char*>(bk.nodes.p), reinterpret_cast<const char*>(s.dev_nodes.data()),
                          s.dev_nodes.size() * sizeof(rt::DevNode)},
                         {reinterpret_cast<char*>(bk.tris.p), reinterpret_cast<const char*>(s.tris.data()),
                          s.tris.size() * sizeof(double)},
                         {reinterpret_cast<char*>(bk.colors.p), reinterpret_cast<const char*>(s.colors.data()),
                          s.colors.size() * sizeof(double)}};
    const size_t total = big[0].bytes + big[1].bytes + big[2].bytes;
    const size_t par_min = opt("RT_UPLOAD_PAR_MIN") ? (size_t)std::atoll(opt("RT_UPLOAD_PAR_MIN")) : (size_t(64) << 20);
    constexpr int kParts = 4;
    if (total < par_min) {
        for (const Span& b : big) HIP_TRY(hipMemcpyAsync(b.dst, b.src, b.bytes, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        return RT_OK;
    }
    std::vector<Span> part[kParts];
    {
        const size_t per = ((total + kParts - 1) / kParts + 4095) & ~size_t(4095);
        int k = 0;
        size_t room = per;
        for (const Span& b : big) {
            size_t off = 0;
            while (off < b.bytes) {
                const size_t n = std::min(b.bytes - off, room);
                part[k].push_back({b.dst + off, b.src + off, n});
                off += n;
                room -= n;
                if (room == 0 && k + 1 < kParts) {
                    ++k;
                    room = per;
                }
            }
        }
    }
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipError_t err[kParts];
    auto run = [&](int k, hipStream_t q) {
        err[k] = hipSuccess;
        for (const Span& sp : part[k])
            if (err[k] == hipSuccess) err[k] = hipMemcpyAsync(sp.dst, sp.src, sp.bytes, hipMemcpyHostToDevice, q);
        const hipError_t e = hipStreamSynchronize(q);
        if (err[k] == hipSuccess) err[k] = e;
    };
    std::thread helpers[kParts - 1];
    for (int k = 1; k < kParts; ++k)
        helpers[k - 1] = std::thread([&, k]() {
            hipStream_t q = nullptr;
            err[k] = hipSetDevice(dev);
            if (err[k] == hipSuccess) err[k] = hipStreamCreateWithFlags(&q, hipStreamNonBlocking);
            if (err[k] != hipSuccess) return;
            run(k, q);
            (void)hipStreamDestroy(q);
        });
    run(0, st);
    for (auto& h : helpers) h.join();
    for (int k = 0; k < kParts; ++k) HIP_TRY(err[k]);
    return RT_OK;
}

void swap_sets(rt_ctx* c, rt_ctx::BackScene& bk) {
    std::swap(c->nodes, bk.nodes);
    std::swap(c->tlas, bk.tlas);
    std::swap(c->slot_root, bk.slot_root);
    std::swap(c->leaf_ext, bk.leaf_ext);
    std::swap(c->tris, bk.tris);
    std::swap(c->colors, bk.colors);
}

// Called on a helper thread: copy `s` into a free staged set.  Returns the set's index, or -1 when no
// set is free or a HIP call failed (the caller then falls back to rt_upload_scene on its own thread).
int stage_scene(rt_ctx* c, const rt::Scene& s) {
    rt_ctx::StagedSet* st = nullptr;
    int idx = -1;
    {
        std::lock_guard<std::mutex> lk(c->staged_mu);
        for (size_t i = 0; i < c->staged.size(); ++i)
            if (c->staged[i]->state == 0) {
                st = c->staged[i].get();
                st->state = 1;
                idx = (int)i;
                break;
            }
    }
    if (!st) return -1;
    auto give_up = [&]() {
        std::lock_guard<std::mutex> lk(c->staged_mu);
        st->state = 0;
        return -1;
    };
    if (hipSetDevice(c->device) != hipSuccess) return give_up();
    // these buffers may have been the current set until recently: wait for the last render that read them
    if (st->used && hipEventSynchronize(st->last_use) != hipSuccess) return give_up();
    st->used = false;
    const double t_copy0 = now_ms();
    if (copy_scene_to(st->b, s, st->stream) != RT_OK) return give_up();
    if (opt("RT_BUILD_TIMING"))
        std::fprintf(stderr, "stage_scene: %.1f MB copied to the device in %.2f ms\n",
                     (s.dev_nodes.size() * sizeof(rt::DevNode) + s.tris.size() * 8 + s.colors.size() * 8) / 1e6,
                     now_ms() - t_copy0);
    st->n_nodes = s.dev_nodes.size();
    st->n_tlas = s.dev_tlas.size();
    st->n_tris = (int64_t)s.prim_mesh.size();
    st->stack_need = s.stack_need;
    st->max_abs_coord = s.max_abs_coord;
    std::lock_guard<std::mutex> lk(c->staged_mu);
    st->state = 2;
    return idx;
}

// Calling thread: make staged set `idx` the current scene (pointer swaps only); the set keeps the
// previous scene's buffers, guarded by the event of the last render that used them.
void activate_staged(rt_ctx* c, int idx) {
    rt_ctx::StagedSet* st = c->staged[(size_t)idx].get();
    (void)retire_front(c);
    swap_sets(c, st->b);
    std::swap(c->front_last_use, st->last_use);
    std::swap(c->front_used, st->used);
    c->front_used = false;
    c->n_nodes = st->n_nodes;
    c->n_tlas = st->n_tlas;
    c->n_tris = st->n_tris;
    c->stack_need = st->stack_need;
    c->max_abs_coord = st->max_abs_coord;
    c->have_scene = true;
    std::lock_guard<std::mutex> lk(c->staged_mu);
    st->state = 0;
}

int ensure_staged_pool(rt_ctx* c, size_t n) {
    std::lock_guard<std::mutex> lk(c->staged_mu);
    while (c->staged.size() < n) {
        std::unique_ptr<rt_ctx::StagedSet> st(new rt_ctx::StagedSet());
        // RT_STAGE_STREAMS=1: one stream per set; default: the first NSTAGE_STREAMS sets' streams are shared by the
        // rest -- the number of HIP streams of a context then does not grow with RT_BUILD_AHEAD
        const size_t n_own = opt("RT_STAGE_STREAMS") ? ~size_t(0) : size_t(1);
        if (c->staged.size() < n_own) {
            HIP_TRY(hipStreamCreateWithFlags(&st->stream, hipStreamNonBlocking));
            st->owns_stream = true;
        } else {
            st->stream = c->staged[c->staged.size() % n_own]->stream;
        }
        HIP_TRY(hipEventCreateWithFlags(&st->last_use, hipEventDisableTiming));
        c->staged.push_back(std::move(st));
    }
    return RT_OK;
}

std::mutex g_global_mu;
rt_ctx* g_global_ctx[64] = {};

// A few persistent helper threads for the frame driver (one build per timestep, one file per frame): starting a
// thread per task (std::async) is 30-50 us each, twice per frame -- most of the host's time per frame of a job of
// small frames.  Tasks run in submission order; the threads live until the process ends.
class TaskThreads {
public:
    explicit TaskThreads(int n) {
        for (int i = 0; i < n; ++i)
            threads_.emplace_back([this] {
                for (;;) {
                    std::function<void()> job;
                    {
                        std::unique_lock<std::mutex> lk(mu_);
                        cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
                        if (queue_.empty()) return;  // (stop_)
                        job = std::move(queue_.front());
                        queue_.pop_front();
                    }
                    job();
                }
            });
    }
    ~TaskThreads() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    template <class F>
    auto submit(F f) -> std::future<decltype(f())> {
        auto task = std::make_shared<std::packaged_task<decltype(f())()>>(std::move(f));
        auto fut = task->get_future();
        {
            std::lock_guard<std::mutex> lk(mu_);
            queue_.emplace_back([task] { (*task)(); });
        }
        cv_.notify_one();
        return fut;
    }

private:
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::function<void()>> queue_;
    bool stop_ = false;
};
TaskThreads& build_threads() {
    static TaskThreads t(8);
    return t;
}
TaskThreads& writer_thread() {
    static TaskThreads t(1);
    return t;
}

// Device address of `scratch` on each GPU (every GPU holds its own copy of the array at its own address).
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
unsigned char* g_scratch_ptr[kMaxDevices] = {};
// Who launched last into each GPU's scratch array (rt_render_static): a call from ANOTHER context while that launch
// has not finished would overwrite the records its kernels are reading -- it is refused with RT_E_BUSY instead.
rt_ctx* g_scratch_owner[kMaxDevices] = {};

// Resolve (once per device, with that device current) the address of this device's copy of `scratch`.
int static_scratch_ptr(int device, unsigned char** out) {
    if (device < 0 || device >= kMaxDevices) return fail(RT_E_INVALID, "static scratch: device index out of range");
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        HIP_TRY(hipSetDevice(device));
        void* ptr = nullptr;
        HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return fail(RT_E_NO_DEVICE, "hipGetSymbolAddress(scratch) returned a null address");
        g_scratch_ptr[device] = static_cast<unsigned char*>(ptr);
    }
    *out = g_scratch_ptr[device];
    return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_last_error(void) { return g_err.c_str(); }
const char* rt_version(void) { return "rtgpu 0.1 (gfx950, fp64 parity path)"; }

int rt_host_threads(void) { return rt::host_threads(); }
const char* rt_option_names(void) {
    static const std::string joined = [] {
        std::string j;
        for (int i = 0; i < kNumOptions; ++i) j += std::string(i ? "," : "") + kOptionNames[i];
        return j;
    }();
    return joined.c_str();
}
int rt_reload_options(void) {
    std::lock_guard<std::mutex> lk(g_options.mu);
    g_options.load();
    return RT_OK;
}
int rt_mt_phi(uint64_t* out312, int recompute) {
    if (!out312) return fail(RT_E_INVALID, "rt_mt_phi: null argument");
    try {
        rt::mt_phi_words(out312, recompute != 0);
    } catch (const std::exception& e) {
        return fail(RT_E_INVALID, std::string("rt_mt_phi: ") + e.what());
    }
    return RT_OK;
}

int64_t rt_partition_selftest(int64_t cases, uint64_t seed, int max_n) {
    try {
        return rt::partition_selftest(cases, seed, max_n);
    } catch (const std::exception& e) {
        return fail(RT_E_INVALID, e.what());
    }
}

void* rt_host_alloc(size_t bytes) {
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        g_err = std::string("rt_host_alloc: hipHostMalloc failed: ") + hipGetErrorString(e);
        return nullptr;
    }
    return p;
}

void rt_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

/* ------------------------------------------------------------------ host scene */
rt_scene* rt_scene_build(int n_meshes, const double* const* coords, const double* const* colors,
                         const int64_t* elem_counts, int timestep, int n_threads) {
    try {
        if (!coords || !colors || !elem_counts) {
            g_err = "rt_scene_build: null argument";
            return nullptr;
        }
        std::unique_ptr<rt_scene> h(new rt_scene());
        // n_threads < 0: the frame driver's mode (several builds in flight, one thread per tree)
        h->s.reset(rt::build_scene(n_meshes, coords, colors, elem_counts, timestep, std::max(n_threads, 0), n_threads < 0));
        return h.release();
    } catch (const std::exception& e) {
        g_err = std::string("rt_scene_build: ") + e.what();
        return nullptr;
    }
}

void rt_scene_free(rt_scene* scene) { delete scene; }

int rt_scene_info(const rt_scene* scene, rt_scene_info_t* out) {
    if (!scene || !out) return fail(RT_E_INVALID, "rt_scene_info: null argument");
    const rt::Scene& s = *scene->s;
    out->n_meshes = (int32_t)s.slots.size();
    out->n_tlas_nodes = (int32_t)s.tlas.size();
    out->n_blas_nodes = s.n_blas_nodes;
    out->n_tris = (int64_t)s.prim_mesh.size();
    out->n_dev_nodes = (int64_t)s.dev_nodes.size();
    out->stack_need = s.stack_need;
    out->max_leaf_tris = s.max_leaf_tris;
    out->midpoint_fallbacks = s.midpoint_fallbacks;
    out->build_seconds = s.build_seconds;
    return RT_OK;
}

int rt_scene_tlas(const rt_scene* scene, double* boxes6, int32_t* blas_count, int32_t* left, int32_t* min_blas_idx) {
    if (!scene) return fail(RT_E_INVALID, "rt_scene_tlas: null scene");
    const rt::Scene& s = *scene->s;
    for (size_t i = 0; i < s.tlas.size(); ++i) {
        if (boxes6) {
            std::memcpy(boxes6 + 6 * i, s.tlas[i].mn, 3 * sizeof(double));
            std::memcpy(boxes6 + 6 * i + 3, s.tlas[i].mx, 3 * sizeof(double));
        }
        if (blas_count) blas_count[i] = (int32_t)s.tlas[i].count;
        if (left) left[i] = s.tlas[i].left;
        if (min_blas_idx) min_blas_idx[i] = s.tlas[i].first;
    }
    return RT_OK;
}

int rt_scene_slot_info(const rt_scene* scene, int slot, int32_t* mesh, int32_t* n_nodes, int32_t* n_elems,
                       int64_t* tri_base) {
    if (!scene) return fail(RT_E_INVALID, "rt_scene_slot_info: null scene");
    const rt::Scene& s = *scene->s;
    if (slot < 0 || slot >= (int)s.slots.size()) return fail(RT_E_INVALID, "rt_scene_slot_info: bad slot");
    if (mesh) *mesh = s.slots[slot].mesh;
    if (n_nodes) *n_nodes = (int32_t)s.slots[slot].nodes.size();
    if (n_elems) *n_elems = (int32_t)s.slots[slot].sorted_to_elem.size();
    if (tri_base) *tri_base = s.slot_tri_base[slot];
    return RT_OK;
}

int64_t rt_scene_slot_dev_root(const rt_scene* scene, int slot) {
    if (!scene) return fail(RT_E_INVALID, "rt_scene_slot_dev_root: null scene");
    const rt::Scene& s = *scene->s;
    if (slot < 0 || slot >= (int)s.slots.size()) return fail(RT_E_INVALID, "rt_scene_slot_dev_root: bad slot");
    return s.slot_root[(size_t)slot];
}

int rt_scene_slot_nodes(const rt_scene* scene, int slot, double* boxes6, int32_t* left, int64_t* count,
                        int32_t* min_idx) {
    if (!scene) return fail(RT_E_INVALID, "rt_scene_slot_nodes: null scene");
    const rt::Scene& s = *scene->s;
    if (slot < 0 || slot >= (int)s.slots.size()) return fail(RT_E_INVALID, "rt_scene_slot_nodes: bad slot");
    const auto& nodes = s.slots[slot].nodes;
    for (size_t i = 0; i < nodes.size(); ++i) {
        if (boxes6) {
            std::memcpy(boxes6 + 6 * i, nodes[i].mn, 3 * sizeof(double));
            std::memcpy(boxes6 + 6 * i + 3, nodes[i].mx, 3 * sizeof(double));
        }
        if (left) left[i] = nodes[i].left;
        if (count) count[i] = nodes[i].count;
        if (min_idx) min_idx[i] = nodes[i].first;
    }
    return RT_OK;
}

int rt_scene_slot_indices(const rt_scene* scene, int slot, int32_t* sorted_to_elem) {
    if (!scene || !sorted_to_elem) return fail(RT_E_INVALID, "rt_scene_slot_indices: null argument");
    const rt::Scene& s = *scene->s;
    if (slot < 0 || slot >= (int)s.slots.size()) return fail(RT_E_INVALID, "rt_scene_slot_indices: bad slot");
    const auto& v = s.slots[slot].sorted_to_elem;
    std::memcpy(sorted_to_elem, v.data(), v.size() * sizeof(int32_t));
    return RT_OK;
}

int rt_scene_prim_tables(const rt_scene* scene, int32_t* prim_mesh, int32_t* prim_elem) {
    if (!scene) return fail(RT_E_INVALID, "rt_scene_prim_tables: null scene");
    const rt::Scene& s = *scene->s;
    if (prim_mesh) std::memcpy(prim_mesh, s.prim_mesh.data(), s.prim_mesh.size() * sizeof(int32_t));
    if (prim_elem) std::memcpy(prim_elem, s.prim_elem.data(), s.prim_elem.size() * sizeof(int32_t));
    return RT_OK;
}

/* ------------------------------------------------------------------ sampler (host part) */
int rt_mt19937_state_at(uint32_t seed, uint64_t word_position, uint32_t* state624) {
    if (!state624) return fail(RT_E_INVALID, "rt_mt19937_state_at: null output");
    try {
        rt::MtState s, o;
        rt::mt_seed(s, seed);
        rt::mt_apply(rt::mt_jump_poly(word_position), s, o);
        std::memcpy(state624, o.w, sizeof(o.w));
        return RT_OK;
    } catch (const std::exception& e) {
        return fail(RT_E_INVALID, e.what());
    }
}

int rt_mt19937_stream_states(uint32_t seed, uint64_t base_position, uint64_t stride, int n, uint32_t* states) {
    if (!states || n < 1) return fail(RT_E_INVALID, "rt_mt19937_stream_states: bad argument");
    try {
        rt::MtState s, b;
        rt::mt_seed(s, seed);
        rt::mt_apply(rt::mt_jump_poly(base_position), s, b);
        std::vector<rt::MtState> out;
        rt::mt_stream_states(b, stride, n, rt::host_threads(), out);
        std::memcpy(states, out.data(), size_t(n) * sizeof(rt::MtState));
        return RT_OK;
    } catch (const std::exception& e) {
        return fail(RT_E_INVALID, e.what());
    }
}

/* ------------------------------------------------------------------ context */
rt_ctx* rt_create(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_err = "rt_create: no HIP device available (this library has no CPU fallback)";
        return nullptr;
    }
    if (device < 0 || device >= n) {
        g_err = "rt_create: device index out of range";
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        g_err = "rt_create: hipSetDevice failed";
        return nullptr;
    }
    std::unique_ptr<rt_ctx> c(new rt_ctx());
    c->device = device;
    // the sampler stream gets the highest priority: its small kernels (jumps, jitter of the NEXT frame)
    // must be dispatched between the workgroups of the long traversal kernel, not after it
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->front_last_use, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->back_last_use, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->out_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->prev_out_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->prev2_out_done, hipEventDisableTiming) != hipSuccess) {
        g_err = "rt_create: copy stream / events";
        return nullptr;
    }
    const char* pe = opt("RT_RNG_PRIO");  // tuning knob: "lo" = same priority as the traversal stream
    const int prio_rng = (pe && pe[0] == 'l') ? prio_lo : prio_hi;
    if (hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_lo) != hipSuccess ||
        hipStreamCreateWithPriority(&c->rng_stream, hipStreamNonBlocking, prio_rng) != hipSuccess) {
        g_err = "rt_create: hipStreamCreate failed";
        return nullptr;
    }
    // the prep + cut kernels of the NEXT launch likewise go between the workgroups of the running traversal kernel
    if (hipStreamCreateWithPriority(&c->prep_stream, hipStreamNonBlocking, prio_hi) != hipSuccess) {
        g_err = "rt_create: hipStreamCreate failed";
        return nullptr;
    }
    for (auto& cs : c->cam)
        if (hipEventCreateWithFlags(&cs.ready, hipEventDisableTiming) != hipSuccess) {
            g_err = "rt_create: hipEventCreate failed";
            return nullptr;
        }
    for (auto& ev : c->ev)
        if (hipEventCreate(&ev) != hipSuccess) {
            g_err = "rt_create: hipEventCreate failed";
            return nullptr;
        }
    for (auto& sl : c->slot)
        if (hipEventCreate(&sl.done) != hipSuccess || hipEventCreate(&sl.t0) != hipSuccess ||
            hipEventCreate(&sl.t1) != hipSuccess) {
            g_err = "rt_create: hipEventCreate failed";
            return nullptr;
        }
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(rtk::mt_jump_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)kJumpLds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(rtk::mt_jump_tree_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)kJumpLds) != hipSuccess) {
        g_err = "rt_create: cannot reserve LDS for the jump kernels";
        return nullptr;
    }
    if (c->d_state.ensure(rt::MT_N) != RT_OK) return nullptr;
    // rt_render_static must not call anything that could allocate: its scratch address is resolved here
    if (static_scratch_ptr(device, &c->static_scratch) != RT_OK) return nullptr;
    if (hipEventCreateWithFlags(&c->static_done, hipEventDisableTiming) != hipSuccess) {
        g_err = "rt_create: event for the static scratch array";
        return nullptr;
    }
    (void)opt("RT_NO_CUT");  // the RT_* options are taken from the environment here, once (see OptionTable)
    rt_ctx* raw = c.release();
    if (rt_rng_seed(raw, 5489u) != RT_OK) {
        rt_destroy(raw);
        return nullptr;
    }
    return raw;
}

void rt_destroy(rt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    {
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        if (c->device >= 0 && c->device < kMaxDevices && g_scratch_owner[c->device] == c) g_scratch_owner[c->device] = nullptr;
    }
    if (c->static_done) (void)hipEventDestroy(c->static_done);
    if (c->rng_stream) (void)hipStreamSynchronize(c->rng_stream);
    c->nodes.release();
    c->tlas.release();
    c->slot_root.release();
    c->leaf_ext.release();
    c->tris.release();
    c->colors.release();
    if (c->adapt_job.valid()) c->adapt_job.wait();
    if (c->prep_stream) (void)hipStreamSynchronize(c->prep_stream);
    for (auto& cs : c->cam) {
        cs.nodes.release();
        cs.tlas.release();
        cs.tris.release();
        cs.cuts.release();
        if (cs.ready) (void)hipEventDestroy(cs.ready);
    }
    c->nodes32.release();
    c->tlas32.release();
    c->tris32.release();
    c->frame_params_buf[0].release();
    c->frame_params_buf[1].release();
    c->d_states.release();
    c->jitter.release();
    c->d_state.release();
    c->d_tmp.release();
    for (auto& kv : c->d_polys) (void)hipFree(kv.second);
    c->d_polys.clear();
    for (auto& sl : c->slot) {
        sl.buf.release();
        sl.d_states.release();
        sl.d_end.release();
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.t0) (void)hipEventDestroy(sl.t0);
        if (sl.t1) (void)hipEventDestroy(sl.t1);
    }
    c->frame_intensity.release();
    c->frame_hit_t.release();
    c->frame_rgb.release();
    c->frame_hit_prim.release();
    c->intensity.release();
    c->hit_t.release();
    c->rgb.release();
    c->hit_prim.release();
    c->counters.release();
    c->gstack.release();
    c->tr_o.release();
    c->tr_d.release();
    c->tr_t.release();
    c->tr_b.release();
    c->tr_c.release();
    c->tr_p.release();
    for (auto& ev : c->ev)
        if (ev) (void)hipEventDestroy(ev);
    for (auto& st : c->staged) {
        st->b.nodes.release();
        st->b.tlas.release();
        st->b.slot_root.release();
        st->b.leaf_ext.release();
        st->b.tris.release();
        st->b.colors.release();
        if (st->last_use) (void)hipEventDestroy(st->last_use);
        if (st->stream && st->owns_stream) (void)hipStreamDestroy(st->stream);
    }
    c->staged.clear();
    c->back.nodes.release();
    c->back.tlas.release();
    c->back.slot_root.release();
    c->back.leaf_ext.release();
    c->back.tris.release();
    c->back.colors.release();
    if (c->out_done) (void)hipEventDestroy(c->out_done);
    if (c->prev_out_done) (void)hipEventDestroy(c->prev_out_done);
    if (c->prev2_out_done) (void)hipEventDestroy(c->prev2_out_done);
    c->prev_intensity.release();
    c->prev_rgb.release();
    c->prev2_intensity.release();
    c->prev2_rgb.release();
    if (c->front_last_use) (void)hipEventDestroy(c->front_last_use);
    if (c->back_last_use) (void)hipEventDestroy(c->back_last_use);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->rng_stream) (void)hipStreamDestroy(c->rng_stream);
    if (c->prep_stream) (void)hipStreamDestroy(c->prep_stream);
    delete c;
}

int rt_sync(rt_ctx* c) {
    if (!c) return fail(RT_E_INVALID, "rt_sync: null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->rng_stream));
    HIP_TRY(hipStreamSynchronize(c->prep_stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_upload_scene(rt_ctx* c, const rt_scene* scene) {
    if (!c || !scene) return fail(RT_E_INVALID, "rt_upload_scene: null argument");
    const rt::Scene& s = *scene->s;
    HIP_TRY(hipSetDevice(c->device));
    // The upload goes into the back set on its own stream, so a render launched before this call keeps
    // running on the current set.  The back set was the current one two uploads ago: wait for the last
    // render that used it.
    if (c->back_used) HIP_TRY(hipEventSynchronize(c->back_last_use));
    int rc;
    if ((rc = copy_scene_to(c->back, s, c->copy_stream))) return rc;  // waits for the copies only
    if ((rc = retire_front(c))) return rc;
    swap_sets(c, c->back);
    std::swap(c->front_last_use, c->back_last_use);
    std::swap(c->front_used, c->back_used);
    c->front_used = false;
    c->n_nodes = s.dev_nodes.size();
    c->n_tlas = s.dev_tlas.size();
    c->stack_need = s.stack_need;
    c->max_abs_coord = s.max_abs_coord;
    c->n_tris = (int64_t)s.prim_mesh.size();
    c->have_scene = true;
    return RT_OK;
}

/* ------------------------------------------------------------------ sampler (context part) */
int rt_rng_seed(rt_ctx* c, uint32_t seed) {
    if (!c) return fail(RT_E_INVALID, "rt_rng_seed: null context");
    HIP_TRY(hipSetDevice(c->device));
    // the previous upload of h_seed_state must have been consumed before it is overwritten
    HIP_TRY(hipStreamSynchronize(c->rng_stream));
    c->seed = seed;
    rt::mt_seed(c->h_seed_state, seed);
    HIP_TRY(hipMemcpyAsync(c->d_state.p, c->h_seed_state.w, sizeof(rt::MtState), hipMemcpyHostToDevice, c->rng_stream));
    c->pos = 0;
    c->state_pos = 0;
    for (auto& sl : c->slot) sl.valid = sl.states_ok = false;  // slots belong to the old seed's stream
    c->have_last_start = false;
    return RT_OK;
}

int rt_rng_seek(rt_ctx* c, uint64_t word_position) {
    if (!c) return fail(RT_E_INVALID, "rt_rng_seek: null context");
    if (word_position == c->pos) return RT_OK;
    HIP_TRY(hipSetDevice(c->device));
    try {
        int rc;
        if (word_position > c->pos) {
            if ((rc = dev_jump(c, c->d_state.p, c->d_state.p, word_position - c->state_pos))) return rc;
        } else {
            HIP_TRY(hipStreamSynchronize(c->rng_stream));
            rt::mt_seed(c->h_seed_state, c->seed);
            HIP_TRY(hipMemcpyAsync(c->d_state.p, c->h_seed_state.w, sizeof(rt::MtState), hipMemcpyHostToDevice,
                                   c->rng_stream));
            if ((rc = dev_jump(c, c->d_state.p, c->d_state.p, word_position))) return rc;
        }
    } catch (const std::exception& e) {
        return fail(RT_E_INVALID, e.what());
    }
    // slots prefetched relative to other positions stay valid: they are keyed by absolute position
    c->pos = word_position;
    c->state_pos = word_position;
    return RT_OK;
}

uint64_t rt_rng_tell(const rt_ctx* c) { return c ? c->pos : 0; }

int rt_rng_jitter(rt_ctx* c, int64_t n, int n_streams, double* out_host) {
    if (!c || !out_host || n < 1 || n_streams < 1) return fail(RT_E_INVALID, "rt_rng_jitter: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    try {
        StreamPlan pl;
        const uint64_t per = (uint64_t(n) + uint64_t(n_streams) - 1) / uint64_t(n_streams);
        pl.stride = per * 2;
        pl.n = int((uint64_t(n) + per - 1) / per);
        pl.total_words = uint64_t(n) * 2;
        int rc = c->jitter.ensure(size_t(n));
        if (rc) return rc;
        if ((rc = c->d_states.ensure(size_t(pl.n) * rt::MT_N))) return rc;
        if ((rc = sync_state(c))) return rc;
        if ((rc = dev_jump(c, c->d_state.p, c->d_states.p, 0))) return rc;
        if ((rc = launch_jitter(c, c->d_states.p, pl, c->jitter.p))) return rc;
        if ((rc = dev_jump(c, c->d_state.p, c->d_state.p, pl.total_words))) return rc;
        HIP_TRY(hipMemcpyAsync(out_host, c->jitter.p, size_t(n) * sizeof(double), hipMemcpyDeviceToHost, c->rng_stream));
        HIP_TRY(hipStreamSynchronize(c->rng_stream));
        c->pos += pl.total_words;
        c->state_pos = c->pos;
    } catch (const std::exception& e) {
        return fail(RT_E_INVALID, e.what());
    }
    return RT_OK;
}

/* Test seam: the device-resident generator state (624 raw words preceding the current position). */
int rt_rng_state(rt_ctx* c, uint32_t* state624) {
    if (!c || !state624) return fail(RT_E_INVALID, "rt_rng_state: null argument");
    HIP_TRY(hipSetDevice(c->device));
    try {
        int rc = sync_state(c);
        if (rc) return rc;
    } catch (const std::exception& e) {
        return fail(RT_E_INVALID, e.what());
    }
    HIP_TRY(hipMemcpyAsync(state624, c->d_state.p, sizeof(rt::MtState), hipMemcpyDeviceToHost, c->rng_stream));
    HIP_TRY(hipStreamSynchronize(c->rng_stream));
    return RT_OK;
}

/* ------------------------------------------------------------------ render */
}  // extern "C"

namespace {

// Tile side (log2) of a traversal launch.  >= 256 rays per 8x8 tile from 4 spp on.
// Small images (the reference's own 400x225 frames): one workgroup per 8x8 tile leaves most of the 2048
// workgroup slots of the chip empty.  The packet kernels take any tile size, so halve the tile while there
// are fewer than 8192 tiles and a wave (a quarter of the tile) still has >= 128 rays to walk.
int pick_tile_log2(bool packet_like, int spp, int W, int rows) {
    int tile_log2 = (spp >= 4) ? 3 : 4;
    const bool small_tiles = opt("RT_NO_SMALL_TILES") == nullptr;
    if (small_tiles && packet_like)
        while (tile_log2 > 1 && int64_t((W + (1 << tile_log2) - 1) >> tile_log2) * ((rows + (1 << tile_log2) - 1) >> tile_log2) < 8192 &&
               (int64_t(1) << (2 * (tile_log2 - 1) - 2)) * spp >= 128)
            --tile_log2;
    if (const char* e = opt("RT_TILE_LOG2")) {  // tuning knob (packet kernels: 1..4, per-lane kernel: 3 or 4)
        const int v = std::atoi(e);
        if (v >= (packet_like ? 1 : 3) && v <= 4) tile_log2 = v;
    }
    return tile_log2;
}

// Device memory one packet launch touches besides the uploaded scene: the camera-relative records the prep
// kernels write (scratch), the jitter pairs (input) and the outputs.  rt_render passes the context's own
// buffers, rt_render_static a partition of the file-scope scratch array and the caller's pointers.
struct PacketBuffers {
    rt::DevNode* cam_nodes;
    rt::DevNode* cam_tlas;
    rtk::CamTri* cam_tris;
    const double* jitter;
    double* intensity;
    uint8_t* rgb;
    int32_t* hit_prim;
    double* hit_t;
    unsigned long long* counters;
    uint32_t* cuts;  // plan_cuts(...).total_words words for the cut levels, or null: every batch walks from the root
};

// Camera-relative node / TLAS / triangle records of a frame in one launch (tris_out = null: boxes only).
void launch_prep(rt_ctx* c, hipStream_t st, rt::DevNode* cam_nodes, rt::DevNode* cam_tlas, rtk::CamTri* cam_tris, double ox,
                 double oy, double oz) {
    rtk::PrepArgs pa;
    pa.nodes_in = c->nodes.p;
    pa.nodes_out = cam_nodes;
    pa.n_nodes = (long long)c->n_nodes;
    pa.tlas_in = c->tlas.p;
    pa.tlas_out = cam_tlas;
    pa.n_tlas = (long long)c->n_tlas;
    pa.tris = c->tris.p;
    pa.tris_out = cam_tris;
    pa.n_tris = cam_tris ? (long long)c->n_tris : 0;
    pa.ox = ox;
    pa.oy = oy;
    pa.oz = oz;
    pa.nb_nodes = (unsigned)((4 * c->n_nodes + 255) / 256);
    pa.nb_tlas = (unsigned)((4 * c->n_tlas + 255) / 256);
    const unsigned nb_tris = (unsigned)((pa.n_tris + 255) / 256);
    hipLaunchKernelGGL(rtk::prep_frame_kernel, dim3(pa.nb_nodes + pa.nb_tlas + nb_tris), dim3(256), 0, st, pa);
}

// What a render kernel is told about the cuts of its launch (rtk::PacketArgs / rtk32::Packet32Args).
struct CutTables {
    const uint32_t* lvl[rtk::CUT_MAX_LEVELS] = {};
    int32_t tx[rtk::CUT_MAX_LEVELS] = {};
    int32_t eff[rtk::CUT_MAX_LEVELS] = {0, 1, 2, 3, 4};
    int32_t lo = -1;  // < 0: no cuts, every batch walks from the root
};

// The cut kernels of one launch (cut_level_kernel, level after level) on stream `st`, from the camera-relative boxes
// the prep kernels have written.  `allowed`: not with visit counters (they count the reference's visit rule), not
// without min <= max boxes and tame coordinates (the cut's argument rests on the NaN-free slab test).  Allocates
// nothing; `cuts` = plan_cuts(...).total_words words, or null.
int make_cuts(rt_ctx* c, hipStream_t st, const rt_render_params* p, int tile_log2, const rtk::FrameParams& fp,
              const rt::DevNode* cam_nodes, const rt::DevNode* cam_tlas, const rtk::CamTri* cam_tris, uint32_t* cuts,
              bool allowed, CutTables& out) {
    out = CutTables{};
    const int W = p->image_width, spp = p->samples, rows = p->row_end - p->row_begin;
    const bool no_cut = opt("RT_NO_CUT") != nullptr;  // (comparison / test knob, read per launch)
    // The cut kernels are chains of dependent steps (~0.3 ms however small the image) and save a third to a half of
    // the traversal.  Measured: 2048^2 x 4 spp (16.7 M rays, 1 M triangles) 1.47 -> 1.32 ms per frame with cuts,
    // 400 x 225 x 50 spp (4.5 M rays) 0.30 -> 0.49 ms.
    if (!cuts || !allowed || no_cut || cut_mode(W, rows, spp, tile_log2, c->n_tris) == 0) return RT_OK;
    const CutPlan cp = plan_cuts(W, rows, spp, tile_log2, c->n_tris);
    rtk::CutArgs ca;
    ca.nodes = cam_nodes;
    ca.tlas = cam_tlas;
    ca.slot_root = c->slot_root.p;
    ca.fp = fp;
    ca.W = W;
    ca.rows = rows;
    ca.row0 = p->row_begin;
    ca.parent = nullptr;
    ca.parent_tiles_x = 0;
    ca.mag = rtk::cut_magnitude(ca.fp, W, p->row_begin, rows);
    if (cp.single && !opt("RT_CUT1_SERIAL")) {
        // one level straight from the root, several lanes per block (cut_single_kernel)
        const int l = cp.lo;
        ca.lvl = l;
        ca.parent_shift = 0;
        ca.cap = cut_cap(l, cp);
        ca.tiles_x = cp.tiles_x[l];
        ca.n_tiles = (int32_t)cp.tiles[l];
        ca.cuts = cuts + cp.offset_words[l];
        ca.tris = opt("RT_NO_TRIM") ? nullptr : cam_tris;
        constexpr int per_wave = 64 / rtk::CUT1_LANES;
        hipLaunchKernelGGL(rtk::cut_single_kernel, dim3((unsigned)((cp.tiles[l] + per_wave - 1) / per_wave)), dim3(64), 0, st, ca);
        HIP_TRY(hipGetLastError());
        for (int q = cp.lo; q <= tile_log2 && q < rtk::CUT_MAX_LEVELS; ++q) {
            out.lvl[q] = cuts + cp.offset_words[l];
            out.tx[q] = cp.tiles_x[l];
            out.eff[q] = l;
        }
        out.lo = cp.lo;
        return RT_OK;
    }
    int parent_lvl = -1;
    for (int l = cp.top; l >= cp.lo; --l) {  // each level from its parent's cuts, the top one from the root
        if (!cp.present[l]) continue;
        ca.lvl = l;
        ca.parent_shift = parent_lvl < 0 ? 0 : parent_lvl - l;
        parent_lvl = l;
        ca.cap = cut_cap(l, cp);
        ca.tiles_x = cp.tiles_x[l];
        ca.n_tiles = (int32_t)cp.tiles[l];
        ca.cuts = cuts + cp.offset_words[l];
        // the finest level's leaves lose the triangles no ray of the block can hit (cut_trim); RT_NO_TRIM: comparison knob
        ca.tris = (l == cp.lo && !cp.single && !opt("RT_NO_TRIM")) ? cam_tris : nullptr;
        const int64_t patches = int64_t((cp.tiles_x[l] + 7) / 8) * ((cp.tiles[l] / cp.tiles_x[l] + 7) / 8);
        const int per_wg = rtk::CUT_THREADS / 64;  // one 8 x 8 patch of blocks per wave
        hipLaunchKernelGGL(rtk::cut_level_kernel, dim3((unsigned)((patches + per_wg - 1) / per_wg)), dim3(rtk::CUT_THREADS),
                           0, st, ca);
        ca.parent = ca.cuts;
        ca.parent_tiles_x = ca.tiles_x;
    }
    HIP_TRY(hipGetLastError());
    // what a batch finds at level l: that level's cuts, or those of the next coarser level that was made
    for (int l = cp.lo; l <= tile_log2 && l < rtk::CUT_MAX_LEVELS; ++l) {
        int e = l;
        while (e < cp.top && !cp.present[e]) ++e;  // the top level (>= tile_log2) always is
        out.lvl[l] = cuts + cp.offset_words[e];
        out.tx[l] = cp.tiles_x[e];
        out.eff[l] = e;
    }
    out.lo = cp.lo;
    return RT_OK;
}

// What the prep + cut kernels of a packet launch leave for its traversal kernel.
struct PacketPrep {
    rtk::FrameParams fpv;
    CutTables ct;
    int fast_ok = 0;
};

// Prep kernel + cut kernels of one packet launch on stream `st` (c->stream, or prep_stream with the caller's
// events around it).  Allocates nothing.
int launch_packet_prep(rt_ctx* c, hipStream_t st, const rt_render_params* p, int tile_log2, const PacketBuffers& pb,
                       PacketPrep& pp) {
    const bool count = pb.counters != nullptr;
    const double ox = p->camera_center[0], oy = p->camera_center[1], oz = p->camera_center[2];
    launch_prep(c, st, pb.cam_nodes, pb.cam_tlas, pb.cam_tris, ox, oy, oz);
    HIP_TRY(hipGetLastError());
    for (int i = 0; i < 3; ++i) {
        pp.fpv.cam[i] = p->camera_center[i];
        pp.fpv.p00[i] = p->pixel_00_center[i];
    }
    for (int i = 0; i < 6; ++i) pp.fpv.sp[i] = p->pixel_spacing[i];
    // finite coordinates => every node box (>= 1 triangle, rt_scene_build rejects empty meshes) has
    // min <= max, and so has its camera-relative copy (subtraction rounds monotonically): level 2
    pp.fast_ok = (c->max_abs_coord + std::fabs(ox) + std::fabs(oy) + std::fabs(oz)) < 1e100 ? 2 : 0;
    // (visit counters count the reference's visit rule, so a counting launch walks from the root; RT_COUNT_WITH_CUTS=1
    // counts what the cut walk really fetches and tests instead -- profiling aid)
    const bool count_cuts = opt("RT_COUNT_WITH_CUTS") != nullptr;
    return make_cuts(c, st, p, tile_log2, pp.fpv, pb.cam_nodes, pb.cam_tlas, pb.cam_tris, pb.cuts,
                     (!count || count_cuts) && pp.fast_ok == 2, pp.ct);
}

// render_packet_kernel on c->stream, bracketed by the context's timing events.  Allocates nothing.
int launch_packet_kernel(rt_ctx* c, const rt_render_params* p, int tile_log2, int tiles_x, int64_t n_tiles,
                         const PacketBuffers& pb, const PacketPrep& pp, bool* cuts_made = nullptr) {
    const int W = p->image_width, spp = p->samples, rows = p->row_end - p->row_begin;
    const bool count = pb.counters != nullptr;
    rtk::PacketArgs a;
    a.sc.nodes = pb.cam_nodes;
    a.sc.tlas = pb.cam_tlas;
    a.sc.slot_root = c->slot_root.p;
    a.sc.leaf_ext = c->leaf_ext.p;
    a.sc.tris = pb.cam_tris;
    a.sc.colors = c->colors.p;
    a.fpv = pp.fpv;
    a.W = W;
    a.spp = spp;
    a.row0 = p->row_begin;
    a.rows = rows;
    a.tile_log2 = tile_log2;
    a.tiles_x = tiles_x;
    a.n_tiles = (int32_t)n_tiles;
    a.jitter = pb.jitter;
    a.intensity = pb.intensity;
    a.rgb = pb.rgb;
    a.hit_prim = pb.hit_prim;
    a.hit_t = pb.hit_t;
    a.counters = pb.counters;
    a.pair64 = 0;  // 1 + log2(spp / 64) for spp = 64, 128, 256, ...: the one-pixel-per-batch instantiation
    // (an even number of batches per wave: 16 or 4 pixels per wave, or one pixel of >= 128 samples)
    if (spp >= 64 && (spp & (spp - 1)) == 0 && (tile_log2 == 3 || tile_log2 == 2 || (tile_log2 == 1 && spp >= 128)) &&
        c->stack_need <= rtk::PAIR64_STACK &&
        !opt("RT_NO_PAIR64"))
        for (int q = spp >> 6; q; q >>= 1) ++a.pair64;
    // Sample counts a little below 64 * 2^n (from 49/64 of it): the same instantiation with the last lanes idle beats
    // batches that straddle pixels (their cuts are those of bigger blocks, and their index arithmetic is per lane):
    // 2048^2, 1 M triangles, 50 spp 5.85 -> 5.21 ms, 56 spp 5.99 -> 5.29, 63 spp 7.01 -> 5.40; 48 spp 5.08 -> 5.17 and
    // 40 spp 4.57 -> 5.05 stay generic
    const int pad_from = opt("RT_PAD64_FROM") ? std::atoi(opt("RT_PAD64_FROM")) : 49;  // (tuning knob, per 64)
    if (a.pair64 == 0 && spp > 32 && c->stack_need <= rtk::PAIR64_STACK && !opt("RT_NO_PAIR64")) {
        int padded = 64, lg = 0;
        while (padded < spp) {
            padded <<= 1;
            ++lg;
        }
        // (two or more batches per pixel: the idle lanes are all in the last one; 100 of 128 lost 2 %, 120 won 9 %)
        const int from = lg == 0 ? pad_from : std::max(pad_from, 56);
        if (spp != padded && int64_t(spp) * 64 >= int64_t(padded) * from &&
            (tile_log2 == 3 || tile_log2 == 2 || (tile_log2 == 1 && padded >= 128)))
            a.pair64 = 1 + lg;
    }
    a.interleave = opt("RT_NO_INTERLEAVE") ? 0 : (a.pair64 ? 1 : 0);
    if (const char* e = opt("RT_INTERLEAVE")) a.interleave = std::atoi(e);  // experiment: 0, 1, 2, 4, ...
    if (a.interleave && ((a.interleave & (a.interleave - 1)) || a.interleave > ((1 << (2 * tile_log2)) >> 2))) a.interleave = 0;
    a.fast_ok = pp.fast_ok;
    for (int l = 0; l < rtk::CUT_MAX_LEVELS; ++l) {
        a.cut_lvl[l] = pp.ct.lvl[l];
        a.cut_tx[l] = pp.ct.tx[l];
        a.cut_eff[l] = pp.ct.eff[l];
    }
    a.cut_lo = pp.ct.lo;
    a.spp_magic = 65536u / (uint32_t)std::max(spp, 1) + 1u;
    c->stats.finest_cut_level = a.cut_lo;  // (test seam: -1 = this launch walks from the root)
    if (cuts_made) *cuts_made = a.cut_lo >= 0;
    if (!c->skip_timing_events) HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    if (count && a.pair64)
        hipLaunchKernelGGL((rtk::render_packet_kernel<true, true>), dim3((unsigned)n_tiles), dim3(rtk::NT), 0, c->stream, a);
    else if (count)
        hipLaunchKernelGGL((rtk::render_packet_kernel<true, false>), dim3((unsigned)n_tiles), dim3(rtk::NT), 0, c->stream, a);
    else if (a.pair64)
        hipLaunchKernelGGL((rtk::render_packet_kernel<false, true>), dim3((unsigned)n_tiles), dim3(rtk::NT), 0, c->stream, a);
    else
        hipLaunchKernelGGL((rtk::render_packet_kernel<false, false>), dim3((unsigned)n_tiles), dim3(rtk::NT), 0, c->stream, a);
    HIP_TRY(hipGetLastError());
    if (!c->skip_timing_events) HIP_TRY(hipEventRecord(c->ev[1], c->stream));
    return RT_OK;
}

// Both on c->stream (rt_render_static).
int launch_packet(rt_ctx* c, const rt_render_params* p, int tile_log2, int tiles_x, int64_t n_tiles, const PacketBuffers& pb,
                  bool* cuts_made = nullptr) {
    PacketPrep pp;
    int rc = launch_packet_prep(c, c->stream, p, tile_log2, pb, pp);
    if (rc) return rc;
    return launch_packet_kernel(c, p, tile_log2, tiles_x, n_tiles, pb, pp, cuts_made);
}

// Called when the previous launch is known to be complete (its kernel time resolved): learn the stream count of
// the look-ahead fills of very large frames from that launch's sampler and traversal times (see rt_ctx::adapt_n).
void adapt_streams(rt_ctx* c, int rows, int W, int spp) {
    if (opt("RT_NO_ADAPT_STREAMS") || opt("RT_JITTER_STREAMS")) return;  // (tuning knobs)
    if (!c->kernel_ms_known || !c->jitter_timed || c->out_rows != rows || c->out_w != W || c->out_spp != spp) return;
    if (uint64_t(rows) * uint64_t(W) * uint64_t(spp) * 4u < adapt_min_words()) return;
    const rt_ctx::JitterSlot& sl = c->slot[c->last_slot];  // the slot that launch read: its fill is complete
    if (sl.rows != rows || sl.W != W || sl.spp != spp || sl.frames != 1 || sl.states_n < 1) return;
    float sampler_ms = 0.f;
    if (hipEventElapsedTime(&sampler_ms, sl.t0, sl.t1) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    const double render_ms = c->stats.render_kernel_ms;
    if (!(sampler_ms > 0.f) || !(render_ms > 0.0)) return;
    const char* re = opt("RT_ADAPT_RATIO");  // (test knob)
    // (0.95, was 0.8: the sampler's workgroups hold wave slots and its jumps are work, so the fewer streams the
    // better as long as the fill still ends before the traversal it hides behind -- 4096^2 x 64: 21.0 -> 20.5 ms per
    // step, same box; rounding up to a multiple of 64 streams is the margin)
    const double ratio = re ? std::max(1e-6, std::atof(re)) : 0.95;
    double want = double(sl.states_n) * double(sampler_ms) / (ratio * render_ms);
    want = std::min(1024.0, std::max(128.0, std::ceil(want / 64.0) * 64.0));
    if (c->adapt_job.valid()) {
        if (c->adapt_job.wait_for(std::chrono::seconds(0)) != std::future_status::ready) return;
        try {
            c->adapt_job.get();
            c->adapt_n = c->adapt_pending_n;
            c->adapt_rows = c->adapt_pending_rows;
            c->adapt_w = c->adapt_pending_w;
            c->adapt_spp = c->adapt_pending_spp;
        } catch (const std::exception&) {  // no new polynomials: the old count stays
        }
        return;  // (the next launch measures the new count before anything else is decided)
    }
    const int cur = (c->adapt_n > 0 && c->adapt_rows == rows && c->adapt_w == W && c->adapt_spp == spp) ? c->adapt_n : sl.states_n;
    if (want < 0.8 * cur || want > 1.25 * cur) {  // (hysteresis: a new count costs a tree of jumps)
        const StreamPlan pl = plan_streams(rows, W, spp, 1, (int)want);
        c->adapt_pending_n = (int)want;
        c->adapt_pending_rows = rows;
        c->adapt_pending_w = W;
        c->adapt_pending_spp = spp;
        c->adapt_job = std::async(std::launch::async, [pl]() {
            int levels = 0;
            while ((1 << levels) < pl.n) ++levels;
            for (int k = 0; k < levels; ++k) (void)rt::mt_jump_poly(pl.stride << k);
        });
    }
}

// rt_render.  `pipelined` (rt_render_scene only, plain MT19937 frames without per-sample outputs): the
// launch is enqueued behind the previous one instead of waiting for it, so the GPU goes from frame to
// frame without a gap while the host reads the previous frame back.  Everything a launch touches is
// then either ordered on c->stream, double-buffered (frame buffers, camera block) or guarded by an
// event (jitter slots).
int render_impl(rt_ctx* c, const rt_render_params* p, bool pipelined) {
    if (!c || !p) return fail(RT_E_INVALID, "rt_render: null argument");
    if (!c->have_scene) return fail(RT_E_NO_SCENE, "rt_render: no scene uploaded");
    const int H = p->image_height, W = p->image_width, spp = p->samples;
    if (H < 1 || W < 1 || spp < 1) return fail(RT_E_INVALID, "rt_render: image size and samples must be >= 1");
    if (p->row_begin < 0 || p->row_end > H || p->row_begin >= p->row_end)
        return fail(RT_E_INVALID, "rt_render: bad row range");
    if (p->sampler != RT_SAMPLER_MT19937 && p->sampler != RT_SAMPLER_EXPLICIT && p->sampler != RT_SAMPLER_PHILOX)
        return fail(RT_E_INVALID, "rt_render: unknown sampler");
    if (p->sampler == RT_SAMPLER_EXPLICIT && !p->jitter)
        return fail(RT_E_INVALID, "rt_render: RT_SAMPLER_EXPLICIT needs a jitter array");
    if ((int64_t)spp * 256 > (int64_t)1 << 30) return fail(RT_E_LIMIT, "rt_render: samples per pixel too large");
    const int rows = p->row_end - p->row_begin;
    const size_t n_pix = size_t(rows) * size_t(W);
    const size_t n_smp = n_pix * size_t(spp);
    HIP_TRY(hipSetDevice(c->device));
    pipelined = pipelined && p->sampler == RT_SAMPLER_MT19937 && p->flags == 0;
    int rc;
    // ---- traversal launch shape ----
    // the packet kernels address nodes and triangle records with 32-bit scalar byte offsets: both arrays < 4 GB
    // (6.7e7 nodes of 64 bytes, 3.3e7 triangles of 128)
    const bool packet = c->stack_need <= rtk::PACKET_STACK && !force_per_lane() && c->n_nodes < (size_t(1) << 26) &&
                        c->n_tris < (int64_t(1) << 25);
    const int tile_log2 = pick_tile_log2(packet || (p->flags & RT_FLAG_FP32), spp, W, rows);
    const int ts = 1 << tile_log2;
    const int tiles_x = (W + ts - 1) / ts;
    const int tiles_y = (rows + ts - 1) / ts;
    const int64_t n_tiles = int64_t(tiles_x) * tiles_y;
    if (n_tiles > 0x7fffffff) return fail(RT_E_LIMIT, "rt_render: too many tiles");
    const bool count = (p->flags & RT_FLAG_COUNTERS) != 0;
    if ((rc = c->counters.ensure(8))) return rc;
    // ---- prep + cut kernels of a packet launch: before anything waits for the previous launch, on prep_stream,
    // into the camera set that launch is not reading -- they run while its traversal kernel still does
    const bool packet_launch = packet && !(p->flags & RT_FLAG_FP32);
    PacketBuffers pb{};
    PacketPrep pp;
    rt_ctx::CamSet* cs = nullptr;
    CutPlan cp;
    // (small frames of small scenes: the two cross-stream dependencies cost more than the ~10 us of prep they hide)
    // pipelined launches (the frame driver's, which never asks for a launch's statistics): no timing events around the kernel
    c->skip_timing_events = pipelined;
    // (a frame that gets cuts always overlaps: the cut kernels are chains of dependent steps, ~0.3 ms of latency)
    const bool overlap = opt("RT_NO_PREP_OVERLAP") == nullptr &&  // (comparison knob, read per launch)
                         (n_smp >= (size_t(1) << 24) || c->n_tris >= (1 << 18) || cut_mode(W, rows, spp, tile_log2, c->n_tris) != 0);
    if (packet_launch) {
        c->cam_idx ^= 1;
        cs = &c->cam[c->cam_idx];
        if ((rc = cs->nodes.ensure(c->n_nodes))) return rc;
        if ((rc = cs->tlas.ensure(c->n_tlas))) return rc;
        if ((rc = cs->tris.ensure((size_t)c->n_tris))) return rc;
        cp = plan_cuts(W, rows, spp, tile_log2, c->n_tris);
        if (cp.tiles[cp.lo] > 0x7fffffff) return fail(RT_E_LIMIT, "rt_render: too many pixel blocks");
        if ((rc = cs->cuts.ensure(size_t(cp.total_words)))) return rc;
        pb.cam_nodes = cs->nodes.p;
        pb.cam_tlas = cs->tlas.p;
        pb.cam_tris = cs->tris.p;
        pb.cuts = cs->cuts.p;
        pb.counters = count ? c->counters.p : nullptr;
        const hipStream_t st = overlap ? c->prep_stream : c->stream;
        // the set was last read by the launch before the previous one (or by an fp32 launch)
        if (overlap && cs->read_pending) HIP_TRY(hipStreamWaitEvent(st, done_event_for(c, cs->read_seq), 0));
        if ((rc = launch_packet_prep(c, st, p, tile_log2, pb, pp))) return rc;
        if (overlap) HIP_TRY(hipEventRecord(cs->ready, st));
    }
    // outputs / jitter of the previous call may still be in flight
    if (!pipelined) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        // the previous launch is complete: its kernel time goes into the running total now, so that a loop of
        // rt_render calls can be timed launch by launch without a stats call (and its sync) in between
        int trc = resolve_kernel_ms(c);
        if (trc) return trc;
        adapt_streams(c, rows, W, spp);
    }
    // the other samplers write slot 0's buffer on the traversal stream: no look-ahead fill may be using it.  (MT19937
    // launches need no such wait: everything they touch of the sampler is ordered on rng_stream or guarded by the
    // slots' events, and a fill of several frames must not hold up the frames the slot in use still serves.)
    if (p->sampler != RT_SAMPLER_MT19937) HIP_TRY(hipStreamSynchronize(c->rng_stream));

    // the previous launch's frame moves to prev_* (its kernel may still be writing it: only the
    // pointers move); this launch writes the other pair
    // (three pairs in rotation: the frame driver reads a frame back while the TWO launches after it are queued)
    std::swap(c->prev_intensity, c->prev2_intensity);
    std::swap(c->prev_rgb, c->prev2_rgb);
    std::swap(c->prev_out_done, c->prev2_out_done);
    std::swap(c->prev_out_seq, c->prev2_out_seq);
    std::swap(c->intensity, c->prev_intensity);
    std::swap(c->rgb, c->prev_rgb);
    std::swap(c->out_done, c->prev_out_done);
    std::swap(c->out_seq, c->prev_out_seq);
    c->fp_idx ^= 1;
    DevBuf<rtk::FrameParams>& frame_params = c->frame_params_buf[c->fp_idx];
    rtk::FrameParams& h_frame_params = c->h_frame_params_buf[c->fp_idx];
    if ((rc = c->intensity.ensure(n_pix))) return rc;
    if ((rc = c->rgb.ensure(n_pix * 3))) return rc;
    if (p->flags & RT_FLAG_HIT_IDS)
        if ((rc = c->hit_prim.ensure(n_smp))) return rc;
    if (p->flags & RT_FLAG_HIT_T)
        if ((rc = c->hit_t.ensure(n_smp))) return rc;
    if ((rc = c->counters.ensure(8))) return rc;

    c->stats = rt_stats{};
    c->stats.rays = n_smp;
    c->jitter_timed = false;
    c->kernel_ms_known = false;
    c->timing_pending = false;

    // ---- jitter ----
    int use = 0;
    int slot_frame = 0;  // which of the slot's frames this launch reads
    const uint64_t start_pos = c->pos;
    if (p->sampler == RT_SAMPLER_EXPLICIT) {
        use = 0;
        c->slot[0].valid = false;
        if ((rc = c->slot[0].buf.ensure(n_smp * 2))) return rc;
        HIP_TRY(hipMemcpyAsync(c->slot[0].buf.p, p->jitter, n_smp * 2 * sizeof(double), hipMemcpyHostToDevice,
                               c->stream));
    } else if (p->sampler == RT_SAMPLER_PHILOX) {
        // counter-based fast sampler: one fully parallel kernel on the traversal stream, keyed by the
        // context's seed and the global sample number (stream position / 4)
        use = 0;
        c->slot[0].valid = false;
        if ((rc = c->slot[0].buf.ensure(n_smp * 2))) return rc;
        hipLaunchKernelGGL(rtk::philox_jitter_kernel, dim3((unsigned)((n_smp + 255) / 256)), dim3(256), 0, c->stream,
                           c->slot[0].buf.p, (unsigned long long)n_smp, (unsigned long long)(c->pos / 4u), c->seed, 0u);
        HIP_TRY(hipGetLastError());
        try {  // the MT19937 state follows the position, so the samplers can be mixed within one stream
            if ((rc = sync_state(c))) return rc;
            if ((rc = dev_jump(c, c->d_state.p, c->d_state.p, uint64_t(n_smp) * 4u))) return rc;
        } catch (const std::exception& e) {
            return fail(RT_E_INVALID, std::string("rt_render: sampler: ") + e.what());
        }
        c->pos += uint64_t(n_smp) * 4u;
        c->state_pos = c->pos;
    } else {
        try {
            use = -1;
            const uint64_t fw = uint64_t(n_smp) * 4u;
            for (int k = 0; k < 2; ++k) {
                const auto& sl = c->slot[k];
                if (sl.valid && sl.rows == rows && sl.W == W && sl.spp == spp && c->pos >= sl.pos &&
                    (c->pos - sl.pos) % fw == 0 && (c->pos - sl.pos) / fw < uint64_t(sl.frames)) {
                    use = k;
                    slot_frame = (int)((c->pos - sl.pos) / fw);
                }
            }
            if (use < 0) {
                use = 1 - c->last_slot;
                slot_frame = 0;
                if ((rc = fill_slot(c, use, 0, rows, W, spp))) return rc;
                c->stats.host_jump_ms = c->slot[use].host_ms;
            }
            HIP_TRY(hipStreamWaitEvent(c->stream, c->slot[use].done, 0));
            c->jitter_timed = true;
            c->stats.jitter_streams = (uint64_t)c->slot[use].states_n;
            c->pos += fw;
            if (slot_frame == c->slot[use].frames - 1) {
                // the slot's last frame: its end state is the generator state at the new position
                HIP_TRY(hipMemcpyAsync(c->d_state.p, c->slot[use].d_end.p, sizeof(rt::MtState), hipMemcpyDeviceToDevice,
                                       c->rng_stream));
                c->state_pos = c->pos;
                c->slot[use].valid = false;  // consumed
            }  // else d_state stays behind (state_pos): nothing needs it while the slot serves the frames
            c->last_slot = use;
        } catch (const std::exception& e) {
            return fail(RT_E_INVALID, std::string("rt_render: sampler: ") + e.what());
        }
    }
    const double* d_jitter = c->slot[use].buf.p + size_t(slot_frame) * n_smp * 2;

    // ---- traversal ----
    if (count) HIP_TRY(hipMemsetAsync(c->counters.p, 0, 8 * sizeof(unsigned long long), c->stream));
    c->stats_counters = count ? c->counters.p : nullptr;
    if (p->flags & RT_FLAG_FP32) {
        // fp32 FAST PATH (not the parity path): same walk in binary32, see rt_kernels_fp32.hip.hpp
        if (c->stack_need > rtk::PACKET_STACK || c->n_nodes >= (size_t(1) << 26))
            return fail(RT_E_LIMIT, "rt_render: RT_FLAG_FP32 needs a traversal stack <= 128 entries and < 2^26 nodes");
        if ((rc = c->nodes32.ensure(c->n_nodes))) return rc;
        if ((rc = c->tlas32.ensure(c->n_tlas))) return rc;
        if ((rc = c->tris32.ensure((size_t)c->n_tris))) return rc;
        const double ox = p->camera_center[0], oy = p->camera_center[1], oz = p->camera_center[2];
        hipLaunchKernelGGL(rtk32::prep32_nodes_kernel, dim3((unsigned)((c->n_nodes + 255) / 256)), dim3(256), 0,
                           c->stream, c->nodes.p, c->nodes32.p, (long long)c->n_nodes, ox, oy, oz);
        hipLaunchKernelGGL(rtk32::prep32_tlas_kernel, dim3((unsigned)((c->n_tlas + 255) / 256)), dim3(256), 0, c->stream,
                           c->tlas.p, c->tlas32.p, (long long)c->n_tlas, ox, oy, oz);
        hipLaunchKernelGGL(rtk32::prep32_tris_kernel, dim3((unsigned)((c->n_tris + 255) / 256)), dim3(256), 0, c->stream,
                           c->tris.p, c->tris32.p, (long long)c->n_tris, ox, oy, oz);
        HIP_TRY(hipGetLastError());
        if ((rc = frame_params.ensure(1))) return rc;
        for (int i = 0; i < 3; ++i) {
            h_frame_params.cam[i] = p->camera_center[i];
            h_frame_params.p00[i] = p->pixel_00_center[i];
        }
        for (int i = 0; i < 6; ++i) h_frame_params.sp[i] = p->pixel_spacing[i];
        HIP_TRY(hipMemcpyAsync(frame_params.p, &h_frame_params, sizeof(rtk::FrameParams), hipMemcpyHostToDevice,
                               c->stream));
        // cuts: made from the exact camera-relative boxes (the fp32 records share the node numbering)
        CutTables ct;
        {
            // (all on c->stream, in the set of the last launch: ordered behind that launch's kernel)
            cp = plan_cuts(W, rows, spp, tile_log2, c->n_tris);
            cs = &c->cam[c->cam_idx];
            if ((rc = cs->nodes.ensure(c->n_nodes))) return rc;
            if ((rc = cs->tlas.ensure(c->n_tlas))) return rc;
            if ((rc = cs->cuts.ensure(size_t(cp.total_words)))) return rc;
            if ((rc = cs->tris.ensure((size_t)c->n_tris))) return rc;  // (f64 records: the cuts' triangle trimming)
            launch_prep(c, c->stream, cs->nodes.p, cs->tlas.p, cs->tris.p, ox, oy, oz);
            const bool tame = (c->max_abs_coord + std::fabs(ox) + std::fabs(oy) + std::fabs(oz)) < 1e30;
            if ((rc = make_cuts(c, c->stream, p, tile_log2, h_frame_params, cs->nodes.p, cs->tlas.p, cs->tris.p, cs->cuts.p,
                                !count && tame, ct)))
                return rc;
            c->cut_plan = cp;
            c->cut_plan_valid = ct.lo >= 0;
        }
        rtk32::Packet32Args a;
        for (int l = 0; l < rtk::CUT_MAX_LEVELS; ++l) {
            a.cut_lvl[l] = ct.lvl[l];
            a.cut_tx[l] = ct.tx[l];
            a.cut_eff[l] = ct.eff[l];
        }
        a.cut_lo = ct.lo;
        a.nodes = c->nodes32.p;
        a.tlas = c->tlas32.p;
        a.slot_root = c->slot_root.p;
        a.leaf_ext = c->leaf_ext.p;
        a.tris = c->tris32.p;
        a.colors = c->colors.p;
        a.fp = frame_params.p;
        a.W = W;
        a.spp = spp;
        a.row0 = p->row_begin;
        a.rows = rows;
        a.tile_log2 = tile_log2;
        a.tiles_x = tiles_x;
        a.n_tiles = (int32_t)n_tiles;
        a.pair64 = (spp == 64 && (tile_log2 == 3 || tile_log2 == 2) && c->stack_need <= rtk::PAIR64_STACK && !opt("RT_NO_PAIR64")) ? 1 : 0;
        a.interleave = (opt("RT_NO_INTERLEAVE") || !a.pair64) ? 0 : 1;
        a.jitter = d_jitter;
        a.intensity = c->intensity.p;
        a.rgb = c->rgb.p;
        a.hit_prim = (p->flags & RT_FLAG_HIT_IDS) ? c->hit_prim.p : nullptr;
        a.hit_t = (p->flags & RT_FLAG_HIT_T) ? c->hit_t.p : nullptr;
        a.counters = count ? c->counters.p : nullptr;
        HIP_TRY(hipEventRecord(c->ev[0], c->stream));
        if (count)
            hipLaunchKernelGGL(rtk32::render_packet32_kernel<true>, dim3((unsigned)n_tiles), dim3(rtk::NT), 0, c->stream, a);
        else
            hipLaunchKernelGGL(rtk32::render_packet32_kernel<false>, dim3((unsigned)n_tiles), dim3(rtk::NT), 0, c->stream, a);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->ev[1], c->stream));
    } else if (packet) {
        // Wave-packet traversal on camera-relative records (see rt_kernels.hip.hpp); they and the cuts are on their
        // way or done (above)
        pb.jitter = d_jitter;
        pb.intensity = c->intensity.p;
        pb.rgb = c->rgb.p;
        pb.hit_prim = (p->flags & RT_FLAG_HIT_IDS) ? c->hit_prim.p : nullptr;
        pb.hit_t = (p->flags & RT_FLAG_HIT_T) ? c->hit_t.p : nullptr;
        if (overlap) HIP_TRY(hipStreamWaitEvent(c->stream, cs->ready, 0));
        bool cuts_made = false;
        if ((rc = launch_packet_kernel(c, p, tile_log2, tiles_x, n_tiles, pb, pp, &cuts_made))) return rc;
        c->cut_plan = cp;
        c->cut_plan_valid = cuts_made;
    } else {
        // Per-lane traversal: trees deeper than the packet kernel's 64 pending bits, or forced.
        rtk::RenderArgs a;
        a.sc = dev_scene(c);
        for (int i = 0; i < 3; ++i) {
            a.cam[i] = p->camera_center[i];
            a.p00[i] = p->pixel_00_center[i];
        }
        for (int i = 0; i < 6; ++i) a.sp[i] = p->pixel_spacing[i];
        a.W = W;
        a.spp = spp;
        a.row0 = p->row_begin;
        a.rows = rows;
        a.tile_log2 = tile_log2;
        a.tiles_x = tiles_x;
        a.n_tiles = (int32_t)n_tiles;
        a.jitter = d_jitter;
        a.intensity = c->intensity.p;
        a.rgb = c->rgb.p;
        a.hit_prim = (p->flags & RT_FLAG_HIT_IDS) ? c->hit_prim.p : nullptr;
        a.hit_t = (p->flags & RT_FLAG_HIT_T) ? c->hit_t.p : nullptr;
        a.counters = count ? c->counters.p : nullptr;
        a.gstack = nullptr;
        a.gstack_cap = 0;
        const int stack = pick_stack(c->stack_need);
        if (stack == 0) {
            if ((rc = c->gstack.ensure(size_t(n_tiles) * rtk::NT * size_t(c->stack_need)))) return rc;
            a.gstack = c->gstack.p;
            a.gstack_cap = c->stack_need;
        }
        HIP_TRY(hipEventRecord(c->ev[0], c->stream));
        if (count)
            launch_render<true>(stack, dim3((unsigned)n_tiles), c->stream, a);
        else
            launch_render<false>(stack, dim3((unsigned)n_tiles), c->stream, a);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->ev[1], c->stream));
    }
    // "This launch's kernel has finished" is what four parties wait for: the camera set's next prep, whoever reuses the
    // scene set, the read-back of this frame, the refill of the jitter slot.  Every event record is a packet the
    // traversal stream works off before it starts the next launch's kernel (~5 us each: most of the gap between the
    // kernels of a loop of small frames), so ONE event is recorded per launch -- out_done, which rotates with the
    // output buffers -- and the camera set and the jitter slot remember the launch number that read them last
    // (done_event_for finds the rotating event that covers it); the scene set's event is recorded when the set is
    // retired (retire_front).
    HIP_TRY(hipEventRecord(c->out_done, c->stream));
    c->out_seq = ++c->launch_seq;
    if (cs) {
        cs->read_seq = c->out_seq;
        cs->read_pending = true;
    }
    c->front_used = true;
    c->slot[use].read_seq = c->out_seq;
    c->slot[use].read_pending = true;
    c->timing_pending = !c->skip_timing_events;
    c->out_rows = rows;
    c->out_w = W;
    c->out_spp = spp;
    c->out_flags = p->flags;

    // ---- look-ahead: while the GPU traverses, generate the jitter of the predicted next call
    // (same shape; same distance between starts as the last two calls, else contiguous) into the
    // other slot.  A wrong guess costs nothing but the wasted work.
    if (p->sampler == RT_SAMPLER_MT19937 && !opt("RT_NO_LOOKAHEAD")) {
        try {
            const uint64_t fw = uint64_t(n_smp) * 4u;
            uint64_t next = c->pos;
            if (c->have_last_start && start_pos > c->last_start) {
                const uint64_t delta = start_pos - c->last_start;
                if (delta >= fw) next = start_pos + delta;
            }
            // Small frames: a slot holds several consecutive frames (one set of jumps and one sampler launch per
            // group instead of per frame -- the jumps alone take as long as such a frame's traversal), and the group
            // after the one in use is generated while the first of its frames renders.
            int group = 1;
            if (next == c->pos && !opt("RT_NO_FRAME_GROUPS"))
                group = (int)std::max<uint64_t>(1, std::min<uint64_t>(16, (uint64_t(1) << 25) / std::max<size_t>(n_smp, 1)));
            const rt_ctx::JitterSlot& cur_sl = c->slot[use];
            uint64_t target = next;
            if (cur_sl.valid && next == c->pos) target = cur_sl.pos + uint64_t(cur_sl.frames) * fw;  // frames left in it
            bool covered = false;
            for (int k = 0; k < 2; ++k) {
                const auto& sl = c->slot[k];
                if (sl.valid && sl.rows == rows && sl.W == W && sl.spp == spp && target >= sl.pos &&
                    (target - sl.pos) % fw == 0 && (target - sl.pos) / fw < uint64_t(sl.frames))
                    covered = true;
            }
            const int other = 1 - use;
            // slot `other` was last read by an earlier call's kernel: fill_slot makes rng_stream wait
            // for it (read_seq); the slot's own jitter kernel is ordered on rng_stream.
            const int hidden_n = (c->adapt_n > 0 && c->adapt_rows == rows && c->adapt_w == W && c->adapt_spp == spp) ? c->adapt_n : 0;
            if (!covered && (rc = fill_slot(c, other, target - c->pos, rows, W, spp, group, hidden_n))) return rc;
        } catch (const std::exception&) {
            c->slot[1 - use].valid = false;
        }
        c->have_last_start = true;
        c->last_start = start_pos;
    }
    return RT_OK;
}

// Rows per band when a launch's jitter slot (16 B per sample, two slots per context) would exceed the budget
// (RT_JITTER_SLOT_BYTES, default 32 GiB), else 0: the launch is rendered in one piece.
int band_rows_for(const rt_render_params* p) {
    if (!p || p->sampler == RT_SAMPLER_EXPLICIT) return 0;  // (the caller's array: the caller bands)
    const int W = p->image_width, spp = p->samples, rows = p->row_end - p->row_begin;
    if (W < 1 || spp < 1 || rows < 16) return 0;
    double budget = 32.0 * 1024.0 * 1024.0 * 1024.0;
    if (const char* e = opt("RT_JITTER_SLOT_BYTES")) budget = std::max(1.0, std::atof(e));
    const double row_bytes = double(W) * spp * 16.0;
    if (double(rows) * row_bytes <= budget) return 0;
    const int64_t fit = (int64_t)(budget / row_bytes);
    const int band = (int)std::max<int64_t>(8, fit / 8 * 8);
    return band < rows ? band : 0;
}

// rt_render of any size: a launch whose jitter would not fit the budget is rendered as consecutive row bands -- the
// same machinery as the multi-GPU bands: band after band at consecutive stream positions, the same frame bit for
// bit -- each band's outputs copied (device to device, on the traversal stream) into whole-frame buffers that then
// take the place of the context's output buffers.  The look-ahead sampler works across bands as it does across
// frames.  Waits for the last band (the bands' kernel times and counters are summed for rt_last_stats).
int render_frame(rt_ctx* c, const rt_render_params* p, bool pipelined) {
    const int band = c && c->have_scene ? band_rows_for(p) : 0;
    if (band == 0) return render_impl(c, p, pipelined);
    const int H = p->image_height, W = p->image_width, spp = p->samples;
    if (H < 1 || p->row_begin < 0 || p->row_end > H) return render_impl(c, p, pipelined);  // (reports the error)
    const int rows = p->row_end - p->row_begin;
    const size_t n_pix = size_t(rows) * size_t(W), n_smp = n_pix * size_t(spp);
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = c->frame_intensity.ensure(n_pix))) return rc;
    if ((rc = c->frame_rgb.ensure(n_pix * 3))) return rc;
    if ((p->flags & RT_FLAG_HIT_IDS) && (rc = c->frame_hit_prim.ensure(n_smp))) return rc;
    if ((p->flags & RT_FLAG_HIT_T) && (rc = c->frame_hit_t.ensure(n_smp))) return rc;
    const bool count = (p->flags & RT_FLAG_COUNTERS) != 0;
    unsigned long long sums[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double jitter_ms = 0.0, host_jump_ms = 0.0;
    // (the kernel time of a launch still pending from an earlier call belongs to that call)
    HIP_TRY(hipStreamSynchronize(c->stream));
    if ((rc = resolve_kernel_ms(c))) return rc;
    const double k0 = c->kernel_ms_total;
    for (int r = p->row_begin; r < p->row_end; r += band) {
        rt_render_params q = *p;
        q.row_begin = r;
        q.row_end = std::min(r + band, p->row_end);
        if ((rc = render_impl(c, &q, false))) return rc;
        const size_t pix0 = size_t(r - p->row_begin) * size_t(W), pix_b = size_t(q.row_end - r) * size_t(W);
        HIP_TRY(hipMemcpyAsync(c->frame_intensity.p + pix0, c->intensity.p, pix_b * sizeof(double), hipMemcpyDeviceToDevice,
                               c->stream));
        HIP_TRY(hipMemcpyAsync(c->frame_rgb.p + pix0 * 3, c->rgb.p, pix_b * 3, hipMemcpyDeviceToDevice, c->stream));
        if (p->flags & RT_FLAG_HIT_IDS)
            HIP_TRY(hipMemcpyAsync(c->frame_hit_prim.p + pix0 * spp, c->hit_prim.p, pix_b * spp * sizeof(int32_t),
                                   hipMemcpyDeviceToDevice, c->stream));
        if (p->flags & RT_FLAG_HIT_T)
            HIP_TRY(hipMemcpyAsync(c->frame_hit_t.p + pix0 * spp, c->hit_t.p, pix_b * spp * sizeof(double),
                                   hipMemcpyDeviceToDevice, c->stream));
        if (count || r + band >= p->row_end) {  // counters are per launch: read them band by band
            rt_stats st;
            if ((rc = rt_last_stats(c, &st))) return rc;
            sums[0] += st.nodes_visited;
            sums[1] += st.tris_tested;
            sums[2] += st.rays_hit;
            sums[3] += st.wave_node_fetches;
            sums[4] += st.wave_tri_fetches;
            jitter_ms += st.jitter_kernel_ms;
        }
        host_jump_ms += c->stats.host_jump_ms;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::swap(c->intensity, c->frame_intensity);
    std::swap(c->rgb, c->frame_rgb);
    if (p->flags & RT_FLAG_HIT_IDS) std::swap(c->hit_prim, c->frame_hit_prim);
    if (p->flags & RT_FLAG_HIT_T) std::swap(c->hit_t, c->frame_hit_t);
    c->out_rows = rows;
    c->stats.rays = n_smp;
    c->stats.render_kernel_ms = c->kernel_ms_total - k0;
    c->stats.jitter_kernel_ms = jitter_ms;
    c->stats.host_jump_ms = host_jump_ms;
    if (count) {
        c->stats.nodes_visited = sums[0];
        c->stats.tris_tested = sums[1];
        c->stats.rays_hit = sums[2];
        c->stats.wave_node_fetches = sums[3];
        c->stats.wave_tri_fetches = sums[4];
    }
    c->stats_counters = nullptr;  // (already summed)
    c->timing_pending = false;
    return RT_OK;
}

// rt_render_scene: the frame of the launch BEFORE the last one (pipelined launches), read back on the
// copy stream while the last launch is still running
int download_prev_rgb(rt_ctx* c, uint8_t* rgb, size_t n_bytes, int age) {  // age 1: the previous launch's, 2: the one before
    HIP_TRY(hipStreamWaitEvent(c->copy_stream, age == 2 ? c->prev2_out_done : c->prev_out_done, 0));
    HIP_TRY(hipMemcpyAsync(rgb, age == 2 ? c->prev2_rgb.p : c->prev_rgb.p, n_bytes, hipMemcpyDeviceToHost, c->copy_stream));
    HIP_TRY(hipStreamSynchronize(c->copy_stream));
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_render(rt_ctx* c, const rt_render_params* p) { return render_frame(c, p, false); }

// rt_render for callers that cannot allocate: same prep + traversal kernels, the camera-relative records and
// the counters in this GPU's copy of the file-scope `scratch`, jitter and outputs in the caller's device memory.
// No hipMalloc / hipFree / DevBuf::ensure anywhere below.
int rt_render_static(rt_ctx* c, const rt_render_params* p, const double* d_jitter, double* d_intensity, uint8_t* d_rgb,
                     int32_t* d_hit_prim) {
    if (!c || !p) return fail(RT_E_INVALID, "rt_render_static: null argument");
    if (!d_jitter || !d_intensity || !d_rgb)
        return fail(RT_E_INVALID, "rt_render_static: jitter, intensity and rgb device pointers are required");
    if (!c->have_scene) return fail(RT_E_NO_SCENE, "rt_render_static: no scene uploaded");
    const int H = p->image_height, W = p->image_width, spp = p->samples;
    if (H < 1 || W < 1 || spp < 1) return fail(RT_E_INVALID, "rt_render_static: image size and samples must be >= 1");
    if (p->row_begin < 0 || p->row_end > H || p->row_begin >= p->row_end)
        return fail(RT_E_INVALID, "rt_render_static: bad row range");
    if (p->sampler != RT_SAMPLER_EXPLICIT)
        return fail(RT_E_INVALID, "rt_render_static: only RT_SAMPLER_EXPLICIT (caller-supplied device jitter); the "
                                  "MT19937 sampler keeps per-launch jitter and stream-state buffers");
    if (p->flags & ~(RT_FLAG_HIT_IDS | RT_FLAG_COUNTERS))
        return fail(RT_E_INVALID, "rt_render_static: only RT_FLAG_HIT_IDS and RT_FLAG_COUNTERS are supported");
    if ((p->flags & RT_FLAG_HIT_IDS) && !d_hit_prim)
        return fail(RT_E_INVALID, "rt_render_static: RT_FLAG_HIT_IDS needs a hit-id device pointer");
    if ((int64_t)spp * 256 > (int64_t)1 << 30) return fail(RT_E_LIMIT, "rt_render_static: samples per pixel too large");
    if (c->stack_need > rtk::PACKET_STACK || c->n_nodes >= (size_t(1) << 26) || c->n_tris >= (int64_t(1) << 25))
        return fail(RT_E_LIMIT, "rt_render_static: the tree needs the per-lane kernel (traversal stack > 128 entries)");
    // partition of the scratch array: [counters 256 B][cam_tlas][cam_nodes][cam_tris], each 256-byte aligned
    auto up256 = [](size_t v) { return (v + 255) & ~size_t(255); };
    const size_t off_tlas = 256;
    const size_t off_nodes = off_tlas + up256(c->n_tlas * sizeof(rt::DevNode));
    const size_t off_tris = off_nodes + up256(c->n_nodes * sizeof(rt::DevNode));
    const size_t off_cuts = off_tris + up256(size_t(c->n_tris) * sizeof(rtk::CamTri));
    const size_t need = off_cuts;  // the records are required; the tile cuts are used when they fit behind them
    if (need > size_t(RT_STATIC_SCRATCH_BYTES))
        return fail(RT_E_LIMIT, "rt_render_static: scene needs " + std::to_string(need) + " bytes of scratch (" +
                                    std::to_string(c->n_tris) + " triangles, " + std::to_string(c->n_nodes) +
                                    " nodes), the static array holds " + std::to_string(size_t(RT_STATIC_SCRATCH_BYTES)) +
                                    "; use rt_render");
    unsigned char* const ptr = c->static_scratch;  // resolved in rt_create with hipGetSymbolAddress
    if (!ptr) return fail(RT_E_NO_DEVICE, "rt_render_static: scratch address not resolved");
    const int rows = p->row_end - p->row_begin;
    HIP_TRY(hipSetDevice(c->device));
    // One array per GPU, shared by every context on it.  The lock is held until this call's launches and its
    // `static_done` event are enqueued, so two threads cannot interleave their launches; a launch of ANOTHER context
    // that has not finished yet still reads the array: refuse instead of overwriting what it reads.
    std::lock_guard<std::mutex> scratch_lock(g_scratch_mu);
    if (rt_ctx* owner = g_scratch_owner[c->device]) {
        if (owner != c) {
            const hipError_t q = hipEventQuery(owner->static_done);
            if (q == hipErrorNotReady)
                return fail(RT_E_BUSY, "rt_render_static: another context's launch is still using this GPU's scratch array "
                                       "(rt_sync that context first, or serialise the callers)");
            if (q != hipSuccess) return fail(RT_E_NO_DEVICE, std::string("rt_render_static: hipEventQuery: ") + hipGetErrorString(q));
        }
    }
    // the previous launch of this context may still be reading the scratch records
    HIP_TRY(hipStreamSynchronize(c->stream));
    int rc = resolve_kernel_ms(c);
    if (rc) return rc;
    c->stats = rt_stats{};
    c->stats.rays = uint64_t(rows) * uint64_t(W) * uint64_t(spp);
    c->jitter_timed = false;
    c->kernel_ms_known = false;
    c->timing_pending = false;
    const int tile_log2 = pick_tile_log2(true, spp, W, rows);
    const int ts = 1 << tile_log2;
    const int tiles_x = (W + ts - 1) / ts;
    const int64_t n_tiles = int64_t(tiles_x) * ((rows + ts - 1) / ts);
    if (n_tiles > 0x7fffffff) return fail(RT_E_LIMIT, "rt_render_static: too many tiles");
    const bool count = (p->flags & RT_FLAG_COUNTERS) != 0;
    // the array keeps the last call's contents: the counters must start at zero
    if (count) HIP_TRY(hipMemsetAsync(ptr, 0, 8 * sizeof(unsigned long long), c->stream));
    c->stats_counters = count ? reinterpret_cast<unsigned long long*>(ptr) : nullptr;
    PacketBuffers pb;
    pb.cam_tlas = reinterpret_cast<rt::DevNode*>(ptr + off_tlas);
    pb.cam_nodes = reinterpret_cast<rt::DevNode*>(ptr + off_nodes);
    pb.cam_tris = reinterpret_cast<rtk::CamTri*>(ptr + off_tris);
    pb.jitter = d_jitter;
    pb.intensity = d_intensity;
    pb.rgb = d_rgb;
    pb.hit_prim = (p->flags & RT_FLAG_HIT_IDS) ? d_hit_prim : nullptr;
    pb.hit_t = nullptr;
    pb.counters = count ? reinterpret_cast<unsigned long long*>(ptr) : nullptr;
    // tile cuts only speed the walk up (same results without): used when the image's tiles fit the array too
    pb.cuts = (off_cuts + size_t(plan_cuts(W, rows, spp, tile_log2, c->n_tris).total_words) * sizeof(uint32_t) <=
               size_t(RT_STATIC_SCRATCH_BYTES))
                  ? reinterpret_cast<uint32_t*>(ptr + off_cuts)
                  : nullptr;
    g_scratch_owner[c->device] = c;  // (before the launch: even a launch that fails half-way may have enqueued kernels)
    rc = launch_packet(c, p, tile_log2, tiles_x, n_tiles, pb);
    HIP_TRY(hipEventRecord(c->static_done, c->stream));
    if (rc) return rc;
    HIP_TRY(hipEventRecord(c->front_last_use, c->stream));  // rt_upload_scene waits for it before reusing the set
    c->front_used = true;
    c->timing_pending = true;
    return RT_OK;
}

int rt_last_stats(const rt_ctx* cc, rt_stats* out) {
    if (!cc || !out) return fail(RT_E_INVALID, "rt_last_stats: null argument");
    rt_ctx* c = const_cast<rt_ctx*>(cc);
    HIP_TRY(hipSetDevice(c->device));
    if (c->timing_pending) {
        HIP_TRY(hipEventSynchronize(c->ev[1]));
        float ms = 0.f;
        int trc = resolve_kernel_ms(c);
        if (trc) return trc;
        if (c->jitter_timed) {
            rt_ctx::JitterSlot& sl = c->slot[c->last_slot];
            HIP_TRY(hipEventSynchronize(sl.t1));
            HIP_TRY(hipEventElapsedTime(&ms, sl.t0, sl.t1));
            c->stats.jitter_kernel_ms = ms;
        }
        if (c->stats_counters) {
            unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            HIP_TRY(hipMemcpy(h, c->stats_counters, sizeof(h), hipMemcpyDeviceToHost));
            c->stats.wave_tri_fetches = h[4];
            c->stats.nodes_visited = h[0];
            c->stats.tris_tested = h[1];
            c->stats.rays_hit = h[2];
            c->stats.wave_node_fetches = h[3];
        }
        c->timing_pending = false;
    }
    c->stats.render_kernel_ms_total = c->kernel_ms_total;
    c->stats.render_launches_timed = c->kernel_launches_timed;
    *out = c->stats;
    return RT_OK;
}

int rt_download(rt_ctx* c, double* intensity, uint8_t* rgb, int32_t* hit_prim, double* hit_t) {
    if (!c) return fail(RT_E_INVALID, "rt_download: null context");
    if (c->out_rows == 0) return fail(RT_E_INVALID, "rt_download: nothing rendered yet");
    HIP_TRY(hipSetDevice(c->device));
    const size_t n_pix = size_t(c->out_rows) * size_t(c->out_w);
    const size_t n_smp = n_pix * size_t(c->out_spp);
    if (hit_prim && !(c->out_flags & RT_FLAG_HIT_IDS))
        return fail(RT_E_INVALID, "rt_download: hit ids were not requested (RT_FLAG_HIT_IDS)");
    if (hit_t && !(c->out_flags & RT_FLAG_HIT_T))
        return fail(RT_E_INVALID, "rt_download: hit t was not requested (RT_FLAG_HIT_T)");
    if (intensity)
        HIP_TRY(hipMemcpyAsync(intensity, c->intensity.p, n_pix * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (rgb) HIP_TRY(hipMemcpyAsync(rgb, c->rgb.p, n_pix * 3, hipMemcpyDeviceToHost, c->stream));
    if (hit_prim)
        HIP_TRY(hipMemcpyAsync(hit_prim, c->hit_prim.p, n_smp * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (hit_t) HIP_TRY(hipMemcpyAsync(hit_t, c->hit_t.p, n_smp * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_tile_cuts_host(const rt_scene* scene, const rt_render_params* p, int level, uint32_t* words,
                      int64_t capacity_words, int32_t* tiles_x_out, int32_t* level_lo, int32_t* level_top) {
    if (!scene || !p) return fail(RT_E_INVALID, "rt_tile_cuts_host: null argument");
    const rt::Scene& s = *scene->s;
    const int W = p->image_width, spp = p->samples, rows = p->row_end - p->row_begin;
    if (W < 1 || spp < 1 || rows < 1) return fail(RT_E_INVALID, "rt_tile_cuts_host: bad image");
    const int tile_log2 = pick_tile_log2(true, spp, W, rows);
    const CutPlan cp = plan_cuts(W, rows, spp, tile_log2, (int64_t)s.prim_mesh.size());
    if (level_lo) *level_lo = cp.lo;
    if (level_top) *level_top = cp.top;
    if (level < cp.lo || level > cp.top || !cp.present[level]) return fail(RT_E_INVALID, "rt_tile_cuts_host: no such level");
    if (tiles_x_out) *tiles_x_out = cp.tiles_x[level];
    if (cp.tiles[level] > 0x7fffffff) return fail(RT_E_LIMIT, "rt_tile_cuts_host: too many blocks");
    if (!words) return (int)cp.tiles[level];
    if (capacity_words < cp.tiles[level] * rtk::CUT_WORDS) return fail(RT_E_INVALID, "rt_tile_cuts_host: buffer too small");
    // camera-relative boxes, as prep_frame_kernel computes them
    auto relative = [&](const rt::DevNode* in, size_t n) {
        std::vector<rt::DevNode> out(in, in + n);
        for (auto& v : out)
            for (int k = 0; k < 3; ++k) {
                v.mn[k] = v.mn[k] - p->camera_center[k];
                v.mx[k] = v.mx[k] - p->camera_center[k];
            }
        return out;
    };
    const std::vector<rt::DevNode> nodes = relative(s.dev_nodes.data(), s.dev_nodes.size());
    const std::vector<rt::DevNode> tlas = relative(s.dev_tlas.data(), s.dev_tlas.size());
    // camera-relative triangle records, likewise
    std::vector<rtk::CamTri> tris(s.prim_mesh.size());
    for (size_t i = 0; i < tris.size(); ++i) {
        const double* tp = s.tris.data() + i * 9;
        rtk::CamTri& t = tris[i];
        for (int k = 0; k < 3; ++k) {
            t.e0[k] = tp[3 + k] - tp[k];
            t.e2[k] = tp[6 + k] - tp[k];
            t.tv[k] = p->camera_center[k] - tp[k];
        }
        t.q[0] = t.tv[1] * t.e0[2] - t.tv[2] * t.e0[1];
        t.q[1] = t.tv[2] * t.e0[0] - t.tv[0] * t.e0[2];
        t.q[2] = t.tv[0] * t.e0[1] - t.tv[1] * t.e0[0];
        t.tnum = (t.e2[0] * t.q[0] + t.e2[1] * t.q[1]) + t.e2[2] * t.q[2];
        t.pad[0] = t.pad[1] = t.pad[2] = 0.0;
    }
    rtk::CutArgs ca;
    ca.nodes = nodes.data();
    ca.tlas = tlas.data();
    ca.slot_root = s.slot_root.data();
    for (int i = 0; i < 3; ++i) {
        ca.fp.cam[i] = p->camera_center[i];
        ca.fp.p00[i] = p->pixel_00_center[i];
    }
    for (int i = 0; i < 6; ++i) ca.fp.sp[i] = p->pixel_spacing[i];
    ca.W = W;
    ca.rows = rows;
    ca.row0 = p->row_begin;
    ca.parent = nullptr;
    ca.parent_tiles_x = 0;
    ca.mag = rtk::cut_magnitude(ca.fp, W, p->row_begin, rows);
    std::vector<uint32_t> prev, cur;
    uint32_t lists[rtk::CUT_WORDS];
    int parent_lvl = -1;
    for (int l = cp.top; l >= level; --l) {
        if (!cp.present[l]) continue;
        cur.assign(size_t(cp.tiles[l]) * rtk::CUT_WORDS, 0u);
        ca.lvl = l;
        ca.parent_shift = parent_lvl < 0 ? 0 : parent_lvl - l;
        parent_lvl = l;
        ca.cap = cut_cap(l, cp);
        ca.tiles_x = cp.tiles_x[l];
        ca.n_tiles = (int32_t)cp.tiles[l];
        ca.cuts = cur.data();
        const bool rounds = cp.single && !opt("RT_CUT1_SERIAL");  // (the breadth-first kernel's twin)
        ca.tris = (l == cp.lo && (rounds || !cp.single) && !opt("RT_NO_TRIM")) ? tris.data() : nullptr;
        for (int64_t t = 0; t < cp.tiles[l]; ++t) {
            if (rounds) {
                rtk::cut_single_serial(ca, (int)t);
                continue;
            }
            rtk::cut_refine(ca, (int)t, lists, 1);
            if (ca.tris) rtk::cut_trim(ca, (int)t, lists, 1);
        }
        prev.swap(cur);
        ca.parent = prev.data();
        ca.parent_tiles_x = ca.tiles_x;
    }
    std::memcpy(words, prev.data(), prev.size() * sizeof(uint32_t));
    return (int)cp.tiles[level];
}

int rt_last_cuts(rt_ctx* c, int level, uint32_t* words, int64_t capacity_words, int32_t* tiles_x, int32_t* level_lo,
                 int32_t* level_top) {
    if (!c) return fail(RT_E_INVALID, "rt_last_cuts: null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!c->cut_plan_valid) return 0;
    const CutPlan& cp = c->cut_plan;
    if (level_lo) *level_lo = cp.lo;
    if (level_top) *level_top = cp.top;
    if (level < cp.lo || level > cp.top || !cp.present[level]) return fail(RT_E_INVALID, "rt_last_cuts: no such level");
    if (tiles_x) *tiles_x = cp.tiles_x[level];
    const int64_t n = cp.tiles[level] * rtk::CUT_WORDS;
    if (words) {
        if (capacity_words < n) return fail(RT_E_INVALID, "rt_last_cuts: buffer too small");
        HIP_TRY(hipMemcpy(words, c->cam[c->cam_idx].cuts.p + cp.offset_words[level], size_t(n) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    return (int)std::min<int64_t>(cp.tiles[level], 0x7fffffff);
}

int rt_device_outputs(rt_ctx* c, void** intensity, void** rgb, void** hit_prim) {
    if (!c) return fail(RT_E_INVALID, "rt_device_outputs: null context");
    if (intensity) *intensity = c->out_rows ? c->intensity.p : nullptr;
    if (rgb) *rgb = c->out_rows ? c->rgb.p : nullptr;
    if (hit_prim) *hit_prim = (c->out_rows && (c->out_flags & RT_FLAG_HIT_IDS)) ? c->hit_prim.p : nullptr;
    return RT_OK;
}

/* ------------------------------------------------------------------ arbitrary rays */
int rt_trace_rays(rt_ctx* c, int64_t n, const double* origins, const double* dirs, double* t, double* bary_uvw,
                  double* color, int32_t* prim) {
    if (!c || !origins || !dirs || n < 0) return fail(RT_E_INVALID, "rt_trace_rays: bad argument");
    if (!c->have_scene) return fail(RT_E_NO_SCENE, "rt_trace_rays: no scene uploaded");
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    int rc;
    const size_t N = size_t(n);
    if ((rc = c->tr_o.ensure(N * 3))) return rc;
    if ((rc = c->tr_d.ensure(N * 3))) return rc;
    if ((rc = c->tr_t.ensure(N))) return rc;
    if ((rc = c->tr_b.ensure(N * 3))) return rc;
    if ((rc = c->tr_c.ensure(N * 3))) return rc;
    if ((rc = c->tr_p.ensure(N))) return rc;
    HIP_TRY(hipMemcpyAsync(c->tr_o.p, origins, N * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->tr_d.p, dirs, N * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    rtk::TraceArgs a;
    a.sc = dev_scene(c);
    a.n = n;
    a.origins = c->tr_o.p;
    a.dirs = c->tr_d.p;
    a.t = c->tr_t.p;
    a.bary = c->tr_b.p;
    a.color = c->tr_c.p;
    a.prim = c->tr_p.p;
    a.gstack = nullptr;
    const int64_t blocks = (n + rtk::NT - 1) / rtk::NT;
    const int stack = pick_stack(c->stack_need);
    if (stack == 0) {
        if ((rc = c->gstack.ensure(size_t(blocks) * rtk::NT * size_t(c->stack_need)))) return rc;
        a.gstack = c->gstack.p;
    }
    launch_trace(stack, dim3((unsigned)blocks), c->stream, a);
    HIP_TRY(hipGetLastError());
    if (t) HIP_TRY(hipMemcpyAsync(t, c->tr_t.p, N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (bary_uvw) HIP_TRY(hipMemcpyAsync(bary_uvw, c->tr_b.p, N * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (color) HIP_TRY(hipMemcpyAsync(color, c->tr_c.p, N * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (prim) HIP_TRY(hipMemcpyAsync(prim, c->tr_p.p, N * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RT_OK;
}

/* ------------------------------------------------------------------ PPM + whole-job driver */
int rt_write_ppm(const char* path, int width, int height, const uint8_t* rgb) {
    if (!path || !rgb || width < 1 || height < 1) return fail(RT_E_INVALID, "rt_write_ppm: bad argument");
    FILE* f = std::fopen(path, "wb");
    if (!f) {
        std::fprintf(stderr, "Failed to open the output file.\n");  // cpp/rtrender.cpp:72-75
        return fail(RT_E_IO, std::string("cannot open ") + path);
    }
    std::fprintf(f, "P6\n%d %d\n255\n", width, height);
    std::fwrite(rgb, 1, size_t(width) * size_t(height) * 3, f);
    std::fclose(f);
    std::printf("\r Done. \n");  // cpp/rtrender.cpp:81
    std::fflush(stdout);
    return RT_OK;
}

int64_t rt_shard_frames(int timestep_count, int n_cameras, int rank, int world, uint64_t stream_base,
                        uint64_t frame_words, int64_t capacity, int64_t* frames, uint64_t* positions) {
    if (timestep_count < 0 || n_cameras < 0 || world < 1 || rank < 0 || rank >= world)
        return fail(RT_E_INVALID, "rt_shard_frames: bad counts");
    const int64_t total = int64_t(timestep_count) * n_cameras;
    // Whole timesteps go to one rank when there are enough of them (t % world == rank): the per-timestep host
    // BVH build is then done once, not once per camera on different ranks.  With fewer timesteps than ranks
    // single frames are dealt out instead (f % world == rank).
    const bool by_timestep = timestep_count >= world && n_cameras > 0;
    int64_t n = 0;
    for (int64_t f = 0; f < total; ++f) {
        const int64_t owner = by_timestep ? (f / n_cameras) % world : f % world;
        if (owner != rank) continue;
        if (frames && n < capacity) frames[n] = f;
        if (positions && n < capacity) positions[n] = stream_base + uint64_t(f) * frame_words;
        ++n;
    }
    return n;
}

uint64_t rt_driver_stream_position(int device) {
    if (device < 0 || device >= 64) return 0;
    std::lock_guard<std::mutex> lk(g_global_mu);
    return g_global_ctx[device] ? g_global_ctx[device]->pos : 0;
}

int rt_shard_rows(int image_height, int image_width, int number_of_samples, int rank, int world, uint64_t frame_base,
                  int* row_begin, int* row_end, uint64_t* stream_position) {
    if (image_height < 1 || image_width < 1 || number_of_samples < 1 || world < 1 || rank < 0 || rank >= world ||
        !row_begin || !row_end)
        return fail(RT_E_INVALID, "rt_shard_rows: bad arguments");
    // contiguous bands, boundaries on multiples of 8 rows (the traversal kernel's tile height) when the
    // image is tall enough to give every rank a tile row; the last band takes the remainder
    auto cut = [&](int r) -> int {
        if (r >= world) return image_height;
        const int64_t exact = int64_t(image_height) * r / world;
        if (image_height < 8 * world) return (int)exact;
        return (int)std::min<int64_t>(image_height, (exact + 4) / 8 * 8);
    };
    *row_begin = cut(rank);
    *row_end = cut(rank + 1);
    if (stream_position)
        *stream_position = frame_base + uint64_t(*row_begin) * uint64_t(image_width) * uint64_t(number_of_samples) * 4u;
    return RT_OK;
}

int rt_render_scene(int image_height, int image_width, int number_of_samples, const char* output_directory,
                    int timestep_count, int n_meshes, const double* const* coords, const double* const* colors,
                    const int64_t* elem_counts, int n_cameras, const double* camera_centers,
                    const double* pixel_00_centers, const double* pixel_spacings, int device, int rank, int world) {
    if (!output_directory || !coords || !colors || !elem_counts || !camera_centers || !pixel_00_centers ||
        !pixel_spacings)
        return fail(RT_E_INVALID, "rt_render_scene: null argument");
    if (timestep_count < 0 || n_cameras < 0 || world < 1 || rank < 0 || rank >= world)
        return fail(RT_E_INVALID, "rt_render_scene: bad counts");
    if (device < 0 || device >= 64) return fail(RT_E_INVALID, "rt_render_scene: bad device");
    rt_ctx* c;
    {
        std::lock_guard<std::mutex> lk(g_global_mu);
        if (!g_global_ctx[device]) g_global_ctx[device] = rt_create(device);
        c = g_global_ctx[device];
    }
    if (!c) return RT_E_NO_DEVICE;  // message set by rt_create

    const uint64_t frame_words = uint64_t(image_height) * uint64_t(image_width) * uint64_t(number_of_samples) * 4u;
    const uint64_t base = rt_rng_tell(c);
    const size_t rgb_bytes = size_t(image_height) * size_t(image_width) * 3;
    int rc = RT_OK;
    const int64_t n_mine = rt_shard_frames(timestep_count, n_cameras, rank, world, base, frame_words, 0, nullptr, nullptr);
    std::vector<int64_t> my_frames((size_t)std::max<int64_t>(n_mine, 1));
    std::vector<uint64_t> my_pos((size_t)std::max<int64_t>(n_mine, 1));
    rt_shard_frames(timestep_count, n_cameras, rank, world, base, frame_words, n_mine, my_frames.data(), my_pos.data());
    // The host BVH build of the NEXT timestep this rank renders runs on a helper thread while the GPU
    // renders the current one (the build is per frame and would otherwise serialise with the render).
    std::vector<int> my_steps;
    for (int64_t q = 0; q < n_mine; ++q) {
        const int t = (int)(my_frames[q] / n_cameras);
        if (my_steps.empty() || my_steps.back() != t) my_steps.push_back(t);
    }
    struct Built {
        rt_scene* sc = nullptr;
        int staged = -1;  // index of the device buffer set the helper thread filled, or -1
        std::string err;
    };
    // Builds run ahead of the GPU on helper threads: up to `depth` timesteps in flight (each build is
    // itself parallel over meshes), fewer for big scenes whose flattened copies are large.
    int64_t tris_per_step = 0;
    for (int m = 0; m < n_meshes; ++m) tris_per_step += elem_counts[m];
    size_t depth = tris_per_step > 4000000 ? 1 : (tris_per_step > 500000 ? 2 : 6);
    if (const char* e = opt("RT_BUILD_AHEAD")) depth = (size_t)std::max(1, std::atoi(e));
    // enough meshes in flight to occupy every core: one thread per tree (least CPU time per timestep, see rt_bvh.cpp)
    const bool serial_trees = int64_t(n_meshes) * int64_t(depth) >= int64_t(rt::host_threads()) && !opt("RT_NO_SERIAL_TREES");
    auto build_async = [&](int t) {
        const int64_t ticket = rt::draw_build_ticket();  // (drawn here, in timestep order: the pool serves the oldest build first)
        return build_threads().submit([=]() {
            Built b;
            const rt::AdoptBuildTicket adopt(ticket);
            try {
                std::unique_ptr<rt_scene> h(new rt_scene());
                h->s.reset(rt::build_scene(n_meshes, coords, colors, elem_counts, t, 0, serial_trees));
                b.sc = h.release();
            } catch (const std::exception& e) {
                b.err = std::string("rt_scene_build: ") + e.what();
            }
            if (!b.sc) {
                if (b.err.empty()) b.err = "rt_scene_build failed";
            }
            else
                b.staged = stage_scene(c, *b.sc->s);
            return b;
        });
    };
    // RT_DRIVER_TIMING=1: per-phase wall time of this call on stderr (tuning aid); =2: also the host's timeline frame by
    // frame (when each enqueue, read-back and wait for a build began and ended)
    const bool timing = opt("RT_DRIVER_TIMING") != nullptr;
    const bool trace = timing && std::atoi(opt("RT_DRIVER_TIMING")) >= 2;
    double t_wait = 0, t_upload = 0, t_render = 0, t_down = 0, t_write = 0;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_job0 = now();
    std::deque<std::future<Built>> pending;
    size_t next_step = 0;
    auto top_up = [&]() {
        while (pending.size() < depth && next_step < my_steps.size()) pending.push_back(build_async(my_steps[next_step++]));
    };
    auto drain = [&]() {
        for (auto& f : pending) {
            Built b = f.get();
            if (b.staged >= 0) {
                std::lock_guard<std::mutex> lk(c->staged_mu);
                c->staged[(size_t)b.staged]->state = 0;
            }
            rt_scene_free(b.sc);
        }
        pending.clear();
    };
    if ((rc = ensure_staged_pool(c, depth + 2))) return rc;
    // P6 files are written by a helper thread while the next frame renders (two host frame buffers)
    std::vector<uint8_t> rgb2[2] = {std::vector<uint8_t>(rgb_bytes), std::vector<uint8_t>(rgb_bytes)};
    std::future<void> writer;
    int wbuf = 0;
    bool pipeline = opt("RT_NO_PIPELINE") == nullptr;  // (tuning knob)
    {  // frames rendered band by band (render_frame) end with a wait and swap whole-frame buffers in: not pipelined
        rt_render_params probe;
        std::memset(&probe, 0, sizeof(probe));
        probe.image_height = image_height;
        probe.image_width = image_width;
        probe.samples = number_of_samples;
        probe.row_end = image_height;
        probe.sampler = RT_SAMPLER_MT19937;
        if (band_rows_for(&probe) != 0) pipeline = false;
    }
    std::deque<std::pair<int, int>> in_flight;  // (timestep, camera) of the frames queued and not yet read back, oldest first
    // (comparison knob.  2 -- read a frame back when TWO younger ones are queued -- buys nothing, measured)
    const int ahead = opt("RT_FRAMES_AHEAD") ? std::max(1, std::min(2, std::atoi(opt("RT_FRAMES_AHEAD")))) : 1;
    int64_t pending_sah_lines = 0;
    auto frame_path = [&](int tt, int cc) {
        // cpp/rtmain.cpp:58-60: rtimage_<timestep>_cam<camera>.ppm, zero-based
        return std::string(output_directory) + "/rtimage_" + std::to_string(tt) + "_cam" + std::to_string(cc) + ".ppm";
    };
    auto write_async = [&](const std::string& path, uint8_t* src) {
        // like the reference, an unopenable file is reported and the job goes on
        writer = writer_thread().submit([path, image_width, image_height, src]() {
            (void)rt_write_ppm(path.c_str(), image_width, image_height, src);
        });
        wbuf ^= 1;
    };
    top_up();
    int built_t = -1;
    for (int64_t q = 0; q < n_mine; ++q) {
        const int t = (int)(my_frames[q] / n_cameras), cam = (int)(my_frames[q] % n_cameras);
        auto bail = [&](int code) {
            drain();
            if (writer.valid()) writer.get();
            return code;
        };
        // frames arrive timestep-major: one build + upload per timestep this rank touches.  Returns 0 or
        // an error code (after cleaning up).
        auto advance_scene = [&](int t_new) -> int {
            double t0 = now();
            Built b = pending.front().get();
            pending.pop_front();
            t_wait += now() - t0;
            if (trace) std::fprintf(stderr, "    wait for build of t%d %.3f .. %.3f ms\n", t_new, (t0 - t_job0) * 1e3, (now() - t_job0) * 1e3);
            top_up();
            if (!b.sc) {
                const std::string msg = b.err;
                bail(0);
                return fail(RT_E_INVALID, msg);
            }
            // cpp/rtbvh.cpp:128 prints one line per fallback while the timestep's trees are built, i.e. before the
            // timestep's first "Rendering frame": this build ran ahead on a helper thread, so the lines are held back
            // until that frame is announced
            pending_sah_lines += b.sc->s->midpoint_fallbacks;
            t0 = now();
            int urc = RT_OK;
            if (b.staged >= 0)
                activate_staged(c, b.staged);
            else
                urc = rt_upload_scene(c, b.sc);
            t_upload += now() - t0;
            rt_scene_free(b.sc);
            if (urc) return bail(urc);
            built_t = t_new;
            return RT_OK;
        };
        if (t != built_t && (rc = advance_scene(t))) return rc;
        if ((rc = rt_rng_seek(c, my_pos[q]))) return bail(rc);
        for (; pending_sah_lines > 0; --pending_sah_lines)
            std::printf("SAH splitting failed. Trying midpoint instead.\n");  // cpp/rtbvh.cpp:128
        std::printf("Rendering frame %d/%d\n", t + 1, timestep_count);  // cpp/rtmain.cpp:61
        std::fflush(stdout);
        rt_render_params p;
        std::memset(&p, 0, sizeof(p));
        std::memcpy(p.camera_center, camera_centers + 3 * cam, 3 * sizeof(double));
        std::memcpy(p.pixel_00_center, pixel_00_centers + 3 * cam, 3 * sizeof(double));
        std::memcpy(p.pixel_spacing, pixel_spacings + 6 * cam, 6 * sizeof(double));
        p.image_height = image_height;
        p.image_width = image_width;
        p.samples = number_of_samples;
        p.row_begin = 0;
        p.row_end = image_height;
        p.sampler = RT_SAMPLER_MT19937;
        double t1 = now();
        // pipelined: this frame is enqueued behind the previous one; the previous frame is read back and
        // written while this one renders
        if ((rc = render_frame(c, &p, pipeline))) return bail(rc);
        t_render += now() - t1;
        if (trace) std::fprintf(stderr, "  f%lld t%d c%d: enqueue %.3f .. %.3f ms\n", (long long)q, t, cam, (t1 - t_job0) * 1e3, (now() - t_job0) * 1e3);
        if (pipeline) {
            // this frame is queued; the oldest of the frames in flight is read back once `ahead` younger ones are queued
            // behind it
            in_flight.emplace_back(t, cam);
            if ((int)in_flight.size() > ahead) {
                t1 = now();
                if (writer.valid()) writer.get();
                uint8_t* dst = rgb2[wbuf].data();
                if ((rc = download_prev_rgb(c, dst, rgb_bytes, (int)in_flight.size() - 1))) return bail(rc);
                t_down += now() - t1;
                if (trace) std::fprintf(stderr, "      download of an earlier frame %.3f .. %.3f ms\n", (t1 - t_job0) * 1e3, (now() - t_job0) * 1e3);
                t1 = now();
                write_async(frame_path(in_flight.front().first, in_flight.front().second), dst);
                in_flight.pop_front();
                t_write += now() - t1;
            }
        }
        // the tree of the next timestep becomes current (pointer swap; its helper thread has already
        // copied it to the GPU) while this timestep's last frame is still rendering
        if (q + 1 < n_mine) {
            const int t_next = (int)(my_frames[q + 1] / n_cameras);
            if (t_next != built_t && (rc = advance_scene(t_next))) return rc;
        }
        if (!pipeline) {
            t1 = now();
            if (writer.valid()) writer.get();  // previous file done (it was being written while this frame rendered)
            uint8_t* dst = rgb2[wbuf].data();
            if ((rc = rt_download(c, nullptr, dst, nullptr, nullptr))) return bail(rc);
            t_down += now() - t1;
            t1 = now();
            write_async(frame_path(t, cam), dst);
            t_write += now() - t1;
        }
    }
    while (pipeline && !in_flight.empty()) {  // the last frames, oldest first
        if (writer.valid()) writer.get();
        uint8_t* dst = rgb2[wbuf].data();
        const int age = (int)in_flight.size() - 1;
        rc = age == 0 ? rt_download(c, nullptr, dst, nullptr, nullptr) : download_prev_rgb(c, dst, rgb_bytes, age);
        if (rc) {
            drain();
            return rc;
        }
        write_async(frame_path(in_flight.front().first, in_flight.front().second), dst);
        in_flight.pop_front();
    }
    if (writer.valid()) writer.get();
    if (timing)
        std::fprintf(stderr, "rt_render_scene: %lld frames: wait-for-build %.1f ms, upload %.1f, render-call %.1f, "
                             "download(sync) %.1f, ppm %.1f\n",
                     (long long)n_mine, t_wait * 1e3, t_upload * 1e3, t_render * 1e3, t_down * 1e3, t_write * 1e3);
    // every rank leaves the global stream where the unsharded job would have left it
    return rt_rng_seek(c, base + uint64_t(timestep_count) * uint64_t(n_cameras) * frame_words);
}

}  // extern "C"
