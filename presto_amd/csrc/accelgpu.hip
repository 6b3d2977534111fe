// accelgpu.hip -- C ABI of libaccelgpu.so (see include/accelgpu.h).
//
// Data layout in HBM (one context = one search geometry at a time):
//   fft       complex64[numbins]              the input spectrum (.fft layout)
//   bank      complex64[numz][fftlen]         FFT'd z-response templates
//   blockfft  complex64[nblk][fftlen/2]       forward FFT of each normalised block window, times 1/fftlen
//   plane     float[numz][rlen]               fundamental powers, rlen = 2*(highestbin+7470)
//   hits      {float,u32,i64}[cap]            compacted threshold crossings
// Blocks follow the reference grid: pre-population blocks at r = 8 + k*3735 below
// rlo, search blocks at r = rlo + k*3735.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <deque>
#include <map>
#include <mutex>
#include <vector>

#include "accel_host.hpp"
#include "accel_kernels.hpp"
#include "spectrum_kernels.hpp"
#include "optimize_kernels.hpp"
#include "accelgpu.h"

using accelhost::set_error;
using accelk::c32;
using accelk::DevHit;

namespace {

typedef afft::Plan<16, 16, 16, 256> P4K;
typedef afft::Plan<32, 16, 16, 256> P8K;
typedef afft::Plan<32, 16, 32, 512> P16K;


#define HIPCHK(expr)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,   \
                      __LINE__);                                                         \
            return ACCELGPU_ERR_HIP;                                                     \
        }                                                                                \
    } while (0)

struct Twiddles {
    c32 *tw1 = nullptr, *tw2 = nullptr;
};

struct PowTables { // the two short tables of afft::TwoLevel, per thread column
    c32 *pw1 = nullptr, *pw2 = nullptr;
};

struct EventPair {
    hipEvent_t a, b;
};

// Device allocation that is released on every way out of a function.
template <class T> struct DevTemp {
    T *p = nullptr;
    ~DevTemp() { if (p) (void) hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc((void **) &p, n * sizeof(T)); }
    T *release() { T *q = p; p = nullptr; return q; }
};

} // namespace

// Working memory of accelgpu_fund_blocks_noalloc(), the entry point for callers that
// cannot allocate: block descriptors and the block spectra of up to kNoAllocBlocks blocks
// of the longest plan.  Every GPU the library runs on holds its own copy of this array,
// at its own address, until the process ends.  Host code never uses the array's name as
// an address: accelgpu_create() asks the runtime for the device address once per device
// (with that device current) and keeps it in g_scratch_ptr[device].
constexpr int kNoAllocBlocks = accelk::DESC_MAX;
constexpr size_t kScratchDescBytes = 512;                                  // lobin, col, numdata, ncols
constexpr size_t kScratchBlockfftBytes = (size_t) kNoAllocBlocks * 8192 * sizeof(accelk::c32);
constexpr size_t kScratchBytes = kScratchDescBytes + kScratchBlockfftBytes; // 1 MiB + 512 B
static __device__ __align__(256) unsigned char scratch[kScratchBytes];
constexpr int kMaxDevices = 64;
// A launch's hit counter is word 0 of a small block; words 4..8 tell the host what became of the coarse
// bound's list (written by k_flagged_exact).
constexpr size_t kCountBytes = 64;
static unsigned char *g_scratch_ptr[kMaxDevices];
static std::mutex g_scratch_mutex;

// Launch-shape knobs.  Defaults are the measured optima; accelgpu_create() reads the
// ACCELGPU_* environment overrides once, accelgpu_set_tuning() changes one by name
// (A/B sweeps and tests).  Nothing on the launch path reads the environment.
struct Tuning {
    int corr_rr = 1;        // plane generation: 1 register-twiddle kernel, 0 table form
    int corr_zl = 1;        // register form loops over templates (1) or over blocks (0)
    int corr_chunk = 0;     // transforms per workgroup; 0: 16, fewer on small grids
    int corr_tz = -1;       // XCD tile: owned items per tile; -1: 4 on large grids, plain grid otherwise; 0: plain grid
    int corr_tc = 16;       // XCD tile: chunks per tile
    int corr_lds = 0;       // plane generation: extra bytes of dynamic LDS (caps the workgroups per CU; experiments)
    int corr_pair = 1;      // plane generation, 8192 points: outputs stored in adjacent pairs (0: singly)
    int hs_xcd_remap = 1;   // harmonic sum: each XCD takes one contiguous run of columns
    int hs_lds = 0;         // harmonic sum: bytes of dynamic LDS (caps the waves per CU)
    int hs_lean_from = 6;   // harmonic sum: register-lean form from this many stages on
    int hs_j = -1;          // harmonic sum: column pairs per thread; -1: by number of stages
    int hs_chunk_blocks = 0; // fill + sum in chunks of this many search blocks; 0: whole plane at once
    int std_table = 0;      // standard mode: table-form correlate also at the fundamental's length
    int std_blocks = 0;     // standard mode: search blocks per pass; 0: as many as the scratch holds
    int opt_nt = 512;       // candidate optimisation: threads per workgroup
    int use_graph = 1;      // small searches: one captured graph per launch instead of nine stream operations
    int hs_bound = 1;       // harmonic sum over 16-bit upper bounds of the plane, exact where a bound crosses
    int pf_list_cap = 0;    // entries of the coarse bound's list of (row, group)s; 0: 1/64 of all of them (tests lower it)
    int pf_limit_ppm = 15000; // the coarse bound's list is of no use beyond this many entries per million (row, group)s: the
                            // walk is given up and the bounded sum runs over the whole plane (16 harmonics on a 2^23-point
                            // zmax 200 plane: 1.5 % listed 5.3 ms per search against 6.4, 2.7 % 6.8 against 6.4)
    int pf_sample = 1;      // ask a sample of the (row, group)s how long the list would be before walking them all
    int pf_memory = 1;      // a context whose coarse bound was given up leaves it out of its next trials (4, 8, ... 64 of them)
    int pf_nt = 0;          // threads per workgroup of that kernel: 0 by the number of groups, 256 / 512 / 1024 that many
    int pf_groups = 0;      // groups of four columns per lane in the coarse bound's kernel: 0 two up to 8 harmonics, one with 16; 1 / 2: always,
                            // 3: one group per lane, 0: never
                            // (0: off, float plane only; 1: on; 2 / 3: on, one / two columns per lane forced)
};

struct accelgpu_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    Tuning tune;

    // kernel bank
    int zmax = -1, numz = 0, fftlen = 0, halfwidth = 0;
    c32 *d_bank = nullptr;
    c32 *d_bankf = nullptr; // the bank folded for the register-form correlate kernels (k_fold_bank): second half of d_bank's allocation
    Twiddles tw_inv, tw_fwd_full, tw_fwd_half;
    PowTables pw_inv;

    // spectrum
    // d_fft is the spectrum the next launch reads: a caller's device buffer or one of
    // the library's own.  There are several of those so that the next trial can be
    // uploaded and prepared while queued searches still read theirs and while the
    // candidates of the last collected search are optimised against its spectrum.
    static constexpr int kOwn = 3;
    const c32 *d_fft = nullptr;
    c32 *d_own[kOwn] = {nullptr, nullptr, nullptr};
    size_t own_cap[kOwn] = {0, 0, 0};
    int cur_own = -1;                                // index of d_fft in d_own, -1: caller's buffer
    // accelgpu_load_fft_async(): the copy into d_own[j] runs on the side stream; own_ready[j]
    // is recorded behind it and own_pending[j] stays set until the host has waited for it
    hipEvent_t own_ready[kOwn] = {nullptr, nullptr, nullptr};
    bool own_pending[kOwn] = {false, false, false};
    int64_t numbins = 0;
    struct {
        const c32 *fft = nullptr;
        int64_t numbins = 0;
        bool valid = false;
    } collected;                                     // spectrum of the most recently collected search
    hipStream_t side_stream = nullptr;               // uploads and preparation beside queued searches (see SideScope)
    hipStream_t opt_stream = nullptr;                // candidate optimisation beside them: its own stream, so the
                                                     // next trial's upload does not queue behind it

    // one candidate optimisation between its launch and its collect
    struct OptJob {
        bool active = false;
        int polish = 0;
        int lobin = 0;
        size_t np = 0;                               // device problems (one per candidate or per harmonic)
        const c32 *fft = nullptr;                    // the spectrum the queued kernel reads
        std::vector<accelgpu_cand> cands;
        std::vector<float> sigma;                    // per candidate, from the set-up at launch
        optk::Problem *h_prob = nullptr;             // pinned
        optk::Result *h_res = nullptr;               // pinned
        size_t h_cap = 0;
        hipEvent_t done = nullptr;
    } opt;

    // search geometry
    bool have_search = false;
    accelgpu_setup setup{};
    int norm_type = 0;
    float nph = 0.f;
    int nblk = 0, nprepop = 0, nsearch = 0;
    int nblk_active = 0;      // blocks generated by fill (all of them unless a block range is set)
    bool partial_fill = false; // a block range is set: a fill generates some blocks, or some rows of them, only
    int range_first = 0, range_count = 0; // search blocks this context searches
    std::vector<int64_t> h_blk_lobin, h_blk_col; // host copies of the full block grid
    std::vector<int> h_blk_numdata, h_blk_ncols;
    accelhost::BlockGrid grid_host;
    int2 *d_blk_zrange = nullptr; // per generated block: the z rows [x, y) a fill computes (a block range is set)
    int64_t col0 = 0, ncols = 0; // first search column (2*rlo) and number of search columns
    int64_t *d_blk_lobin = nullptr, *d_blk_col = nullptr;
    int *d_blk_numdata = nullptr, *d_blk_ncols = nullptr;
    size_t blk_cap = 0;
    c32 *d_blockfft = nullptr;
    size_t blockfft_cap = 0;
    float *d_plane = nullptr;
    size_t plane_cap = 0;
    // The plane in its other layout (hs_bound != 0): two planes of 16-bit values, the upper and the
    // lower halves of every power's float, instead of d_plane.  The harmonic sum reads the upper
    // halves as bounds (see k_harmsum_b); d_flags[0] is set when a host writes a negative power.
    bool split = false;
    unsigned short *d_hi = nullptr, *d_lo = nullptr;
    size_t hilo_cap = 0;               // cells in each
    unsigned int *d_flags = nullptr;
    bool host_wrote_outside = false;   // a host wrote plane cells that a whole plane generation does not rewrite
    int64_t *d_rowoff = nullptr; // [numz][NH] byte offsets of the sub-harmonic source rows
    size_t rowoff_cap = 0;
    uint32_t *d_rowchg = nullptr; // [numz] bit i: source row of harmonic i differs from row zi-1
    size_t rowchg_cap = 0;
    int64_t *d_rowhalf = nullptr; // [numz][NH] the same source rows as byte offsets into a plane of 16-bit halves
    // the coarse bound in front of the bounded sum (k_prefilter): the plane of group maxima, the source
    // rows as byte offsets into it, the list of (row, group)s to sum exactly and its two control words
    unsigned short *d_mx = nullptr;
    size_t mx_cap = 0;
    int64_t *d_rowmx = nullptr;
    size_t rowmx_cap = 0;
    accelk::WorkItem *d_work = nullptr;
    size_t work_cap = 0;
    unsigned int *d_ctl = nullptr;
    accelk::GroupGrid grid{};
    bool pf_ready = false;    // the three buffers above are sized for the search that is set
    bool mx_valid = false;    // the maxima describe the plane as it is (written by the last whole plane generation)
    bool mx_fix_pending = false; // ... once k_group_fix has run
    // What the coarse bound's last outcomes say about the next trial (the trials of one batch are alike): after a
    // launch whose list was given up, the next pf_backoff launches go without it, then one tries again.
    int pf_backoff = 0, pf_skip = 0;
    bool pf_off = false;      // this launch goes without (decided when it is launched)
    int64_t pf_last_listed = 0; // the last collected launch: entries listed, -1 given up, -2 the coarse bound did not run
    size_t rowhalf_cap = 0;

    // hits
    // Two hit buffers so that one trial's hits can be collected while the next
    // trial's kernels are already queued.  d_hits/h_hits/d_count/h_count alias
    // the slot currently being launched into or collected from.
    struct HitSlot {
        DevHit *d_hits = nullptr, *h_hits = nullptr; // device buffer, pinned host mirror
        unsigned int *d_count = nullptr, *h_count = nullptr;
        hipEvent_t done = nullptr;                   // recorded after the slot's last copy
        hipEvent_t kernels = nullptr;                // recorded after the slot's last kernel
        const c32 *fft = nullptr;                    // the spectrum this launch reads
        int64_t numbins = 0;
        size_t prefetch = 0;                         // hits copied back speculatively at launch
        size_t cap = 0;                              // room in d_hits / h_hits
        uint64_t serial = 0;                         // which launch this is (see plane_serial)
        bool pf_counted = false;                     // the coarse bound's outcome has been taken note of
    } slots[2];
    int head = 0, tail = 0, inflight = 0;            // FIFO of launched, uncollected trials
    uint64_t launch_serial = 0;                      // launches so far
    uint64_t plane_serial = 0;                       // the launch whose fundamental the plane holds
    // Small searches are launch-bound: their whole launch (count reset, block preparation,
    // plane generation, harmonic sum, copies of the count and of the first hits) is captured
    // once per (hit slot, spectrum buffer) and replayed with one call.
    struct LaunchGraph {
        int slot = 0;
        bool pf_off = false;
        bool mx_valid_after = false, mx_fix_after = false; // the state of the group maxima once the graph has run
        const c32 *fft = nullptr;
        int64_t numbins = 0;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        uint64_t used = 0;
    };
    std::vector<LaunchGraph> graphs;
    uint64_t graph_clock = 0;
    hipStream_t copy_stream = nullptr;               // late D2H copies that must not queue behind kernels
    DevHit *d_hits = nullptr;
    DevHit *h_hits = nullptr;
    unsigned int *d_count = nullptr, *h_count = nullptr;
    size_t hits_cap = 0;                             // of the selected slot

    // scratch for the single-block getters
    float *d_tmp = nullptr;
    size_t tmp_cap = 0;
    int64_t *d_one_lobin = nullptr, *d_one_col = nullptr;
    int *d_one_numdata = nullptr, *d_one_ncols = nullptr;
    c32 *d_one_blockfft = nullptr;

    // standard (not in-memory) mode: no plane; per-harmonic template banks,
    // twiddle tables per transform length, block scratch
    bool inmem = true;
    struct SubBank {
        int numharm = 0, harmnum = 0, zmax = 0, numkern = 0;
        int fftlen = 0;    // the reference's calc_fftlen()
        int xlen = 0;      // transform length used here (>= fftlen)
        int halfwidth = 0; // kern[0].kern_half_width
        c32 *d_bank = nullptr;
        c32 *d_fold = nullptr; // folded copy (banks of the fundamental's length only), in the same allocation
        bool owned = false;
    };
    std::deque<SubBank> subbanks;
    std::map<int, Twiddles> tw_cache; // key 2*n + (dir > 0)
    static constexpr int kStdBufs = 12;
    struct StdBuf {
        void *p = nullptr;
        size_t cap = 0;
    } std_buf[kStdBufs];
    // chirp-z filter spectrum for time series whose length is not a power of two
    c32 *d_blu_filter = nullptr;
    size_t blu_cap = 0;
    int64_t blu_L = 0;
    // red-noise filter tables currently on the device
    struct RedKey {
        int64_t numamps;
        int startwidth, endwidth, program_form, pad;
        double endfreq, T;
    } red_key{};
    bool red_valid = false;
    int red_nb = 0, red_ns = 0;
    size_t red_off[4] = {0, 0, 0, 0};

    // profiling
    bool profiling = false;
    std::vector<EventPair> ev[3];
    std::vector<EventPair> ev_pool;
    int64_t prof_launches[3] = {0, 0, 0};
    double prof_ms[3] = {0, 0, 0};
};

namespace {

void free_subbanks(accelgpu_ctx *ctx);
bool want_split(const accelgpu_ctx *ctx, int numharmstages);
int set_search_impl(accelgpu_ctx *ctx, const accelgpu_setup *s, int norm_type, float nph, bool inmem);

int use(accelgpu_ctx *ctx)
{
    if (!ctx) {
        set_error("null context");
        return ACCELGPU_ERR_ARG;
    }
    HIPCHK(hipSetDevice(ctx->device));
    return ACCELGPU_OK;
}

// Captured launches hold kernel arguments (buffers, geometry, knobs): anything that changes
// one of those drops them all, and the next launch captures afresh.
void drop_graphs(accelgpu_ctx *ctx)
{
    for (auto &g : ctx->graphs) {
        if (g.exec) (void) hipGraphExecDestroy(g.exec);
        if (g.graph) (void) hipGraphDestroy(g.graph);
    }
    ctx->graphs.clear();
}

template <class T> int dev_reserve(T **p, size_t *cap, size_t want, bool zero, hipStream_t st)
{
    if (*cap >= want && *p)
        return ACCELGPU_OK;
    if (*p)
        HIPCHK(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    hipError_t e = hipMalloc((void **) p, want * sizeof(T));
    if (e != hipSuccess) {
        set_error("hipMalloc of %zu bytes failed: %s", want * sizeof(T), hipGetErrorString(e));
        return ACCELGPU_ERR_NOMEM;
    }
    *cap = want;
    if (zero)
        HIPCHK(hipMemsetAsync(*p, 0, want * sizeof(T), st));
    return ACCELGPU_OK;
}

// Spectrum loading and preparation and the candidate optimisation do not touch what a
// queued in-memory search uses.  With searches in flight they run on the side stream,
// overlapping the searches instead of waiting behind them; every such call still ends
// with a host wait for its own work, so launches that follow see it complete.
struct SideScope {
    accelgpu_ctx *ctx;
    hipStream_t saved;
    explicit SideScope(accelgpu_ctx *c, bool optimisation = false) : ctx(c), saved(c->stream)
    {
        if (c->inflight > 0)
            c->stream = optimisation ? c->opt_stream : c->side_stream;
    }
    ~SideScope() { ctx->stream = saved; }
};

// Does queued device work (a launched search, a launched optimisation) read spectrum p?
bool in_flight_reads(const accelgpu_ctx *ctx, const c32 *p)
{
    if (!p)
        return false;
    for (int k = 0, si = ctx->head; k < ctx->inflight; k++, si ^= 1)
        if (ctx->slots[si].fft == p)
            return true;
    return ctx->opt.active && ctx->opt.fft == p;
}

// Which of the library's buffers the next upload goes to.
int pick_own(accelgpu_ctx *ctx, int *idx)
{
    const int cur = std::max(ctx->cur_own, 0);
    if (ctx->inflight == 0 && !in_flight_reads(ctx, ctx->d_own[cur])) {
        *idx = cur;
    } else {
        int best = -1;
        for (int pass = 0; pass < 2 && best < 0; pass++)
            for (int j = 0; j < accelgpu_ctx::kOwn && best < 0; j++) {
                const int k = (std::max(ctx->cur_own, 0) + j) % accelgpu_ctx::kOwn; // current first
                if (in_flight_reads(ctx, ctx->d_own[k]))
                    continue;
                if (pass == 0 && ctx->collected.valid && ctx->d_own[k] && ctx->d_own[k] == ctx->collected.fft)
                    continue; // keep the collected search's spectrum for its optimisation if possible
                best = k;
            }
        if (best < 0) {
            set_error("every spectrum buffer is read by a launched search or optimisation; collect one first");
            return ACCELGPU_ERR_ARG;
        }
        *idx = best;
    }
    if (ctx->d_own[*idx] && ctx->collected.fft == ctx->d_own[*idx])
        ctx->collected.valid = false;
    if (ctx->own_pending[*idx]) { // an asynchronous upload into this buffer that nothing waited for
        HIPCHK(hipEventSynchronize(ctx->own_ready[*idx]));
        ctx->own_pending[*idx] = false;
    }
    return ACCELGPU_OK;
}

template <class P> int make_twiddles(Twiddles *tw, int dir, hipStream_t st)
{
    std::vector<c32> t1((size_t) P::R1 * P::M1), t2((size_t) P::R2 * P::R3);
    for (int n1 = 0; n1 < P::R1; n1++)
        for (int b = 0; b < P::M1; b++) {
            const double a = dir * 2.0 * M_PI * (double) ((long long) b * n1 % P::N) / (double) P::N;
            t1[(size_t) n1 * P::M1 + b] = c32{(float) std::cos(a), (float) std::sin(a)};
        }
    for (int e = 0; e < P::R2; e++)
        for (int d = 0; d < P::R3; d++) {
            const double a = dir * 2.0 * M_PI * (double) (e * d % P::M1) / (double) P::M1;
            t2[(size_t) e * P::R3 + d] = c32{(float) std::cos(a), (float) std::sin(a)};
        }
    if (tw->tw1) HIPCHK(hipFree(tw->tw1));
    if (tw->tw2) HIPCHK(hipFree(tw->tw2));
    HIPCHK(hipMalloc((void **) &tw->tw1, t1.size() * sizeof(c32)));
    HIPCHK(hipMalloc((void **) &tw->tw2, t2.size() * sizeof(c32)));
    HIPCHK(hipMemcpyAsync(tw->tw1, t1.data(), t1.size() * sizeof(c32), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(tw->tw2, t2.data(), t2.size() * sizeof(c32), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st)); // host vectors die at return
    return ACCELGPU_OK;
}

template <class P> int make_pow_tables(PowTables *pw, int dir, hipStream_t st)
{
    // rows: w^(4*hi), hi = 1..R/4-1, then w^lo, lo = 1..3 (see afft::TwoLevel)
    const int n1 = P::R1 / 4 + 2, n2 = P::R2 / 4 + 2;
    std::vector<c32> t1((size_t) n1 * P::M1), t2((size_t) n2 * P::R3);
    auto expo = [](int row, int R) { return row < R / 4 - 1 ? 4 * (row + 1) : row - (R / 4 - 1) + 1; };
    for (int k = 0; k < n1; k++)
        for (int b = 0; b < P::M1; b++) {
            const double a = dir * 2.0 * M_PI * (double) (((long long) b * expo(k, P::R1)) % P::N) / (double) P::N;
            t1[(size_t) k * P::M1 + b] = c32{(float) std::cos(a), (float) std::sin(a)};
        }
    for (int k = 0; k < n2; k++)
        for (int d = 0; d < P::R3; d++) {
            const double a = dir * 2.0 * M_PI * (double) ((d * expo(k, P::R2)) % P::M1) / (double) P::M1;
            t2[(size_t) k * P::R3 + d] = c32{(float) std::cos(a), (float) std::sin(a)};
        }
    if (pw->pw1) HIPCHK(hipFree(pw->pw1));
    if (pw->pw2) HIPCHK(hipFree(pw->pw2));
    HIPCHK(hipMalloc((void **) &pw->pw1, t1.size() * sizeof(c32)));
    HIPCHK(hipMalloc((void **) &pw->pw2, t2.size() * sizeof(c32)));
    HIPCHK(hipMemcpyAsync(pw->pw1, t1.data(), t1.size() * sizeof(c32), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(pw->pw2, t2.data(), t2.size() * sizeof(c32), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    return ACCELGPU_OK;
}

template <class K> int allow_lds(K kernel, int bytes)
{
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return ACCELGPU_OK;
}

int prof_begin(accelgpu_ctx *ctx, int which, EventPair *ep)
{
    if (!ctx->profiling)
        return ACCELGPU_OK;
    if (!ctx->ev_pool.empty()) {
        *ep = ctx->ev_pool.back();
        ctx->ev_pool.pop_back();
    } else {
        HIPCHK(hipEventCreate(&ep->a));
        HIPCHK(hipEventCreate(&ep->b));
    }
    HIPCHK(hipEventRecord(ep->a, ctx->stream));
    (void) which;
    return ACCELGPU_OK;
}

int prof_end(accelgpu_ctx *ctx, int which, EventPair *ep)
{
    if (!ctx->profiling)
        return ACCELGPU_OK;
    HIPCHK(hipEventRecord(ep->b, ctx->stream));
    ctx->ev[which].push_back(*ep);
    return ACCELGPU_OK;
}

// Block descriptors exactly as subharm_ffdot_plane() derives them for
// numharm = harmnum = 1 (reference src/accel_utils.c:903-907, 929-934).
struct BlockDesc {
    int rlo;
    int64_t lobin;
    int numdata;
};

BlockDesc describe_block(double startr, int halfwidth)
{
    const double fullrhi = startr + ACCELGPU_USELEN * 0.5 - 0.5;
    const double drlo = accelhost::calc_required_r(1.0, startr);
    const double drhi = accelhost::calc_required_r(1.0, fullrhi);
    BlockDesc d;
    d.rlo = (int) std::floor(drlo);
    d.lobin = (int64_t) d.rlo - halfwidth;
    const int hibin = (int) std::ceil(drhi) + halfwidth;
    d.numdata = (int) (hibin - d.lobin + 1);
    return d;
}

int launch_block_prep(accelgpu_ctx *ctx, int nblk, const int64_t *d_lobin, const int *d_numdata,
                      c32 *d_out, const int2 *d_zrange = nullptr)
{
    const double nph_norm = ctx->nph > 0.f ? 1.0 / std::sqrt((double) ctx->nph) : 0.0;
    EventPair ep;
    int rc = prof_begin(ctx, 0, &ep);
    if (rc) return rc;
    if (ctx->fftlen == 8192) {
        hipLaunchKernelGGL(accelk::k_block_prep<P4K>, dim3(nblk), dim3(P4K::NT),
                           P4K::LDS_BYTES + 64, ctx->stream, ctx->d_fft, ctx->numbins,
                           ctx->setup.lobin, d_lobin, d_numdata, d_out, nph_norm,
                           ctx->norm_type == 1 ? 1 : 0, (const int *) nullptr,
                           ctx->tw_fwd_half.tw1, ctx->tw_fwd_half.tw2, d_zrange);
    } else {
        hipLaunchKernelGGL(accelk::k_block_prep<P8K>, dim3(nblk), dim3(P8K::NT),
                           P8K::LDS_BYTES + 64, ctx->stream, ctx->d_fft, ctx->numbins,
                           ctx->setup.lobin, d_lobin, d_numdata, d_out, nph_norm,
                           ctx->norm_type == 1 ? 1 : 0, (const int *) nullptr,
                           ctx->tw_fwd_half.tw1, ctx->tw_fwd_half.tw2, d_zrange);
    }
    HIPCHK(hipGetLastError());
    return prof_end(ctx, 0, &ep);
}

// Blocks per correlate workgroup (they share the template and twiddle registers): the
// preferred number, fewer when templates x blocks would not give every CU several
// workgroups (one template row and 140 blocks at 16 per workgroup is 9 workgroups).
int corr_chunk(int preferred, int numz, int nblk)
{
    const long long work = (long long) numz * nblk;
    return (int) std::max<long long>(1, std::min<long long>(preferred, work / 2048));
}

// Grid of the register-twiddle correlate kernel over numz templates x nchunks chunks of
// blocks.  Large grids are one-dimensional and tiled: the hardware deals consecutive
// workgroups round-robin over the 8 XCDs, and the kernel maps them so that the workgroups
// one XCD holds at a time share tile_z template rows and tile_c chunks (4 x 16 measured
// best on a 201 x 71 grid: 3.08 against 3.24 ms; 8 x 8 3.11, 2 x 32 3.08, 16 x 16 3.26).
// tile_z = 0: plain two-dimensional grid.  Tuning corr_tz / corr_tc override.
dim3 corr_grid(const Tuning &tn, int numz, int nchunks, int *tile_z, int *tile_c)
{
    int tz = (long long) numz * nchunks >= 4096 ? 4 : 0;
    if (tn.corr_tz >= 0) tz = tn.corr_tz;
    const int tc = std::max(1, tn.corr_tc);
    *tile_z = tz;
    *tile_c = tc;
    if (tz <= 0)
        return dim3((unsigned) numz, (unsigned) nchunks);
    const long long tiles = (long long) ((numz + tz - 1) / tz) * ((nchunks + tc - 1) / tc);
    return dim3((unsigned) (((tiles + 7) / 8) * 8 * tz * tc));
}

// Output: rows of `rlen` floats at d_plane, or -- d_hi, d_lo given -- rows of the two planes of
// 16-bit halves (register form looping over templates only; see k_harmsum_b).
int launch_correlate(accelgpu_ctx *ctx, int nblk, const c32 *d_blockfft, const int64_t *d_col,
                     const int *d_ncols, float *d_plane, int64_t rlen, unsigned short *d_hi = nullptr,
                     unsigned short *d_lo = nullptr, unsigned short *d_mx = nullptr, int64_t mxlen = 0,
                     const int2 *d_zrange = nullptr)
{
    // Register-resident twiddle form for both plans (measured on one device: 3.3 against
    // 5.8 ms per 2^23-point zmax 200 search, 83 against 116 ms per 2^24-point zmax 1200
    // search).  Tuning corr_rr = 0 selects the table form for A/B runs.
    const Tuning &tn = ctx->tune;
    const bool rr = tn.corr_rr != 0 || d_hi || d_zrange; // (only the register form writes split planes and takes row ranges)
    int chunk = corr_chunk(rr ? 16 : 8, ctx->numz, nblk);   // blocks per workgroup (templates in the ZL form)
    if (tn.corr_chunk > 0)
        chunk = tn.corr_chunk;
    const dim3 grid(ctx->numz, (nblk + chunk - 1) / chunk);
    if (2 * ctx->halfwidth > 2048) { // the kernels' range-checked stores assume this
        set_error("template half width %d is beyond what the correlate kernels store", ctx->halfwidth);
        return ACCELGPU_ERR_ARG;
    }
    EventPair ep;
    int rc = prof_begin(ctx, 1, &ep);
    if (rc) return rc;
    // Register form: a workgroup owns one block and loops over a chunk of templates (the
    // block's spectrum stays in registers: a third less L2->L1 traffic per transform than
    // one template looping over blocks; 2.67 against 2.78 ms on a 201 x 1123 grid, chunks
    // of 16 templates against 8: 2.75, 32: 2.81).  Tuning corr_zl = 0 selects the other loop.
    const bool zl = tn.corr_zl != 0 || d_hi || d_zrange; // (row ranges per block: only the template loop knows them)
    int tile_z = 0, tile_c = 0;
    dim3 rgrid;
    if (zl) {
        chunk = corr_chunk(16, nblk, ctx->numz);
        if (tn.corr_chunk > 0)
            chunk = tn.corr_chunk;
        rgrid = corr_grid(tn, nblk, (ctx->numz + chunk - 1) / chunk, &tile_z, &tile_c);
    } else {
        rgrid = corr_grid(tn, ctx->numz, (int) grid.y, &tile_z, &tile_c);
    }
#define LAUNCH_RR(PLAN, ZL, SPLIT, PAIR)                                                       \
    hipLaunchKernelGGL((accelk::k_correlate_rr<PLAN, false, 16, ZL, afft::NoStamp, SPLIT, PAIR>), rgrid, dim3(PLAN::NT),   \
                       PLAN::LDS_BYTES + std::min(std::max(tn.corr_lds, 0), 20000), ctx->stream, ctx->d_bankf, d_blockfft, d_col, d_ncols,   \
                       nblk, chunk, d_plane, rlen, ctx->halfwidth, ctx->pw_inv.pw1,            \
                       ctx->pw_inv.pw2, ctx->numz, tile_z, tile_c, afft::NoStamp(), d_hi, d_lo, (unsigned short *) nullptr, (int64_t) 0, zl ? d_zrange : nullptr)
    // 8192 points, loop over templates: the last pass hands a thread adjacent outputs, stored two
    // at a time (corr_pair = 0: one at a time, for comparisons; the powers are the same bits)
    const bool pair = tn.corr_pair != 0;
    if (rr && zl && ctx->fftlen == 8192 && pair && d_hi && d_mx) {
        // with the largest upper half of every four outputs, for the coarse bound in front of the harmonic sum
        hipLaunchKernelGGL((accelk::k_correlate_rr<P8K, false, 16, true, afft::NoStamp, true, true, true>), rgrid, dim3(P8K::NT),
                           P8K::LDS_BYTES + accelk::kMaximaLdsBytes, ctx->stream, ctx->d_bankf, d_blockfft, d_col,
                           d_ncols, nblk, chunk, d_plane, rlen, ctx->halfwidth, ctx->pw_inv.pw1, ctx->pw_inv.pw2, ctx->numz,
                           tile_z, tile_c, afft::NoStamp(), d_hi, d_lo, d_mx, mxlen, d_zrange);
    } else if (rr && zl && ctx->fftlen == 16384 && d_hi && d_mx) {
        // the same from the 16384-point plan's single stores (a group's four outputs in four neighbouring lanes)
        hipLaunchKernelGGL((accelk::k_correlate_rr<P16K, false, 16, true, afft::NoStamp, true, false, true>), rgrid, dim3(P16K::NT),
                           P16K::LDS_BYTES + accelk::kMaximaLdsBytes16k, ctx->stream, ctx->d_bankf, d_blockfft, d_col,
                           d_ncols, nblk, chunk, d_plane, rlen, ctx->halfwidth, ctx->pw_inv.pw1, ctx->pw_inv.pw2, ctx->numz,
                           tile_z, tile_c, afft::NoStamp(), d_hi, d_lo, d_mx, mxlen, d_zrange);
    } else if (rr && zl && ctx->fftlen == 8192 && pair) {
        if (d_hi) LAUNCH_RR(P8K, true, true, true);
        else LAUNCH_RR(P8K, true, false, true);
    } else if (rr && zl && d_hi) {
        if (ctx->fftlen == 8192) LAUNCH_RR(P8K, true, true, false);
        else LAUNCH_RR(P16K, true, true, false);
    } else if (rr && ctx->fftlen == 8192) {
        if (zl) LAUNCH_RR(P8K, true, false, false);
        else LAUNCH_RR(P8K, false, false, false);
    } else if (rr) {
        if (zl) LAUNCH_RR(P16K, true, false, false);
        else LAUNCH_RR(P16K, false, false, false);
#undef LAUNCH_RR
    } else if (ctx->fftlen == 8192) {
        hipLaunchKernelGGL((accelk::k_correlate<P8K, false>), grid, dim3(P8K::NT), P8K::LDS_BYTES,
                           ctx->stream, ctx->d_bank, d_blockfft, d_col, d_ncols, nblk, chunk,
                           d_plane, rlen, ctx->halfwidth, ctx->tw_inv.tw1, ctx->tw_inv.tw2);
    } else {
        hipLaunchKernelGGL((accelk::k_correlate<P16K, false>), grid, dim3(P16K::NT), P16K::LDS_BYTES,
                           ctx->stream, ctx->d_bank, d_blockfft, d_col, d_ncols, nblk, chunk,
                           d_plane, rlen, ctx->halfwidth, ctx->tw_inv.tw1, ctx->tw_inv.tw2);
    }
    HIPCHK(hipGetLastError());
    return prof_end(ctx, 1, &ep);
}

// Is the plane kept as two planes of 16-bit halves (and the harmonic sum the bounded one)?
bool bounds_in_use(const accelgpu_ctx *ctx)
{
    return ctx->inmem && ctx->split && ctx->d_hi && ctx->d_lo;
}

// Where a harmonic-sum launch compacts its threshold crossings.
struct HitTarget {
    DevHit *hits;
    unsigned int *count;
    size_t cap;
};

// J = column pairs per thread of the harmonic-sum kernel.
template <bool DUMP, int J>
int launch_harmsum_j(accelgpu_ctx *ctx, int nstages, int64_t col0, int64_t ncols, float *dump,
                     int dump_stage, const HitTarget &out, bool timed)
{
    accelk::HarmParams prm;
    for (int i = 0; i < 6; i++)
        prm.powcut[i] = i < ctx->setup.numharmstages ? ctx->setup.powcut[i] : INFINITY;
    constexpr int C = J > 0 ? 2 * J : 1;
    const int remap = ctx->tune.hs_xcd_remap ? 1 : 0;
    const int lds_pad = ctx->tune.hs_lds; // occupancy experiments
    unsigned ntiles = (unsigned) ((ncols + 256 * C - 1) / (256 * C));
    if (remap)
        ntiles = ((ntiles + 7) / 8) * 8;
    const dim3 grid(ntiles);
    const unsigned int cap = (unsigned int) std::min<size_t>(out.cap, 0xFFFFFFFFu);
    EventPair ep;
    timed = timed && !DUMP;
    if (timed) {
        int rc = prof_begin(ctx, 2, &ep);
        if (rc) return rc;
    }
    // the offset table is laid out for the configured number of stages; a dump
    // of an earlier stage still runs the full-depth kernel and picks its stage
    if (nstages != ctx->setup.numharmstages) {
        set_error("harmonic-sum launch for %d stages but the table holds %d", nstages,
                  ctx->setup.numharmstages);
        return ACCELGPU_ERR_ARG;
    }
#define LAUNCH_HS(NS)                                                                        \
    hipLaunchKernelGGL((accelk::k_harmsum<NS, C, DUMP>), grid, dim3(256), lds_pad, ctx->stream,     \
                       ctx->d_plane, ctx->setup.rlen, ctx->numz, col0, ncols, ctx->d_rowoff,   \
                       ctx->d_rowchg,                                                         \
                       prm, out.hits, out.count, cap, dump, dump_stage, remap)
#define LAUNCH_HS_LEAN(NS)                                                                   \
    hipLaunchKernelGGL((accelk::k_harmsum_lean<NS, DUMP>), grid, dim3(256), lds_pad, ctx->stream,   \
                       ctx->d_plane, ctx->setup.rlen, ctx->numz, col0, ncols, ctx->d_rowoff,   \
                       ctx->d_rowchg,                                                         \
                       prm, out.hits, out.count, cap, dump, dump_stage, remap)
    // One column per thread and 31 sub-harmonics: the register-lean form keeps 7 waves
    // per SIMD in flight instead of 4 (measured 58 -> 78-80 % of the HBM peak); with 15
    // or fewer the plain form already runs 7-8 waves and the two time the same.
    const bool lean = C == 1 && nstages >= ctx->tune.hs_lean_from && nstages >= 3;
    switch (nstages) {
    case 1: LAUNCH_HS(1); break;
    case 2: LAUNCH_HS(2); break;
    case 3: if (lean) LAUNCH_HS_LEAN(3); else LAUNCH_HS(3); break;
    case 4: if (lean) LAUNCH_HS_LEAN(4); else LAUNCH_HS(4); break;
    case 5: if (lean) LAUNCH_HS_LEAN(5); else LAUNCH_HS(5); break;
    case 6: if (lean) LAUNCH_HS_LEAN(6); else LAUNCH_HS(6); break;
    default:
        set_error("bad number of harmonic stages %d", nstages);
        return ACCELGPU_ERR_ARG;
    }
#undef LAUNCH_HS
#undef LAUNCH_HS_LEAN
    HIPCHK(hipGetLastError());
    if (timed)
        return prof_end(ctx, 2, &ep);
    return ACCELGPU_OK;
}

// The plane of group maxima, the row table into it and the work list, sized for the search that is set.
int ensure_prefilter(accelgpu_ctx *ctx)
{
    if (ctx->pf_ready)
        return ACCELGPU_OK;
    const accelgpu_setup &su = ctx->setup;
    accelk::GroupGrid g{};
    g.prepop_col0 = ctx->nprepop > 0 ? ctx->h_blk_col[0] : ctx->col0;
    g.search_col0 = ctx->col0;
    g.nprepop = ctx->nprepop;
    g.nblk = ctx->nblk;
    g.a = (2 * ctx->halfwidth) & 3;
    g.mxlen = (int64_t) ctx->nblk * accelk::kGpb;
    ctx->grid = g;
    int rc;
    const size_t mx_elems = (size_t) ctx->numz * (size_t) g.mxlen + 64;
    if ((rc = dev_reserve(&ctx->d_mx, &ctx->mx_cap, mx_elems, false, ctx->stream))) return rc;
    const int nh = (1 << (su.numharmstages - 1)) - 1;
    if (nh > 0) {
        std::vector<int64_t> half((size_t) nh * ctx->numz), tab(half.size());
        HIPCHK(hipMemcpy(half.data(), ctx->d_rowhalf, half.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
        // a source's row in the plane it is bounded from, as an address (the walk spends no scalar additions on it): the
        // plane of upper halves for harmonics up to 1/4, the plane of maxima for the others (accelk::pf_from_cells).
        // Both planes stay where they are until the search is set again, which builds this table again.
        std::vector<char> cells;
        for (int s = 1; s < su.numharmstages; s++)
            for (int h = 1; h < (1 << s); h += 2)
                cells.push_back(accelk::pf_from_cells(s, h));
        std::vector<uint32_t> chg((size_t) ctx->numz);
        HIPCHK(hipMemcpy(chg.data(), ctx->d_rowchg, chg.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        tab.assign((size_t) (nh + 1) * ctx->numz, 0);
        for (int zi = 0; zi < ctx->numz; zi++) {
            for (int i = 0; i < nh; i++) {
                const int64_t hb = half[(size_t) zi * nh + i];
                tab[(size_t) zi * (nh + 1) + i] = cells[(size_t) i] ? (int64_t) (uintptr_t) ctx->d_hi + hb
                                                                  : (int64_t) (uintptr_t) ctx->d_mx + hb / (2 * su.rlen) * (2 * g.mxlen);
            }
            tab[(size_t) zi * (nh + 1) + nh] = (int64_t) chg[(size_t) zi]; // which sources' rows moved since the row before
        }
        if ((rc = dev_reserve(&ctx->d_rowmx, &ctx->rowmx_cap, tab.size(), false, ctx->stream))) return rc;
        HIPCHK(hipMemcpy(ctx->d_rowmx, tab.data(), tab.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    const size_t ngroups = (size_t) ctx->nsearch * accelk::kGpb;
    // (pf_list_cap above the default makes room for more: measurements of what long lists cost)
    const size_t want = std::max<size_t>(std::max<size_t>((size_t) 1 << 20, (size_t) ctx->numz * ngroups / 64),
                                         (size_t) std::max(ctx->tune.pf_list_cap, 0));
    if ((rc = dev_reserve(&ctx->d_work, &ctx->work_cap, want, false, ctx->stream))) return rc;
    if (!ctx->d_ctl)
        HIPCHK(hipMalloc((void **) &ctx->d_ctl, accelk::kPfCtlWords * sizeof(unsigned int)));
    ctx->pf_ready = true;
    return ACCELGPU_OK;
}

// Does the coarse bound go in front of the bounded sum of this search?  (The 8192-point paired plane
// generation writes the group maxima; planes too small to fill the device with a quarter of the
// lanes, two-stage searches and chunked sums keep the plain bounded sum.)  hs_bound = 4 forces it where
// it can run at all (the maxima are then taken from the plane if plane generation did not write them).
// The shape part of that question (accelgpu_coarse_bound_applies).
bool coarse_bound_shape_ok(int numharmstages, int64_t search_columns, int64_t rlen, int fftlen, bool forced)
{
    if (numharmstages < 2 || numharmstages > (forced ? 6 : 5))
        return false;
    if (rlen <= 0 || rlen >= ((int64_t) 1 << 32))
        return false; // its kernels count columns, groups and entries in 32 bits (the sums proper index in 64)
    if (forced)
        return true; // (the maxima are then taken from the plane if plane generation does not write them)
    // 2 to 16 harmonics: with 32 the sum of 32 maxima is too loose a bound (most groups are listed).  (With 2 and 4 it
    // pays since a lane sums two adjacent groups from 8-byte loads: 0.40 against 0.93 ms and 0.48 against 1.27 ms
    // on a 2^23-point, zmax 200 plane, for 0.15-0.19 ms more of plane generation.)
    // 16384-point transforms (zmax above 300) write the maxima from single stores, which costs plane generation a
    // tenth to a sixth more: 16 and 8 harmonics gain by it (zmax 400, 2^23 points: 18.2 -> 15.0 and 15.3 -> 13.8 ms per
    // search; zmax 1200, 16 harmonics: 54.6 -> 45.6), fewer do not
    return search_columns >= ((int64_t) 1 << 22) && (fftlen == 8192 || (fftlen == 16384 && numharmstages >= 4));
}
bool prefilter_wanted(const accelgpu_ctx *ctx)
{
    if (!bounds_in_use(ctx) || ctx->tune.hs_chunk_blocks > 0)
        return false;
    const bool forced = ctx->tune.hs_bound == 4;
    if (!coarse_bound_shape_ok(ctx->setup.numharmstages, ctx->ncols, ctx->setup.rlen, ctx->fftlen, forced))
        return false;
    if (forced)
        return true;
    if (ctx->pf_off)
        return false; // it was given up on this context's last trials
    return ctx->tune.hs_bound == 1 && (ctx->fftlen == 16384 || ctx->tune.corr_pair != 0);
}
bool maxima_from_plane_generation(const accelgpu_ctx *ctx)
{
    return prefilter_wanted(ctx) && (ctx->fftlen == 16384 || (ctx->fftlen == 8192 && ctx->tune.corr_pair != 0));
}

int launch_prefilter(accelgpu_ctx *ctx, int nstages, const accelk::HarmParams &prm, const HitTarget &out, unsigned int cap,
                     int remap)
{
    int rc = ensure_prefilter(ctx);
    if (rc) return rc;
    const accelk::GroupGrid &g = ctx->grid;
    HIPCHK(hipMemsetAsync(ctx->d_ctl, 0, accelk::kPfCtlWords * sizeof(unsigned int), ctx->stream));
    if (!ctx->mx_valid) { // the plane was not written by a plane generation that keeps the maxima: from the plane
        // (blocks folded into grid.x: a long series has more of them than grid.y may count)
        const dim3 grid((unsigned) ((accelk::kGpb + 255) / 256) * (unsigned) ctx->nblk, 1u, (unsigned) ctx->numz);
        hipLaunchKernelGGL(accelk::k_group_max, grid, dim3(256), 0, ctx->stream, ctx->d_hi, ctx->setup.rlen, ctx->d_blk_col,
                           ctx->d_blk_ncols, g, ctx->d_mx);
        ctx->mx_valid = true;
        ctx->mx_fix_pending = false;
    } else if (ctx->mx_fix_pending) {
        hipLaunchKernelGGL(accelk::k_group_fix, dim3((unsigned) ((ctx->nblk + 255) / 256), (unsigned) ctx->numz), dim3(256), 0,
                           ctx->stream, ctx->d_mx, g, ctx->numz);
        ctx->mx_fix_pending = false;
    }
    const int64_t ngroups = (int64_t) ctx->range_count * accelk::kGpb; // of the blocks this context searches
    const uint32_t group0 = (uint32_t) ctx->range_first * (uint32_t) accelk::kGpb;
    // Groups per lane: with up to 8 harmonics a lane takes two groups 256 apart (the row's scalar work is shared by
    // twice the loads), with 16 one (two cost it its eight waves per SIMD).  pf_groups: 0 as said, 1 / 2 always that many.
    const int pp = ctx->tune.pf_groups;
    const int per_lane = pp == 1 || pp == 2 ? pp : nstages <= 4 ? 2 : 1;
    // Threads per workgroup: 1024 where that still makes several workgroups per CU (the waves of a workgroup walk
    // neighbouring columns and share the lines of the low harmonics' sources: 1.24 against 1.39 ms with 256 on a
    // 2^23-point, zmax 200 plane); pf_nt = 256 / 512 / 1024 forces one.
    const int pn = ctx->tune.pf_nt;
    const int wg = pn == 256 || pn == 512 || pn == 1024 ? pn : ngroups >= (int64_t) 1024 * per_lane * 1024 ? 1024 : ngroups >= (int64_t) 512 * per_lane * 1024 ? 512 : 256;
    unsigned nt = (unsigned) ((ngroups + wg * per_lane - 1) / (wg * per_lane));
    if (remap)
        nt = ((nt + 7) / 8) * 8;
    unsigned int wcap = (unsigned int) std::min<size_t>(ctx->work_cap, 0xFFFFFFFFu);
    if (ctx->tune.pf_list_cap > 0)
        wcap = std::min(wcap, (unsigned int) ctx->tune.pf_list_cap);
    const unsigned int room = wcap; // what the list can hold
    // Beyond this many entries the list costs more than the bounded sum over the whole plane it replaces (every
    // entry is four exact column sums): the kernels give the walk up as soon as they see it coming.
    const double all_entries = (double) ctx->numz * (double) ngroups;
    if (ctx->tune.hs_bound != 4) { // (forced: as long a list as there is room for)
        wcap = (unsigned int) std::min<double>((double) wcap, std::max(1.0, all_entries * 1e-6 * std::max(ctx->tune.pf_limit_ppm, 0)));
        // ... and as long as the workgroups' own lists hold it: six entries per thread, i.e. per group with one group per
        // lane and per two groups with two.  Past four fifths of that on average the fuller workgroups overflow into the
        // global list, an atomic per wave and row (8 harmonics, two groups per lane, 201 rows: 1.2 % listed 4.9 ms
        // against the bounded sum's 5.3, 1.9 % 11.9 ms against 5.4)
        wcap = (unsigned int) std::min<double>((double) wcap, std::max(1.0, 0.8 * 6.0 * (double) ngroups / per_lane));
    }
    // ... by asking a sample of the (row, group)s before the walk (k_prefilter_sample): 2^15 of them put the list's
    // length within a few per cent where it matters (a list at the limit has hundreds of them)
    const unsigned int nsamples = (unsigned int) std::min<double>(32768.0, all_entries);
    unsigned int gstep = (unsigned int) ((double) ngroups * 0.6180339887) | 1u;
    const unsigned int sample_limit = (unsigned int) std::min<double>(4294967295.0, std::ceil((double) wcap * (double) nsamples / all_entries));
    // The limit is the sample's question.  A walk that was started is given up only half as much again past it (or
    // when the list has no room): a list a little over the limit still costs about what the bounded sum would, while
    // giving up after the walk costs the walk and the bounded sum (8 harmonics, 1.24 % listed against a limit of
    // 1.19 %: 4.9 ms finished, 6.7 given up).
    if (ctx->tune.pf_sample)
        wcap = (unsigned int) std::min<double>((double) room, 1.5 * (double) wcap);
#define LAUNCH_WALK_AS(NS, NG, NT)                                                                             \
    hipLaunchKernelGGL((accelk::k_prefilter_lanes<NS, NG, NT>), dim3(nt), dim3(NT), 0, ctx->stream, ctx->d_mx, ctx->numz, g,  \
                       ngroups, ctx->d_rowmx, prm, ctx->d_work, ctx->d_ctl, wcap, remap, ctx->d_flags, group0)
#define LAUNCH_WALK(NS)                                                                                        \
    do {                                                                                                       \
        if (per_lane == 2 && wg == 1024) LAUNCH_WALK_AS(NS, 2, 1024);                                          \
        else if (per_lane == 2 && wg == 512) LAUNCH_WALK_AS(NS, 2, 512);                                       \
        else if (per_lane == 2) LAUNCH_WALK_AS(NS, 2, 256);                                                    \
        else if (wg == 1024) LAUNCH_WALK_AS(NS, 1, 1024);                                                      \
        else if (wg == 512) LAUNCH_WALK_AS(NS, 1, 512);                                                        \
        else LAUNCH_WALK_AS(NS, 1, 256);                                                                       \
    } while (0)
#define LAUNCH_PF(NS)                                                                                          \
    do {                                                                                                       \
        if (ctx->tune.pf_sample)                                                                               \
            hipLaunchKernelGGL((accelk::k_prefilter_sample<NS>), dim3((nsamples + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_mx, \
                               ctx->d_hi, ctx->numz, g, ngroups, ctx->d_rowmx, prm, ctx->d_ctl, nsamples, gstep, sample_limit,     \
                               ctx->d_flags, group0);                                                          \
        LAUNCH_WALK(NS);                                                                                       \
        hipLaunchKernelGGL((accelk::k_flagged_exact<NS>), dim3(4096), dim3(256), 0, ctx->stream, ctx->d_hi, ctx->d_lo,       \
                           ctx->setup.rlen, g, ctx->d_rowhalf, prm, ctx->d_work, ctx->d_ctl, wcap, out.hits, out.count, cap); \
    } while (0)
    switch (nstages) {
    case 2: LAUNCH_PF(2); break;
    case 3: LAUNCH_PF(3); break;
    case 4: LAUNCH_PF(4); break;
    case 5: LAUNCH_PF(5); break;
    default: LAUNCH_PF(6); break;
    }
#undef LAUNCH_PF
#undef LAUNCH_WALK
#undef LAUNCH_WALK_AS
    HIPCHK(hipGetLastError());
    return ACCELGPU_OK;
}

template <bool DUMP>
int launch_harmsum(accelgpu_ctx *ctx, int nstages, int64_t col0, int64_t ncols, float *dump,
                   int dump_stage, const HitTarget &out, bool timed = true)
{
    // Few harmonics: little data per thread, so take it in 8-byte pieces, two
    // pairs per thread, to keep enough bytes in flight; many harmonics: one
    // column per thread keeps registers low and waves plentiful.
    // (measured on the 2^23-point, zmax 200 plane: 1-2 harmonics 80/78 % of the HBM peak
    // with two pairs against 44/49 % with one column; 4 harmonics 71.5 % with one pair
    // against 60.7 %; 8 harmonics 75 % with one column against 66 % with pairs)
    if (bounds_in_use(ctx) && nstages != ctx->setup.numharmstages) {
        set_error("harmonic-sum launch for %d stages but the table holds %d", nstages, ctx->setup.numharmstages);
        return ACCELGPU_ERR_ARG;
    }
    if constexpr (DUMP) {
        if (bounds_in_use(ctx)) {
            const dim3 grid((unsigned) ((ncols + 255) / 256));
#define LAUNCH_DUMP(NS)                                                                                   \
    hipLaunchKernelGGL((accelk::k_dump_split<NS>), grid, dim3(256), 0, ctx->stream, ctx->d_hi, ctx->d_lo, \
                       ctx->setup.rlen, ctx->numz, col0, ncols, ctx->d_rowoff, dump, dump_stage)
            switch (nstages) {
            case 1: LAUNCH_DUMP(1); break;
            case 2: LAUNCH_DUMP(2); break;
            case 3: LAUNCH_DUMP(3); break;
            case 4: LAUNCH_DUMP(4); break;
            case 5: LAUNCH_DUMP(5); break;
            default: LAUNCH_DUMP(6); break;
            }
#undef LAUNCH_DUMP
            HIPCHK(hipGetLastError());
            return ACCELGPU_OK;
        }
    }
    if constexpr (!DUMP) {
        if (bounds_in_use(ctx)) {
            accelk::HarmParams prm;
            for (int i = 0; i < 6; i++)
                prm.powcut[i] = i < ctx->setup.numharmstages ? ctx->setup.powcut[i] : INFINITY;
            const int remap = ctx->tune.hs_xcd_remap ? 1 : 0;
            unsigned ntiles = (unsigned) ((ncols + 255) / 256);
            if (remap)
                ntiles = ((ntiles + 7) / 8) * 8;
            const unsigned int cap = (unsigned int) std::min<size_t>(out.cap, 0xFFFFFFFFu);
            EventPair ep;
            if (timed) {
                int rc = prof_begin(ctx, 2, &ep);
                if (rc) return rc;
            }
            // The coarse bound first: group maxima summed per group of four columns, the
            // (row, group)s that could cross listed and summed exactly; the plain bounded sum below then
            // runs only if the list was of no use (its workgroups return at once otherwise).
            const unsigned int *gate = nullptr;
            if (prefilter_wanted(ctx) && col0 == ctx->col0 + (int64_t) ctx->range_first * ACCELGPU_USELEN &&
                ncols == (int64_t) ctx->range_count * ACCELGPU_USELEN) { // the columns this context searches
                int rc = launch_prefilter(ctx, nstages, prm, out, cap, remap);
                if (rc) return rc;
                gate = ctx->d_ctl;
            }
#define LAUNCH_HSB(NS)                                                                                    \
    hipLaunchKernelGGL((accelk::k_harmsum_b<NS>), dim3(ntiles), dim3(256), ctx->tune.hs_lds, ctx->stream, \
                       ctx->d_hi, ctx->d_lo, ctx->setup.rlen, ctx->numz, col0, ncols, ctx->d_rowhalf, \
                       ctx->d_rowchg, prm, out.hits, out.count, cap, remap, ctx->d_flags, gate)
#define LAUNCH_HSBW(NS, NC)                                                                                \
    do {                                                                                                   \
        unsigned nt = (unsigned) ((ncols + 256 * NC - 1) / (256 * NC));                                    \
        if (remap)                                                                                         \
            nt = ((nt + 7) / 8) * 8;                                                                       \
        hipLaunchKernelGGL((accelk::k_harmsum_bw<NS, NC>), dim3(nt), dim3(256), ctx->tune.hs_lds, ctx->stream, \
                           ctx->d_hi, ctx->d_lo, ctx->setup.rlen, ctx->numz, col0, ncols, ctx->d_rowhalf, \
                           ctx->d_rowchg, prm, out.hits, out.count, cap, remap, ctx->d_flags, gate);             \
    } while (0)
#define LAUNCH_HSBL(NS, NC)                                                                                \
    do {                                                                                                   \
        unsigned nt = (unsigned) ((ncols + 256 * NC - 1) / (256 * NC));                                    \
        if (remap)                                                                                         \
            nt = ((nt + 7) / 8) * 8;                                                                       \
        hipLaunchKernelGGL((accelk::k_harmsum_bl<NS, NC>), dim3(nt), dim3(256), ctx->tune.hs_lds, ctx->stream, \
                           ctx->d_hi, ctx->d_lo, ctx->setup.rlen, ctx->numz, col0, ncols, ctx->d_rowhalf, \
                           ctx->d_rowchg, prm, out.hits, out.count, cap, remap, ctx->d_flags, gate);             \
    } while (0)
            // Several columns per lane (as many times the loads in flight) once there are enough columns
            // to fill the device with that many fewer workgroups; as many columns as keep eight waves
            // per SIMD (64 registers).  2^23 points, zmax 200: 16 harmonics 3.49 (two columns) against
            // 4.2 ms (one), 8 harmonics 2.03 against 2.95 ms; 2^21 points, zmax 50: 0.315 against 0.245 ms.
            // hs_bound = 2 / 3 force the one-column / the wide kernel.
            // (with few harmonics the wide form gains most: 1 / 2 / 4 harmonics 0.73 / 0.86 / 1.12 against 1.85 / 2.04 /
            // 2.37 ms on that plane -- what a search runs when the coarse bound does not apply or was given up)
            const bool wide = ctx->tune.hs_bound == 3 || (ctx->tune.hs_bound != 2 && ncols >= ((int64_t) 1 << 22));
            switch (nstages) {
            case 1: if (wide) LAUNCH_HSBW(1, 4); else LAUNCH_HSB(1); break;
            case 2: if (wide) LAUNCH_HSBW(2, 4); else LAUNCH_HSB(2); break;
            case 3: if (wide) LAUNCH_HSBW(3, 4); else LAUNCH_HSB(3); break;
            case 4: if (wide) LAUNCH_HSBW(4, 3); else LAUNCH_HSB(4); break;
            case 5: if (wide) LAUNCH_HSBW(5, 2); else LAUNCH_HSB(5); break;
            // 31 sub-harmonics: the register-lean form, four columns per lane (2^24 points, zmax 1200:
            // 84.5 ms against 101 ms with one column in k_harmsum_b<6> at five waves per SIMD; three
            // columns 86 ms, two 92 ms, one -- eight waves per SIMD -- 136 ms)
            default: if (wide) LAUNCH_HSBL(6, 4); else LAUNCH_HSB(6); break;
            }
#undef LAUNCH_HSB
#undef LAUNCH_HSBW
#undef LAUNCH_HSBL
            HIPCHK(hipGetLastError());
            if (timed)
                return prof_end(ctx, 2, &ep);
            return ACCELGPU_OK;
        }
    }
    int j = nstages <= 2 ? 2 : nstages == 3 ? 1 : 0;
    if (ctx->tune.hs_j >= 0)
        j = ctx->tune.hs_j;
    if (j == 2)
        return launch_harmsum_j<DUMP, 2>(ctx, nstages, col0, ncols, dump, dump_stage, out, timed);
    if (j == 1)
        return launch_harmsum_j<DUMP, 1>(ctx, nstages, col0, ncols, dump, dump_stage, out, timed);
    return launch_harmsum_j<DUMP, 0>(ctx, nstages, col0, ncols, dump, dump_stage, out, timed);
}

HitTarget slot_target(const accelgpu_ctx *ctx)
{
    return HitTarget{ctx->d_hits, ctx->d_count, ctx->hits_cap};
}

void select_slot(accelgpu_ctx *ctx, int i)
{
    ctx->d_hits = ctx->slots[i].d_hits;
    ctx->h_hits = ctx->slots[i].h_hits;
    ctx->d_count = ctx->slots[i].d_count;
    ctx->h_count = ctx->slots[i].h_count;
    ctx->hits_cap = ctx->slots[i].cap;
}

// (Re)allocates one slot's hit buffers for exactly `want` records.  hipFree waits for the
// device, so whatever is queued has finished when this returns.
int resize_slot(accelgpu_ctx *ctx, int i, size_t want)
{
    drop_graphs(ctx);
    accelgpu_ctx::HitSlot &sl = ctx->slots[i];
    if (sl.d_hits) HIPCHK(hipFree(sl.d_hits));
    if (sl.h_hits) HIPCHK(hipHostFree(sl.h_hits));
    sl.d_hits = sl.h_hits = nullptr;
    sl.cap = 0;
    if (hipMalloc((void **) &sl.d_hits, want * sizeof(DevHit)) != hipSuccess ||
        hipHostMalloc((void **) &sl.h_hits, want * sizeof(DevHit)) != hipSuccess) {
        if (sl.d_hits) (void) hipFree(sl.d_hits);
        sl.d_hits = sl.h_hits = nullptr;
        (void) hipGetLastError();
        set_error("cannot allocate room for %zu hits", want);
        return ACCELGPU_ERR_NOMEM;
    }
    sl.cap = want;
    return ACCELGPU_OK;
}

// Both slots to at least `want` records (only with nothing in flight: a launched trial's
// hits live in its slot).
int reserve_hits(accelgpu_ctx *ctx, size_t want)
{
    if (ctx->slots[0].cap >= want && ctx->slots[1].cap >= want)
        return ACCELGPU_OK;
    if (ctx->inflight > 0) {
        set_error("cannot grow the hit buffers while %d launches are in flight", ctx->inflight);
        return ACCELGPU_ERR_ARG;
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 2; i++)
        if (ctx->slots[i].cap < want) {
            int rc = resize_slot(ctx, i, want);
            if (rc) return rc;
        }
    select_slot(ctx, ctx->head);
    return ACCELGPU_OK;
}

int need_plane(accelgpu_ctx *ctx)
{
    if (!ctx->inmem || !(ctx->split ? ctx->d_hi && ctx->d_lo : ctx->d_plane != nullptr)) {
        set_error("this call needs the in-memory plane: set the search up with accelgpu_set_search()");
        return ACCELGPU_ERR_ARG;
    }
    return ACCELGPU_OK;
}

// The current spectrum may still be on its way (accelgpu_load_fft_async).  A launched
// search waits for it on the device; everything else waits here, on the host.
int settle_spectrum(accelgpu_ctx *ctx)
{
    const int j = ctx->cur_own;
    if (j >= 0 && ctx->own_pending[j]) {
        HIPCHK(hipEventSynchronize(ctx->own_ready[j]));
        ctx->own_pending[j] = false;
    }
    return ACCELGPU_OK;
}

int spectrum_ready(accelgpu_ctx *ctx, hipStream_t st)
{
    const int j = ctx->cur_own;
    if (j >= 0 && ctx->own_pending[j])
        HIPCHK(hipStreamWaitEvent(st, ctx->own_ready[j], 0));
    return ACCELGPU_OK;
}

int check_ready(accelgpu_ctx *ctx, bool need_fft, bool settle = true)
{
    if (ctx->zmax < 0) {
        set_error("accelgpu_build_kernels() has not been called");
        return ACCELGPU_ERR_ARG;
    }
    if (!ctx->have_search) {
        set_error("accelgpu_set_search() has not been called");
        return ACCELGPU_ERR_ARG;
    }
    if (need_fft && !ctx->d_fft) {
        set_error("no spectrum loaded (accelgpu_load_fft)");
        return ACCELGPU_ERR_ARG;
    }
    if (ctx->setup.zmax != ctx->zmax) {
        set_error("kernel bank built for zmax=%d but the search uses zmax=%d", ctx->zmax,
                  ctx->setup.zmax);
        return ACCELGPU_ERR_ARG;
    }
    if (need_fft && settle)
        return settle_spectrum(ctx);
    return ACCELGPU_OK;
}

// Does the launch that starts now go without the coarse bound?  (See note_prefilter_outcome.)
void decide_prefilter(accelgpu_ctx *ctx)
{
    ctx->pf_off = false;
    if (ctx->tune.pf_memory && ctx->pf_skip > 0) {
        ctx->pf_off = true;
        ctx->pf_skip--;
    }
}

int queue_fill(accelgpu_ctx *ctx)
{
    if (ctx->nblk_active == 0)
        return ACCELGPU_OK;
    const int2 *zrange = ctx->partial_fill ? ctx->d_blk_zrange : nullptr;
    int rc = launch_block_prep(ctx, ctx->nblk, ctx->d_blk_lobin, ctx->d_blk_numdata, ctx->d_blockfft, zrange);
    if (rc) return rc;
    ctx->mx_valid = false;
    if (bounds_in_use(ctx)) {
        // every cell a host may have written a negative power into is rewritten now: the bounds hold again
        if (!ctx->partial_fill && !ctx->host_wrote_outside)
            HIPCHK(hipMemsetAsync(ctx->d_flags, 0, 64, ctx->stream));
        if (maxima_from_plane_generation(ctx)) {
            if ((rc = ensure_prefilter(ctx))) return rc;
            rc = launch_correlate(ctx, ctx->nblk, ctx->d_blockfft, ctx->d_blk_col, ctx->d_blk_ncols, nullptr,
                                  ctx->setup.rlen, ctx->d_hi, ctx->d_lo, ctx->d_mx, ctx->grid.mxlen, zrange);
            ctx->mx_valid = ctx->mx_fix_pending = rc == ACCELGPU_OK;
            return rc;
        }
        return launch_correlate(ctx, ctx->nblk, ctx->d_blockfft, ctx->d_blk_col, ctx->d_blk_ncols, nullptr,
                                ctx->setup.rlen, ctx->d_hi, ctx->d_lo, nullptr, 0, zrange);
    }
    return launch_correlate(ctx, ctx->nblk, ctx->d_blockfft, ctx->d_blk_col, ctx->d_blk_ncols,
                            ctx->d_plane, ctx->setup.rlen, nullptr, nullptr, nullptr, 0, zrange);
}

// Plane generation and harmonic sum of one whole trial, chunk by chunk in ascending r: the
// fundamental of a chunk of search blocks is generated and summed right away, while it is
// still in the caches (a sub-harmonic's source column lies below the column it is added
// to, so it has been written by this or an earlier chunk).  Only for the whole block grid.
int queue_fill_and_sum_chunked(accelgpu_ctx *ctx, int chunk_blocks, const HitTarget &out)
{
    int rc = launch_block_prep(ctx, ctx->nblk, ctx->d_blk_lobin, ctx->d_blk_numdata, ctx->d_blockfft);
    if (rc) return rc;
    const size_t nf = (size_t) ctx->fftlen / 2;
    for (int s0 = 0; s0 < ctx->nsearch; s0 += chunk_blocks) {
        const int s1 = std::min(s0 + chunk_blocks, ctx->nsearch);
        const int b0 = s0 == 0 ? 0 : ctx->nprepop + s0, b1 = ctx->nprepop + s1; // pre-population blocks go first
        const bool halves = bounds_in_use(ctx);
        if ((rc = launch_correlate(ctx, b1 - b0, ctx->d_blockfft + (size_t) b0 * nf, ctx->d_blk_col + b0,
                                   ctx->d_blk_ncols + b0, halves ? nullptr : ctx->d_plane, ctx->setup.rlen,
                                   halves ? ctx->d_hi : nullptr, halves ? ctx->d_lo : nullptr)))
            return rc;
        if ((rc = launch_harmsum<false>(ctx, ctx->setup.numharmstages, ctx->col0 + (int64_t) s0 * ACCELGPU_USELEN,
                                        (int64_t) (s1 - s0) * ACCELGPU_USELEN, nullptr, 0, out)))
            return rc;
    }
    return ACCELGPU_OK;
}

// Hits copied back speculatively with the count, so that a collect usually finds them all
// on the host already.
size_t prefetch_hits(const accelgpu_ctx *ctx, const accelgpu_ctx::HitSlot &sl)
{
    const bool small = ctx->setup.plane_bytes <= ((int64_t) 64 << 20);
    return std::min<size_t>(sl.cap, small ? 8192 : 65536);
}

// One whole launch of a small search into hit slot `si` as a captured graph, replayed with one
// call: the nine stream operations of the plain path cost more host time than the GPU work
// they queue.  The copies are part of the graph (a few microseconds at these sizes).
int launch_as_graph(accelgpu_ctx *ctx, int si)
{
    accelgpu_ctx::HitSlot &sl = ctx->slots[si];
    accelgpu_ctx::LaunchGraph *g = nullptr;
    for (auto &e : ctx->graphs)
        if (e.slot == si && e.fft == ctx->d_fft && e.numbins == ctx->numbins && e.pf_off == ctx->pf_off)
            g = &e;
    if (!g) {
        if (ctx->graphs.size() >= 12) { // least recently used one out
            size_t old = 0;
            for (size_t k = 1; k < ctx->graphs.size(); k++)
                if (ctx->graphs[k].used < ctx->graphs[old].used) old = k;
            if (ctx->graphs[old].exec) (void) hipGraphExecDestroy(ctx->graphs[old].exec);
            if (ctx->graphs[old].graph) (void) hipGraphDestroy(ctx->graphs[old].graph);
            ctx->graphs.erase(ctx->graphs.begin() + (std::ptrdiff_t) old);
        }
        accelgpu_ctx::LaunchGraph e;
        e.slot = si;
        e.fft = ctx->d_fft;
        e.numbins = ctx->numbins;
        e.pf_off = ctx->pf_off;
        if (prefilter_wanted(ctx)) { // its buffers are allocated and its table copied outside the capture
            int rc = ensure_prefilter(ctx);
            if (rc) return rc;
        }
        // the body below records kernels, it runs none: what it notes about the maxima holds only once the
        // graph has been launched
        const bool mx_valid0 = ctx->mx_valid, mx_fix0 = ctx->mx_fix_pending;
        HIPCHK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
        auto body = [&]() -> int {
            HIPCHK(hipMemsetAsync(sl.d_count, 0, kCountBytes, ctx->stream));
            int rc = queue_fill(ctx);
            if (rc) return rc;
            rc = launch_harmsum<false>(ctx, ctx->setup.numharmstages, ctx->col0, (int64_t) ctx->nsearch * ACCELGPU_USELEN,
                                       nullptr, 0, HitTarget{sl.d_hits, sl.d_count, sl.cap});
            if (rc) return rc;
            HIPCHK(hipMemcpyAsync(sl.h_count, sl.d_count, kCountBytes, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipMemcpyAsync(sl.h_hits, sl.d_hits, prefetch_hits(ctx, sl) * sizeof(DevHit), hipMemcpyDeviceToHost,
                                  ctx->stream));
            return ACCELGPU_OK;
        };
        const int rc = body();
        const hipError_t ee = hipStreamEndCapture(ctx->stream, &e.graph); // always: the stream must leave capture mode
        e.mx_valid_after = ctx->mx_valid;
        e.mx_fix_after = ctx->mx_fix_pending;
        ctx->mx_valid = mx_valid0;
        ctx->mx_fix_pending = mx_fix0;
        if (rc) {
            if (e.graph) (void) hipGraphDestroy(e.graph);
            return rc;
        }
        HIPCHK(ee);
        if (hipGraphInstantiate(&e.exec, e.graph, nullptr, nullptr, 0) != hipSuccess) {
            (void) hipGraphDestroy(e.graph);
            set_error("hipGraphInstantiate failed: %s", hipGetErrorString(hipGetLastError()));
            return ACCELGPU_ERR_HIP;
        }
        ctx->graphs.push_back(e);
        g = &ctx->graphs.back();
    }
    g->used = ++ctx->graph_clock;
    HIPCHK(hipGraphLaunch(g->exec, ctx->stream));
    ctx->mx_valid = g->mx_valid_after; // what a whole fill and sum leave behind
    ctx->mx_fix_pending = g->mx_fix_after;
    return ACCELGPU_OK;
}

int queue_search(accelgpu_ctx *ctx)
{
    HIPCHK(hipMemsetAsync(ctx->d_count, 0, kCountBytes, ctx->stream));
    int rc = launch_harmsum<false>(ctx, ctx->setup.numharmstages,
                                   ctx->col0 + (int64_t) ctx->range_first * ACCELGPU_USELEN,
                                   (int64_t) ctx->range_count * ACCELGPU_USELEN, nullptr, 0,
                                   slot_target(ctx));
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(ctx->h_count, ctx->d_count, kCountBytes, hipMemcpyDeviceToHost,
                          ctx->stream));
    return ACCELGPU_OK;
}

// The reference's visiting order (block, stage, z row, r) as one integer per hit.
inline uint64_t visit_key(int64_t rind, uint32_t stage, uint32_t zind, int64_t col0)
{
    const uint64_t c = (uint64_t) (rind - col0);
    const uint64_t block = c / ACCELGPU_USELEN, col = c % ACCELGPU_USELEN;      // col < 2^13
    return (((block * 8 + stage) * 2048 + zind) << 13) | col;                  // stage < 2^3, zind < 2^11
}

// What the coarse bound of a finished launch did decides whether the next launches try it: given up ->
// the next 4 launches go without (8, 16 ... 64 when it is given up again right after), a list that served ->
// it stays.  The trials of one batch are alike (DM trials of one beam), and a launch that gives up has
// still paid for part of the walk and for the maxima plane generation wrote.
void note_prefilter_outcome(accelgpu_ctx *ctx, const unsigned int *words)
{
    if (!words[8]) {
        ctx->pf_last_listed = -2;
        return;
    }
    if (words[5]) {
        ctx->pf_last_listed = -1;
        ctx->pf_backoff = ctx->tune.pf_memory ? std::min(std::max(2 * ctx->pf_backoff, 4), 64) : 0;
        ctx->pf_skip = ctx->pf_backoff;
    } else {
        ctx->pf_last_listed = (int64_t) words[4];
        ctx->pf_backoff = 0;
    }
}

// Waits for the search launched into slot `si` and returns its hits sorted into the
// reference's visiting order.  If the compaction overflowed the slot's buffer the search
// is run again into a larger one: from the plane if it still holds this trial, after
// regenerating the plane from this trial's spectrum if a later trial has been launched
// since (that trial has finished by then, and its hits wait in the other slot).
int finish_search(accelgpu_ctx *ctx, int si, accelgpu_hit *hits, int64_t max_hits, int64_t *n_hits)
{
    accelgpu_ctx::HitSlot &sl = ctx->slots[si];
    select_slot(ctx, si);
    HIPCHK(hipEventSynchronize(sl.done));
    if (!sl.pf_counted) { // once per launch (a collect can be repeated)
        sl.pf_counted = true;
        note_prefilter_outcome(ctx, sl.h_count);
    }
    size_t n = *sl.h_count;
    size_t have = std::min(sl.prefetch, n);
    sl.prefetch = 0;
    if (n > sl.cap) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->copy_stream));
        int rc = resize_slot(ctx, si, n + n / 4 + 1024);
        if (rc) return rc;
        select_slot(ctx, si);
        const bool prof = ctx->profiling;
        ctx->profiling = false; // a repeat is not one of the timed launches
        if (ctx->plane_serial != sl.serial) {
            const c32 *cur_fft = ctx->d_fft;
            const int64_t cur_bins = ctx->numbins;
            ctx->d_fft = sl.fft;
            ctx->numbins = sl.numbins;
            rc = queue_fill(ctx);
            ctx->d_fft = cur_fft;
            ctx->numbins = cur_bins;
            if (!rc) ctx->plane_serial = sl.serial;
        }
        if (!rc) rc = queue_search(ctx);
        ctx->profiling = prof;
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(ctx->stream));
        n = *ctx->h_count;
        have = 0;
        if (n > sl.cap) {
            set_error("hit count changed between two runs of the same search");
            return ACCELGPU_ERR_HIP;
        }
    }
    if (n > have) {
        HIPCHK(hipMemcpyAsync(ctx->h_hits + have, ctx->d_hits + have, (n - have) * sizeof(DevHit),
                              hipMemcpyDeviceToHost, ctx->copy_stream));
        HIPCHK(hipStreamSynchronize(ctx->copy_stream));
    }
    *n_hits = (int64_t) n;
    if ((int64_t) n > max_hits) {
        set_error("%zu hits but room for %lld: re-run with a larger buffer", n, (long long) max_hits);
        return ACCELGPU_ERR_OVERFLOW;
    }
    // sort one key per hit, then write the hits out in that order
    const int64_t col0 = ctx->col0;
    std::vector<std::pair<uint64_t, uint32_t>> order(n);
    for (size_t i = 0; i < n; i++) {
        const DevHit &d = ctx->h_hits[i];
        order[i] = {visit_key(d.rind, d.stage_zind >> 16, d.stage_zind & 0xFFFF, col0), (uint32_t) i};
    }
    std::sort(order.begin(), order.end());
    for (size_t k = 0; k < n; k++) {
        const DevHit &d = ctx->h_hits[order[k].second];
        accelgpu_hit &h = hits[k];
        h.power = d.power;
        h.stage = (int32_t) (d.stage_zind >> 16);
        h.zind = (int32_t) (d.stage_zind & 0xFFFF);
        h.rind = d.rind;
        h.block = (int32_t) ((d.rind - col0) / ACCELGPU_USELEN);
    }
    return ACCELGPU_OK;
}

// The tuning knobs by name, for accelgpu_set_tuning()/accelgpu_get_tuning() and the
// environment overrides (ACCELGPU_ + the name in capitals) read once per context.
struct TuneName {
    const char *name;
    int Tuning::*field;
};
const TuneName kTuneNames[] = {
    {"corr_rr", &Tuning::corr_rr},           {"corr_zl", &Tuning::corr_zl},
    {"corr_chunk", &Tuning::corr_chunk},     {"corr_tz", &Tuning::corr_tz},
    {"corr_tc", &Tuning::corr_tc},           {"corr_lds", &Tuning::corr_lds},
    {"corr_pair", &Tuning::corr_pair},
            {"hs_xcd_remap", &Tuning::hs_xcd_remap},
    {"hs_lds", &Tuning::hs_lds},             {"hs_lean_from", &Tuning::hs_lean_from},
    {"hs_j", &Tuning::hs_j},                 {"hs_chunk_blocks", &Tuning::hs_chunk_blocks},
    {"std_table", &Tuning::std_table},       {"std_blocks", &Tuning::std_blocks},
    {"opt_nt", &Tuning::opt_nt},             {"use_graph", &Tuning::use_graph},
    {"hs_bound", &Tuning::hs_bound},         {"pf_list_cap", &Tuning::pf_list_cap},
    {"pf_groups", &Tuning::pf_groups},       {"pf_nt", &Tuning::pf_nt},         {"pf_limit_ppm", &Tuning::pf_limit_ppm},
    {"pf_memory", &Tuning::pf_memory},       {"pf_sample", &Tuning::pf_sample},
};

void tuning_from_env(Tuning *t)
{
    for (const TuneName &k : kTuneNames) {
        char var[64] = "ACCELGPU_";
        size_t n = strlen(var);
        for (const char *p = k.name; *p && n + 1 < sizeof(var); p++)
            var[n++] = (char) toupper((unsigned char) *p);
        var[n] = 0;
        if (const char *e = getenv(var))
            t->*(k.field) = atoi(e);
    }
}

} // namespace

// ===========================================================================
extern "C" {

const char *accelgpu_last_error(void) { return accelhost::last_error(); }
const char *accelgpu_version(void) { return "accelgpu 0.1 (gfx950)"; }

int accelgpu_z_resp_halfwidth(double z) { return accelhost::z_resp_halfwidth(z); }
int accelgpu_calc_fftlen(int numharm, int harmnum, int max_zfull)
{
    return accelhost::calc_fftlen(numharm, harmnum, max_zfull);
}
double accelgpu_candidate_sigma(double power, int numsum, double numtrials)
{
    return accelhost::candidate_sigma(power, numsum, numtrials);
}
double accelgpu_power_for_sigma(double sigma, int numsum, double numtrials)
{
    return accelhost::power_for_sigma(sigma, numsum, numtrials);
}
int accelgpu_coarse_bound_applies(int numharmstages, int64_t search_columns, int64_t rlen, int fftlen, int forced)
{
    return coarse_bound_shape_ok(numharmstages, search_columns, rlen, fftlen, forced != 0) ? 1 : 0;
}
int accelgpu_block_range_need(const accelgpu_setup *setup, int32_t first_block, int32_t num_blocks, int32_t *blocks_generated,
                              int64_t *rows_generated)
{
    if (!setup || first_block < 0 || num_blocks < 0 || first_block + num_blocks > setup->num_search_blocks) {
        set_error("accelgpu_block_range_need: null set-up or a range outside its search blocks");
        return ACCELGPU_ERR_ARG;
    }
    accelhost::BlockGrid g;
    accelhost::block_grid(setup->rlo, setup->highestbin, setup->halfwidth, &g);
    std::vector<int> z0, z1;
    int64_t rows = 0;
    const int nb = accelhost::block_range_need(setup, g, first_block, num_blocks, &z0, &z1, &rows);
    if (blocks_generated) *blocks_generated = nb;
    if (rows_generated) *rows_generated = rows;
    return ACCELGPU_OK;
}
int accelgpu_plan(int64_t numbins, int64_t N, double dt, int zmax, int numharm, double sigma,
                  double flo, double fhi, int64_t lobin, accelgpu_setup *out)
{
    if (!out) {
        set_error("accelgpu_plan: null output");
        return ACCELGPU_ERR_ARG;
    }
    return accelhost::plan(numbins, N, dt, zmax, numharm, sigma, flo, fhi, lobin, out);
}
int accelgpu_replay_candidates(const accelgpu_setup *setup, const accelgpu_hit *hits,
                               int64_t n_hits, accelgpu_cand *cands, int64_t max_cands,
                               int64_t *n_cands, float *sigma_out, int32_t *added)
{
    if (!setup || (!hits && n_hits > 0) || !n_cands || (!cands && max_cands > 0)) {
        set_error("accelgpu_replay_candidates: null argument");
        return ACCELGPU_ERR_ARG;
    }
    return accelhost::replay(setup, hits, n_hits, cands, max_cands, n_cands, sigma_out, added);
}

int accelgpu_replay_options(int linear_walk, int threads)
{
    accelhost::replay_options(linear_walk, threads);
    return ACCELGPU_OK;
}

int accelgpu_create(accelgpu_ctx **out, int device)
{
    if (!out) {
        set_error("accelgpu_create: null output");
        return ACCELGPU_ERR_ARG;
    }
    *out = nullptr;
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) {
        set_error("accelgpu_create: device %d of %d", device, ndev);
        return ACCELGPU_ERR_ARG;
    }
    HIPCHK(hipSetDevice(device));
    if (device >= kMaxDevices) {
        set_error("accelgpu_create: device index %d beyond the %d this library keeps state for", device,
                  kMaxDevices);
        return ACCELGPU_ERR_ARG;
    }
    {
        // once per device, with that device current: the address of this device's copy
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ptr[device]) {
            unsigned char *p = nullptr;
            HIPCHK(hipGetSymbolAddress((void **) &p, HIP_SYMBOL(scratch)));
            g_scratch_ptr[device] = p;
        }
    }
    // released on every early return below; handed to the caller at the end
    struct Guard {
        accelgpu_ctx *c;
        ~Guard() { if (c) accelgpu_destroy(c); }
    } guard{new accelgpu_ctx()};
    accelgpu_ctx *ctx = guard.c;
    ctx->device = device;
    tuning_from_env(&ctx->tune);
    HIPCHK(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    HIPCHK(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    {
        // the side stream's small kernels should not wait for a whole search to drain:
        // give them the highest priority the device offers
        int least = 0, greatest = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIPCHK(hipStreamCreateWithPriority(&ctx->side_stream, hipStreamNonBlocking, greatest));
        HIPCHK(hipStreamCreateWithPriority(&ctx->opt_stream, hipStreamNonBlocking, greatest));
    }
    for (auto &sl : ctx->slots) {
        HIPCHK(hipMalloc((void **) &sl.d_count, kCountBytes));
        HIPCHK(hipHostMalloc((void **) &sl.h_count, kCountBytes));
        memset(sl.h_count, 0, kCountBytes);
        HIPCHK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&sl.kernels, hipEventDisableTiming));
    }
    HIPCHK(hipEventCreateWithFlags(&ctx->opt.done, hipEventDisableTiming));
    for (hipEvent_t &e : ctx->own_ready)
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(hipMalloc((void **) &ctx->d_one_lobin, sizeof(int64_t)));
    HIPCHK(hipMalloc((void **) &ctx->d_one_col, sizeof(int64_t)));
    HIPCHK(hipMalloc((void **) &ctx->d_one_numdata, sizeof(int)));
    HIPCHK(hipMalloc((void **) &ctx->d_one_ncols, sizeof(int)));
    int rc;
    if ((rc = allow_lds(accelk::k_block_prep<P4K>, P4K::LDS_BYTES + 64))) return rc;
    if ((rc = allow_lds(accelk::k_block_prep<P8K>, P8K::LDS_BYTES + 64))) return rc;
    if ((rc = allow_lds(accelk::k_correlate<P8K, false>, P8K::LDS_BYTES))) return rc;
    if ((rc = allow_lds(accelk::k_correlate<P16K, false>, P16K::LDS_BYTES))) return rc;
    if ((rc = allow_lds(accelk::k_correlate_rr<P8K, false, 16, false>, P8K::LDS_BYTES + 20000))) return rc;
    if ((rc = allow_lds(accelk::k_correlate_rr<P8K, false, 16, true>, P8K::LDS_BYTES + 20000))) return rc;
    if ((rc = allow_lds(accelk::k_correlate_rr<P16K, false, 16, false>, P16K::LDS_BYTES + 20000))) return rc;
    if ((rc = allow_lds(accelk::k_correlate_rr<P16K, false, 16, true>, P16K::LDS_BYTES + 20000))) return rc;
    if ((rc = allow_lds(accelk::k_correlate_rr<P8K, false, 16, true, afft::NoStamp, true>, P8K::LDS_BYTES + 20000))) return rc;
    if ((rc = allow_lds(accelk::k_correlate_rr<P8K, false, 16, true, afft::NoStamp, true, true>, P8K::LDS_BYTES + 20000))) return rc;
    if ((rc = allow_lds(accelk::k_correlate_rr<P8K, false, 16, true, afft::NoStamp, false, true>, P8K::LDS_BYTES + 20000))) return rc;
    if ((rc = allow_lds(accelk::k_correlate_rr<P8K, false, 16, true, afft::NoStamp, true, true, true>, P8K::LDS_BYTES + 20000))) return rc;
    if ((rc = allow_lds(accelk::k_correlate_rr<P16K, false, 16, true, afft::NoStamp, true>, P16K::LDS_BYTES + 20000))) return rc;
    if ((rc = allow_lds(accelk::k_correlate_rr<P16K, false, 16, true, afft::NoStamp, true, false, true>, P16K::LDS_BYTES + 20000))) return rc;
    if ((rc = allow_lds(accelk::k_fft_rows<P8K, -1>, P8K::LDS_BYTES))) return rc;
    if ((rc = allow_lds(accelk::k_fft_rows<P16K, -1>, P16K::LDS_BYTES))) return rc;
    if ((rc = allow_lds(speck::k_bigfft_cols<P8K>, P8K::LDS_BYTES))) return rc;
    if ((rc = allow_lds(speck::k_bigfft_cols<P16K>, P16K::LDS_BYTES))) return rc;
    if ((rc = allow_lds(speck::k_bigfft_rows<P8K>, P8K::LDS_BYTES))) return rc;
    if ((rc = allow_lds(speck::k_bigfft_rows<P16K>, P16K::LDS_BYTES))) return rc;
    if ((rc = reserve_hits(ctx, (size_t) 1 << 20))) return rc;
    select_slot(ctx, 0);
    guard.c = nullptr;
    *out = ctx;
    return ACCELGPU_OK;
}

// Teardown releases what it can: a failing release has nobody left to report to.
void accelgpu_destroy(accelgpu_ctx *ctx)
{
    if (ctx) drop_graphs(ctx);
    if (!ctx)
        return;
    (void) hipSetDevice(ctx->device);
    (void) hipStreamSynchronize(ctx->stream);
    if (ctx->side_stream) (void) hipStreamSynchronize(ctx->side_stream);
    if (ctx->opt_stream) (void) hipStreamSynchronize(ctx->opt_stream);
    if (ctx->copy_stream) (void) hipStreamSynchronize(ctx->copy_stream);
    for (auto &v : ctx->ev)
        for (auto &e : v) { (void) hipEventDestroy(e.a); (void) hipEventDestroy(e.b); }
    for (auto &e : ctx->ev_pool) { (void) hipEventDestroy(e.a); (void) hipEventDestroy(e.b); }
    void *dev[] = {ctx->pw_inv.pw1, ctx->pw_inv.pw2, ctx->d_bank, ctx->tw_inv.tw1, ctx->tw_inv.tw2, ctx->tw_fwd_full.tw1,
                   ctx->tw_fwd_full.tw2, ctx->tw_fwd_half.tw1, ctx->tw_fwd_half.tw2,
                   ctx->d_own[0], ctx->d_own[1], ctx->d_own[2], ctx->d_blk_lobin, ctx->d_blk_col, ctx->d_blk_numdata,
                   ctx->d_blk_ncols, ctx->d_blockfft, ctx->d_plane, ctx->d_hi, ctx->d_lo, ctx->d_flags, ctx->d_rowoff, ctx->d_rowchg, ctx->d_rowhalf, ctx->d_tmp, ctx->d_one_lobin, ctx->d_one_col,
                   ctx->d_one_numdata, ctx->d_one_ncols, ctx->d_one_blockfft, ctx->d_mx, ctx->d_rowmx, ctx->d_work, ctx->d_ctl, ctx->d_blk_zrange};
    for (void *p : dev)
        if (p) (void) hipFree(p);
    free_subbanks(ctx);
    if (ctx->d_blu_filter) (void) hipFree(ctx->d_blu_filter);
    for (auto &kv : ctx->tw_cache) {
        if (kv.second.tw1) (void) hipFree(kv.second.tw1);
        if (kv.second.tw2) (void) hipFree(kv.second.tw2);
    }
    for (auto &b : ctx->std_buf)
        if (b.p) (void) hipFree(b.p);
    for (auto &sl : ctx->slots) {
        if (sl.d_hits) (void) hipFree(sl.d_hits);
        if (sl.d_count) (void) hipFree(sl.d_count);
        if (sl.h_hits) (void) hipHostFree(sl.h_hits);
        if (sl.h_count) (void) hipHostFree(sl.h_count);
        if (sl.done) (void) hipEventDestroy(sl.done);
        if (sl.kernels) (void) hipEventDestroy(sl.kernels);
    }
    if (ctx->opt.h_prob) (void) hipHostFree(ctx->opt.h_prob);
    if (ctx->opt.h_res) (void) hipHostFree(ctx->opt.h_res);
    if (ctx->opt.done) (void) hipEventDestroy(ctx->opt.done);
    for (hipEvent_t e : ctx->own_ready)
        if (e) (void) hipEventDestroy(e);
    if (ctx->copy_stream) (void) hipStreamDestroy(ctx->copy_stream);
    if (ctx->side_stream) (void) hipStreamDestroy(ctx->side_stream);
    if (ctx->opt_stream) (void) hipStreamDestroy(ctx->opt_stream);
    if (ctx->own_stream) (void) hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int accelgpu_set_tuning(accelgpu_ctx *ctx, const char *name, int value)
{
    int rc = use(ctx);
    if (rc) return rc;
    if (!name) {
        set_error("accelgpu_set_tuning: null name");
        return ACCELGPU_ERR_ARG;
    }
    if (ctx->inflight > 0) {
        set_error("accelgpu_set_tuning: collect the launched trial(s) first");
        return ACCELGPU_ERR_ARG;
    }
    if (!strcmp(name, "hits_cap")) {
        // the device hit buffers, both of them, at exactly this size (tests of the regrow path)
        if (value < 1) {
            set_error("accelgpu_set_tuning: hits_cap must be positive");
            return ACCELGPU_ERR_ARG;
        }
        HIPCHK(hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < 2; i++)
            if ((rc = resize_slot(ctx, i, (size_t) value))) return rc;
        select_slot(ctx, ctx->head);
        return ACCELGPU_OK;
    }
    for (const TuneName &k : kTuneNames)
        if (!strcmp(name, k.name)) {
            ctx->tune.*(k.field) = value;
            drop_graphs(ctx);
            if (k.field == &Tuning::pf_list_cap)
                ctx->pf_ready = false; // the list is sized again at the next launch
            if (k.field == &Tuning::pf_memory || k.field == &Tuning::hs_bound)
                ctx->pf_skip = ctx->pf_backoff = 0;
            if (k.field == &Tuning::hs_bound && ctx->have_search && ctx->inmem &&
                want_split(ctx, ctx->setup.numharmstages) != ctx->split) {
                // the plane changes its layout: set up afresh (an empty plane, as after accelgpu_set_search)
                const accelgpu_setup again = ctx->setup;
                return set_search_impl(ctx, &again, ctx->norm_type, ctx->nph, true);
            }
            return ACCELGPU_OK;
        }
    set_error("accelgpu_set_tuning: no knob named '%s'", name);
    return ACCELGPU_ERR_ARG;
}

int accelgpu_get_tuning(accelgpu_ctx *ctx, const char *name, int *value)
{
    if (!ctx || !name || !value) {
        set_error("accelgpu_get_tuning: null argument");
        return ACCELGPU_ERR_ARG;
    }
    if (!strcmp(name, "hits_cap")) {
        *value = (int) std::min<size_t>(std::min(ctx->slots[0].cap, ctx->slots[1].cap), 0x7FFFFFFF);
        return ACCELGPU_OK;
    }
    if (!strcmp(name, "pf_listed")) { // read-only: (row, group)s the last coarse bound listed; -1: list of no use
        unsigned int c[accelk::kPfCtlWords] = {};
        if (ctx->d_ctl) {
            HIPCHK(hipStreamSynchronize(ctx->stream));
            HIPCHK(hipMemcpy(c, ctx->d_ctl, sizeof(c), hipMemcpyDeviceToHost));
        }
        *value = c[accelk::kPfFlag] ? -1 : (int) std::min<unsigned int>(c[0], 0x7FFFFFFFu);
        return ACCELGPU_OK;
    }
    if (!strcmp(name, "pf_last_listed")) { // read-only: the same for the last COLLECTED launch, without touching the device; -2: it did not run
        *value = (int) std::min<int64_t>(ctx->pf_last_listed, 0x7FFFFFFF);
        return ACCELGPU_OK;
    }
    if (!strcmp(name, "pf_skip")) { // read-only: launches still to go without the coarse bound after it was given up
        *value = ctx->pf_skip;
        return ACCELGPU_OK;
    }
    if (!strcmp(name, "hs_prefilter_active")) { // read-only: does the coarse bound (group maxima) run in front of the sum?
        *value = ctx->have_search && prefilter_wanted(ctx) ? 1 : 0;
        return ACCELGPU_OK;
    }
    if (!strcmp(name, "hs_bound_active")) { // read-only: is the plane kept as halves and summed over its bounds?
        *value = ctx->have_search && bounds_in_use(ctx) ? 1 : 0;
        return ACCELGPU_OK;
    }
    for (const TuneName &k : kTuneNames)
        if (!strcmp(name, k.name)) {
            *value = ctx->tune.*(k.field);
            return ACCELGPU_OK;
        }
    set_error("accelgpu_get_tuning: no knob named '%s'", name);
    return ACCELGPU_ERR_ARG;
}

int accelgpu_set_stream(accelgpu_ctx *ctx, void *hip_stream)
{
    int rc = use(ctx);
    if (rc) return rc;
    drop_graphs(ctx);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t) hip_stream : ctx->own_stream;
    return ACCELGPU_OK;
}

int accelgpu_bind_thread(accelgpu_ctx *ctx)
{
    return use(ctx);
}

int accelgpu_synchronize(accelgpu_ctx *ctx)
{
    int rc = use(ctx);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (hipStream_t st : {ctx->copy_stream, ctx->side_stream, ctx->opt_stream})
        if (st) HIPCHK(hipStreamSynchronize(st));
    for (int j = 0; j < accelgpu_ctx::kOwn; j++)
        if (ctx->own_pending[j]) {
            HIPCHK(hipEventSynchronize(ctx->own_ready[j]));
            ctx->own_pending[j] = false;
        }
    return ACCELGPU_OK;
}

int accelgpu_build_kernels(accelgpu_ctx *ctx, int zmax)
{
    int rc = use(ctx);
    if (rc) return rc;
    if (zmax < 0 || zmax > 1200) {
        set_error("accelgpu_build_kernels: zmax %d outside 0..1200", zmax);
        return ACCELGPU_ERR_ARG;
    }
    if (zmax % 2)
        zmax = (zmax / 2 + 1) * 2;
    if (zmax == ctx->zmax)
        return ACCELGPU_OK;
    if (ctx->inflight > 0) {
        set_error("accelgpu_build_kernels: collect the %d launched trial(s) first", ctx->inflight);
        return ACCELGPU_ERR_ARG;
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    drop_graphs(ctx);
    const int numz = (zmax / 2) * 2 + 1;
    const int fftlen = accelhost::calc_fftlen(1, 1, zmax);
    if (fftlen != 8192 && fftlen != 16384) {
        set_error("unsupported FFT length %d", fftlen);
        return ACCELGPU_ERR_ARG;
    }
    if (fftlen != ctx->fftlen) {
        if (fftlen == 8192) {
            if ((rc = make_twiddles<P8K>(&ctx->tw_inv, +1, ctx->stream))) return rc;
            if ((rc = make_pow_tables<P8K>(&ctx->pw_inv, +1, ctx->stream))) return rc;
            if ((rc = make_twiddles<P8K>(&ctx->tw_fwd_full, -1, ctx->stream))) return rc;
            if ((rc = make_twiddles<P4K>(&ctx->tw_fwd_half, -1, ctx->stream))) return rc;
        } else {
            if ((rc = make_twiddles<P16K>(&ctx->tw_inv, +1, ctx->stream))) return rc;
            if ((rc = make_pow_tables<P16K>(&ctx->pw_inv, +1, ctx->stream))) return rc;
            if ((rc = make_twiddles<P16K>(&ctx->tw_fwd_full, -1, ctx->stream))) return rc;
            if ((rc = make_twiddles<P8K>(&ctx->tw_fwd_half, -1, ctx->stream))) return rc;
        }
    }
    std::vector<int> hw(numz);
    int maxhw = 0;
    for (int i = 0; i < numz; i++) {
        hw[i] = accelhost::z_resp_halfwidth((double) (-zmax + 2 * i));
        maxhw = std::max(maxhw, hw[i]);
    }
    free_subbanks(ctx);
    if (ctx->d_bank) HIPCHK(hipFree(ctx->d_bank));
    ctx->d_bank = ctx->d_bankf = nullptr;
    const size_t bank_elems = (size_t) numz * fftlen;
    if (hipMalloc((void **) &ctx->d_bank, 2 * bank_elems * sizeof(c32)) != hipSuccess) { // the bank and its folded copy
        set_error("cannot allocate the kernel bank (%zu bytes)", 2 * bank_elems * sizeof(c32));
        return ACCELGPU_ERR_NOMEM;
    }
    ctx->d_bankf = ctx->d_bank + bank_elems;
    DevTemp<int> hwbuf;
    HIPCHK(hwbuf.alloc((size_t) numz));
    int *d_hw = hwbuf.p;
    HIPCHK(hipMemcpyAsync(d_hw, hw.data(), numz * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_bank, 0, bank_elems * sizeof(c32), ctx->stream));
    const int maxkern = 4 * maxhw;
    hipLaunchKernelGGL(accelk::k_gen_zresp, dim3((maxkern + 127) / 128, numz), dim3(128), 0,
                       ctx->stream, ctx->d_bank, d_hw, zmax, fftlen, maxkern);
    HIPCHK(hipGetLastError());
    if (fftlen == 8192)
        hipLaunchKernelGGL((accelk::k_fft_rows<P8K, -1>), dim3(numz), dim3(P8K::NT),
                           P8K::LDS_BYTES, ctx->stream, ctx->d_bank, ctx->tw_fwd_full.tw1,
                           ctx->tw_fwd_full.tw2);
    else
        hipLaunchKernelGGL((accelk::k_fft_rows<P16K, -1>), dim3(numz), dim3(P16K::NT),
                           P16K::LDS_BYTES, ctx->stream, ctx->d_bank, ctx->tw_fwd_full.tw1,
                           ctx->tw_fwd_full.tw2);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(accelk::k_fold_bank, dim3((unsigned) ((fftlen / 2 + 255) / 256), (unsigned) numz), dim3(256), 0, ctx->stream,
                       ctx->d_bank, ctx->d_bankf, fftlen);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->zmax = zmax;
    ctx->numz = numz;
    ctx->fftlen = fftlen;
    ctx->halfwidth = hw[0];
    return ACCELGPU_OK;
}

int accelgpu_get_kernel(accelgpu_ctx *ctx, int zind, float *out)
{
    int rc = use(ctx);
    if (rc) return rc;
    if (ctx->zmax < 0 || zind < 0 || zind >= ctx->numz || !out) {
        set_error("accelgpu_get_kernel: no bank or zind %d out of range", zind);
        return ACCELGPU_ERR_ARG;
    }
    HIPCHK(hipMemcpyAsync(out, ctx->d_bank + (size_t) zind * ctx->fftlen,
                          (size_t) ctx->fftlen * sizeof(c32), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return ACCELGPU_OK;
}

int accelgpu_load_fft(accelgpu_ctx *ctx, const float *host, int64_t numbins)
{
    int rc = use(ctx);
    if (rc) return rc;
    if (!host || numbins <= 0) {
        set_error("accelgpu_load_fft: null spectrum or numbins <= 0");
        return ACCELGPU_ERR_ARG;
    }
    SideScope side(ctx);
    int j = 0;
    if ((rc = pick_own(ctx, &j))) return rc;
    if ((rc = dev_reserve(&ctx->d_own[j], &ctx->own_cap[j], (size_t) numbins, false, ctx->stream)))
        return rc;
    HIPCHK(hipMemcpyAsync(ctx->d_own[j], host, (size_t) numbins * sizeof(c32),
                          hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream)); // the caller may free `host` on return
    ctx->own_pending[j] = false;
    ctx->cur_own = j;
    ctx->d_fft = ctx->d_own[j];
    ctx->numbins = numbins;
    return ACCELGPU_OK;
}

int accelgpu_host_alloc(void **ptr, size_t bytes)
{
    if (!ptr || bytes == 0) {
        set_error("accelgpu_host_alloc: null output or zero size");
        return ACCELGPU_ERR_ARG;
    }
    *ptr = nullptr;
    if (hipHostMalloc(ptr, bytes, hipHostMallocDefault) != hipSuccess) {
        set_error("cannot page-lock %zu bytes of host memory", bytes);
        return ACCELGPU_ERR_NOMEM;
    }
    return ACCELGPU_OK;
}

int accelgpu_host_free(void *ptr)
{
    if (ptr)
        HIPCHK(hipHostFree(ptr));
    return ACCELGPU_OK;
}

int accelgpu_load_fft_async(accelgpu_ctx *ctx, const float *host, int64_t numbins)
{
    int rc = use(ctx);
    if (rc) return rc;
    if (!host || numbins <= 0) {
        set_error("accelgpu_load_fft_async: null spectrum or numbins <= 0");
        return ACCELGPU_ERR_ARG;
    }
    // a pageable source would make the "asynchronous" copy a staged, blocking one
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, host) != hipSuccess || attr.type != hipMemoryTypeHost) {
        (void) hipGetLastError();
        set_error("accelgpu_load_fft_async: the spectrum must be in page-locked host memory "
                  "(accelgpu_host_alloc); use accelgpu_load_fft for ordinary memory");
        return ACCELGPU_ERR_ARG;
    }
    int j = 0;
    if ((rc = pick_own(ctx, &j))) return rc;
    if ((rc = dev_reserve(&ctx->d_own[j], &ctx->own_cap[j], (size_t) numbins, false, ctx->side_stream)))
        return rc;
    // always the side stream: the copy engine works beside whatever the search stream runs,
    // and the launch that reads this buffer waits for own_ready[j] on the device
    HIPCHK(hipMemcpyAsync(ctx->d_own[j], host, (size_t) numbins * sizeof(c32),
                          hipMemcpyHostToDevice, ctx->side_stream));
    HIPCHK(hipEventRecord(ctx->own_ready[j], ctx->side_stream));
    ctx->own_pending[j] = true;
    ctx->cur_own = j;
    ctx->d_fft = ctx->d_own[j];
    ctx->numbins = numbins;
    return ACCELGPU_OK;
}

int accelgpu_load_fft_device(accelgpu_ctx *ctx, const void *dev, int64_t numbins)
{
    int rc = use(ctx);
    if (rc) return rc;
    if (!dev || numbins <= 0) {
        set_error("accelgpu_load_fft_device: null spectrum or numbins <= 0");
        return ACCELGPU_ERR_ARG;
    }
    ctx->d_fft = (const c32 *) dev;
    ctx->cur_own = -1;
    ctx->numbins = numbins;
    return ACCELGPU_OK;
}

} // extern "C"

namespace {

// Which layout an in-memory search of this many stages keeps its plane in: two planes of 16-bit
// halves for the bounded harmonic sum (hs_bound != 0; every number of stages has a bounded
// kernel), floats otherwise.  Both take 4 bytes per cell.
bool want_split(const accelgpu_ctx *ctx, int numharmstages)
{
    return ctx->tune.hs_bound != 0 && numharmstages >= 1 && numharmstages <= 6;
}

int release_plane(accelgpu_ctx *ctx)
{
    if (ctx->d_plane) HIPCHK(hipFree(ctx->d_plane));
    ctx->d_plane = nullptr;
    ctx->plane_cap = 0;
    if (ctx->d_hi) HIPCHK(hipFree(ctx->d_hi));
    if (ctx->d_lo) HIPCHK(hipFree(ctx->d_lo));
    ctx->d_hi = ctx->d_lo = nullptr;
    ctx->hilo_cap = 0;
    if (ctx->d_mx) HIPCHK(hipFree(ctx->d_mx));
    if (ctx->d_work) HIPCHK(hipFree(ctx->d_work));
    ctx->d_mx = nullptr;
    ctx->d_work = nullptr;
    ctx->mx_cap = ctx->work_cap = 0;
    ctx->pf_ready = ctx->mx_valid = false;
    return ACCELGPU_OK;
}

int set_search_impl(accelgpu_ctx *ctx, const accelgpu_setup *s, int norm_type, float nph, bool inmem)
{
    int rc = use(ctx);
    if (rc) return rc;
    if (!s) {
        set_error("accelgpu_set_search: null setup");
        return ACCELGPU_ERR_ARG;
    }
    if (ctx->zmax < 0 || s->zmax != ctx->zmax) {
        set_error("accelgpu_set_search: build the kernel bank for zmax=%d first", s->zmax);
        return ACCELGPU_ERR_ARG;
    }
    if (norm_type != 0 && norm_type != 1) {
        set_error("accelgpu_set_search: norm_type is 0 (block median) or 1 (-locpow)");
        return ACCELGPU_ERR_ARG;
    }
    if (s->numharmstages < 1 || s->numharmstages > ACCELGPU_MAXSTAGES || s->rlo < 0 ||
        s->highestbin < s->rlo) {
        set_error("accelgpu_set_search: inconsistent setup");
        return ACCELGPU_ERR_ARG;
    }
    if (ctx->inflight > 0) { // their hits are sorted and labelled with the geometry they were launched with
        set_error("accelgpu_set_search: collect the %d launched trial(s) first", ctx->inflight);
        return ACCELGPU_ERR_ARG;
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    drop_graphs(ctx);
    const bool split = inmem && want_split(ctx, s->numharmstages);
    const bool same_geometry = ctx->have_search && ctx->inmem && inmem && ctx->split == split && ctx->setup.rlo == s->rlo &&
                               ctx->setup.highestbin == s->highestbin &&
                               ctx->setup.lobin == s->lobin && ctx->setup.zmax == s->zmax;
    ctx->pf_ready = ctx->mx_valid = ctx->mx_fix_pending = false;
    ctx->setup = *s;
    ctx->setup.numz = ctx->numz;
    ctx->setup.fftlen = ctx->fftlen;
    ctx->setup.halfwidth = ctx->halfwidth;
    ctx->setup.rlen = (s->highestbin + ACCELGPU_USELEN) * 2;
    ctx->setup.plane_bytes = inmem ? ctx->setup.rlen * (int64_t) ctx->numz * 4 : 0;
    ctx->have_search = false;
    ctx->inmem = inmem;
    ctx->norm_type = norm_type;
    ctx->nph = nph;

    // block grid (reference src/accelsearch.c:143-161 and 165-218)
    accelhost::BlockGrid bg;
    accelhost::block_grid(s->rlo, s->highestbin, ctx->halfwidth, &bg);
    const std::vector<int64_t> &lobin = bg.lobin, &col = bg.col;
    const std::vector<int> &numdata = bg.numdata, &ncols = bg.ncols;
    const int nprepop = bg.nprepop, nsearch = bg.nsearch;
    const int64_t search_col0 = bg.search_col0;
    const int nblk = nprepop + nsearch;
    for (int i = 0; i < nblk; i++)
        if (numdata[i] > ctx->fftlen / 2) {
            set_error("block window of %d bins exceeds fftlen/2 = %d", numdata[i], ctx->fftlen / 2);
            return ACCELGPU_ERR_ARG;
        }
    ctx->nblk = nblk;
    ctx->nblk_active = nblk;
    ctx->partial_fill = false;
    ctx->range_first = 0;
    ctx->range_count = nsearch;
    ctx->h_blk_lobin = lobin;
    ctx->h_blk_col = col;
    ctx->h_blk_numdata = numdata;
    ctx->h_blk_ncols = ncols;
    ctx->grid_host = bg;
    ctx->nprepop = nprepop;
    ctx->nsearch = nsearch;
    ctx->setup.num_prepop_blocks = nprepop;
    ctx->setup.num_search_blocks = nsearch;
    ctx->col0 = search_col0;
    ctx->ncols = (int64_t) nsearch * ACCELGPU_USELEN;
    if (!inmem) {
        // standard mode: no plane, no block grid on the device
        if ((rc = release_plane(ctx))) return rc;
        ctx->split = false;
        ctx->have_search = true;
        return ACCELGPU_OK;
    }

    if (nblk > 0) {
        const size_t want = (size_t) nblk;
        if (ctx->blk_cap < want) {
            for (void *p : {(void *) ctx->d_blk_lobin, (void *) ctx->d_blk_col,
                            (void *) ctx->d_blk_numdata, (void *) ctx->d_blk_ncols, (void *) ctx->d_blk_zrange})
                if (p) HIPCHK(hipFree(p));
            ctx->d_blk_lobin = ctx->d_blk_col = nullptr;
            ctx->d_blk_numdata = ctx->d_blk_ncols = nullptr;
            ctx->d_blk_zrange = nullptr;
            HIPCHK(hipMalloc((void **) &ctx->d_blk_zrange, want * sizeof(int2)));
            HIPCHK(hipMalloc((void **) &ctx->d_blk_lobin, want * sizeof(int64_t)));
            HIPCHK(hipMalloc((void **) &ctx->d_blk_col, want * sizeof(int64_t)));
            HIPCHK(hipMalloc((void **) &ctx->d_blk_numdata, want * sizeof(int)));
            HIPCHK(hipMalloc((void **) &ctx->d_blk_ncols, want * sizeof(int)));
            ctx->blk_cap = want;
        }
        HIPCHK(hipMemcpy(ctx->d_blk_lobin, lobin.data(), want * sizeof(int64_t), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(ctx->d_blk_col, col.data(), want * sizeof(int64_t), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(ctx->d_blk_numdata, numdata.data(), want * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(ctx->d_blk_ncols, ncols.data(), want * sizeof(int), hipMemcpyHostToDevice));
        if ((rc = dev_reserve(&ctx->d_blockfft, &ctx->blockfft_cap,
                              (size_t) nblk * (ctx->fftlen / 2), false, ctx->stream)))
            return rc;
    }

    // the plane: zero-filled so that cells below r = 8, which nothing writes, read as 0
    const size_t plane_elems = (size_t) ctx->setup.rlen * ctx->numz;
    if (!ctx->d_flags) {
        HIPCHK(hipMalloc((void **) &ctx->d_flags, 64));
        HIPCHK(hipMemsetAsync(ctx->d_flags, 0, 64, ctx->stream));
    }
    if (split) {
        if (ctx->d_plane && (rc = release_plane(ctx))) return rc; // the other layout's memory goes first
        const bool fresh = ctx->hilo_cap < plane_elems || !ctx->d_hi || !ctx->d_lo;
        if (fresh) {
            if ((rc = release_plane(ctx))) return rc;
            // (the several-columns-per-lane sum reads up to 192 cells past a row's searched part: inside
            // the row; 64 bytes of slack all the same)
            const size_t bytes = plane_elems * sizeof(unsigned short) + 64;
            if (hipMalloc((void **) &ctx->d_hi, bytes) != hipSuccess || hipMalloc((void **) &ctx->d_lo, bytes) != hipSuccess) {
                (void) hipGetLastError();
                (void) release_plane(ctx);
                set_error("hipMalloc of the plane (2 x %zu bytes) failed", bytes);
                return ACCELGPU_ERR_NOMEM;
            }
            ctx->hilo_cap = plane_elems;
        }
        if (fresh || !same_geometry) {
            HIPCHK(hipMemsetAsync(ctx->d_hi, 0, plane_elems * sizeof(unsigned short), ctx->stream));
            HIPCHK(hipMemsetAsync(ctx->d_lo, 0, plane_elems * sizeof(unsigned short), ctx->stream));
            HIPCHK(hipMemsetAsync(ctx->d_flags, 0, 64, ctx->stream));
            ctx->host_wrote_outside = false;
        }
    } else {
        if (ctx->d_hi && (rc = release_plane(ctx))) return rc;
        const bool fresh = ctx->plane_cap < plane_elems || !ctx->d_plane;
        if ((rc = dev_reserve(&ctx->d_plane, &ctx->plane_cap, plane_elems, false, ctx->stream)))
            return rc;
        if (fresh || !same_geometry)
            HIPCHK(hipMemsetAsync(ctx->d_plane, 0, plane_elems * sizeof(float), ctx->stream));
    }
    ctx->split = split;

    // sub-harmonic source rows, in summing order (stage ascending, odd harmonic
    // ascending), as byte offsets into the plane: [numz][nh]
    const int nh = (1 << (s->numharmstages - 1)) - 1;
    if (nh > 0) {
        std::vector<int64_t> tab((size_t) nh * ctx->numz);
        int i = 0;
        for (int st = 1; st < s->numharmstages; st++)
            for (int h = 1; h < (1 << st); h += 2, i++) {
                const double f = (double) h / (double) (1 << st);
                for (int zi = 0; zi < ctx->numz; zi++) {
                    const int zz = (int) s->zlo + zi * 2;
                    const int subz = accelhost::calc_required_z(f, zz);
                    const int zind = accelhost::index_from_z(subz, s->zlo);
                    tab[(size_t) zi * nh + i] = (int64_t) zind * ctx->setup.rlen * 4;
                }
            }
        if ((rc = dev_reserve(&ctx->d_rowoff, &ctx->rowoff_cap, tab.size(), false, ctx->stream)))
            return rc;
        HIPCHK(hipMemcpy(ctx->d_rowoff, tab.data(), tab.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        std::vector<int64_t> half(tab.size());
        for (size_t k = 0; k < tab.size(); k++)
            half[k] = tab[k] / 2;
        if ((rc = dev_reserve(&ctx->d_rowhalf, &ctx->rowhalf_cap, half.size(), false, ctx->stream)))
            return rc;
        HIPCHK(hipMemcpy(ctx->d_rowhalf, half.data(), half.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        std::vector<uint32_t> chg((size_t) ctx->numz);
        for (int zi = 0; zi < ctx->numz; zi++) {
            uint32_t m = 0;
            for (int k = 0; k < nh; k++)
                if (zi == 0 || tab[(size_t) zi * nh + k] != tab[(size_t) (zi - 1) * nh + k])
                    m |= 1u << k;
            chg[zi] = m;
        }
        if ((rc = dev_reserve(&ctx->d_rowchg, &ctx->rowchg_cap, chg.size(), false, ctx->stream)))
            return rc;
        HIPCHK(hipMemcpy(ctx->d_rowchg, chg.data(), chg.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    } else if ((rc = dev_reserve(&ctx->d_rowoff, &ctx->rowoff_cap, 1, false, ctx->stream))) {
        return rc;
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->have_search = true;
    return ACCELGPU_OK;
}
} // namespace

extern "C" {

int accelgpu_set_search(accelgpu_ctx *ctx, const accelgpu_setup *s, int norm_type, float nph)
{
    return set_search_impl(ctx, s, norm_type, nph, true);
}

int accelgpu_set_block_range(accelgpu_ctx *ctx, int32_t first_block, int32_t num_blocks,
                             int32_t *blocks_generated)
{
    int rc = use(ctx);
    if (rc) return rc;
    if ((rc = check_ready(ctx, false))) return rc;
    if ((rc = need_plane(ctx))) return rc;
    if (ctx->inflight > 0) {
        set_error("accelgpu_set_block_range: collect the launched trial(s) first");
        return ACCELGPU_ERR_ARG;
    }
    if (first_block < 0 || num_blocks < 0 || first_block + num_blocks > ctx->nsearch) {
        set_error("accelgpu_set_block_range: [%d, %d) outside the %d search blocks", first_block,
                  first_block + num_blocks, ctx->nsearch);
        return ACCELGPU_ERR_ARG;
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    drop_graphs(ctx);
    std::vector<int> z0, z1;
    accelhost::block_range_need(&ctx->setup, ctx->grid_host, first_block, num_blocks, &z0, &z1, nullptr);
    const bool whole = first_block == 0 && num_blocks == ctx->nsearch; // every block, every row: no row ranges needed
    // The block grid on the device stays whole (the plane of group maxima is indexed by a block's place in it);
    // a fill takes the rows of each block from this table, and blocks with none cost it a workgroup that
    // returns at once.
    std::vector<int2> zr((size_t) ctx->nblk);
    size_t n = 0;
    for (int b = 0; b < ctx->nblk; b++) {
        zr[b] = whole ? make_int2(0, ctx->numz) : make_int2(z0[b], z1[b]);
        n += zr[b].y > zr[b].x;
    }
    if (ctx->nblk > 0)
        HIPCHK(hipMemcpy(ctx->d_blk_zrange, zr.data(), zr.size() * sizeof(int2), hipMemcpyHostToDevice));
    ctx->nblk_active = (int) n;
    ctx->partial_fill = !whole;
    ctx->range_first = first_block;
    ctx->range_count = num_blocks;
    if (blocks_generated)
        *blocks_generated = (int32_t) n;
    return ACCELGPU_OK;
}

int accelgpu_fill_plane(accelgpu_ctx *ctx)
{
    int rc = use(ctx);
    if (rc) return rc;
    if ((rc = check_ready(ctx, true))) return rc;
    if ((rc = need_plane(ctx))) return rc;
    ctx->plane_serial = 0; // no launched trial's plane any more: a repeat of one regenerates its own
    decide_prefilter(ctx);
    if ((rc = queue_fill(ctx))) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return ACCELGPU_OK;
}

int accelgpu_search_plane(accelgpu_ctx *ctx, accelgpu_hit *hits, int64_t max_hits, int64_t *n_hits)
{
    int rc = use(ctx);
    if (rc) return rc;
    if ((rc = check_ready(ctx, false))) return rc;
    if ((rc = need_plane(ctx))) return rc;
    if (!n_hits || (!hits && max_hits > 0)) {
        set_error("accelgpu_search_plane: null output");
        return ACCELGPU_ERR_ARG;
    }
    if (ctx->inflight > 0) {
        set_error("accelgpu_search_plane: collect the %d launched trial(s) first", ctx->inflight);
        return ACCELGPU_ERR_ARG;
    }
    *n_hits = 0;
    if (ctx->range_count == 0)
        return ACCELGPU_OK;
    const int si = ctx->head;
    select_slot(ctx, si);
    ctx->slots[si].prefetch = 0;
    ctx->slots[si].serial = ctx->plane_serial; // searches the plane as it stands
    ctx->slots[si].pf_counted = false;
    if ((rc = queue_search(ctx))) return rc;
    HIPCHK(hipEventRecord(ctx->slots[si].done, ctx->stream));
    return finish_search(ctx, si, hits, max_hits, n_hits);
}

int accelgpu_run_launch(accelgpu_ctx *ctx)
{
    int rc = use(ctx);
    if (rc) return rc;
    if ((rc = check_ready(ctx, true, false))) return rc;
    if ((rc = need_plane(ctx))) return rc;
    if (ctx->inflight >= 2) {
        set_error("accelgpu_run_launch: two launches are already in flight; collect one first");
        return ACCELGPU_ERR_ARG;
    }
    const int si = ctx->tail;
    accelgpu_ctx::HitSlot &sl = ctx->slots[si];
    select_slot(ctx, si);
    sl.prefetch = 0;
    sl.fft = ctx->d_fft;
    sl.numbins = ctx->numbins;
    sl.serial = ++ctx->launch_serial;
    sl.pf_counted = false;
    decide_prefilter(ctx);
    if (ctx->range_count > 0) {
        if ((rc = spectrum_ready(ctx, ctx->stream))) return rc;
        const bool whole_grid = !ctx->partial_fill && ctx->range_first == 0 && ctx->range_count == ctx->nsearch;
        const bool small = ctx->setup.plane_bytes <= ((int64_t) 64 << 20);
        if (ctx->tune.use_graph && small && whole_grid && ctx->tune.hs_chunk_blocks == 0 && !ctx->profiling &&
            ctx->stream == ctx->own_stream) {
            if ((rc = launch_as_graph(ctx, si))) return rc;
            ctx->plane_serial = sl.serial;
            sl.prefetch = prefetch_hits(ctx, sl);
            HIPCHK(hipEventRecord(sl.done, ctx->stream));
            ctx->tail ^= 1;
            ctx->inflight++;
            return ACCELGPU_OK;
        }
        HIPCHK(hipMemsetAsync(ctx->d_count, 0, kCountBytes, ctx->stream));
        if (ctx->tune.hs_chunk_blocks > 0 && whole_grid) {
            if ((rc = queue_fill_and_sum_chunked(ctx, ctx->tune.hs_chunk_blocks, slot_target(ctx)))) return rc;
            ctx->plane_serial = sl.serial;
        } else {
            if ((rc = queue_fill(ctx))) return rc;
            ctx->plane_serial = sl.serial;
            if ((rc = launch_harmsum<false>(ctx, ctx->setup.numharmstages,
                                            ctx->col0 + (int64_t) ctx->range_first * ACCELGPU_USELEN,
                                            (int64_t) ctx->range_count * ACCELGPU_USELEN, nullptr, 0,
                                            slot_target(ctx))))
                return rc;
        }
        // the copy stream takes over once this trial's last kernel is done, so the next
        // trial's kernels never queue behind PCIe transfers
        HIPCHK(hipEventRecord(sl.kernels, ctx->stream));
        HIPCHK(hipStreamWaitEvent(ctx->copy_stream, sl.kernels, 0));
        HIPCHK(hipMemcpyAsync(sl.h_count, sl.d_count, kCountBytes, hipMemcpyDeviceToHost,
                              ctx->copy_stream));
        // speculative copy of the head of the hit buffer: usually all of them
        const size_t pre = prefetch_hits(ctx, sl);
        HIPCHK(hipMemcpyAsync(sl.h_hits, sl.d_hits, pre * sizeof(DevHit), hipMemcpyDeviceToHost,
                              ctx->copy_stream));
        sl.prefetch = pre;
        HIPCHK(hipEventRecord(sl.done, ctx->copy_stream));
    } else {
        HIPCHK(hipEventRecord(sl.done, ctx->stream));
    }
    ctx->tail ^= 1;
    ctx->inflight++;
    return ACCELGPU_OK;
}

int accelgpu_run_collect(accelgpu_ctx *ctx, accelgpu_hit *hits, int64_t max_hits, int64_t *n_hits)
{
    int rc = use(ctx);
    if (rc) return rc;
    if (ctx->inflight == 0) {
        set_error("accelgpu_run_collect: nothing launched");
        return ACCELGPU_ERR_ARG;
    }
    if (!n_hits || (!hits && max_hits > 0)) {
        set_error("accelgpu_run_collect: null output");
        return ACCELGPU_ERR_ARG;
    }
    const int si = ctx->head;
    *n_hits = 0;
    if (ctx->range_count == 0) {
        HIPCHK(hipEventSynchronize(ctx->slots[si].done));
    } else {
        // The trial stays at the head of the queue until its hits have been handed over:
        // after any failure (a caller's buffer too small: ACCELGPU_ERR_OVERFLOW with the
        // count in *n_hits) the same call can simply be made again.
        rc = finish_search(ctx, si, hits, max_hits, n_hits);
        if (rc) return rc;
    }
    ctx->head ^= 1;
    ctx->inflight--;
    ctx->collected.fft = ctx->slots[si].fft;
    ctx->collected.numbins = ctx->slots[si].numbins;
    ctx->collected.valid = true;
    return ACCELGPU_OK;
}

int accelgpu_run(accelgpu_ctx *ctx, accelgpu_hit *hits, int64_t max_hits, int64_t *n_hits)
{
    if (ctx && ctx->inflight > 0) { // the collect below would return the oldest launch, not this one
        set_error("accelgpu_run: collect the %d launched trial(s) first", ctx->inflight);
        return ACCELGPU_ERR_ARG;
    }
    int rc = accelgpu_run_launch(ctx);
    if (rc) return rc;
    rc = accelgpu_run_collect(ctx, hits, max_hits, n_hits);
    if (rc && ctx->inflight > 0) {
        // This call is the whole search or nothing: a trial whose hits could not be handed
        // over (caller's buffer too small, *n_hits says how many) does not stay queued, so
        // that calling accelgpu_run() again with a larger buffer starts from a clean queue.
        // It is the most recent launch: the one at the tail.
        (void) hipStreamSynchronize(ctx->stream);
        (void) hipStreamSynchronize(ctx->copy_stream);
        ctx->tail ^= 1;
        ctx->inflight--;
    }
    return rc;
}

int accelgpu_fund_block(accelgpu_ctx *ctx, double startr, float *powers, int32_t *rlo_out)
{
    int rc = use(ctx);
    if (rc) return rc;
    if ((rc = check_ready(ctx, true))) return rc;
    if (!powers) {
        set_error("accelgpu_fund_block: null output");
        return ACCELGPU_ERR_ARG;
    }
    const BlockDesc d = describe_block(startr, ctx->halfwidth);
    const int64_t zero = 0;
    const int ncols = ACCELGPU_USELEN;
    const size_t elems = (size_t) ctx->numz * ACCELGPU_USELEN;
    if ((rc = dev_reserve(&ctx->d_tmp, &ctx->tmp_cap, elems, false, ctx->stream))) return rc;
    if (!ctx->d_one_blockfft)
        HIPCHK(hipMalloc((void **) &ctx->d_one_blockfft, (size_t) 8192 * sizeof(c32)));
    HIPCHK(hipMemcpyAsync(ctx->d_one_lobin, &d.lobin, sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->d_one_numdata, &d.numdata, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->d_one_col, &zero, sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->d_one_ncols, &ncols, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream)); // sources are stack variables
    const bool prof = ctx->profiling;
    ctx->profiling = false;
    rc = launch_block_prep(ctx, 1, ctx->d_one_lobin, ctx->d_one_numdata, ctx->d_one_blockfft);
    if (!rc)
        rc = launch_correlate(ctx, 1, ctx->d_one_blockfft, ctx->d_one_col, ctx->d_one_ncols,
                              ctx->d_tmp, ACCELGPU_USELEN);
    ctx->profiling = prof;
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(powers, ctx->d_tmp, elems * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (rlo_out)
        *rlo_out = d.rlo;
    return ACCELGPU_OK;
}

// accelgpu_fund_block() for callers that cannot allocate: up to kNoAllocBlocks blocks per
// call, block spectra and descriptors in this device's `scratch` array, powers into the
// caller's device buffer.  No hipMalloc/hipFree, no host-device copies and no host wait (the
// stream waits for an asynchronous upload of the current spectrum that is still on its way):
// the work is queued on the context's stream.  Calls on one GPU share that GPU's array, so
// calls that overlap in time there (two contexts, two streams) race.
int accelgpu_fund_blocks_noalloc(accelgpu_ctx *ctx, const double *startr, int32_t nblocks,
                                 float *powers_dev, int32_t *rlo_out)
{
    int rc = use(ctx);
    if (rc) return rc;
    if ((rc = check_ready(ctx, true, false))) return rc;
    if (!startr || !powers_dev || nblocks < 1) {
        set_error("accelgpu_fund_blocks_noalloc: null argument or no blocks");
        return ACCELGPU_ERR_ARG;
    }
    if (nblocks > kNoAllocBlocks) {
        set_error("accelgpu_fund_blocks_noalloc: %d blocks in one call, the fixed working memory holds %d",
                  nblocks, kNoAllocBlocks);
        return ACCELGPU_ERR_ARG;
    }
    static_assert(kScratchBlockfftBytes >= (size_t) kNoAllocBlocks * (16384 / 2) * sizeof(c32),
                  "working memory sized for the longest plan");
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, powers_dev) != hipSuccess || attr.type != hipMemoryTypeDevice ||
        attr.device != ctx->device) {
        (void) hipGetLastError();
        set_error("accelgpu_fund_blocks_noalloc: the output must be device memory of the context's GPU (%d)", ctx->device);
        return ACCELGPU_ERR_ARG;
    }
    // a spectrum still on its way (accelgpu_load_fft_async): the stream waits for it, not the host
    if ((rc = spectrum_ready(ctx, ctx->stream))) return rc;
    unsigned char *base = g_scratch_ptr[ctx->device];
    if (!base) {
        set_error("accelgpu_fund_blocks_noalloc: no working memory on device %d", ctx->device);
        return ACCELGPU_ERR_HIP;
    }
    int64_t *d_lobin = reinterpret_cast<int64_t *>(base);
    int64_t *d_col = d_lobin + kNoAllocBlocks;
    int *d_numdata = reinterpret_cast<int *>(d_col + kNoAllocBlocks);
    int *d_ncols = d_numdata + kNoAllocBlocks;
    c32 *d_blockfft = reinterpret_cast<c32 *>(base + kScratchDescBytes);
    static_assert(kNoAllocBlocks * (2 * sizeof(int64_t) + 2 * sizeof(int)) <= kScratchDescBytes, "descriptors fit");

    // output layout [block][z][column]: row stride USELEN, block b starts numz rows further on
    accelk::BlockDescArgs a{};
    for (int b = 0; b < nblocks; b++) {
        const BlockDesc d = describe_block(startr[b], ctx->halfwidth);
        if (d.numdata > ctx->fftlen / 2) {
            set_error("block window of %d bins exceeds fftlen/2 = %d", d.numdata, ctx->fftlen / 2);
            return ACCELGPU_ERR_ARG;
        }
        a.lobin[b] = d.lobin;
        a.col[b] = (int64_t) b * ctx->numz * ACCELGPU_USELEN;
        a.numdata[b] = d.numdata;
        a.ncols[b] = ACCELGPU_USELEN;
        if (rlo_out)
            rlo_out[b] = d.rlo;
    }
    hipLaunchKernelGGL(accelk::k_write_desc, dim3(1), dim3(64), 0, ctx->stream, a, (int) nblocks, d_lobin,
                       d_col, d_numdata, d_ncols);
    HIPCHK(hipGetLastError());
    const bool prof = ctx->profiling; // the timing brackets create events
    ctx->profiling = false;
    rc = launch_block_prep(ctx, nblocks, d_lobin, d_numdata, d_blockfft);
    if (!rc)
        rc = launch_correlate(ctx, nblocks, d_blockfft, d_col, d_ncols, powers_dev, ACCELGPU_USELEN);
    ctx->profiling = prof;
    return rc;
}

int accelgpu_summed_block(accelgpu_ctx *ctx, int block, int stage, float *powers)
{
    int rc = use(ctx);
    if (rc) return rc;
    if ((rc = check_ready(ctx, false))) return rc;
    if ((rc = need_plane(ctx))) return rc;
    if (block < 0 || block >= ctx->nsearch || stage < 0 || stage >= ctx->setup.numharmstages ||
        !powers) {
        set_error("accelgpu_summed_block: block %d / stage %d out of range", block, stage);
        return ACCELGPU_ERR_ARG;
    }
    const size_t elems = (size_t) ctx->numz * ACCELGPU_USELEN;
    if ((rc = dev_reserve(&ctx->d_tmp, &ctx->tmp_cap, elems, false, ctx->stream))) return rc;
    rc = launch_harmsum<true>(ctx, ctx->setup.numharmstages, ctx->col0 + (int64_t) block * ACCELGPU_USELEN,
                              ACCELGPU_USELEN, ctx->d_tmp, stage, slot_target(ctx));
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(powers, ctx->d_tmp, elems * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return ACCELGPU_OK;
}

// A host is about to write plane columns [col, col + count): does a whole plane generation rewrite them all?
static void note_host_write(accelgpu_ctx *ctx, int64_t col, int64_t count)
{
    const int64_t lo = ctx->nblk > 0 ? ctx->h_blk_col[0] : 0;
    const int64_t hi = ctx->col0 + (int64_t) ctx->nsearch * ACCELGPU_USELEN;
    if (ctx->nblk == 0 || col < lo || col + count > hi)
        ctx->host_wrote_outside = true;
}

static int stage_of(int numharm)
{
    int st = 0;
    for (int n = numharm; n > 1; n >>= 1)
        st++;
    return (1 << st) == numharm ? st : -1;
}

// Uploads a host block into the context's block buffer.
static int upload_block(accelgpu_ctx *ctx, const float *powers)
{
    const size_t elems = (size_t) ctx->numz * ACCELGPU_USELEN;
    int rc = dev_reserve(&ctx->d_tmp, &ctx->tmp_cap, elems, false, ctx->stream);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(ctx->d_tmp, powers, elems * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    return ACCELGPU_OK;
}

int accelgpu_fund_to_ffdotplane(accelgpu_ctx *ctx, const float *powers, int32_t rlo)
{
    int rc = use(ctx);
    if (rc) return rc;
    if ((rc = check_ready(ctx, false))) return rc;
    if ((rc = need_plane(ctx))) return rc;
    const int64_t col = (int64_t) rlo * 2;
    if (!powers || col < 0 || col + ACCELGPU_USELEN > ctx->setup.rlen) {
        set_error("accelgpu_fund_to_ffdotplane: block at rlo=%d does not fit the plane", rlo);
        return ACCELGPU_ERR_ARG;
    }
    ctx->plane_serial = 0;
    ctx->mx_valid = false; // the group maxima no longer describe the plane
    note_host_write(ctx, col, ACCELGPU_USELEN);
    if (ctx->split) {
        // through the block buffer: the floats are cut into the two planes of halves on the device
        if ((rc = upload_block(ctx, powers))) return rc;
        hipLaunchKernelGGL(accelk::k_split_span, dim3((ACCELGPU_USELEN + 255) / 256, ctx->numz), dim3(256), 0, ctx->stream,
                           ctx->d_tmp, (int64_t) ACCELGPU_USELEN, ctx->d_hi + col, ctx->d_lo + col, ctx->setup.rlen,
                           (int64_t) ACCELGPU_USELEN, ctx->d_flags);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(ctx->stream));
        return ACCELGPU_OK;
    }
    HIPCHK(hipMemcpy2DAsync(ctx->d_plane + col, (size_t) ctx->setup.rlen * sizeof(float), powers,
                            ACCELGPU_USELEN * sizeof(float), ACCELGPU_USELEN * sizeof(float),
                            ctx->numz, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return ACCELGPU_OK;
}

int accelgpu_inmem_add_ffdotpows(accelgpu_ctx *ctx, float *powers, int32_t rlo, int numharm,
                                 int harmnum)
{
    int rc = use(ctx);
    if (rc) return rc;
    if ((rc = check_ready(ctx, false))) return rc;
    if ((rc = need_plane(ctx))) return rc;
    const int st = stage_of(numharm);
    if (!powers || st < 1 || st > 5 || harmnum < 1 || harmnum >= numharm || rlo < 0 ||
        (int64_t) rlo * 2 + ACCELGPU_USELEN > ctx->setup.rlen) {
        set_error("accelgpu_inmem_add_ffdotpows: bad numharm/harmnum/rlo (%d, %d, %d)", numharm,
                  harmnum, rlo);
        return ACCELGPU_ERR_ARG;
    }
    std::vector<int> zind(ctx->numz);
    const double f = (double) harmnum / (double) numharm;
    for (int zi = 0; zi < ctx->numz; zi++) {
        const int zz = (int) ctx->setup.zlo + zi * 2;
        zind[zi] = accelhost::index_from_z(accelhost::calc_required_z(f, zz), ctx->setup.zlo);
    }
    DevTemp<int> zbuf;
    HIPCHK(zbuf.alloc(zind.size()));
    int *d_zind = zbuf.p;
    HIPCHK(hipMemcpyAsync(d_zind, zind.data(), zind.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    if ((rc = upload_block(ctx, powers))) return rc;
    if (ctx->split)
        hipLaunchKernelGGL(accelk::k_block_add_harm_split, dim3((ACCELGPU_USELEN + 255) / 256, ctx->numz),
                           dim3(256), 0, ctx->stream, ctx->d_tmp, ctx->d_hi, ctx->d_lo, ctx->setup.rlen, rlo, st,
                           harmnum, d_zind);
    else
        hipLaunchKernelGGL(accelk::k_block_add_harm, dim3((ACCELGPU_USELEN + 255) / 256, ctx->numz),
                           dim3(256), 0, ctx->stream, ctx->d_tmp, ctx->d_plane, ctx->setup.rlen, rlo, st,
                           harmnum, d_zind);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(powers, ctx->d_tmp, (size_t) ctx->numz * ACCELGPU_USELEN * sizeof(float),
                          hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return ACCELGPU_OK;
}

int accelgpu_search_ffdotpows(accelgpu_ctx *ctx, const float *powers, int32_t rlo, int numharm,
                              accelgpu_hit *hits, int64_t max_hits, int64_t *n_hits)
{
    int rc = use(ctx);
    if (rc) return rc;
    if ((rc = check_ready(ctx, false))) return rc;
    const int st = stage_of(numharm);
    if (!powers || !n_hits || (!hits && max_hits > 0) || st < 0 || st >= ctx->setup.numharmstages) {
        set_error("accelgpu_search_ffdotpows: bad arguments (numharm %d)", numharm);
        return ACCELGPU_ERR_ARG;
    }
    if (ctx->inflight > 0) { // the hit buffers hold the launched trials' hits
        set_error("accelgpu_search_ffdotpows: collect the %d launched trial(s) first", ctx->inflight);
        return ACCELGPU_ERR_ARG;
    }
    const int64_t bcol = (int64_t) rlo * 2;
    for (;;) {
        if ((rc = upload_block(ctx, powers))) return rc;
        HIPCHK(hipMemsetAsync(ctx->d_count, 0, sizeof(unsigned int), ctx->stream));
        const unsigned int cap = (unsigned int) std::min<size_t>(ctx->hits_cap, 0xFFFFFFFFu);
        hipLaunchKernelGGL(accelk::k_block_threshold, dim3((ACCELGPU_USELEN + 255) / 256, ctx->numz),
                           dim3(256), 0, ctx->stream, ctx->d_tmp, ctx->setup.powcut[st], st, bcol,
                           ctx->d_hits, ctx->d_count, cap);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(ctx->h_count, ctx->d_count, sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (*ctx->h_count <= ctx->hits_cap)
            break;
        if ((rc = reserve_hits(ctx, (size_t) *ctx->h_count + 1024))) return rc;
    }
    const size_t n = *ctx->h_count;
    *n_hits = (int64_t) n;
    if ((int64_t) n > max_hits) {
        set_error("%zu hits but room for %lld: re-run with a larger buffer", n, (long long) max_hits);
        return ACCELGPU_ERR_OVERFLOW;
    }
    if (n) {
        HIPCHK(hipMemcpyAsync(ctx->h_hits, ctx->d_hits, n * sizeof(DevHit), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    const int64_t rel = bcol - ctx->col0;
    const int32_t block = (rel >= 0 && rel % ACCELGPU_USELEN == 0) ? (int32_t) (rel / ACCELGPU_USELEN) : -1;
    for (size_t i = 0; i < n; i++) {
        const DevHit &d = ctx->h_hits[i];
        hits[i].power = d.power;
        hits[i].stage = (int32_t) (d.stage_zind >> 16);
        hits[i].zind = (int32_t) (d.stage_zind & 0xFFFF);
        hits[i].rind = d.rind;
        hits[i].block = block;
    }
    std::sort(hits, hits + n, [](const accelgpu_hit &a, const accelgpu_hit &b) {
        if (a.zind != b.zind) return a.zind < b.zind;
        return a.rind < b.rind;
    });
    return ACCELGPU_OK;
}

static int plane_span(accelgpu_ctx *ctx, int zind, int64_t col, int64_t count)
{
    if (!ctx->have_search || !ctx->inmem || !(ctx->split ? ctx->d_hi && ctx->d_lo : ctx->d_plane != nullptr)) {
        set_error("no plane allocated (accelgpu_set_search)");
        return ACCELGPU_ERR_ARG;
    }
    if (zind < 0 || zind >= ctx->numz || col < 0 || count < 0 || col + count > ctx->setup.rlen) {
        set_error("plane span z=%d col=%lld count=%lld out of range", zind, (long long) col,
                  (long long) count);
        return ACCELGPU_ERR_ARG;
    }
    return ACCELGPU_OK;
}

int accelgpu_get_plane(accelgpu_ctx *ctx, int zind, int64_t col, int64_t count, float *out)
{
    int rc = use(ctx);
    if (rc) return rc;
    if ((rc = plane_span(ctx, zind, col, count))) return rc;
    if (ctx->split) {
        if (count == 0)
            return ACCELGPU_OK;
        if ((rc = dev_reserve(&ctx->d_tmp, &ctx->tmp_cap, (size_t) count, false, ctx->stream))) return rc;
        const size_t cell = (size_t) zind * ctx->setup.rlen + col;
        hipLaunchKernelGGL(accelk::k_join_span, dim3((unsigned) ((count + 255) / 256), 1), dim3(256), 0, ctx->stream,
                           ctx->d_tmp, count, ctx->d_hi + cell, ctx->d_lo + cell, ctx->setup.rlen, count);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(out, ctx->d_tmp, (size_t) count * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        return ACCELGPU_OK;
    }
    HIPCHK(hipMemcpyAsync(out, ctx->d_plane + (size_t) zind * ctx->setup.rlen + col,
                          (size_t) count * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return ACCELGPU_OK;
}

int accelgpu_put_plane(accelgpu_ctx *ctx, int zind, int64_t col, int64_t count, const float *in)
{
    int rc = use(ctx);
    if (rc) return rc;
    if ((rc = plane_span(ctx, zind, col, count))) return rc;
    ctx->plane_serial = 0;
    ctx->mx_valid = false; // the group maxima no longer describe the plane
    note_host_write(ctx, col, count);
    if (ctx->split) {
        if (count == 0)
            return ACCELGPU_OK;
        if ((rc = dev_reserve(&ctx->d_tmp, &ctx->tmp_cap, (size_t) count, false, ctx->stream))) return rc;
        const size_t cell = (size_t) zind * ctx->setup.rlen + col;
        HIPCHK(hipMemcpyAsync(ctx->d_tmp, in, (size_t) count * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(accelk::k_split_span, dim3((unsigned) ((count + 255) / 256), 1), dim3(256), 0, ctx->stream,
                           ctx->d_tmp, count, ctx->d_hi + cell, ctx->d_lo + cell, ctx->setup.rlen, count, ctx->d_flags);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(ctx->stream));
        return ACCELGPU_OK;
    }
    HIPCHK(hipMemcpyAsync(ctx->d_plane + (size_t) zind * ctx->setup.rlen + col, in,
                          (size_t) count * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return ACCELGPU_OK;
}

int accelgpu_set_profiling(accelgpu_ctx *ctx, int on)
{
    int rc = use(ctx);
    if (rc) return rc;
    ctx->profiling = on != 0;
    return ACCELGPU_OK;
}

int accelgpu_get_timings(accelgpu_ctx *ctx, int64_t launches[3], double total_ms[3], int reset)
{
    int rc = use(ctx);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 3; k++) {
        for (auto &e : ctx->ev[k]) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, e.a, e.b));
            ctx->prof_ms[k] += ms;
            ctx->prof_launches[k]++;
            ctx->ev_pool.push_back(e);
        }
        ctx->ev[k].clear();
        if (launches) launches[k] = ctx->prof_launches[k];
        if (total_ms) total_ms[k] = ctx->prof_ms[k];
        if (reset) {
            ctx->prof_launches[k] = 0;
            ctx->prof_ms[k] = 0;
        }
    }
    return ACCELGPU_OK;
}

} // extern "C"

#include "accel_standard.inc"
#include "accel_spectrum.inc"
#include "accel_optimize.inc"
