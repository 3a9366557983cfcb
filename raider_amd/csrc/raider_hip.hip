// libraider_hip.so - C ABI (include/raider_hip.h) + the remaining kernels.  gfx950 only.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC raider_hip.hip -o libraider_hip.so
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/raider_hip.h"
#include "raider_kernels.h"

using namespace rdr;

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

constexpr int MAX_SLICES = 512;   // height slices per rdr_raytrace_slices call

enum { SLOT_IN0 = 0, SLOT_IN1, SLOT_IN2, SLOT_IN3, SLOT_IN4, SLOT_IN5, SLOT_IN6, SLOT_OUT0, SLOT_OUT1, SLOT_OUT2, SLOT_AUX,
       SLOT_PT0, SLOT_PT1, SLOT_PT2, SLOT_PT3, SLOT_PT4, SLOT_PT5, SLOT_TMPCUBE, SLOT_TMPAXES,      // rdr_point_delays: points, delays, the intermediate cube
       NSLOT };

struct rdr_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t copy_stream = nullptr;      // host<->device transfers of the pipelined host-buffer ray tracing
    hipStream_t down_stream = nullptr;      // downloads that run under the uploads of the next chunk (rdr_point_delays)
    hipStream_t stream = nullptr;
    int num_cus = 256;
    size_t lds_max = 64u << 10;             // largest dynamic LDS allocation of one workgroup
    size_t total_mem = 0;
    std::string name;
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> evs[4];   // HIP event pairs around every launch, per kernel kind
    size_t ev_used[4] = {0, 0, 0, 0};
    DevBuf slot[NSLOT];
    unsigned long long* d_maxlen = nullptr;   // [MAX_SLICES][MAX_LEVELS]
    int* d_flags = nullptr;                   // [MAX_SLICES]
    int* d_nparts = nullptr;                  // [MAX_LEVELS] (rdr_ray_march's host-given partition)
    int* d_nslow = nullptr;                   // [1] rays sent to the generic kernels by the last pass 1
    int* d_tilectr = nullptr;                 // [4][8] per-XCD tile counters of the four ray-kernel launches of a step
    // value buffers of destroyed cubes, kept for the next cube of the same size: a job that blends / builds a cube per date or per call
    // (cli/raider.py:817-819, the intermediate cubes of the point branch) then allocates nothing - hipMalloc + hipFree of a 400 MB cube
    // are 0.2 ms of host time AND a device-wide synchronisation.  A buffer is handed on with the event recorded when its cube died.
    struct PoolEntry { void* p; size_t bytes; std::vector<hipEvent_t> evs; };   // one event per stream the context has launched on
    std::vector<PoolEntry> cube_pool;
    std::vector<hipStream_t> ext_streams;     // caller streams handed to rdr_set_stream so far (a cube may still be read on any of them)
    bool ext_overflow = false;                // more than 8 of them: destroyed cubes are freed synchronously instead of pooled
    size_t cube_pool_bytes = 0, cube_pool_limit = (size_t)4 << 30;
    DevBuf ws;                                // pass 1 -> pass 2 workspace (field-major ray records, 232 B per ray)
    DevBuf side;                              // level crossings of the generic rays (compact columns of K+1 doubles)
    int64_t side_cap = 0;                     // columns of `side` in the current layout
    int* d_sidectr = nullptr;                 // [1] next free column
    int* h_word = nullptr;                    // [16] page-locked host words: flags read back WITHOUT stalling the host before the final sync
    // Small host inputs (axes, level lists, a cube's three axes) go up through a ring of page-locked buffers: the caller's array is consumed by a
    // memcpy before the call returns and the device copy is REALLY asynchronous (from pageable memory hipMemcpyAsync first waits for everything
    // queued on the stream - ADVICE r5).  A slot is reused only after the event recorded behind its last copy.
    struct PinSlot { void* p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool used = false; };
    static constexpr int PIN_SLOTS = 16;
    static constexpr size_t PIN_MAX = (size_t)1 << 20;
    PinSlot pin[PIN_SLOTS];
    int pin_next = 0;
    hipEvent_t staged_ev = nullptr;           // recorded behind a LARGE host input's copy and waited for: the source is consumed when the entry returns
    // cubes made from DEVICE sources are created without a host synchronisation (round 5): their NaN verdict lands in one of these page-locked
    // words and is read when somebody asks (rdr_cube_has_nan), after the cube's ready event; nan_owner[i] = the cube word i is pending for
    static constexpr int NAN_SLOTS = 256;
    int* nan_words = nullptr;
    const struct rdr_cube* nan_owner[NAN_SLOTS] = {};
    int nan_next = 0;
    int64_t side_forced = -1;                 // rdr_set_side_capacity
    int64_t last_nslow = 0;                   // generic rays seen by the last pass 1 whose count the host happened to read back
    int wall_khz = 0;                         // rate of the wall counter (rdr_clock_sample_begin)
    int last_nan_output = -1;                 // rdr_build_cube (host arrays): 1 / 0 = its last result holds / does not hold a NaN; -1 unknown
    size_t ws_limit = (size_t)48 << 30;       // cap on that workspace; bigger batches are marched in chunks
    // which ray batch the stored records belong to (a later rdr_ray_march reuses them only for the identical batch)
    struct { const void* cube = nullptr; const void* vals = nullptr; LccParams proj = {0, 0, 0, 0, 0, 0, 0, 0}; int64_t n = -1; double ht = 0, zref = 0; const void* a = nullptr; const void* b = nullptr; const void* c = nullptr; const void* d = nullptr; int K = 0; bool valid = false; } wsig;
    std::string err;
};

struct rdr_cube {
    rdr_ctx* ctx = nullptr;
    int64_t ny = 0, nx = 0, nz = 0;
    int dtype = RDR_F32;
    void* d_vals = nullptr;     // interleaved (wet,hydro), (y,x,z) z fastest
    double* d_axes = nullptr;   // ys | xs | zs ascending
    std::vector<double> ys, xs, zs;
    int uni[3] = {0, 0, 0};
    int exact[2] = {0, 0};
    double inv_d[3] = {0, 0, 0};
    LccParams proj = {0, 0, 0, 0, 0, 0, 0, 0};   // kind 0: the cube axes are lon/lat degrees
    // corner-quad copy for large random point sets (cube_kernels.h): built on demand, owned by the cube
    mutable void* d_quad = nullptr;
    mutable size_t quad_bytes = 0;
    mutable int quad_nblk = 0;
    mutable int big_point_calls = 0;             // rdr_interp3 calls that would have profited
    mutable std::mutex quad_mutex;               // one builder of the quad copy per cube
    mutable bool has_nan = false;                // a NaN among the fields (seen while packing; blends: either source's)
    // asynchronous creation (device sources): `ready_ev` is recorded on `ready_stream` after the last kernel that writes the buffers; any use
    // on another stream waits for it (cube_wait); nan_slot >= 0: the packing kernel's verdict is still in ctx->nan_words[nan_slot]
    mutable hipEvent_t ready_ev = nullptr;
    hipStream_t ready_stream = nullptr;
    mutable int nan_slot = -1;
    size_t alloc_bytes = 0;                      // bytes of the ONE allocation behind d_vals (values | axes); 0: not owned (scratch cube, view)
    // views (rdr_cube_view): `base` != NULL marks a handle that shares base's buffers and corner-quad copy and owns only its projection
    const rdr_cube* base = nullptr;
    mutable int views = 0;                       // live views of THIS cube (guarded by g_view_mutex)
    mutable bool doomed = false;                 // destroyed while views were alive: the last view frees the buffers
    mutable std::atomic<bool> foreign{false};    // used from a context other than its own: its buffers are freed synchronously, never pooled
    // caller streams (rdr_set_stream) this cube was read or written on: its buffers are pooled behind events on THESE streams, not on every
    // stream the context ever saw (guarded by g_view_mutex; last_stream: the common case - the same stream again - without the lock)
    mutable std::vector<hipStream_t> used_streams;
    mutable std::atomic<hipStream_t> last_stream{nullptr};
};
// One lock for the state that a cube shares with its OWNING context and that another context's thread may touch (a cube is usable from any
// context): view counts, the pending-NaN-verdict slots (nan_owner) and the pool of buffers of destroyed cubes.  Recursive: settling a verdict
// happens inside a creation.
static std::recursive_mutex g_view_mutex;
static inline const rdr_cube* root(const rdr_cube* q) { return q->base ? q->base : q; }
// every entry point that reads a cube says so: work enqueued by ANOTHER context is not covered by the events the owner records at destroy
static inline void note_use(const rdr_ctx* c, const rdr_cube* q) {
    if (!c || !q) return;
    const rdr_cube* r = root(q);
    if (r->ctx != c) r->foreign.store(true, std::memory_order_relaxed);
    if (r->last_stream.load(std::memory_order_relaxed) != c->stream) {
        std::lock_guard<std::recursive_mutex> guard(g_view_mutex);
        if (std::find(r->used_streams.begin(), r->used_streams.end(), c->stream) == r->used_streams.end()) r->used_streams.push_back(c->stream);
        r->last_stream.store(c->stream, std::memory_order_relaxed);
    }
    // a cube created asynchronously on another stream: this stream's work is ordered after its completion
    if (r->ready_ev && r->ready_stream != c->stream) { (void)hipStreamWaitEvent(c->stream, r->ready_ev, 0); (void)hipGetLastError(); }
}
// the pending NaN verdict of an asynchronously created cube (waits for the cube's ready event: the only host wait of such a cube)
static void cube_resolve_nan(const rdr_cube* r) {
    std::lock_guard<std::recursive_mutex> guard(g_view_mutex);         // (a cube may be asked from two contexts' threads at once)
    if (r->nan_slot < 0) return;
    rdr_ctx* c = r->ctx;
    if (r->ready_ev) (void)hipEventSynchronize(r->ready_ev);
    r->has_nan = c->nan_words[r->nan_slot] != 0;
    c->nan_owner[r->nan_slot] = nullptr;
    r->nan_slot = -1;
}


static int fail(rdr_ctx* ctx, int code, const std::string& msg) {
    g_err = msg;
    if (ctx) ctx->err = msg;
    return code;
}

#define HIPCHECK(ctx, expr)                                                                         \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) {                                                                     \
            (void)hipGetLastError();   /* reported here: a later launch check must not find it again */ \
            return fail(ctx, _e == hipErrorOutOfMemory ? RDR_ERR_OOM : RDR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
        }                                                                                           \
    } while (0)

// mark `q` complete-on-stream: everything enqueued on c->stream so far finishes before any other stream touches it
static int cube_mark_ready(rdr_ctx* c, rdr_cube* q) {
    HIPCHECK(c, hipEventCreateWithFlags(&q->ready_ev, hipEventDisableTiming));
    HIPCHECK(c, hipEventRecord(q->ready_ev, c->stream));
    q->ready_stream = c->stream;
    return RDR_OK;
}

static void pool_drain(rdr_ctx* c);
// hipMalloc that, when the device is out of memory, first gives back what the context itself is hoarding - the pooled buffers of destroyed
// cubes (up to 4 GiB) - and tries once more: a job that would fit must not fail on the library's own cache.
static int dev_malloc(rdr_ctx* ctx, void** p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory && !ctx->cube_pool.empty()) {
        (void)hipGetLastError();
        pool_drain(ctx);
        e = hipMalloc(p, bytes);
    }
    if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); *p = nullptr; return fail(ctx, RDR_ERR_OOM, "out of device memory (" + std::to_string(bytes >> 20) + " MiB asked)"); }
    if (e != hipSuccess) { *p = nullptr; return fail(ctx, RDR_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    return RDR_OK;
}

static int ensure(rdr_ctx* ctx, int s, size_t bytes, void** out) {
    DevBuf& b = ctx->slot[s];
    if (bytes > ((size_t)1 << 46)) return fail(ctx, RDR_ERR_INVALID, "a size argument is negative or beyond 64 TiB");
    if (b.cap < bytes) {
        if (b.p) { HIPCHECK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHECK(ctx, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
        size_t cap = std::max(bytes, (size_t)1 << 16);
        int rc = dev_malloc(ctx, &b.p, cap); if (rc) return rc;
        b.cap = cap;
    }
    *out = b.p;
    return RDR_OK;
}

// Host bytes -> device, stream-ordered, with the source CONSUMED when this returns whatever memory it lives in (pageable or page-locked):
// up to PIN_MAX through the ring of page-locked buffers (the host never waits for earlier work on the stream); larger ones copied from where
// they are and waited for (one event behind the copy - which is what a pageable copy costs anyway).
// (several host pieces that land back to back on the device go up as ONE copy: every copy is a packet of its own on the stream - four of
// them per step showed as +20 us on the 0.13 ms step of BASELINE configs[1])
struct HostPart { const void* p; size_t bytes; };
static int upload(rdr_ctx* ctx, void* dst, const void* src, size_t bytes);
static int upload_parts(rdr_ctx* ctx, void* dst, const HostPart* parts, int nparts) {
    size_t bytes = 0;
    for (int i = 0; i < nparts; ++i) bytes += parts[i].bytes;
    if (bytes == 0) return RDR_OK;
    if (bytes > rdr_ctx::PIN_MAX) {               // too large for the ring: piece by piece
        size_t o = 0;
        for (int i = 0; i < nparts; ++i) { const int rc = upload(ctx, static_cast<char*>(dst) + o, parts[i].p, parts[i].bytes); if (rc) return rc; o += parts[i].bytes; }
        return RDR_OK;
    }
    std::vector<char> tmp;                        // (small: at most PIN_MAX; the ring path of upload() consumes it before returning)
    tmp.resize(bytes);
    size_t o = 0;
    for (int i = 0; i < nparts; ++i) { std::memcpy(tmp.data() + o, parts[i].p, parts[i].bytes); o += parts[i].bytes; }
    return upload(ctx, dst, tmp.data(), bytes);
}

static int upload(rdr_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return RDR_OK;
    if (bytes <= rdr_ctx::PIN_MAX) {
        rdr_ctx::PinSlot& ps = ctx->pin[ctx->pin_next];
        ctx->pin_next = (ctx->pin_next + 1) % rdr_ctx::PIN_SLOTS;
        bool ok = true;
        if (ps.used) { ok = hipEventSynchronize(ps.ev) == hipSuccess; ps.used = false; }
        if (ok && ps.cap < bytes) {
            if (ps.p) { (void)hipHostFree(ps.p); ps.p = nullptr; ps.cap = 0; }
            const size_t cap = std::max(bytes, (size_t)1 << 14);
            ok = hipHostMalloc(&ps.p, cap, hipHostMallocDefault) == hipSuccess;
            if (ok) ps.cap = cap; else ps.p = nullptr;
        }
        if (ok && !ps.ev) ok = hipEventCreateWithFlags(&ps.ev, hipEventDisableTiming) == hipSuccess;
        if (ok) {
            std::memcpy(ps.p, src, bytes);
            HIPCHECK(ctx, hipMemcpyAsync(dst, ps.p, bytes, hipMemcpyHostToDevice, ctx->stream));
            HIPCHECK(ctx, hipEventRecord(ps.ev, ctx->stream));
            ps.used = true;
            return RDR_OK;
        }
        (void)hipGetLastError();                  // (no page-locked memory to be had: the plain copy below)
    }
    HIPCHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (!ctx->staged_ev) HIPCHECK(ctx, hipEventCreateWithFlags(&ctx->staged_ev, hipEventDisableTiming));
    HIPCHECK(ctx, hipEventRecord(ctx->staged_ev, ctx->stream));
    HIPCHECK(ctx, hipEventSynchronize(ctx->staged_ev));
    return RDR_OK;
}

// input staging: host -> scratch slot (the host buffer is consumed when this returns), device -> passthrough
static int stage_in(rdr_ctx* ctx, int s, const void* src, size_t bytes, int loc, const void** dev) {
    if (!src) { *dev = nullptr; return RDR_OK; }
    if (loc == RDR_DEVICE) { *dev = src; return RDR_OK; }
    void* d;
    int rc = ensure(ctx, s, bytes, &d);
    if (rc) return rc;
    rc = upload(ctx, d, src, bytes);
    if (rc) return rc;
    *dev = d;
    return RDR_OK;
}

static int stage_out(rdr_ctx* ctx, int s, void* dst, size_t bytes, int loc, void** dev) {
    if (loc == RDR_DEVICE) { *dev = dst; return RDR_OK; }
    return ensure(ctx, s, bytes, dev);
}

static int finish_out(rdr_ctx* ctx, void* dst, const void* dev, size_t bytes, int loc) {
    if (loc == RDR_DEVICE || !dst) return RDR_OK;
    HIPCHECK(ctx, hipMemcpyAsync(dst, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return RDR_OK;
}

struct KTimer {
    rdr_ctx* c; int which; hipEvent_t stop = nullptr;
    KTimer(rdr_ctx* ctx, int w) : c(ctx), which(w) {
        if (!c->profiling) return;
        auto& v = c->evs[which];
        if (c->ev_used[which] == v.size()) {
            hipEvent_t a = nullptr, b = nullptr;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            v.emplace_back(a, b);
        }
        auto& pr = v[c->ev_used[which]++];
        (void)hipEventRecord(pr.first, c->stream);
        stop = pr.second;
    }
    ~KTimer() { if (stop) (void)hipEventRecord(stop, c->stream); }
};

static inline int grid_for(int64_t n, int block, int max_blocks) {
    int64_t g = (n + block - 1) / block;
    return (int)std::max<int64_t>(1, std::min<int64_t>(g, max_blocks));
}


// Launch with `sm` bytes of dynamic LDS; allocations beyond the 64 KB default need the kernel's limit raised first (long
// non-uniform axes: their (node, 1/spacing) tables live in LDS, 16 B per node).
template <typename... KArgs, typename... Args>
static hipError_t launch_lds(void (*kern)(KArgs...), dim3 g, dim3 b, size_t sm, hipStream_t s, Args&&... args) {
    if (sm > (size_t)(48u << 10)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, g, b, sm, s, std::forward<Args>(args)...);
    return hipGetLastError();
}

// ---- kernels other than the two ray passes (raider_kernels.h) ---------------------------------------------------------
#include "cube_kernels.h"
#include "producer_kernels.h"
#include "orbit_kernels.h"
#include "native_kernels.h"
#include "proj_kernels.h"

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
// (definitions below get C linkage from their extern "C" declarations in include/raider_hip.h)

int rdr_version(void) { return 101; }

// sha256[:16] over the sources of this translation unit, handed in by the build recipe (-DRDR_SOURCE_HASH=...; __graft_entry__.build):
// lets the loader prove that the binary it opened was compiled from the tree it sits in (mtimes do not survive a copy).
#ifndef RDR_SOURCE_HASH
#define RDR_SOURCE_HASH "unknown"
#endif
static const char k_source_hash[] = "rdr-source-hash:" RDR_SOURCE_HASH;      // (the marker lets the build recipe read it without dlopen)
const char* rdr_source_hash(void) { return k_source_hash + 16; }

const char* rdr_last_error(rdr_ctx* ctx) { return (ctx && !ctx->err.empty()) ? ctx->err.c_str() : g_err.c_str(); }

int rdr_create(int device, rdr_ctx** out) {
    if (!out) return fail(nullptr, RDR_ERR_INVALID, "rdr_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, RDR_ERR_NODEVICE, std::string("rdr_create: no HIP device available (") +
                    (e != hipSuccess ? hipGetErrorString(e) : "device count 0") + "); raider_amd has no CPU fallback");
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    if (device >= count) return fail(nullptr, RDR_ERR_INVALID, "rdr_create: device index out of range");
    rdr_ctx* c = new rdr_ctx();
    c->device = device;
    const int rc = [&]() -> int {
        HIPCHECK(nullptr, hipSetDevice(device));
        hipDeviceProp_t prop;
        HIPCHECK(nullptr, hipGetDeviceProperties(&prop, device));
        c->num_cus = prop.multiProcessorCount;
        c->total_mem = prop.totalGlobalMem;
        c->lds_max = std::max<size_t>(prop.sharedMemPerBlock, prop.maxSharedMemoryPerMultiProcessor);
        c->name = prop.name;
        if (c->name.empty()) c->name = std::string("AMD ") + prop.gcnArchName;   // amdgpu.ids may be absent on the box
        HIPCHECK(nullptr, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
        HIPCHECK(nullptr, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        HIPCHECK(nullptr, hipStreamCreateWithFlags(&c->down_stream, hipStreamNonBlocking));
        c->stream = c->own_stream;
        HIPCHECK(nullptr, hipMalloc((void**)&c->d_maxlen, (size_t)MAX_SLICES * MAX_LEVELS * sizeof(unsigned long long)));
        HIPCHECK(nullptr, hipMalloc((void**)&c->d_flags, (MAX_SLICES + 4) * sizeof(int)));      // (+ the cube packer's NaN word)
        HIPCHECK(nullptr, hipMalloc((void**)&c->d_nparts, MAX_LEVELS * sizeof(int)));
        HIPCHECK(nullptr, hipMalloc((void**)&c->d_nslow, sizeof(int)));
        HIPCHECK(nullptr, hipMalloc((void**)&c->d_tilectr, 32 * sizeof(int)));
        HIPCHECK(nullptr, hipMalloc((void**)&c->d_sidectr, sizeof(int)));
        HIPCHECK(nullptr, hipMemset(c->d_sidectr, 0, sizeof(int)));
        HIPCHECK(nullptr, hipHostMalloc((void**)&c->h_word, 16 * sizeof(int), hipHostMallocDefault));
        HIPCHECK(nullptr, hipHostMalloc((void**)&c->nan_words, rdr_ctx::NAN_SLOTS * sizeof(int), hipHostMallocDefault));
        HIPCHECK(nullptr, hipMemset(c->d_nslow, 0, sizeof(int)));
        return RDR_OK;
    }();
    if (rc) { rdr_destroy(c); return rc; }
    if (const char* e = std::getenv("RAIDER_HIP_WORKSPACE_BYTES")) c->ws_limit = (size_t)std::strtoull(e, nullptr, 10);
    if (const char* e = std::getenv("RAIDER_HIP_CUBE_POOL_BYTES")) c->cube_pool_limit = (size_t)std::strtoull(e, nullptr, 10);
    *out = c;
    return RDR_OK;
}

void rdr_destroy(rdr_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& b : c->slot) if (b.p) (void)hipFree(b.p);
    for (auto& e : c->cube_pool) { for (auto ev : e.evs) (void)hipEventDestroy(ev); (void)hipFree(e.p); }
    if (c->d_maxlen) (void)hipFree(c->d_maxlen);
    if (c->d_flags) (void)hipFree(c->d_flags);
    if (c->d_nparts) (void)hipFree(c->d_nparts);
    if (c->d_nslow) (void)hipFree(c->d_nslow);
    if (c->d_tilectr) (void)hipFree(c->d_tilectr);
    if (c->ws.p) (void)hipFree(c->ws.p);
    if (c->side.p) (void)hipFree(c->side.p);
    if (c->d_sidectr) (void)hipFree(c->d_sidectr);
    if (c->h_word) (void)hipHostFree(c->h_word);
    if (c->nan_words) (void)hipHostFree(c->nan_words);
    for (auto& ps : c->pin) { if (ps.ev) (void)hipEventDestroy(ps.ev); if (ps.p) (void)hipHostFree(ps.p); }
    if (c->staged_ev) (void)hipEventDestroy(c->staged_ev);
    for (auto& v : c->evs) for (auto& pr : v) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->down_stream) (void)hipStreamDestroy(c->down_stream);
    delete c;
}

int rdr_set_stream(rdr_ctx* c, void* s) {
    if (!c) return fail(nullptr, RDR_ERR_INVALID, "ctx is NULL");
    // NULL is HIP's legacy default stream (what torch.cuda.current_stream().cuda_stream is unless the caller
    // switched streams); (void*)-1 restores the ctx's private stream
    const hipStream_t before = c->stream;
    const hipStream_t want = (s == (void*)(intptr_t)-1) ? c->own_stream : (hipStream_t)s;
    if (want != before) {
        // the context's scratch slots, flag words and asynchronously created cubes are shared by whatever stream it launches on: work on the
        // new stream is ordered after everything enqueued on the old one (one event; calls no longer end with a host synchronisation)
        hipEvent_t ev = nullptr;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
            if (hipEventRecord(ev, before) == hipSuccess) (void)hipStreamWaitEvent(want, ev, 0);
            (void)hipEventDestroy(ev);
        }
        (void)hipGetLastError();
    }
    if (s == (void*)(intptr_t)-1) c->stream = c->own_stream;
    else {
        c->stream = (hipStream_t)s;
        std::lock_guard<std::recursive_mutex> guard(g_view_mutex);
        if (std::find(c->ext_streams.begin(), c->ext_streams.end(), c->stream) == c->ext_streams.end()) {
            if (c->ext_streams.size() < 8) c->ext_streams.push_back(c->stream); else c->ext_overflow = true;
        }
    }
    return RDR_OK;
}

int rdr_forget_stream(rdr_ctx* c, void* s) {
    if (!c) return fail(nullptr, RDR_ERR_INVALID, "ctx is NULL");
    const hipStream_t st = (hipStream_t)s;
    if (c->stream == st && st != c->own_stream) {           // still the current one: finish its work, go back to the private stream
        HIPCHECK(c, hipStreamSynchronize(st));
        c->stream = c->own_stream;
    }
    std::lock_guard<std::recursive_mutex> guard(g_view_mutex);
    c->ext_streams.erase(std::remove(c->ext_streams.begin(), c->ext_streams.end(), st), c->ext_streams.end());
    if (c->ext_streams.size() < 8) c->ext_overflow = false;
    return RDR_OK;
}

int rdr_synchronize(rdr_ctx* c) {
    if (!c) return fail(nullptr, RDR_ERR_INVALID, "ctx is NULL");
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return RDR_OK;
}

int rdr_device_info(rdr_ctx* c, char* name, int name_len, int* cus, int64_t* mem) {
    if (!c) return fail(nullptr, RDR_ERR_INVALID, "ctx is NULL");
    if (name && name_len > 0) { std::strncpy(name, c->name.c_str(), name_len - 1); name[name_len - 1] = 0; }
    if (cus) *cus = c->num_cus;
    if (mem) *mem = (int64_t)c->total_mem;
    return RDR_OK;
}

// Page-locked host memory for result arrays: a device -> host copy into it runs at the link rate without first-touch page faults
// (measured on the MI355X box: 512 MB into fresh pageable pages 31 ms, into resident or pinned pages 9.5 ms = 54 GB/s) and truly
// asynchronously, so the download of finished height slices overlaps the kernels of the next ones (rdr_raytrace_slices).
int rdr_host_alloc(int64_t bytes, void** out) {
    if (!out || bytes <= 0) return fail(nullptr, RDR_ERR_INVALID, "rdr_host_alloc: bad argument");
    void* p = nullptr;
    const hipError_t e = hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault);
    if (e != hipSuccess) return fail(nullptr, RDR_ERR_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    *out = p;
    return RDR_OK;
}

int rdr_host_free(void* p) {
    if (!p) return RDR_OK;
    const hipError_t e = hipHostFree(p);
    if (e != hipSuccess) return fail(nullptr, RDR_ERR_HIP, std::string("hipHostFree: ") + hipGetErrorString(e));
    return RDR_OK;
}

int rdr_set_workspace_limit(rdr_ctx* c, int64_t bytes) {
    if (!c || bytes < (int64_t)1 << 20) return fail(c, RDR_ERR_INVALID, "rdr_set_workspace_limit: need at least 1 MiB");
    c->ws_limit = (size_t)bytes;
    c->wsig.valid = false;
    return RDR_OK;
}

// every pooled buffer back to the device (waits for the events they were retired with)
static void pool_drain(rdr_ctx* c) {
    std::lock_guard<std::recursive_mutex> guard(g_view_mutex);
    while (!c->cube_pool.empty()) {
        rdr_ctx::PoolEntry e = c->cube_pool.front();
        c->cube_pool.erase(c->cube_pool.begin());
        c->cube_pool_bytes -= e.bytes;
        for (auto ev : e.evs) { (void)hipEventSynchronize(ev); (void)hipEventDestroy(ev); }
        (void)hipFree(e.p);
    }
    (void)hipGetLastError();
}

int rdr_trim(rdr_ctx* c, int64_t keep_bytes, int64_t* released) {
    if (!c || keep_bytes < 0) return fail(c, RDR_ERR_INVALID, "rdr_trim: NULL context or a negative size");
    HIPCHECK(c, hipSetDevice(c->device));
    HIPCHECK(c, hipStreamSynchronize(c->down_stream));
    HIPCHECK(c, hipStreamSynchronize(c->copy_stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    if (c->stream != c->own_stream) HIPCHECK(c, hipStreamSynchronize(c->own_stream));
    const size_t keep = (size_t)keep_bytes;
    size_t freed = 0;
    auto drop = [&](DevBuf& b) { if (b.p && b.cap > keep) { if (hipFree(b.p) == hipSuccess) freed += b.cap; b.p = nullptr; b.cap = 0; } };
    for (auto& b : c->slot) drop(b);
    if ((c->ws.p && c->ws.cap > keep) || (c->side.p && c->side.cap > keep)) c->wsig.valid = false;      // (the stored ray records go with them)
    drop(c->ws);
    if (c->side.p && c->side.cap > keep) { drop(c->side); c->side_cap = 0; }
    std::lock_guard<std::recursive_mutex> guard(g_view_mutex);
    while (!c->cube_pool.empty() && c->cube_pool_bytes > keep) {          // oldest first
        rdr_ctx::PoolEntry e = c->cube_pool.front();
        c->cube_pool.erase(c->cube_pool.begin());
        c->cube_pool_bytes -= e.bytes;
        for (auto ev : e.evs) { (void)hipEventSynchronize(ev); (void)hipEventDestroy(ev); }
        if (hipFree(e.p) == hipSuccess) freed += e.bytes;
    }
    (void)hipGetLastError();
    if (released) *released = (int64_t)freed;
    return RDR_OK;
}

int rdr_set_side_capacity(rdr_ctx* c, int64_t columns) {
    if (!c) return fail(nullptr, RDR_ERR_INVALID, "ctx is NULL");
    c->side_forced = columns < 0 ? -1 : columns;
    c->wsig.valid = false;
    return RDR_OK;
}

int64_t rdr_generic_ray_count(rdr_ctx* c) { return c ? c->last_nslow : -1; }

int rdr_set_profiling(rdr_ctx* c, int on) {
    if (!c) return fail(nullptr, RDR_ERR_INVALID, "ctx is NULL");
    c->profiling = on != 0;
    for (auto& u : c->ev_used) u = 0;
    return RDR_OK;
}

// Shader clock while other kernels run: ONE wave on the copy stream sleeps for `ms` of wall time and reads the shader-clock counter
// (s_memtime) and the 100 MHz wall counter (s_memrealtime) before and after - cycles per wall tick = the clock the chip actually ran
// at during those milliseconds (under fp64-dense kernels it sits near 2.0 GHz, not at the 2.4 GHz of the data sheet).
__global__ void clock_sample_kernel(long long wall_ticks, long long* out) {
    if (threadIdx.x != 0) return;
    const long long c0 = clock64(), w0 = wall_clock64();
    while (wall_clock64() - w0 < wall_ticks) __builtin_amdgcn_s_sleep(127);
    const long long c1 = clock64(), w1 = wall_clock64();
    out[0] = c1 - c0; out[1] = w1 - w0;
}

int rdr_clock_sample_begin(rdr_ctx* c, double ms) {
    if (!c || !(ms > 0) || ms > 60000.0) return fail(c, RDR_ERR_INVALID, "rdr_clock_sample_begin: NULL context or a duration outside (0, 60000] ms");
    HIPCHECK(c, hipSetDevice(c->device));
    int khz = 0;
    HIPCHECK(c, hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device));
    if (khz <= 0) return fail(c, RDR_ERR_HIP, "rdr_clock_sample_begin: the device reports no wall-clock rate");
    long long* out = reinterpret_cast<long long*>(c->h_word + 4);      // (page-locked and device-visible: the wave writes it directly)
    out[0] = 0; out[1] = 0;
    c->wall_khz = khz;
    hipLaunchKernelGGL(clock_sample_kernel, dim3(1), dim3(64), 0, c->copy_stream, (long long)(ms * khz), out);
    HIPCHECK(c, hipGetLastError());
    return RDR_OK;
}

int rdr_clock_sample_end(rdr_ctx* c, double* ghz) {
    if (!c || !ghz) return fail(c, RDR_ERR_INVALID, "rdr_clock_sample_end: NULL argument");
    HIPCHECK(c, hipSetDevice(c->device));
    HIPCHECK(c, hipStreamSynchronize(c->copy_stream));
    const long long* out = reinterpret_cast<const long long*>(c->h_word + 4);
    if (out[1] <= 0 || c->wall_khz <= 0) return fail(c, RDR_ERR_INVALID, "rdr_clock_sample_end: no sample was started on this context");
    *ghz = (double)out[0] / ((double)out[1] / (c->wall_khz * 1e3)) / 1e9;
    return RDR_OK;
}

int rdr_profile_get(rdr_ctx* c, int which, int* count, float* total_ms) {
    if (!c || which < 0 || which > 3 || !count || !total_ms) return fail(c, RDR_ERR_INVALID, "rdr_profile_get: bad argument");
    float tot = 0;
    for (size_t i = 0; i < c->ev_used[which]; ++i) {
        float ms = 0;
        HIPCHECK(c, hipEventSynchronize(c->evs[which][i].second));
        HIPCHECK(c, hipEventElapsedTime(&ms, c->evs[which][i].first, c->evs[which][i].second));
        tot += ms;
    }
    *count = (int)c->ev_used[which];
    *total_ms = tot;
    return RDR_OK;
}

// ---- cube ---------------------------------------------------------------------------------------
static int axis_check(const double* g, int64_t n, int* flip) {
    if (n < 2) return -1;
    bool asc = true, desc = true;
    for (int64_t i = 1; i < n; ++i) {
        if (!(g[i] > g[i - 1])) asc = false;
        if (!(g[i] < g[i - 1])) desc = false;
    }
    if (!asc && !desc) return -1;
    *flip = desc ? 1 : 0;
    return 0;
}

static void axis_uniformity(const std::vector<double>& g, int* uni, double* inv_d, int* exact = nullptr) {
    const int64_t n = (int64_t)g.size();
    const double span = g[n - 1] - g[0];
    *inv_d = (double)(n - 1) / span;
    double worst = 0;
    for (int64_t i = 0; i < n; ++i) worst = std::max(worst, std::fabs((g[i] - g[0]) * (*inv_d) - (double)i));
    *uni = worst < 0.25 ? 1 : 0;   // guess lands within +-1 cell; fix-up loops make it exact
    if (exact) *exact = worst < 1e-11 ? 1 : 0;   // uniform to round-off: cell index and weight from arithmetic alone (cell_xy)
}

template <typename T2>
static CubeView<T2> make_view(const rdr_cube* q) {
    CubeView<T2> v;
    v.v = (const T2*)q->d_vals;
    v.axes = q->d_axes;
    v.ny = (int)q->ny; v.nx = (int)q->nx; v.nz = (int)q->nz;
    v.y_lo = q->ys.front(); v.y_hi = q->ys.back();
    v.x_lo = q->xs.front(); v.x_hi = q->xs.back();
    v.z_lo = q->zs.front(); v.z_hi = q->zs.back();
    v.inv_dy = q->inv_d[0]; v.inv_dx = q->inv_d[1]; v.inv_dz = q->inv_d[2];
    v.uni_y = q->uni[0]; v.uni_x = q->uni[1]; v.uni_z = q->uni[2];
    v.exact_y = q->exact[0]; v.exact_x = q->exact[1];
    v.small = (q->ny * q->nx < (1 << 24)) && (q->nz < (1 << 24)) &&
              ((uint64_t)q->ny * q->nx * q->nz * sizeof(T2) < (1ULL << 32));
    return v;
}

// the zenith / station kernels stage the axes in LDS when they fit 48 KB, else read them from global memory
static bool axes_fit_lds(const rdr_cube* q) { return (size_t)(q->ny + q->nx + q->nz) * sizeof(double) <= (48u << 10); }
static size_t axes_smem(const rdr_cube* q) { return axes_fit_lds(q) ? (size_t)(q->ny + q->nx + q->nz) * sizeof(double) : 0; }

// The three axes of a cube from DEVICE copies of the arrays they were given as (any of them possibly descending): one tiny launch instead of
// one more host copy on the stream (rdr_build_cube_to_cube: the AOI axes went up for the build already).
struct DevAxes { const double* y = nullptr; const double* x = nullptr; const double* z = nullptr; int fy = 0, fx = 0, fz = 0; };
__global__ void axes_from_device_kernel(DevAxes a, int ny, int nx, int nz, double* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ny + nx + nz; i += gridDim.x * blockDim.x) {
        double v;
        if (i < ny) v = a.y[a.fy ? ny - 1 - i : i];
        else if (i < ny + nx) { const int j = i - ny; v = a.x[a.fx ? nx - 1 - j : j]; }
        else { const int j = i - ny - nx; v = a.z[a.fz ? nz - 1 - j : j]; }
        out[i] = v;
    }
}

static int cube_alloc(rdr_ctx* c, rdr_cube* q, const DevAxes* dev_axes = nullptr) {
    const size_t esz = q->dtype == RDR_F32 ? 8 : 16;
    const size_t total = (size_t)q->ny * q->nx * q->nz;
    // ONE allocation: the values, then (256 B aligned) the three axes
    const size_t vbytes = (total * esz + 255) / 256 * 256;
    const size_t bytes = vbytes + (size_t)(q->ny + q->nx + q->nz) * sizeof(double);
    q->alloc_bytes = bytes;
    std::unique_lock<std::recursive_mutex> guard(g_view_mutex);
    for (size_t i = 0; i < c->cube_pool.size(); ++i) {
        if (c->cube_pool[i].bytes != bytes) continue;
        const rdr_ctx::PoolEntry e = c->cube_pool[i];
        c->cube_pool.erase(c->cube_pool.begin() + (long)i);
        c->cube_pool_bytes -= bytes;
        // (its previous cube's last work on every stream this context launches on was enqueued before these events: whatever stream
        // builds the new cube waits for them)
        hipError_t w = hipSuccess;
        for (auto ev : e.evs) { if (w == hipSuccess) w = hipStreamWaitEvent(c->stream, ev, 0); (void)hipEventDestroy(ev); }
        if (w != hipSuccess) { (void)hipDeviceSynchronize(); (void)hipFree(e.p); (void)hipGetLastError(); break; }
        q->d_vals = e.p;
        break;
    }
    if (!q->d_vals) { const int rc = dev_malloc(c, &q->d_vals, bytes); if (rc) { q->alloc_bytes = 0; return rc; } }     // (out of memory: the pool is drained, then once more)
    guard.unlock();
    q->d_axes = reinterpret_cast<double*>(static_cast<char*>(q->d_vals) + vbytes);
    std::vector<double> ax;
    ax.insert(ax.end(), q->ys.begin(), q->ys.end());
    ax.insert(ax.end(), q->xs.begin(), q->xs.end());
    ax.insert(ax.end(), q->zs.begin(), q->zs.end());
    // (stream-ordered: a pooled buffer may still be read by work enqueued before the event above; through the page-locked ring: the host
    // does not wait for what the stream is still doing - creation from device sources really is asynchronous)
    if (dev_axes && dev_axes->y) {
        hipLaunchKernelGGL(axes_from_device_kernel, dim3((unsigned)((ax.size() + 255) / 256)), dim3(256), 0, c->stream, *dev_axes, (int)q->ny, (int)q->nx, (int)q->nz, q->d_axes);
        HIPCHECK(c, hipGetLastError());
    } else { const int rc = upload(c, q->d_axes, ax.data(), ax.size() * sizeof(double)); if (rc) return rc; }
    axis_uniformity(q->ys, &q->uni[0], &q->inv_d[0], &q->exact[0]);
    axis_uniformity(q->xs, &q->uni[1], &q->inv_d[1], &q->exact[1]);
    axis_uniformity(q->zs, &q->uni[2], &q->inv_d[2]);
    return RDR_OK;
}

static int cube_create_impl(rdr_ctx* c, const double* ys, int64_t ny, const double* xs, int64_t nx, const double* zs, int64_t nz,
                            const void* wet, const void* hydro, int dtype, int64_t sy, int64_t sx, int64_t sz, int loc,
                            rdr_cube** out, const double* const* dev_yxz);

int rdr_cube_create(rdr_ctx* c, const double* ys, int64_t ny, const double* xs, int64_t nx, const double* zs, int64_t nz,
                    const void* wet, const void* hydro, int dtype, int64_t sy, int64_t sx, int64_t sz, int loc,
                    rdr_cube** out) {
    return cube_create_impl(c, ys, ny, xs, nx, zs, nz, wet, hydro, dtype, sy, sx, sz, loc, out, nullptr);
}

// dev_yxz: device copies of the three axis arrays AS GIVEN (before any flip), or NULL: the axes go up from the host
static int cube_create_impl(rdr_ctx* c, const double* ys, int64_t ny, const double* xs, int64_t nx, const double* zs, int64_t nz,
                            const void* wet, const void* hydro, int dtype, int64_t sy, int64_t sx, int64_t sz, int loc,
                            rdr_cube** out, const double* const* dev_yxz) {
    if (!c || !out || !ys || !xs || !zs || !wet || !hydro) return fail(c, RDR_ERR_INVALID, "rdr_cube_create: NULL argument");
    const bool swapped = (dtype & RDR_BYTESWAPPED) != 0;       // the source fields are in the other byte order (NetCDF-3: big-endian)
    dtype &= ~RDR_BYTESWAPPED;
    if (dtype != RDR_F32 && dtype != RDR_F64) return fail(c, RDR_ERR_INVALID, "rdr_cube_create: dtype must be RDR_F32 or RDR_F64 (optionally | RDR_BYTESWAPPED)");
    if (nz > MAX_LEVELS) return fail(c, RDR_ERR_INVALID, "rdr_cube_create: more than 512 z levels");
    if (ny + nx + nz > 100000) return fail(c, RDR_ERR_INVALID, "rdr_cube_create: axes too long");
    int fy, fx, fz;
    if (axis_check(ys, ny, &fy) || axis_check(xs, nx, &fx) || axis_check(zs, nz, &fz))
        return fail(c, RDR_ERR_INVALID, "The points in each dimension must be strictly ascending or descending (and >= 2)");
    HIPCHECK(c, hipSetDevice(c->device));
    rdr_cube* q = new rdr_cube();
    q->ctx = c; q->ny = ny; q->nx = nx; q->nz = nz; q->dtype = dtype;
    q->ys.assign(ys, ys + ny); q->xs.assign(xs, xs + nx); q->zs.assign(zs, zs + nz);
    if (fy) std::reverse(q->ys.begin(), q->ys.end());
    if (fx) std::reverse(q->xs.begin(), q->xs.end());
    if (fz) std::reverse(q->zs.begin(), q->zs.end());
    DevAxes da;
    if (dev_yxz) { da.y = dev_yxz[0]; da.x = dev_yxz[1]; da.z = dev_yxz[2]; da.fy = fy; da.fx = fx; da.fz = fz; }
    int rc = cube_alloc(c, q, dev_yxz ? &da : nullptr);
    if (rc) { rdr_cube_destroy(q); return rc; }
    const size_t total = (size_t)ny * nx * nz;
    const size_t esz = dtype == RDR_F32 ? 4 : 8;
    // span of the strided source (non-negative strides)
    if (sy < 0 || sx < 0 || sz < 0) { rdr_cube_destroy(q); return fail(c, RDR_ERR_INVALID, "rdr_cube_create: negative strides not supported"); }
    const size_t span = (size_t)((ny - 1) * sy + (nx - 1) * sx + (nz - 1) * sz + 1);
    const void *dw, *dh;
    rc = stage_in(c, SLOT_IN0, wet, span * esz, loc, &dw); if (rc) { rdr_cube_destroy(q); return rc; }
    rc = stage_in(c, SLOT_IN1, hydro, span * esz, loc, &dh); if (rc) { rdr_cube_destroy(q); return rc; }
    const int g = grid_for((int64_t)total, 256, c->num_cus * 8);
    int* const nf = c->d_flags + MAX_SLICES;      // (a word of the flag array no ray batch uses)
    hipError_t e = hipMemsetAsync(nf, 0, sizeof(int), c->stream);
    if (e != hipSuccess) { rdr_cube_destroy(q); return fail(c, RDR_ERR_HIP, hipGetErrorString(e)); }
#define RDR_PACK(T, T2, SW) hipLaunchKernelGGL((pack_cube_kernel<T, T2, SW>), dim3(g), dim3(256), 0, c->stream, (const T*)dw, (const T*)dh, \
                                               (T2*)q->d_vals, ny, nx, nz, sy, sx, sz, fy, fx, fz, nf)
    // x-contiguous sources (file order (z, y, x); the planar results behind an intermediate delay cube): the LDS-transposing kernel
    const bool xfast = sx == 1 && nx >= 8 && nz >= 8;
    const int gx = (int)std::max<int64_t>(1, std::min<int64_t>(ny * ((nx + 31) / 32) * ((nz + 31) / 32), (int64_t)c->num_cus * 16));
#define RDR_PACKX(T, T2, SW) hipLaunchKernelGGL((pack_cube_xfast_kernel<T, T2, SW>), dim3(gx), dim3(256), 0, c->stream, (const T*)dw, (const T*)dh, \
                                                (T2*)q->d_vals, ny, nx, nz, sy, sz, fy, fx, fz, nf)
    {
        KTimer t(c, 2);       // (the packing of an intermediate delay cube is part of the point branch's step: bench.py --workload c2 counts it)
        if (xfast) {
            if (dtype == RDR_F32) { if (swapped) RDR_PACKX(float, float2, true); else RDR_PACKX(float, float2, false); }
            else { if (swapped) RDR_PACKX(double, double2, true); else RDR_PACKX(double, double2, false); }
        } else if (dtype == RDR_F32) { if (swapped) RDR_PACK(float, float2, true); else RDR_PACK(float, float2, false); }
        else { if (swapped) RDR_PACK(double, double2, true); else RDR_PACK(double, double2, false); }
    }
#undef RDR_PACKX
#undef RDR_PACK
    e = hipGetLastError();
    if (e != hipSuccess) { rdr_cube_destroy(q); return fail(c, RDR_ERR_HIP, hipGetErrorString(e)); }
    if (loc == RDR_DEVICE) {
        // DEVICE sources (intermediate delay cubes, cubes made from tensors): no host synchronisation.  The verdict travels into a page-locked
        // word and is read when asked for (rdr_cube_has_nan -> cube_resolve_nan); other streams wait for the ready event (note_use).
        std::unique_lock<std::recursive_mutex> guard(g_view_mutex);                // (the slots are also released by whoever destroys a cube)
        int slot = c->nan_next; c->nan_next = (c->nan_next + 1) % rdr_ctx::NAN_SLOTS;
        if (c->nan_owner[slot]) cube_resolve_nan(c->nan_owner[slot]);              // (a cube 256 creations ago that nobody asked: settle it now)
        c->nan_words[slot] = 0;
        e = hipMemcpyAsync(&c->nan_words[slot], nf, sizeof(int), hipMemcpyDeviceToHost, c->stream);
        if (e != hipSuccess) { guard.unlock(); rdr_cube_destroy(q); return fail(c, RDR_ERR_HIP, hipGetErrorString(e)); }
        rc = cube_mark_ready(c, q);
        if (rc) { guard.unlock(); rdr_cube_destroy(q); return rc; }
        q->nan_slot = slot; c->nan_owner[slot] = q;
        *out = q;
        return RDR_OK;
    }
    // HOST sources: the staging slots are free again and the cube is complete when the call returns
    int has_nan = 0;
    e = hipMemcpyAsync(&has_nan, nf, sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { rdr_cube_destroy(q); return fail(c, RDR_ERR_HIP, hipGetErrorString(e)); }
    q->has_nan = has_nan != 0;
    *out = q;
    return RDR_OK;
}

static int cone_params(rdr_ctx* c, const char* who, int kind, const double* p, int np, LccParams& out);

// The buffers of a dead cube.  Pooled (no synchronisation) with one event per stream the owning context has ever launched on - its own
// three and every caller stream handed to rdr_set_stream (a torch stream adopted earlier may still be reading the cube) - or, when that
// cannot be established (another context used the cube, more caller streams than are tracked, an event that cannot be recorded - e.g.
// on a stream the caller has destroyed), freed after a device-wide synchronisation.
static void cube_release(rdr_cube* q) {
    rdr_ctx* c = q->ctx;
    if (c) (void)hipSetDevice(c->device);
    // (may run on a thread of ANOTHER context - the last view of a destroyed source: the owner's slots and pool are under the shared lock)
    std::lock_guard<std::recursive_mutex> guard(g_view_mutex);
    if (c && q->nan_slot >= 0) { c->nan_owner[q->nan_slot] = nullptr; q->nan_slot = -1; }     // (nobody asked: the word is free again)
    if (q->ready_ev) { (void)hipEventDestroy(q->ready_ev); q->ready_ev = nullptr; }
    bool pooled = false;
    const bool foreign = q->foreign.load(std::memory_order_relaxed);
    if (c && q->d_vals && q->alloc_bytes > 0 && !foreign && !c->ext_overflow && c->cube_pool.size() < 6 &&
        c->cube_pool_bytes + q->alloc_bytes <= c->cube_pool_limit) {
        // the context's own three streams, and of the caller streams only those this cube was used on AND the context still knows
        // (rdr_forget_stream drops a stream the caller is about to destroy: no event is ever recorded on a dead handle)
        std::vector<hipStream_t> st = {c->own_stream, c->copy_stream, c->down_stream};
        for (auto s_ : c->ext_streams)
            if (std::find(q->used_streams.begin(), q->used_streams.end(), s_) != q->used_streams.end()) st.push_back(s_);
        if (std::find(st.begin(), st.end(), c->stream) == st.end()) st.push_back(c->stream);
        std::vector<hipEvent_t> evs;
        bool ok = true;
        for (auto s_ : st) {
            hipEvent_t ev = nullptr;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { ok = false; break; }
            evs.push_back(ev);
            if (hipEventRecord(ev, s_) != hipSuccess) { ok = false; break; }
        }
        if (ok) {
            // (the corner-quad copy is freed, not pooled: wait for the same events first - every stream that may still be reading it)
            if (q->d_quad) for (auto ev : evs) (void)hipEventSynchronize(ev);
            c->cube_pool.push_back({q->d_vals, q->alloc_bytes, evs});
            c->cube_pool_bytes += q->alloc_bytes;
            pooled = true;
        } else { for (auto ev : evs) (void)hipEventDestroy(ev); (void)hipGetLastError(); }
    }
    if (!pooled) (void)hipDeviceSynchronize();
    if (!pooled && q->d_vals && q->alloc_bytes > 0) (void)hipFree(q->d_vals);       // (the axes live in the same allocation)
    if (q->d_quad) (void)hipFree(q->d_quad);
    delete q;
}

void rdr_cube_destroy(rdr_cube* q) {
    if (!q) return;
    const rdr_cube* dead_root = nullptr;
    {
        std::lock_guard<std::recursive_mutex> guard(g_view_mutex);
        if (q->base) {                                  // a view: only the handle goes; the last view of a destroyed source frees the buffers
            const rdr_cube* r = q->base;
            delete q;
            if (--r->views == 0 && r->doomed) dead_root = r;
            q = nullptr;
        } else if (q->views > 0) { q->doomed = true; q = nullptr; }
    }
    if (dead_root) cube_release(const_cast<rdr_cube*>(dead_root));
    if (q) cube_release(q);
}

int rdr_cube_view(rdr_ctx* c, const rdr_cube* src, int kind, const double* p, int np, rdr_cube** out) {
    if (!c || !src || !out) return fail(c, RDR_ERR_INVALID, "rdr_cube_view: NULL argument");
    *out = nullptr;
    LccParams L = src->proj;
    if (kind == RDR_PROJ_LONLAT) L = LccParams{0, 0, 0, 0, 0, 0, 0, 0};
    else if (kind != -1) { const int rc = cone_params(c, "rdr_cube_view", kind, p, np, L); if (rc) return rc; }
    const rdr_cube* r = root(src);
    if (c->device != r->ctx->device) return fail(c, RDR_ERR_INVALID, "rdr_cube_view: the context is on another device than the cube");
    rdr_cube* v = new rdr_cube();
    v->ctx = c; v->ny = r->ny; v->nx = r->nx; v->nz = r->nz; v->dtype = r->dtype;
    v->d_vals = r->d_vals; v->d_axes = r->d_axes; v->ys = r->ys; v->xs = r->xs; v->zs = r->zs;
    for (int i = 0; i < 3; ++i) { v->uni[i] = r->uni[i]; v->inv_d[i] = r->inv_d[i]; }
    v->exact[0] = r->exact[0]; v->exact[1] = r->exact[1];
    v->proj = L; v->alloc_bytes = 0; v->base = r;
    note_use(c, r);
    { std::lock_guard<std::recursive_mutex> guard(g_view_mutex); ++r->views; }
    *out = v;
    return RDR_OK;
}

int rdr_cube_has_nan(const rdr_cube* q) {
    if (!q) return -1;
    const rdr_cube* r = root(q);                  // (a view's verdict is its source's)
    cube_resolve_nan(r);
    return r->has_nan ? 1 : 0;
}

int rdr_cube_shape(const rdr_cube* q, int64_t* ny, int64_t* nx, int64_t* nz, int* dtype) {
    if (!q) return fail(nullptr, RDR_ERR_INVALID, "cube is NULL");
    if (ny) *ny = q->ny; if (nx) *nx = q->nx; if (nz) *nz = q->nz; if (dtype) *dtype = q->dtype;
    return RDR_OK;
}

int rdr_cube_axes(const rdr_cube* q, double* ys, double* xs, double* zs) {
    if (!q) return fail(nullptr, RDR_ERR_INVALID, "cube is NULL");
    if (ys) std::copy(q->ys.begin(), q->ys.end(), ys);
    if (xs) std::copy(q->xs.begin(), q->xs.end(), xs);
    if (zs) std::copy(q->zs.begin(), q->zs.end(), zs);
    return RDR_OK;
}

// LccParams of a conic model CRS from the C ABI's parameter array (shared by rdr_cube_set_projection / rdr_transform_cone)
static int cone_params(rdr_ctx* c, const char* who, int kind, const double* p, int np, LccParams& out) {
    const std::string w(who);
    if (kind == RDR_PROJ_STERE) {
        if (!p || np < 8) return fail(c, RDR_ERR_INVALID, w + ": STERE needs 8 parameters (a, es, lat_0, lat_ts, k_0, lon_0, x_0, y_0)");
        if (!(p[0] > 0) || p[1] < 0 || p[1] >= 1 || std::fabs(std::fabs(p[2]) - 90.0) > 1e-9 || (p[3] == p[3] && (std::fabs(p[3]) > 90 || p[3] * p[2] < 0)) || !(p[4] > 0))
            return fail(c, RDR_ERR_INVALID, w + ": only the POLAR stereographic aspect is supported (lat_0 = +-90, lat_ts in the same hemisphere, k_0 > 0)");
        out = stere_setup(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]);
        return RDR_OK;
    }
    if (kind != RDR_PROJ_LCC || !p || np < 8) return fail(c, RDR_ERR_INVALID, w + ": LCC needs 8 parameters (a, es, lat_1, lat_2, lat_0, lon_0, x_0, y_0)");
    if (!(p[0] > 0) || p[1] < 0 || p[1] >= 1 || std::fabs(p[2]) >= 90 || std::fabs(p[3]) >= 90 || std::fabs(p[2] + p[3]) < 1e-10)
        return fail(c, RDR_ERR_INVALID, w + ": invalid LCC parameters");
    out = lcc_setup(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]);
    return RDR_OK;
}

int rdr_cube_set_projection(rdr_cube* q, int kind, const double* p, int np) {
    if (!q) return fail(nullptr, RDR_ERR_INVALID, "cube is NULL");
    if (kind == RDR_PROJ_LONLAT) { q->proj = LccParams{0, 0, 0, 0, 0, 0, 0, 0}; return RDR_OK; }
    LccParams L;
    const int rc = cone_params(q->ctx, "rdr_cube_set_projection", kind, p, np, L);
    if (rc) return rc;
    q->proj = L;
    return RDR_OK;
}

int rdr_transform_cone(rdr_ctx* c, int kind, const double* p, int np, int direction, const double* in_a, const double* in_b, int64_t n,
                       double* out_a, double* out_b, int loc) {
    if (!c || (n > 0 && (!in_a || !in_b || !out_a || !out_b))) return fail(c, RDR_ERR_INVALID, "rdr_transform_cone: NULL argument");
    if (direction != 0 && direction != 1) return fail(c, RDR_ERR_INVALID, "rdr_transform_cone: direction is 0 (forward) or 1 (inverse)");
    LccParams L;
    int rc = cone_params(c, "rdr_transform_cone", kind, p, np, L); if (rc) return rc;
    if (n < 0) return fail(c, RDR_ERR_INVALID, "rdr_transform_cone: negative count");
    if (n == 0) return RDR_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    const void *da, *db; void *oa, *ob;
    rc = stage_in(c, SLOT_IN0, in_a, (size_t)n * 8, loc, &da); if (rc) return rc;
    rc = stage_in(c, SLOT_IN1, in_b, (size_t)n * 8, loc, &db); if (rc) return rc;
    rc = stage_out(c, SLOT_OUT0, out_a, (size_t)n * 8, loc, &oa); if (rc) return rc;
    rc = stage_out(c, SLOT_OUT1, out_b, (size_t)n * 8, loc, &ob); if (rc) return rc;
    hipLaunchKernelGGL(cone_kernel, dim3(grid_for(n, 256, c->num_cus * 8)), dim3(256), 0, c->stream, L, direction, (const double*)da, (const double*)db, n,
                       (double*)oa, (double*)ob);
    HIPCHECK(c, hipGetLastError());
    rc = finish_out(c, out_a, oa, (size_t)n * 8, loc); if (rc) return rc;
    rc = finish_out(c, out_b, ob, (size_t)n * 8, loc); if (rc) return rc;
    if (loc == RDR_HOST) HIPCHECK(c, hipStreamSynchronize(c->stream));
    return RDR_OK;
}

int rdr_project_points(rdr_ctx* c, const rdr_cube* q, const double* lat, const double* lon, int64_t n, double* y, double* x, int loc) {
    if (!c || !q || !lat || !lon || !y || !x) return fail(c, RDR_ERR_INVALID, "rdr_project_points: NULL argument");
    note_use(c, q);
    if (n < 0) return fail(c, RDR_ERR_INVALID, "rdr_project_points: negative count");
    if (n == 0) return RDR_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    if (q->proj.kind == 0) {     // lon/lat cube: identity
        const hipMemcpyKind k = loc == RDR_HOST ? hipMemcpyHostToHost : hipMemcpyDeviceToDevice;
        HIPCHECK(c, hipMemcpyAsync(y, lat, (size_t)n * 8, k, c->stream));
        HIPCHECK(c, hipMemcpyAsync(x, lon, (size_t)n * 8, k, c->stream));
        if (loc == RDR_HOST) HIPCHECK(c, hipStreamSynchronize(c->stream));
        return RDR_OK;
    }
    const void *a, *b; void *oy, *ox;
    int rc = stage_in(c, SLOT_IN0, lat, (size_t)n * 8, loc, &a); if (rc) return rc;
    rc = stage_in(c, SLOT_IN1, lon, (size_t)n * 8, loc, &b); if (rc) return rc;
    rc = stage_out(c, SLOT_OUT0, y, (size_t)n * 8, loc, &oy); if (rc) return rc;
    rc = stage_out(c, SLOT_OUT1, x, (size_t)n * 8, loc, &ox); if (rc) return rc;
    hipLaunchKernelGGL(lcc_kernel, dim3(grid_for(n, 256, c->num_cus * 8)), dim3(256), 0, c->stream, q->proj, (const double*)a, (const double*)b, n,
                       (double*)oy, (double*)ox);
    HIPCHECK(c, hipGetLastError());
    rc = finish_out(c, y, oy, (size_t)n * 8, loc); if (rc) return rc;
    rc = finish_out(c, x, ox, (size_t)n * 8, loc); if (rc) return rc;
    if (loc == RDR_HOST) HIPCHECK(c, hipStreamSynchronize(c->stream));
    return RDR_OK;
}

int rdr_transform_tm(rdr_ctx* c, const double* p, int np, int direction, const double* in_a, const double* in_b, int64_t n, double* out_a,
                     double* out_b, int loc) {
    if (!c || !p || np < 7 || (n > 0 && (!in_a || !in_b || !out_a || !out_b))) return fail(c, RDR_ERR_INVALID, "rdr_transform_tm: NULL argument / 7 parameters (a, es, lat_0, lon_0, k_0, x_0, y_0)");
    if (!(p[0] > 0) || p[1] < 0 || p[1] >= 1 || std::fabs(p[2]) > 90 || !(p[4] > 0)) return fail(c, RDR_ERR_INVALID, "rdr_transform_tm: invalid transverse-Mercator parameters");
    if (direction != 0 && direction != 1) return fail(c, RDR_ERR_INVALID, "rdr_transform_tm: direction is 0 (forward) or 1 (inverse)");
    if (n < 0) return fail(c, RDR_ERR_INVALID, "rdr_transform_tm: negative count");
    if (n == 0) return RDR_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    const TmParams T = tm_setup(p[0], p[1], p[2], p[3], p[4], p[5], p[6]);
    const void *da, *db; void *oa, *ob;
    int rc = stage_in(c, SLOT_IN0, in_a, (size_t)n * 8, loc, &da); if (rc) return rc;
    rc = stage_in(c, SLOT_IN1, in_b, (size_t)n * 8, loc, &db); if (rc) return rc;
    rc = stage_out(c, SLOT_OUT0, out_a, (size_t)n * 8, loc, &oa); if (rc) return rc;
    rc = stage_out(c, SLOT_OUT1, out_b, (size_t)n * 8, loc, &ob); if (rc) return rc;
    hipLaunchKernelGGL(tm_kernel, dim3(grid_for(n, 256, c->num_cus * 8)), dim3(256), 0, c->stream, T, direction, (const double*)da, (const double*)db, n,
                       (double*)oa, (double*)ob);
    HIPCHECK(c, hipGetLastError());
    rc = finish_out(c, out_a, oa, (size_t)n * 8, loc); if (rc) return rc;
    rc = finish_out(c, out_b, ob, (size_t)n * 8, loc); if (rc) return rc;
    if (loc == RDR_HOST) HIPCHECK(c, hipStreamSynchronize(c->stream));
    return RDR_OK;
}

int rdr_cube_blend(rdr_ctx* c, const rdr_cube* a, double w1, const rdr_cube* b, double w2, rdr_cube** out) {
    if (!c || !a || !b || !out) return fail(c, RDR_ERR_INVALID, "rdr_cube_blend: NULL argument");
    note_use(c, a); note_use(c, b);
    if (a->ny != b->ny || a->nx != b->nx || a->nz != b->nz || a->dtype != b->dtype || a->ys != b->ys || a->xs != b->xs || a->zs != b->zs)
        return fail(c, RDR_ERR_INVALID, "rdr_cube_blend: cubes are not on the same grid / dtype");
    HIPCHECK(c, hipSetDevice(c->device));
    rdr_cube* q = new rdr_cube();
    q->ctx = c; q->ny = a->ny; q->nx = a->nx; q->nz = a->nz; q->dtype = a->dtype;
    q->ys = a->ys; q->xs = a->xs; q->zs = a->zs; q->proj = a->proj;
    q->has_nan = rdr_cube_has_nan(a) == 1 || rdr_cube_has_nan(b) == 1;
    int rc = cube_alloc(c, q);
    if (rc) { rdr_cube_destroy(q); return rc; }
    const int64_t nscal = 2 * a->ny * a->nx * a->nz;                  // (wet, hydro) pairs as one flat array
    const int vec = a->dtype == RDR_F32 ? 4 : 2;
    // one pass: a grid that covers the array once (4 vectors per lane) streams at 6.4 TB/s; a persistent 8-16 blocks per CU grid
    // looping over it measured 4.8-5.4 TB/s (tools/probes/blend_probe.hip)
    const int g = grid_for((nscal / vec + 3) / 4, 256, 1 << 22);
    {
        KTimer t(c, 3);
        if (a->dtype == RDR_F32)
            hipLaunchKernelGGL((blend_kernel<float>), dim3(g), dim3(256), 0, c->stream, (const float*)a->d_vals, (float)w1,
                               (const float*)b->d_vals, (float)w2, (float*)q->d_vals, nscal);
        else
            hipLaunchKernelGGL((blend_kernel<double>), dim3(g), dim3(256), 0, c->stream, (const double*)a->d_vals, w1,
                               (const double*)b->d_vals, w2, (double*)q->d_vals, nscal);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { rdr_cube_destroy(q); return fail(c, RDR_ERR_HIP, hipGetErrorString(e)); }
    rc = cube_mark_ready(c, q);                    // (no host synchronisation: device sources, device result - other streams wait for the event)
    if (rc) { rdr_cube_destroy(q); return rc; }
    *out = q;
    return RDR_OK;
}

int rdr_inverse_time_weights(rdr_ctx* c, const double* az, int64_t n, const double* dates, int32_t nd, double window_s, double regularizer,
                             double* weights, int loc) {
    if (!c || !az || !dates || !weights) return fail(c, RDR_ERR_INVALID, "rdr_inverse_time_weights: NULL argument");
    if (nd < 1) return fail(c, RDR_ERR_INVALID, "No dates provided");
    if (nd > 8) return fail(c, RDR_ERR_INVALID, "rdr_inverse_time_weights: at most 8 dates");
    DateSet D; D.nd = nd; D.reg = regularizer;
    for (int i = 0; i < nd; ++i) {
        D.date[i] = dates[i];
        for (int j = 0; j < i; ++j) if (dates[j] == dates[i]) return fail(c, RDR_ERR_INVALID, "Dates provided must be unique");
    }
    if (window_s < 0) {                                  // s1_azimuth_timing.py:375-376: infer the model time step
        if (nd < 2) return fail(c, RDR_ERR_INVALID, "rdr_inverse_time_weights: the temporal window cannot be inferred from one date");
        window_s = std::fabs(dates[1] - dates[0]);
        for (int i = 2; i < nd; ++i) window_s = std::min(window_s, std::fabs(dates[i] - dates[0]));
    }
    D.window = window_s;
    if (n < 0) return fail(c, RDR_ERR_INVALID, "rdr_inverse_time_weights: negative count");
    if (n == 0) return RDR_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    const void* da; void* dw;
    int rc = stage_in(c, SLOT_IN0, az, (size_t)n * 8, loc, &da); if (rc) return rc;
    rc = stage_out(c, SLOT_OUT0, weights, (size_t)n * nd * 8, loc, &dw); if (rc) return rc;
    HIPCHECK(c, hipMemsetAsync(c->d_flags, 0, sizeof(int), c->stream));
    hipLaunchKernelGGL(time_weights_kernel, dim3(grid_for(n, 256, c->num_cus * 8)), dim3(256), 0, c->stream, D, (const double*)da, n, (double*)dw, c->d_flags);
    HIPCHECK(c, hipGetLastError());
    rc = finish_out(c, weights, dw, (size_t)n * nd * 8, loc); if (rc) return rc;
    int f = 0;
    HIPCHECK(c, hipMemcpyAsync(&f, c->d_flags, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    if (!f) return fail(c, RDR_ERR_INVALID, "No dates provided are within temporal window");
    return RDR_OK;
}

int rdr_cube_blend_weighted(rdr_ctx* c, const rdr_cube* const* cubes, int32_t nd, const double* weights, int loc, rdr_cube** out) {
    if (!c || !cubes || !weights || !out) return fail(c, RDR_ERR_INVALID, "rdr_cube_blend_weighted: NULL argument");
    if (nd < 1 || nd > 8) return fail(c, RDR_ERR_INVALID, "rdr_cube_blend_weighted: 1..8 cubes");
    const rdr_cube* a = cubes[0];
    CubeSet S; S.nd = nd;
    for (int i = 0; i < nd; ++i) {
        const rdr_cube* b = cubes[i];
        if (!b) return fail(c, RDR_ERR_INVALID, "rdr_cube_blend_weighted: NULL cube");
        note_use(c, b);
        if (a->ny != b->ny || a->nx != b->nx || a->nz != b->nz || a->dtype != b->dtype || a->ys != b->ys || a->xs != b->xs || a->zs != b->zs)
            return fail(c, RDR_ERR_INVALID, "rdr_cube_blend_weighted: cubes are not on the same grid / dtype");
        S.v[i] = b->d_vals;
    }
    HIPCHECK(c, hipSetDevice(c->device));
    const int64_t total = a->ny * a->nx * a->nz;
    const void* dwt;
    int rc = stage_in(c, SLOT_IN0, weights, (size_t)total * nd * 8, loc, &dwt); if (rc) return rc;
    rdr_cube* q = new rdr_cube();
    q->ctx = c; q->ny = a->ny; q->nx = a->nx; q->nz = a->nz; q->dtype = RDR_F64;
    q->ys = a->ys; q->xs = a->xs; q->zs = a->zs; q->proj = a->proj;
    rc = cube_alloc(c, q);
    if (rc) { rdr_cube_destroy(q); return rc; }
    const int g = grid_for(total, 256, c->num_cus * 8);
    {
        KTimer t(c, 3);
        if (a->dtype == RDR_F32)
            hipLaunchKernelGGL((blend_weighted_kernel<float2>), dim3(g), dim3(256), 0, c->stream, S, (const double*)dwt, a->ny, a->nx, a->nz, (double2*)q->d_vals);
        else
            hipLaunchKernelGGL((blend_weighted_kernel<double2>), dim3(g), dim3(256), 0, c->stream, S, (const double*)dwt, a->ny, a->nx, a->nz, (double2*)q->d_vals);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { rdr_cube_destroy(q); return fail(c, RDR_ERR_HIP, hipGetErrorString(e)); }
    *out = q;
    return RDR_OK;
}

// ---- GUNW phase conversion (aria/calcGUNW.py:54-59): delay [m] -> interferometric phase [rad] ------------------------
// ds['wet'] * phase2range with phase2range = -4 pi / wavelength; a Python float is a weak scalar, so f32 delays are
// multiplied in f32 by the f32-rounded factor and f64 delays in f64.
template <typename T>
__global__ __launch_bounds__(256) void phase_kernel(const T* __restrict__ wet, const T* __restrict__ hyd, int64_t n, T factor,
                                                    T* __restrict__ owet, T* __restrict__ ohyd) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        owet[i] = wet[i] * factor;
        ohyd[i] = hyd[i] * factor;
    }
}

int rdr_delays_to_phase(rdr_ctx* c, const void* wet, const void* hydro, int64_t n, int dtype, double wavelength, void* wet_out, void* hydro_out, int loc) {
    if (!c || !wet || !hydro || !wet_out || !hydro_out) return fail(c, RDR_ERR_INVALID, "rdr_delays_to_phase: NULL argument");
    if (dtype != RDR_F32 && dtype != RDR_F64) return fail(c, RDR_ERR_INVALID, "rdr_delays_to_phase: dtype must be RDR_F32 or RDR_F64");
    if (!(wavelength > 0.0)) return fail(c, RDR_ERR_INVALID, "rdr_delays_to_phase: wavelength must be positive");
    if (n < 0) return fail(c, RDR_ERR_INVALID, "rdr_delays_to_phase: negative count");
    if (n == 0) return RDR_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)n * (dtype == RDR_F32 ? 4 : 8);
    const void *dw, *dh; void *ow, *oh;
    int rc = stage_in(c, SLOT_IN0, wet, bytes, loc, &dw); if (rc) return rc;
    rc = stage_in(c, SLOT_IN1, hydro, bytes, loc, &dh); if (rc) return rc;
    rc = stage_out(c, SLOT_OUT0, wet_out, bytes, loc, &ow); if (rc) return rc;
    rc = stage_out(c, SLOT_OUT1, hydro_out, bytes, loc, &oh); if (rc) return rc;
    const double phase2range = (-4.0 * 3.141592653589793) / wavelength;       // calcGUNW.py:54
    const int g = grid_for(n, 256, c->num_cus * 8);
    if (dtype == RDR_F32)
        hipLaunchKernelGGL(phase_kernel<float>, dim3(g), dim3(256), 0, c->stream, (const float*)dw, (const float*)dh, n, (float)phase2range, (float*)ow, (float*)oh);
    else
        hipLaunchKernelGGL(phase_kernel<double>, dim3(g), dim3(256), 0, c->stream, (const double*)dw, (const double*)dh, n, phase2range, (double*)ow, (double*)oh);
    HIPCHECK(c, hipGetLastError());
    rc = finish_out(c, wet_out, ow, bytes, loc); if (rc) return rc;
    rc = finish_out(c, hydro_out, oh, bytes, loc); if (rc) return rc;
    if (loc == RDR_HOST) HIPCHECK(c, hipStreamSynchronize(c->stream));
    return RDR_OK;
}

int rdr_cube_read(rdr_ctx* c, const rdr_cube* q, void* wet, void* hydro) {
    if (!c || !q || !wet || !hydro) return fail(c, RDR_ERR_INVALID, "rdr_cube_read: NULL argument");
    note_use(c, q);
    HIPCHECK(c, hipSetDevice(c->device));
    const int64_t total = q->ny * q->nx * q->nz;
    const size_t esz = q->dtype == RDR_F32 ? 4 : 8;
    void *dw, *dh;
    int rc = ensure(c, SLOT_OUT0, total * esz, &dw); if (rc) return rc;
    rc = ensure(c, SLOT_OUT1, total * esz, &dh); if (rc) return rc;
    const int g = grid_for(total, 256, c->num_cus * 8);
    if (q->dtype == RDR_F32)
        hipLaunchKernelGGL((unpack_cube_kernel<float, float2>), dim3(g), dim3(256), 0, c->stream, (const float2*)q->d_vals, (float*)dw, (float*)dh, total);
    else
        hipLaunchKernelGGL((unpack_cube_kernel<double, double2>), dim3(g), dim3(256), 0, c->stream, (const double2*)q->d_vals, (double*)dw, (double*)dh, total);
    HIPCHECK(c, hipGetLastError());
    HIPCHECK(c, hipMemcpyAsync(wet, dw, total * esz, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipMemcpyAsync(hydro, dh, total * esz, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return RDR_OK;
}

// ---- zenith / projected ---------------------------------------------------------------------------
// ---- corner-quad copy (cube_kernels.h): when, and how --------------------------------------------------------------------------
static size_t quad_need_bytes(const rdr_cube* q, int* nblk) {
    const int cpb = q->dtype == RDR_F32 ? Quad<float2>::CPB : Quad<double2>::CPB;
    *nblk = (int)((q->nz - 1 + cpb - 1) / cpb);
    return (size_t)(q->ny - 1) * (size_t)(q->nx - 1) * (size_t)*nblk * 128;
}

static int quad_build(rdr_ctx* c, const rdr_cube* q_any) {
    const rdr_cube* const q = root(q_any);              // (the copy belongs to the buffers, i.e. to the source of a view)
    // (one builder per cube: two contexts sharing a cube must not both allocate and publish a copy)
    std::lock_guard<std::mutex> guard(q->quad_mutex);
    if (q->d_quad) return RDR_OK;
    int nblk = 0;
    const size_t need = quad_need_bytes(q, &nblk);
    void* dq = nullptr;
    { const int rc_ = dev_malloc(c, &dq, need); if (rc_) return rc_; }
    struct Pub { const rdr_cube* q; void* p; size_t need; int nblk; bool ok = false;
                 ~Pub() { if (ok) { q->quad_bytes = need; q->quad_nblk = nblk; q->d_quad = p; } else (void)hipFree(p); } } pub{q, dq, need, nblk};
    const int64_t parts = (int64_t)(need / 16);
    const int g = (grid_for(parts, 256, c->num_cus * 32) + 7) / 8 * 8;       // a multiple of 8: one share of the ranges per XCD (quad_build_kernel)
    if (q->dtype == RDR_F32) hipLaunchKernelGGL((quad_build_kernel<float2>), dim3(g), dim3(256), 0, c->stream, (const float2*)q->d_vals, (int)q->ny, (int)q->nx, (int)q->nz, nblk, (uint4*)dq);
    else hipLaunchKernelGGL((quad_build_kernel<double2>), dim3(g), dim3(256), 0, c->stream, (const double2*)q->d_vals, (int)q->ny, (int)q->nx, (int)q->nz, nblk, (uint4*)dq);
    HIPCHECK(c, hipGetLastError());
    // one-time, like rdr_cube_create: the copy must be complete before it is PUBLISHED - any other stream (a context switched to
    // torch's stream, a second context sharing the cube) may read it from then on.  0.7 ms of a build that happens once per cube.
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    pub.ok = true;
    return RDR_OK;
}

// Policy of the automatic build: only for point sets and cubes large enough that the four-lines-per-point gather is what bounds
// the call (>= 256 k points, a cube beyond 192 MB - smaller ones live in L2 / the 256 MB Infinity Cache), only when the copy fits a quarter
// of the free memory.  When: by TIME, from the measured rates (profiles/r03_secondary.json, r04_secondary.json: 5 M stations on the
// 400 MB HRRR-sized cube) - the direct gather moves 572 B per point at 6.5 TB/s (88 ps), the quad gather 168 B at 5.65 TB/s (30 ps),
// the build writes the copy at 3.0 TB/s: building at the FIRST large call pays when n x 58 ps > bytes / 3.0 TB/s, i.e. n x 175 B >
// the copy's bytes (12 M points for that cube; a 5 M-station one-shot is FASTER from the (y,x,z) cube: 0.43 ms against 0.72 + 0.15 -
// fewer HBM bytes per point is not the goal, time is), else from the second large call on the cube (then the copy has paid for itself
// by the end of that call).  RAIDER_HIP_POINT_INDEX=0 never, =1 at the first large call, =2 second call only.
static bool quad_wanted(rdr_ctx* c, const rdr_cube* q_any, int64_t n) {
    const rdr_cube* const q = root(q_any);
    if (q->d_quad) return true;
    static const int env = []() { const char* e = std::getenv("RAIDER_HIP_POINT_INDEX"); return e ? std::atoi(e) : -1; }();
    if (env == 0) return false;
    const size_t cube_bytes = (size_t)q->ny * q->nx * q->nz * (q->dtype == RDR_F32 ? 8 : 16);
    // (round 5: 32 MB -> 192 MB.  A 38.8 MB f64 cube - the intermediate delay cube of BASELINE configs[1] - is Infinity-Cache resident: 10^6
    // points from it take 23.7 us directly, 32.0 us from the copy, which also costs 54 us to build; profiles/r05_c2_counters.json at 458657a3)
    if (n < (1 << 18) || cube_bytes < ((size_t)192 << 20) || q->ny < 2 || q->nx < 2 || q->nz < 2) return false;
    int nblk; const size_t need = quad_need_bytes(q, &nblk);
    const bool pays_now = env != 2 && (double)n * 175.0 > (double)need;
    if (++q->big_point_calls < 2 && env != 1 && !pays_now) return false;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || need > free_b / 4) return false;
    (void)c;
    return true;
}

int rdr_cube_point_index(rdr_ctx* c, rdr_cube* q_any, int mode) {
    if (!c || !q_any) return fail(c, RDR_ERR_INVALID, "rdr_cube_point_index: NULL argument");
    const rdr_cube* const q = root(q_any);
    note_use(c, q);
    HIPCHECK(c, hipSetDevice(c->device));
    if (mode == 0) {
        std::lock_guard<std::mutex> guard(q->quad_mutex);
        if (q->d_quad) { HIPCHECK(c, q->foreign.load() ? hipDeviceSynchronize() : hipStreamSynchronize(c->stream)); HIPCHECK(c, hipFree(q->d_quad)); q->d_quad = nullptr; q->quad_bytes = 0; q->quad_nblk = 0; }
        q->big_point_calls = 0;
        return RDR_OK;
    }
    if (mode != 1) return fail(c, RDR_ERR_INVALID, "rdr_cube_point_index: mode is 0 (free) or 1 (build)");
    if (q->ny < 2 || q->nx < 2 || q->nz < 2) return fail(c, RDR_ERR_INVALID, "rdr_cube_point_index: the cube needs two nodes per axis");
    return quad_build(c, q);
}

int64_t rdr_cube_point_index_bytes(const rdr_cube* q) { return q ? (int64_t)root(q)->quad_bytes : -1; }

// a second epoch blended in at the corners (rdr_interp3_blend): vb == NULL for an ordinary gather
// pair != NULL: the blend was made as a PAIRED cube in the context's scratch (rdr_interp3_blend_cube) - the gather reads that
struct BlendSpec { const void* vb = nullptr; double w1 = 1.0, w2 = 0.0; const void* pair = nullptr; bool want_pair = false; };

static void launch_interp(rdr_ctx* c, const rdr_cube* q, const PointQuery& Qk, int64_t cnt, double* dwk, double* dhk, bool quad, const BlendSpec& B = BlendSpec()) {
    const int g = grid_for(cnt, 256, c->num_cus * 8);
    KTimer t(c, 2);
    if (B.pair) {
        if (q->dtype == RDR_F32)
            hipLaunchKernelGGL((interp_points_pair_kernel<float2>), dim3(g), dim3(256), axes_smem(q), c->stream, make_view<float2>(q), (const float2*)B.pair, Qk, cnt, dwk, dhk,
                               (int)axes_fit_lds(q));
        else
            hipLaunchKernelGGL((interp_points_pair_kernel<double2>), dim3(g), dim3(256), axes_smem(q), c->stream, make_view<double2>(q), (const double2*)B.pair, Qk, cnt, dwk, dhk,
                               (int)axes_fit_lds(q));
    } else if (B.vb) {
        if (q->dtype == RDR_F32)
            hipLaunchKernelGGL((interp_points_blend_kernel<float2>), dim3(g), dim3(256), axes_smem(q), c->stream, make_view<float2>(q), (const float2*)B.vb, B.w1, B.w2,
                               Qk, cnt, dwk, dhk, (int)axes_fit_lds(q));
        else
            hipLaunchKernelGGL((interp_points_blend_kernel<double2>), dim3(g), dim3(256), axes_smem(q), c->stream, make_view<double2>(q), (const double2*)B.vb, B.w1, B.w2,
                               Qk, cnt, dwk, dhk, (int)axes_fit_lds(q));
    } else if (quad) {
        if (q->dtype == RDR_F32)
            hipLaunchKernelGGL((interp_points_quad_kernel<float2>), dim3(g), dim3(256), axes_smem(q), c->stream, make_view<float2>(q),
                               (const uint4*)root(q)->d_quad, root(q)->quad_nblk, Qk, cnt, dwk, dhk, (int)axes_fit_lds(q));
        else
            hipLaunchKernelGGL((interp_points_quad_kernel<double2>), dim3(g), dim3(256), axes_smem(q), c->stream, make_view<double2>(q),
                               (const uint4*)root(q)->d_quad, root(q)->quad_nblk, Qk, cnt, dwk, dhk, (int)axes_fit_lds(q));
    } else if (q->dtype == RDR_F32)
        hipLaunchKernelGGL((interp_points_kernel<float2>), dim3(g), dim3(256), axes_smem(q), c->stream, make_view<float2>(q), Qk, cnt, dwk, dhk,
                           (int)axes_fit_lds(q));
    else
        hipLaunchKernelGGL((interp_points_kernel<double2>), dim3(g), dim3(256), axes_smem(q), c->stream, make_view<double2>(q), Qk, cnt, dwk, dhk,
                           (int)axes_fit_lds(q));
}

// An error return must not leave asynchronous copies into (or out of) the caller's buffers in flight: whoever enqueued on the three
// streams arms this; leaving the scope on any path but the disarmed one waits for all of them.
struct StreamsQuiesce {
    rdr_ctx* c; bool armed = true;
    explicit StreamsQuiesce(rdr_ctx* ctx) : c(ctx) {}
    ~StreamsQuiesce() {
        if (!armed) return;
        (void)hipStreamSynchronize(c->down_stream); (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamSynchronize(c->stream);
        (void)hipGetLastError();
    }
};

// Large HOST point sets: the points go up (copy stream) and the delays come down (download stream) in chunks, the download of chunk k
// under the upload of chunk k+1 (PCIe is full duplex), the gathers on the ctx stream in between - 40-48 B per point cross the link,
// which is all such a call costs (the gather itself: 0.03 ms per 10^6 points).  Uploads on their OWN stream also run under whatever
// the ctx stream is still doing (rdr_point_delays: the build of the intermediate cube).  `slots`: device scratch for y, x, z, proj,
// wet, hydro.  Synchronises all three streams before it returns.  (A download into pageable memory blocks the host thread instead of
// overlapping: still correct.)
static int interp_pipeline(rdr_ctx* c, const char* who, const rdr_cube* q, bool quad, const double* y, const double* x, const double* z, int64_t n,
                           PointQuery Q, const double* proj, double* wet, double* hydro, const int* slots, const BlendSpec& B = BlendSpec()) {
    const bool has_proj = Q.pmode == 1 || Q.pmode == 3;
    const size_t ystride = x ? 1 : 3;
    void *dy, *dx = nullptr, *dz = nullptr, *dp = nullptr, *dw = nullptr, *dh = nullptr;
    int rc = ensure(c, slots[0], (size_t)n * ystride * 8, &dy); if (rc) return rc;
    if (x) { rc = ensure(c, slots[1], (size_t)n * 8, &dx); if (rc) return rc; rc = ensure(c, slots[2], (size_t)n * 8, &dz); if (rc) return rc; }
    if (has_proj) { rc = ensure(c, slots[3], (size_t)n * 8, &dp); if (rc) return rc; }
    if (wet) { rc = ensure(c, slots[4], (size_t)n * 8, &dw); if (rc) return rc; }
    if (hydro) { rc = ensure(c, slots[5], (size_t)n * 8, &dh); if (rc) return rc; }
    // chunks of >= 512 k points (a copy call costs 10-20 us: 1 MB copies would spend as long on calls as on bytes), at most 8
    const int nchunk = (int)std::max<int64_t>(1, std::min<int64_t>(8, n >> 19));
    struct EventList { std::vector<hipEvent_t> v; ~EventList() { for (auto& e : v) if (e) (void)hipEventDestroy(e); } } evs;
    evs.v.assign((size_t)2 * nchunk, nullptr);
    for (auto& e : evs.v)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(c, RDR_ERR_HIP, std::string(who) + ": event creation failed");
    StreamsQuiesce quiesce(c);                  // (declared after the events: they are destroyed only once the streams are idle)
    for (int k = 0; k < nchunk; ++k) {
        const int64_t o = n * k / nchunk, cnt = n * (k + 1) / nchunk - o;
        HIPCHECK(c, hipMemcpyAsync((double*)dy + o * ystride, y + o * ystride, (size_t)cnt * ystride * 8, hipMemcpyHostToDevice, c->copy_stream));
        if (x) {
            HIPCHECK(c, hipMemcpyAsync((double*)dx + o, x + o, (size_t)cnt * 8, hipMemcpyHostToDevice, c->copy_stream));
            HIPCHECK(c, hipMemcpyAsync((double*)dz + o, z + o, (size_t)cnt * 8, hipMemcpyHostToDevice, c->copy_stream));
        }
        if (has_proj) HIPCHECK(c, hipMemcpyAsync((double*)dp + o, proj + o, (size_t)cnt * 8, hipMemcpyHostToDevice, c->copy_stream));
        HIPCHECK(c, hipEventRecord(evs.v[2 * k], c->copy_stream));
        HIPCHECK(c, hipStreamWaitEvent(c->stream, evs.v[2 * k], 0));
        PointQuery Qk = Q;
        Qk.y = (const double*)dy + o * ystride;
        Qk.x = x ? (const double*)dx + o : nullptr; Qk.z = x ? (const double*)dz + o : nullptr;
        Qk.proj = has_proj ? (const double*)dp + o : nullptr;
        launch_interp(c, q, Qk, cnt, dw ? (double*)dw + o : nullptr, dh ? (double*)dh + o : nullptr, quad, B);
        HIPCHECK(c, hipGetLastError());
        HIPCHECK(c, hipEventRecord(evs.v[2 * k + 1], c->stream));
        HIPCHECK(c, hipStreamWaitEvent(c->down_stream, evs.v[2 * k + 1], 0));
        if (wet) HIPCHECK(c, hipMemcpyAsync(wet + o, (double*)dw + o, (size_t)cnt * 8, hipMemcpyDeviceToHost, c->down_stream));
        if (hydro) HIPCHECK(c, hipMemcpyAsync(hydro + o, (double*)dh + o, (size_t)cnt * 8, hipMemcpyDeviceToHost, c->down_stream));
    }
    HIPCHECK(c, hipStreamSynchronize(c->down_stream));
    HIPCHECK(c, hipStreamSynchronize(c->copy_stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    quiesce.armed = false;
    return RDR_OK;
}

static int point_query_args(rdr_ctx* c, const char* who, const double* y, const double* x, const double* z, int64_t n, int pmode, const double* proj,
                            const double* wet, const double* hydro) {
    if (n < 0) return fail(c, RDR_ERR_INVALID, std::string(who) + ": negative count");
    if (n > 0 && (!y || (x && !z) || (!x && z) || (!wet && !hydro))) return fail(c, RDR_ERR_INVALID, std::string(who) + ": NULL argument");
    if (pmode < 0 || pmode > 3) return fail(c, RDR_ERR_INVALID, std::string(who) + ": proj_mode is 0 (none), 1 (incidence array), 2 (one incidence) or 3 (divisor array)");
    if (n > 0 && (pmode == 1 || pmode == 3) && !proj) return fail(c, RDR_ERR_INVALID, std::string(who) + ": proj_mode 1 / 3 need the proj array");
    return RDR_OK;
}

// the point query behind rdr_interp3 / rdr_interp3_project: y/x/z three arrays (x != NULL) or y = packed (n,3); pmode / proj / inc0 as
// PointQuery (cube_kernels.h); either output may be NULL (it is then neither written nor downloaded)
static int interp3_impl(rdr_ctx* c, const char* who, const rdr_cube* q, const double* y, const double* x, const double* z, int64_t n, int pmode,
                        const double* proj, double inc0, double* wet, double* hydro, int loc, const BlendSpec& B_in = BlendSpec()) {
    if (!c || !q) return fail(c, RDR_ERR_INVALID, std::string(who) + ": NULL argument");
    note_use(c, q);
    int rc = point_query_args(c, who, y, x, z, n, pmode, proj, wet, hydro); if (rc) return rc;
    if (n == 0) return RDR_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    BlendSpec B = B_in;
    if (B.want_pair) {
        // the blend as a cube, written ONCE in the layout the gather reads best (blend_pair_kernel), into the context's scratch: it never
        // becomes an rdr_cube, so no other kernel can be handed the paired layout
        const size_t esz = q->dtype == RDR_F32 ? 8 : 16;
        const int64_t npx = (q->nx + 1) / 2;
        void* dp;
        rc = ensure(c, SLOT_TMPCUBE, (size_t)q->ny * npx * q->nz * 2 * esz, &dp); if (rc) return rc;
        const int zv = (q->dtype == RDR_F32 && q->nz % 2 == 0) ? 2 : 1;
        const int g = grid_for(q->ny * npx * (q->nz / zv), 256, 1 << 22);
        {
            KTimer t(c, 3);
            if (q->dtype == RDR_F32 && zv == 2)
                hipLaunchKernelGGL((blend_pair_kernel<float2, 2>), dim3(g), dim3(256), 0, c->stream, (const float2*)q->d_vals, (float)B.w1, (const float2*)B.vb, (float)B.w2,
                                   (float2*)dp, (int)q->ny, (int)q->nx, (int)q->nz);
            else if (q->dtype == RDR_F32)
                hipLaunchKernelGGL((blend_pair_kernel<float2, 1>), dim3(g), dim3(256), 0, c->stream, (const float2*)q->d_vals, (float)B.w1, (const float2*)B.vb, (float)B.w2,
                                   (float2*)dp, (int)q->ny, (int)q->nx, (int)q->nz);
            else
                hipLaunchKernelGGL((blend_pair_kernel<double2, 1>), dim3(g), dim3(256), 0, c->stream, (const double2*)q->d_vals, B.w1, (const double2*)B.vb, B.w2,
                                   (double2*)dp, (int)q->ny, (int)q->nx, (int)q->nz);
        }
        HIPCHECK(c, hipGetLastError());
        B.pair = dp;
    }
    PointQuery Q; std::memset(&Q, 0, sizeof(Q));
    Q.pmode = pmode; Q.inc0 = inc0;
    const bool has_proj = pmode == 1 || pmode == 3;
    const size_t ystride = x ? 1 : 3;                                  // doubles per point behind `y`
    const bool quad = !B.vb && quad_wanted(c, q, n);
    if (quad) { rc = quad_build(c, q); if (rc) return rc; }
    static const bool no_pipeline = std::getenv("RAIDER_HIP_NO_PIPELINE") != nullptr;
    if (loc == RDR_HOST && n >= (1 << 18) && !no_pipeline) {
        static const int slots[6] = {SLOT_IN0, SLOT_IN1, SLOT_IN2, SLOT_IN3, SLOT_OUT0, SLOT_OUT1};
        return interp_pipeline(c, who, q, quad, y, x, z, n, Q, proj, wet, hydro, slots, B);
    }
    const void* d; void *dw = nullptr, *dh = nullptr;
    rc = stage_in(c, SLOT_IN0, y, (size_t)n * ystride * 8, loc, &d); if (rc) return rc; Q.y = (const double*)d;
    if (x) {
        rc = stage_in(c, SLOT_IN1, x, (size_t)n * 8, loc, &d); if (rc) return rc; Q.x = (const double*)d;
        rc = stage_in(c, SLOT_IN2, z, (size_t)n * 8, loc, &d); if (rc) return rc; Q.z = (const double*)d;
    }
    if (has_proj) { rc = stage_in(c, SLOT_IN3, proj, (size_t)n * 8, loc, &d); if (rc) return rc; Q.proj = (const double*)d; }
    if (wet) { rc = stage_out(c, SLOT_OUT0, wet, (size_t)n * 8, loc, &dw); if (rc) return rc; }
    if (hydro) { rc = stage_out(c, SLOT_OUT1, hydro, (size_t)n * 8, loc, &dh); if (rc) return rc; }
    launch_interp(c, q, Q, n, (double*)dw, (double*)dh, quad, B);
    HIPCHECK(c, hipGetLastError());
    rc = finish_out(c, wet, dw, (size_t)n * 8, loc); if (rc) return rc;
    rc = finish_out(c, hydro, dh, (size_t)n * 8, loc); if (rc) return rc;
    if (loc == RDR_HOST) HIPCHECK(c, hipStreamSynchronize(c->stream));
    return RDR_OK;
}

int rdr_interp3(rdr_ctx* c, const rdr_cube* q, const double* pts, int64_t n, double* wet, double* hydro, int loc) {
    return interp3_impl(c, "rdr_interp3", q, pts, nullptr, nullptr, n, 0, nullptr, 0.0, wet, hydro, loc);
}

int rdr_interp3_blend(rdr_ctx* c, const rdr_cube* a, double w1, const rdr_cube* b, double w2, const double* y, const double* x, const double* z, int64_t n,
                      double* wet, double* hydro, int loc) {
    if (!c || !a || !b) return fail(c, RDR_ERR_INVALID, "rdr_interp3_blend: NULL argument");
    note_use(c, b);
    if (a->ny != b->ny || a->nx != b->nx || a->nz != b->nz || a->dtype != b->dtype || a->ys != b->ys || a->xs != b->xs || a->zs != b->zs)
        return fail(c, RDR_ERR_INVALID, "rdr_interp3_blend: the two epochs are not on the same grid / dtype");
    BlendSpec B; B.vb = b->d_vals; B.w1 = w1; B.w2 = w2;
    return interp3_impl(c, "rdr_interp3_blend", a, y, x, z, n, 0, nullptr, 0.0, wet, hydro, loc, B);
}

int rdr_interp3_blend_cube(rdr_ctx* c, const rdr_cube* a, double w1, const rdr_cube* b, double w2, const double* y, const double* x, const double* z, int64_t n,
                           double* wet, double* hydro, int loc) {
    if (!c || !a || !b) return fail(c, RDR_ERR_INVALID, "rdr_interp3_blend_cube: NULL argument");
    note_use(c, b);
    if (a->ny != b->ny || a->nx != b->nx || a->nz != b->nz || a->dtype != b->dtype || a->ys != b->ys || a->xs != b->xs || a->zs != b->zs)
        return fail(c, RDR_ERR_INVALID, "rdr_interp3_blend_cube: the two epochs are not on the same grid / dtype");
    if (a->ny > INT32_MAX || a->nx > INT32_MAX || a->nz > INT32_MAX) return fail(c, RDR_ERR_INVALID, "rdr_interp3_blend_cube: axis longer than 2^31");
    BlendSpec B; B.vb = b->d_vals; B.w1 = w1; B.w2 = w2; B.want_pair = true;
    return interp3_impl(c, "rdr_interp3_blend_cube", a, y, x, z, n, 0, nullptr, 0.0, wet, hydro, loc, B);
}

int rdr_interp3_project(rdr_ctx* c, const rdr_cube* q, const double* y, const double* x, const double* z, int64_t n, int proj_mode,
                        const double* proj, double inc0, double* wet, double* hydro, int loc) {
    return interp3_impl(c, "rdr_interp3_project", q, y, x, z, n, proj_mode, proj, inc0, wet, hydro, loc);
}

// _build_cube (delay.py:196-216).  keep == NULL: the public entry (results to the caller's wet / hydro at `loc`); keep != NULL: the results
// stay in the context's scratch (planar (z,y,x), keep[0] = wet, keep[1] = hydro) for rdr_build_cube_to_cube, nothing is downloaded.
static int build_cube_impl(rdr_ctx* c, const rdr_cube* q, const double* xpts, int64_t nx, const double* ypts, int64_t ny,
                           const double* zpts, int64_t nz, double* wet, double* hydro, int loc, double** keep, const double** dev_yxz = nullptr) {
    if (!c || !q || !xpts || !ypts || !zpts || (!keep && (!wet || !hydro))) return fail(c, RDR_ERR_INVALID, "rdr_build_cube: NULL argument");
    note_use(c, q);
    if (nx < 0 || ny < 0 || nz < 0) return fail(c, RDR_ERR_INVALID, "rdr_build_cube: negative count");
    const int64_t n = nx * ny * nz;
    c->last_nan_output = -1;                  // (also for an empty build: the previous call's verdict is not this one's)
    if (n == 0) return RDR_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    const void *dx, *dy, *dz; void *dw, *dh;
    int rc;
    if (loc == RDR_HOST) {                        // the three axes in one slot, one copy
        void* d;
        rc = ensure(c, SLOT_IN0, (size_t)(nx + ny + nz) * 8, &d); if (rc) return rc;
        const HostPart parts[3] = {{xpts, (size_t)nx * 8}, {ypts, (size_t)ny * 8}, {zpts, (size_t)nz * 8}};
        rc = upload_parts(c, d, parts, 3); if (rc) return rc;
        dx = d; dy = static_cast<const double*>(d) + nx; dz = static_cast<const double*>(d) + nx + ny;
    } else { dx = xpts; dy = ypts; dz = zpts; }
    if (dev_yxz) { dev_yxz[0] = (const double*)dy; dev_yxz[1] = (const double*)dx; dev_yxz[2] = (const double*)dz; }
    rc = stage_out(c, SLOT_OUT0, wet, (size_t)n * 8, keep ? RDR_HOST : loc, &dw); if (rc) return rc;
    rc = stage_out(c, SLOT_OUT1, hydro, (size_t)n * 8, keep ? RDR_HOST : loc, &dh); if (rc) return rc;
    // Setup kernel: per-node and per-height records (24 B / 16 B) into scratch; then the gather: 64 x 4-node tiles x chunks of
    // heights - enough chunks that a small grid still fills the chip, as few as possible otherwise.
    const int64_t nodes = nx * ny;
    void *dn, *dl;
    rc = ensure(c, SLOT_IN3, (size_t)nodes * sizeof(BuildNode), &dn); if (rc) return rc;
    rc = ensure(c, SLOT_IN4, (size_t)nz * sizeof(BuildLevel), &dl); if (rc) return rc;
    const int64_t ntile = ((nx + 63) / 64) * ((ny + 3) / 4);
    const int64_t want_tiles = (int64_t)c->num_cus * 8;
    int64_t nchunks = std::min<int64_t>(nz, std::max<int64_t>(1, (want_tiles + ntile - 1) / ntile));
    nchunks = std::max<int64_t>(nchunks, (nz + BUILD_ZCHUNK_MAX - 1) / BUILD_ZCHUNK_MAX);      // the per-height table of a chunk lives in LDS
    if (nchunks > 65535) return fail(c, RDR_ERR_INVALID, "rdr_build_cube: more than 65535 x 1024 heights");
    const int64_t zchunk = (nz + nchunks - 1) / nchunks;
    nchunks = (nz + zchunk - 1) / zchunk;
    const dim3 g((unsigned)std::max<int64_t>(1, std::min<int64_t>(ntile, (int64_t)c->num_cus * 16)), (unsigned)nchunks);
    // staging area of the tile footprint | per-height z weights (8 B) and cells (4 B) | footprint bounds + flag
    const size_t sm = (size_t)BUILD_STAGE_BYTES + (size_t)zchunk * 12 + 32;
    {
        KTimer t(c, 2);
        const int gs = grid_for(nodes + nz, 256, c->num_cus * 8);
        hipError_t e;
        if (q->dtype == RDR_F32) {
            e = launch_lds(build_cube_setup_kernel<float2>, dim3(gs), dim3(256), axes_smem(q), c->stream, make_view<float2>(q), q->proj, (const double*)dx, nx,
                           (const double*)dy, ny, (const double*)dz, nz, (BuildNode*)dn, (BuildLevel*)dl, (int)axes_fit_lds(q));
            if (e == hipSuccess)
                e = launch_lds(build_cube_kernel<float2>, g, dim3(256), sm, c->stream, (const float2*)q->d_vals, (int)q->ny, (int)q->nx, (int)q->nz,
                               (const BuildNode*)dn, (const BuildLevel*)dl, nx, ny, nz, zchunk, (double*)dw, (double*)dh);
        } else {
            e = launch_lds(build_cube_setup_kernel<double2>, dim3(gs), dim3(256), axes_smem(q), c->stream, make_view<double2>(q), q->proj, (const double*)dx, nx,
                           (const double*)dy, ny, (const double*)dz, nz, (BuildNode*)dn, (BuildLevel*)dl, (int)axes_fit_lds(q));
            if (e == hipSuccess)
                e = launch_lds(build_cube_kernel<double2>, g, dim3(256), sm, c->stream, (const double2*)q->d_vals, (int)q->ny, (int)q->nx, (int)q->nz,
                               (const BuildNode*)dn, (const BuildLevel*)dl, nx, ny, nz, zchunk, (double*)dw, (double*)dh);
        }
        if (e != hipSuccess) return fail(c, RDR_ERR_HIP, std::string("build_cube_kernel launch: ") + hipGetErrorString(e));
    }
    if (keep) { keep[0] = (double*)dw; keep[1] = (double*)dh; return RDR_OK; }
    // np.isnan(result).any() (delay.py:187) answered here, before the outputs leave the device (rdr_last_nan_output)
    int* const nf = c->d_flags + MAX_SLICES + 1;
    if (loc == RDR_HOST) {
        HIPCHECK(c, hipMemsetAsync(nf, 0, sizeof(int), c->stream));
        hipLaunchKernelGGL(nan_scan_kernel, dim3(grid_for(n, 256, c->num_cus * 8)), dim3(256), 0, c->stream, (const double*)dw, (const double*)dh, n, n, nf);
        HIPCHECK(c, hipGetLastError());
    }
    rc = finish_out(c, wet, dw, (size_t)n * 8, loc); if (rc) return rc;
    rc = finish_out(c, hydro, dh, (size_t)n * 8, loc); if (rc) return rc;
    if (loc == RDR_HOST) {
        int f = 0;
        HIPCHECK(c, hipMemcpyAsync(&f, nf, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHECK(c, hipStreamSynchronize(c->stream));
        c->last_nan_output = (f & 64) ? 1 : 0;
    }
    return RDR_OK;
}

int rdr_build_cube(rdr_ctx* c, const rdr_cube* q, const double* xpts, int64_t nx, const double* ypts, int64_t ny,
                   const double* zpts, int64_t nz, double* wet, double* hydro, int loc) {
    return build_cube_impl(c, q, xpts, nx, ypts, ny, zpts, nz, wet, hydro, loc, nullptr);
}

int rdr_last_nan_output(rdr_ctx* c) { return c ? c->last_nan_output : -1; }

// host copy of an axis given at `loc` (the cube keeps its axes on the host as well)
static int axis_to_host(rdr_ctx* c, const double* a, int64_t n, int loc, std::vector<double>& out) {
    out.resize((size_t)std::max<int64_t>(n, 0));
    if (n <= 0) return RDR_OK;
    if (loc == RDR_DEVICE) {
        HIPCHECK(c, hipMemcpyAsync(out.data(), a, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHECK(c, hipStreamSynchronize(c->stream));
    } else std::copy(a, a + n, out.begin());
    return RDR_OK;
}

int rdr_build_cube_to_cube(rdr_ctx* c, const rdr_cube* q, const double* xpts, int64_t nx, const double* ypts, int64_t ny,
                           const double* zpts, int64_t nz, int loc, rdr_cube** out) {
    if (!c || !q || !xpts || !ypts || !zpts || !out) return fail(c, RDR_ERR_INVALID, "rdr_build_cube_to_cube: NULL argument");
    if (nx < 2 || ny < 2 || nz < 2) return fail(c, RDR_ERR_INVALID, "rdr_build_cube_to_cube: the delay cube needs two nodes per axis");
    HIPCHECK(c, hipSetDevice(c->device));
    std::vector<double> hx, hy, hz;
    int rc = axis_to_host(c, xpts, nx, loc, hx); if (rc) return rc;
    rc = axis_to_host(c, ypts, ny, loc, hy); if (rc) return rc;
    rc = axis_to_host(c, zpts, nz, loc, hz); if (rc) return rc;
    double* planar[2] = {nullptr, nullptr};
    const double* dev_yxz[3] = {nullptr, nullptr, nullptr};       // the axes as the build has them on the device (scratch slot / the caller's arrays)
    rc = build_cube_impl(c, q, xpts, nx, ypts, ny, zpts, nz, nullptr, nullptr, loc, planar, dev_yxz); if (rc) return rc;
    // planar (z,y,x) results -> interleaved (y,x,z) cube with axes (ypts, xpts, zpts): what getInterpolators(ds, 'ztd') builds from the
    // Dataset of writeResultsToXarray (delay.py:113, delayFcns.py:40-41), descending axes flipped as scipy does; its NaN scan
    // (delayFcns.py:50-52 on this cube == np.isnan(result).any(), delay.py:187) comes with the packing
    return cube_create_impl(c, hy.data(), ny, hx.data(), nx, hz.data(), nz, planar[0], planar[1], RDR_F64, nx, 1, ny * nx, RDR_DEVICE, out, dev_yxz);
}


// tropo_delay's point branch for a zenith / projected line of sight (delay.py:96-128) in ONE call: _build_cube on the output grid
// (xpts, ypts, zpts) -> the intermediate cube packed as getInterpolators(ds, 'ztd') would wrap it -> trilinear gather at the query
// points -> delay / cos(inc).  Everything is enqueued before anything is waited for: the points travel up (copy stream) while the
// intermediate cube is built (ctx stream), the cube lives in the context's scratch (no allocation per call), its NaN verdict
// (delay.py:187) comes back with the final synchronisation.  Same kernels, same arithmetic as the separate entries: same bits.
int rdr_point_delays(rdr_ctx* c, const rdr_cube* q, const double* xpts, int64_t nx, const double* ypts, int64_t ny, const double* zpts, int64_t nz,
                     const double* y, const double* x, const double* z, int64_t n, int proj_mode, const double* proj, double inc0,
                     double* wet, double* hydro, int32_t* cube_has_nan) {
    if (!c || !q || !xpts || !ypts || !zpts) return fail(c, RDR_ERR_INVALID, "rdr_point_delays: NULL argument");
    if (nx < 2 || ny < 2 || nz < 2) return fail(c, RDR_ERR_INVALID, "rdr_point_delays: the delay cube needs two nodes per axis");
    if (nz > MAX_LEVELS) return fail(c, RDR_ERR_INVALID, "rdr_point_delays: more than 512 height levels");
    if (ny + nx + nz > 100000) return fail(c, RDR_ERR_INVALID, "rdr_point_delays: axes too long");
    int rc = point_query_args(c, "rdr_point_delays", y, x, z, n, proj_mode, proj, wet, hydro); if (rc) return rc;
    int fy, fx, fz;
    if (axis_check(ypts, ny, &fy) || axis_check(xpts, nx, &fx) || axis_check(zpts, nz, &fz))
        return fail(c, RDR_ERR_INVALID, "The points in each dimension must be strictly ascending or descending (and >= 2)");
    HIPCHECK(c, hipSetDevice(c->device));
    // the intermediate cube: a scratch object (values and axes in context slots), never destroyed
    rdr_cube tmp;
    tmp.ctx = c; tmp.ny = ny; tmp.nx = nx; tmp.nz = nz; tmp.dtype = RDR_F64;
    tmp.ys.assign(ypts, ypts + ny); tmp.xs.assign(xpts, xpts + nx); tmp.zs.assign(zpts, zpts + nz);
    if (fy) std::reverse(tmp.ys.begin(), tmp.ys.end());
    if (fx) std::reverse(tmp.xs.begin(), tmp.xs.end());
    if (fz) std::reverse(tmp.zs.begin(), tmp.zs.end());
    axis_uniformity(tmp.ys, &tmp.uni[0], &tmp.inv_d[0], &tmp.exact[0]);
    axis_uniformity(tmp.xs, &tmp.uni[1], &tmp.inv_d[1], &tmp.exact[1]);
    axis_uniformity(tmp.zs, &tmp.uni[2], &tmp.inv_d[2]);
    const size_t total = (size_t)ny * nx * nz;
    void *dvals, *daxes;
    rc = ensure(c, SLOT_TMPCUBE, total * sizeof(double2), &dvals); if (rc) return rc;
    rc = ensure(c, SLOT_TMPAXES, (size_t)(ny + nx + nz) * 8, &daxes); if (rc) return rc;
    tmp.d_vals = dvals; tmp.d_axes = (double*)daxes;
    std::vector<double> ax;
    ax.insert(ax.end(), tmp.ys.begin(), tmp.ys.end()); ax.insert(ax.end(), tmp.xs.begin(), tmp.xs.end()); ax.insert(ax.end(), tmp.zs.begin(), tmp.zs.end());
    StreamsQuiesce quiesce(c);                  // (`ax`, the caller's points and outputs: nothing may still be copying when an error returns)
    HIPCHECK(c, hipMemcpyAsync(daxes, ax.data(), ax.size() * 8, hipMemcpyHostToDevice, c->stream));
    double* planar[2] = {nullptr, nullptr};
    rc = build_cube_impl(c, q, xpts, nx, ypts, ny, zpts, nz, nullptr, nullptr, RDR_HOST, planar); if (rc) return rc;
    int* const nf = c->d_flags + MAX_SLICES + 2;
    HIPCHECK(c, hipMemsetAsync(nf, 0, sizeof(int), c->stream));
    if (nx >= 8 && nz >= 8)        // planar (z, y, x) results: x is contiguous - the LDS-transposing packer (cube_kernels.h)
        hipLaunchKernelGGL((pack_cube_xfast_kernel<double, double2, false>), dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(ny * ((nx + 31) / 32) * ((nz + 31) / 32), (int64_t)c->num_cus * 16))),
                           dim3(256), 0, c->stream, (const double*)planar[0], (const double*)planar[1], (double2*)dvals, ny, nx, nz, nx, ny * nx, fy, fx, fz, nf);
    else
        hipLaunchKernelGGL((pack_cube_kernel<double, double2, false>), dim3(grid_for((int64_t)total, 256, c->num_cus * 8)), dim3(256), 0, c->stream,
                           (const double*)planar[0], (const double*)planar[1], (double2*)dvals, ny, nx, nz, nx, (int64_t)1, ny * nx, fy, fx, fz, nf);
    HIPCHECK(c, hipGetLastError());
    // (into a page-locked word: a pageable destination would stall the host here until the cube is built - and the upload of the
    // points, which is to run UNDER that build, with it)
    c->h_word[0] = 0;
    HIPCHECK(c, hipMemcpyAsync(&c->h_word[0], nf, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (n > 0) {
        PointQuery Q; std::memset(&Q, 0, sizeof(Q));
        Q.pmode = proj_mode; Q.inc0 = inc0;
        static const int slots[6] = {SLOT_PT0, SLOT_PT1, SLOT_PT2, SLOT_PT3, SLOT_PT4, SLOT_PT5};
        rc = interp_pipeline(c, "rdr_point_delays", &tmp, false, y, x, z, n, Q, proj, wet, hydro, slots); if (rc) return rc;
    } else HIPCHECK(c, hipStreamSynchronize(c->stream));
    quiesce.armed = false;                      // (both branches have synchronised)
    if (cube_has_nan) *cube_has_nan = c->h_word[0] != 0;
    // a large job's intermediates (16 B per cell of the cube + 16 B per cell of planar results) are not kept for the life of the context
    static const size_t keep = []() { const char* e = std::getenv("RAIDER_HIP_SCRATCH_KEEP_BYTES"); return e ? (size_t)std::strtoull(e, nullptr, 10) : (size_t)4 << 30; }();
    for (int s_ : {SLOT_TMPCUBE, SLOT_OUT0, SLOT_OUT1}) {
        DevBuf& b = c->slot[s_];
        if (b.p && b.cap > keep) { (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
    }
    return RDR_OK;
}

static int project_impl(rdr_ctx* c, const char* who, double* wet, double* hydro, int pmode, const double* proj, int64_t n, int loc) {
    if (!c || (!wet && !hydro) || !proj) return fail(c, RDR_ERR_INVALID, std::string(who) + ": NULL argument");
    if (n < 0) return fail(c, RDR_ERR_INVALID, std::string(who) + ": negative count");
    if (n == 0) return RDR_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    const void *di, *dwi = nullptr, *dhi = nullptr;
    int rc = stage_in(c, SLOT_IN0, proj, (size_t)n * 8, loc, &di); if (rc) return rc;
    if (wet) { rc = stage_in(c, SLOT_OUT0, wet, (size_t)n * 8, loc, &dwi); if (rc) return rc; }
    if (hydro) { rc = stage_in(c, SLOT_OUT1, hydro, (size_t)n * 8, loc, &dhi); if (rc) return rc; }
    hipLaunchKernelGGL(project_kernel, dim3(grid_for(n, 256, c->num_cus * 8)), dim3(256), 0, c->stream, (double*)dwi, (double*)dhi, pmode,
                       (const double*)di, 0.0, n);
    HIPCHECK(c, hipGetLastError());
    rc = finish_out(c, wet, dwi, (size_t)n * 8, loc); if (rc) return rc;
    rc = finish_out(c, hydro, dhi, (size_t)n * 8, loc); if (rc) return rc;
    if (loc == RDR_HOST) HIPCHECK(c, hipStreamSynchronize(c->stream));
    return RDR_OK;
}

int rdr_project_cosinc(rdr_ctx* c, double* wet, double* hydro, const double* inc, int64_t n, int loc) {
    return project_impl(c, "rdr_project_cosinc", wet, hydro, 1, inc, n, loc);
}

int rdr_project_divide(rdr_ctx* c, double* wet, double* hydro, const double* divisor, int64_t n, int loc) {
    return project_impl(c, "rdr_project_divide", wet, hydro, 3, divisor, n, loc);
}

// ---- rays -------------------------------------------------------------------------------------------
static int levels_host(const std::vector<double>& zs, double ht, double zref, std::vector<double>& lo, std::vector<double>& hi,
                       std::vector<int>& kz) {
    const int nz = (int)zs.size();
    const double ztop = zs[nz - 1];
    for (int zz = 0; zz < nz - 1; ++zz) {
        double l = zs[zz], h = zs[zz + 1];
        if (h == ztop) h -= 0.01;
        if (h < ht || l >= zref) continue;
        if (l < ht) l = ht;
        if (h > zref) h = zref;
        if (std::fabs(h - l) < 1.0) continue;
        lo.push_back(l); hi.push_back(h); kz.push_back(zz);
    }
    return (int)lo.size();
}

int rdr_ray_levels(const rdr_cube* q, double ht, double zref, int32_t* K, double* lo, double* hi, int32_t* kz) {
    if (!q || !K) return fail(nullptr, RDR_ERR_INVALID, "rdr_ray_levels: NULL argument");
    std::vector<double> l, h; std::vector<int> z;
    *K = levels_host(q->zs, ht, zref, l, h, z);
    if (lo) std::copy(l.begin(), l.end(), lo);
    if (hi) std::copy(h.begin(), h.end(), hi);
    if (kz) std::copy(z.begin(), z.end(), kz);
    if (*K == 0) return fail(q->ctx, RDR_ERR_NO_LEVELS, "no weather-model interval contributes to the ray integral (build_ray -> None)");
    return RDR_OK;
}

int rdr_nparts(const double* maxlen, int32_t K, double max_seg, int32_t* nparts) {
    if (!maxlen || !nparts) return fail(nullptr, RDR_ERR_INVALID, "rdr_nparts: NULL argument");
    if (K < 0 || !(max_seg > 0.0)) return fail(nullptr, RDR_ERR_INVALID, "rdr_nparts: K >= 0 and MAX_SEGMENT_LENGTH > 0");
    for (int k = 0; k < K; ++k) {
        const double parts = std::ceil(maxlen[k] / max_seg) + 1;   // delay.py:283
        nparts[k] = (parts >= 1 && parts <= 2147483647.0) ? (int32_t)parts : 2147483647;
    }
    return RDR_OK;
}

static int check_rays(rdr_ctx* c, const rdr_rays* r) {
    if (!r) return fail(c, RDR_ERR_INVALID, "rays is NULL");
    if (r->n < 0) return fail(c, RDR_ERR_INVALID, "rays->n negative");
    if (r->origin_mode == RDR_ORIGIN_GRID) {
        if (!r->xpts || !r->ypts || r->nx * r->ny != r->n) return fail(c, RDR_ERR_INVALID, "GRID rays need xpts, ypts and n == nx*ny");
    } else if (r->origin_mode == RDR_ORIGIN_LLH) {
        if (!r->lat || !r->lon) return fail(c, RDR_ERR_INVALID, "LLH rays need lat and lon");
    } else if (r->origin_mode == RDR_ORIGIN_XYZ) {
        if (!r->xyz) return fail(c, RDR_ERR_INVALID, "XYZ rays need xyz");
        if (r->los_mode != RDR_LOS_VEC && (!r->lat || !r->lon)) return fail(c, RDR_ERR_INVALID, "XYZ rays with inc/heading or zenith LOS need lat and lon too");
    } else return fail(c, RDR_ERR_INVALID, "unknown origin_mode");
    if (r->los_mode == RDR_LOS_VEC) { if (!r->los) return fail(c, RDR_ERR_INVALID, "LOS_VEC needs los"); }
    else if (r->los_mode == RDR_LOS_INC_HD) { if (!r->inc) return fail(c, RDR_ERR_INVALID, "LOS_INC_HD needs inc (and hd, or hd0 for one heading)"); }
    else if (r->los_mode != RDR_LOS_INC_HD_SCALAR && r->los_mode != RDR_LOS_ZENITH) return fail(c, RDR_ERR_INVALID, "unknown los_mode");
    return RDR_OK;
}

// stage every ray array the mode needs; fills P (one slice; los_mult > 1: the look-vector / incidence / heading arrays hold
// los_mult consecutive blocks of n rays, one per height slice)
static int stage_rays(rdr_ctx* c, const rdr_rays* r, RayParams& P, int64_t los_mult = 1) {
    std::memset(&P, 0, sizeof(P));
    P.n = r->n; P.origin_mode = r->origin_mode; P.los_mode = r->los_mode; P.nx = r->nx; P.ny = r->ny;
    P.inc0 = r->inc0; P.hd0 = r->hd0;
    const int loc = r->loc;
    const void* d;
    int rc;
    if (r->origin_mode == RDR_ORIGIN_GRID) {
        rc = stage_in(c, SLOT_IN0, r->xpts, (size_t)r->nx * 8, loc, &d); if (rc) return rc; P.xpts = (const double*)d;
        rc = stage_in(c, SLOT_IN1, r->ypts, (size_t)r->ny * 8, loc, &d); if (rc) return rc; P.ypts = (const double*)d;
    } else {
        rc = stage_in(c, SLOT_IN0, r->lat, (size_t)r->n * 8, loc, &d); if (rc) return rc; P.lat = (const double*)d;
        rc = stage_in(c, SLOT_IN1, r->lon, (size_t)r->n * 8, loc, &d); if (rc) return rc; P.lon = (const double*)d;
        if (r->origin_mode == RDR_ORIGIN_XYZ) { rc = stage_in(c, SLOT_IN2, r->xyz, (size_t)r->n * 24, loc, &d); if (rc) return rc; P.xyz = (const double*)d; }
    }
    const size_t nl = (size_t)r->n * (size_t)los_mult;
    if (r->los_mode == RDR_LOS_VEC) { rc = stage_in(c, SLOT_IN3, r->los, nl * 24, loc, &d); if (rc) return rc; P.los = (const double*)d; }
    else if (r->los_mode == RDR_LOS_INC_HD) {
        rc = stage_in(c, SLOT_IN4, r->inc, nl * 8, loc, &d); if (rc) return rc; P.inc = (const double*)d;
        if (r->hd) { rc = stage_in(c, SLOT_IN5, r->hd, nl * 8, loc, &d); if (rc) return rc; P.hd = (const double*)d; }   // NULL: hd0 for every ray
    }
    if (r->hts) {                                           // per-ray origin heights (one batch = one slice)
        if (los_mult != 1) return fail(c, RDR_ERR_INVALID, "per-ray heights (rays->hts) and height slices exclude each other");
        rc = stage_in(c, SLOT_IN6, r->hts, (size_t)r->n * 8, loc, &d); if (rc) return rc; P.ht_ray = (const double*)d;
    }
    if (r->origin_mode == RDR_ORIGIN_GRID) {
        P.tiles_x = (int)((r->nx + TILE - 1) / TILE);
        P.ntiles = (int64_t)P.tiles_x * ((r->ny + TILE - 1) / TILE);
    } else {
        P.tiles_x = 1;
        P.ntiles = (r->n + BLOCK - 1) / BLOCK;
    }
    P.nslices = 1; P.tiles_per_slice = std::max<int64_t>(P.ntiles, 1); P.hts = nullptr; P.los_stride = 0;
    P.maxlen_bits = c->d_maxlen; P.flags = c->d_flags;
    P.ws = nullptr; P.nslots = 0; P.tile_begin = 0; P.tile_count = P.ntiles; P.nslow = c->d_nslow;
    return RDR_OK;
}

static size_t ray_smem(const rdr_cube* q) {
    return ray_smem_bytes(q->ny, q->nx, q->nz, q->exact[0], q->exact[1]);
}
// dynamic + the static LDS of the ray kernels (march_kernel's staging block for f64 cubes: 4 waves x 48 x 16 B): what must fit the CU
static size_t ray_lds_total(const rdr_cube* q) { return ray_smem(q) + (q->dtype == RDR_F64 ? (size_t)(BLOCK / 64) * 48 * 16 : 16); }

// Persistent grid of `per_cu` workgroups per CU; tiles are handed out dynamically (TileWalk: one device atomic per tile and
// XCD band), so the grid only has to cover the resident workgroups (3-4 per CU) with a little slack.  Measured on the bench
// scene (tools/probe_passes.py, RAIDER_HIP_BLOCKS_PER_CU = 4 .. 24): 4-8 per CU are equal (march 6.52 ms), 24 costs 2 % in
// per-workgroup table set-up.  (With a static equal share per workgroup the same sweep needed 24 per CU to hide the
// imbalance and was still 2-7 % slower.)
static int ray_grid(rdr_ctx* c, int64_t ntiles, int per_cu) {
    static const int forced = []() { const char* e = std::getenv("RAIDER_HIP_BLOCKS_PER_CU"); return e ? std::atoi(e) : 0; }();
    if (forced > 0) per_cu = forced;
    int64_t g = std::min<int64_t>(ntiles, (int64_t)c->num_cus * per_cu);
    g = std::max<int64_t>(8, (g + 7) / 8 * 8);   // multiple of 8 (one share per XCD)
    return (int)g;
}

// Workspace = ray records (WS_NFIELDS doubles per ray slot, every ray) + the side buffer (K+1 crossings per GENERIC ray).
static size_t ws_tile_bytes() { return (size_t)WS_NFIELDS * BLOCK * sizeof(double); }

// Side-buffer columns for a launch of `slots` ray slots: rays the static classification rejects are rare (none at all on most
// scenes), so 1/32 of the batch is provisioned up front; when the host has seen a larger count on this context (dateline or polar
// scenes: every slice of a tropo_delay call repeats it) that count is used.  Rays beyond the capacity are still integrated
// correctly - pass 2 recomputes their crossings.
static int64_t side_columns(rdr_ctx* c, int64_t slots, int K, size_t budget_left) {
    if (c->side_forced >= 0) return std::min<int64_t>(c->side_forced, slots);
    int64_t want = std::max<int64_t>(65536, slots / 32);
    if (c->last_nslow > 0) want = std::max<int64_t>(want, c->last_nslow + c->last_nslow / 16 + 1024);
    want = std::min<int64_t>(want, slots);
    const size_t col = (size_t)(K + 1) * sizeof(double);
    want = std::min<int64_t>(want, (int64_t)(budget_left / col));
    return std::max<int64_t>(want, 0);
}

static size_t ws_budget(rdr_ctx* c) {
    size_t free_b = 0, total_b = 0;
    size_t budget = c->ws_limit;
    const size_t held = c->ws.cap + c->side.cap;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min(budget, std::max(held, free_b / 2 + held));
    return budget;
}

// Largest number of tiles whose records (plus a 1/32 side buffer) fit the workspace limit / half of the free device memory.
static int64_t ws_chunk_tiles(rdr_ctx* c, int K) {
    const size_t per_tile = ws_tile_bytes() + (size_t)(K + 1) * sizeof(double) * BLOCK / 32;
    return (int64_t)std::max<size_t>(1, ws_budget(c) / per_tile);
}

// Reserve records for `tiles` tiles and point P at them (records, side buffer, side counter).
static int ws_reserve(rdr_ctx* c, int64_t tiles, int K, RayParams& P) {
    const size_t need = (size_t)tiles * ws_tile_bytes();
    if (c->ws.cap < need) {
        if (c->ws.p) { HIPCHECK(c, hipStreamSynchronize(c->stream)); HIPCHECK(c, hipFree(c->ws.p)); c->ws.p = nullptr; c->ws.cap = 0; }
        { const int rc_ = dev_malloc(c, &c->ws.p, need); if (rc_) return rc_; }
        c->ws.cap = need;
    }
    const size_t budget = ws_budget(c);
    const int64_t cols = side_columns(c, tiles * BLOCK, K, budget > need ? budget - need : 0);
    const size_t sneed = (size_t)cols * (K + 1) * sizeof(double);
    if (c->side.cap < sneed) {
        if (c->side.p) { HIPCHECK(c, hipStreamSynchronize(c->stream)); HIPCHECK(c, hipFree(c->side.p)); c->side.p = nullptr; c->side.cap = 0; }
        { const int rc_ = dev_malloc(c, &c->side.p, sneed); if (rc_) return rc_; }
        c->side.cap = sneed;
    }
    c->side_cap = cols;
    P.ws = (double*)c->ws.p;
    P.side = cols > 0 ? (double*)c->side.p : nullptr; P.side_cap = cols; P.side_ctr = c->d_sidectr;
    return RDR_OK;
}

// the records written by the last pass 1 (wsig_match): same layout, nothing re-reserved
static void ws_attach(rdr_ctx* c, RayParams& P) {
    P.ws = (double*)c->ws.p;
    P.side = c->side_cap > 0 ? (double*)c->side.p : nullptr; P.side_cap = c->side_cap; P.side_ctr = c->d_sidectr;
}

static void wsig_set(rdr_ctx* c, const rdr_cube* q, const rdr_rays* r, double ht, double zref, int K, bool valid) {
    c->wsig.cube = q; c->wsig.vals = q->d_vals; c->wsig.proj = q->proj; c->wsig.n = r->n; c->wsig.ht = ht; c->wsig.zref = zref; c->wsig.K = K; c->wsig.valid = valid;
    c->wsig.a = r->origin_mode == RDR_ORIGIN_GRID ? (const void*)r->xpts : (r->origin_mode == RDR_ORIGIN_XYZ ? (const void*)r->xyz : (const void*)r->lat);
    c->wsig.b = r->origin_mode == RDR_ORIGIN_GRID ? (const void*)r->ypts : (const void*)r->lon;
    c->wsig.c = r->los_mode == RDR_LOS_VEC ? (const void*)r->los : (const void*)r->inc;
    c->wsig.d = (const void*)r->hts;
}

static bool wsig_match(rdr_ctx* c, const rdr_cube* q, const rdr_rays* r, double ht, double zref, int K) {
    if (!c->wsig.valid || c->wsig.cube != q || c->wsig.vals != q->d_vals || std::memcmp(&c->wsig.proj, &q->proj, sizeof(LccParams)) != 0 || c->wsig.n != r->n || c->wsig.ht != ht || c->wsig.zref != zref || c->wsig.K != K) return false;
    if (r->loc != RDR_DEVICE) return false;   // host arrays may have been rewritten in place between the two calls
    const void* a = r->origin_mode == RDR_ORIGIN_GRID ? (const void*)r->xpts : (r->origin_mode == RDR_ORIGIN_XYZ ? (const void*)r->xyz : (const void*)r->lat);
    const void* b = r->origin_mode == RDR_ORIGIN_GRID ? (const void*)r->ypts : (const void*)r->lon;
    const void* cc = r->los_mode == RDR_LOS_VEC ? (const void*)r->los : (const void*)r->inc;
    return a == c->wsig.a && b == c->wsig.b && cc == c->wsig.c && (const void*)r->hts == c->wsig.d;
}

// pass 1 over tiles [tb, tb+tc): optional reduction (P.maxlen_bits != null) and/or record store (P.ws != null)
// nslots_total > 0: the launch is one chunk of a larger record buffer (field stride nslots_total, P.ws already offset to the
// chunk's first slot) and the slow-ray count accumulates over the chunks (reset_nslow only for the first one).
static int launch_crossings(rdr_ctx* c, const rdr_cube* q, RayParams P, int64_t tb, int64_t tc, int64_t nslots_total = 0, bool reset_nslow = true) {
    P.tile_begin = tb; P.tile_count = tc; P.nslots = nslots_total > 0 ? nslots_total : tc * BLOCK;
    const int g = ray_grid(c, tc, 8);
    if (ray_lds_total(q) > c->lds_max)
        return fail(c, RDR_ERR_INVALID, "ray tracing: the cube's non-uniform horizontal axes and level tables need " + std::to_string(ray_lds_total(q)) +
                    " B of LDS per workgroup, the device offers " + std::to_string(c->lds_max) + " (resample the cube to uniform axes or crop it)");
    if (reset_nslow) {
        HIPCHECK(c, hipMemsetAsync(c->d_nslow, 0, sizeof(int), c->stream));
        HIPCHECK(c, hipMemsetAsync(c->d_sidectr, 0, sizeof(int), c->stream));
    }
    HIPCHECK(c, hipMemsetAsync(c->d_tilectr, 0, 16 * sizeof(int), c->stream));
    P.tile_ctr = c->d_tilectr;
    const size_t sm = ray_smem(q);
    hipError_t e = hipSuccess;
    {
        KTimer t(c, 0);
        const bool lcc = q->proj.kind == 1;
        // input form of the batch, fixed at compile time for the two hot ones (crossings_kernel's OM parameter)
        const int om = P.origin_mode != RDR_ORIGIN_GRID ? 0 : (P.los_mode == RDR_LOS_VEC ? 1 : 2);
        const dim3 G(g), B(BLOCK);
#define RDR_LAUNCH_X(T2, LCC_, OM_) e = (P.ht_ray ? launch_lds(crossings_kernel<T2, false, LCC_, OM_, true>, G, B, sm, c->stream, make_view<T2>(q), P, q->proj) \
                                                  : launch_lds(crossings_kernel<T2, false, LCC_, OM_, false>, G, B, sm, c->stream, make_view<T2>(q), P, q->proj))
#define RDR_LAUNCH_X_OM(T2, LCC_) do { if (om == 1) RDR_LAUNCH_X(T2, LCC_, 1); else if (om == 2) RDR_LAUNCH_X(T2, LCC_, 2); else RDR_LAUNCH_X(T2, LCC_, 0); } while (0)
        if (q->dtype == RDR_F32) { if (lcc) RDR_LAUNCH_X_OM(float2, true); else RDR_LAUNCH_X_OM(float2, false); }
        else { if (lcc) RDR_LAUNCH_X_OM(double2, true); else RDR_LAUNCH_X_OM(double2, false); }
#undef RDR_LAUNCH_X_OM
#undef RDR_LAUNCH_X
    }
    if (e != hipSuccess) return fail(c, RDR_ERR_HIP, std::string("crossings_kernel launch: ") + hipGetErrorString(e));
    // generic-geodesy mop-up of the rays the classification rejected (returns at once when there are none)
    P.tile_ctr = c->d_tilectr + 8;
    if (q->dtype == RDR_F32) e = launch_lds(crossings_kernel<float2, true>, dim3(g), dim3(BLOCK), sm, c->stream, make_view<float2>(q), P, q->proj);
    else e = launch_lds(crossings_kernel<double2, true>, dim3(g), dim3(BLOCK), sm, c->stream, make_view<double2>(q), P, q->proj);
    if (e != hipSuccess) return fail(c, RDR_ERR_HIP, std::string("crossings_kernel launch: ") + hipGetErrorString(e));
    return RDR_OK;
}

static int launch_march(rdr_ctx* c, const rdr_cube* q, RayParams P, int64_t tb, int64_t tc, int64_t nslots_total = 0) {
    P.tile_begin = tb; P.tile_count = tc; P.nslots = nslots_total > 0 ? nslots_total : tc * BLOCK;
    static const int stage_f64 = []() { const char* e = std::getenv("RAIDER_HIP_F64_STAGE"); return e ? std::atoi(e) : 1; }();
    P.stage_f64 = stage_f64;
    const int g = ray_grid(c, tc, 8);
    HIPCHECK(c, hipMemsetAsync(c->d_tilectr + 16, 0, 16 * sizeof(int), c->stream));
    P.tile_ctr = c->d_tilectr + 16;
    const size_t sm = ray_smem(q);
    const dim3 G(g), B(BLOCK);
    hipError_t e = hipSuccess;
    {
        KTimer t(c, 1);
        const auto v32 = make_view<float2>(q);
        const bool small = v32.small && (q->dtype == RDR_F32 || make_view<double2>(q).small);
        const int grid = !small ? 0 : (q->exact[0] && q->exact[1]) ? 1 : (!q->exact[0] && !q->exact[1] && q->uni[0] && q->uni[1]) ? 2 : 0;
#define RDR_LAUNCH_M(T2, V) (P.ht_ray ? (grid == 1 ? launch_lds(march_kernel<T2, false, 1, true>, G, B, sm, c->stream, V, P, q->proj)   \
                                                    : launch_lds(march_kernel<T2, false, 0, true>, G, B, sm, c->stream, V, P, q->proj)) \
                             : grid == 1 ? launch_lds(march_kernel<T2, false, 1>, G, B, sm, c->stream, V, P, q->proj)   \
                             : grid == 2 ? launch_lds(march_kernel<T2, false, 2>, G, B, sm, c->stream, V, P, q->proj) \
                                         : launch_lds(march_kernel<T2, false, 0>, G, B, sm, c->stream, V, P, q->proj))
        if (q->dtype == RDR_F32) e = RDR_LAUNCH_M(float2, v32);
        else e = RDR_LAUNCH_M(double2, make_view<double2>(q));
#undef RDR_LAUNCH_M
    }
    if (e != hipSuccess) return fail(c, RDR_ERR_HIP, std::string("march_kernel launch: ") + hipGetErrorString(e));
    P.tile_ctr = c->d_tilectr + 24;
    if (q->dtype == RDR_F32) e = launch_lds(march_kernel<float2, true>, G, B, sm, c->stream, make_view<float2>(q), P, q->proj);
    else e = launch_lds(march_kernel<double2, true>, G, B, sm, c->stream, make_view<double2>(q), P, q->proj);
    if (e != hipSuccess) return fail(c, RDR_ERR_HIP, std::string("march_kernel launch: ") + hipGetErrorString(e));
    return RDR_OK;
}

// Registers / LDS / scratch of the light ray kernel a GRID + look-vector batch on this cube launches, read from the LOADED code
// object (hipFuncGetAttributes) - what bench.py prints, instead of a profiler's metadata column.
int rdr_ray_kernel_attributes(rdr_ctx* c, const rdr_cube* q, int which, int32_t* vgprs, int32_t* static_lds, int32_t* dynamic_lds,
                              int32_t* scratch, int32_t* max_threads) {
    if (!c || !q || which < 0 || which > 3) return fail(c, RDR_ERR_INVALID, "rdr_ray_kernel_attributes: bad argument");
    const void* fn = nullptr;
    const bool lcc = q->proj.kind == 1;
    const bool pr = which >= 2;                    // 2 / 3: the per-ray-height instantiations of pass 1 / pass 2
    if ((which & 1) == 0) {
        if (q->dtype == RDR_F32) fn = pr ? (lcc ? (const void*)crossings_kernel<float2, false, true, 1, true> : (const void*)crossings_kernel<float2, false, false, 1, true>)
                                         : (lcc ? (const void*)crossings_kernel<float2, false, true, 1> : (const void*)crossings_kernel<float2, false, false, 1>);
        else fn = pr ? (lcc ? (const void*)crossings_kernel<double2, false, true, 1, true> : (const void*)crossings_kernel<double2, false, false, 1, true>)
                     : (lcc ? (const void*)crossings_kernel<double2, false, true, 1> : (const void*)crossings_kernel<double2, false, false, 1>);
    } else {
        const auto v32 = make_view<float2>(q);
        const bool small = v32.small && (q->dtype == RDR_F32 || make_view<double2>(q).small);
        const int grid = !small ? 0 : (q->exact[0] && q->exact[1]) ? 1 : (!q->exact[0] && !q->exact[1] && q->uni[0] && q->uni[1]) ? 2 : 0;
        if (pr) {
            if (q->dtype == RDR_F32) fn = grid == 1 ? (const void*)march_kernel<float2, false, 1, true> : (const void*)march_kernel<float2, false, 0, true>;
            else fn = grid == 1 ? (const void*)march_kernel<double2, false, 1, true> : (const void*)march_kernel<double2, false, 0, true>;
        } else if (q->dtype == RDR_F32) fn = grid == 1 ? (const void*)march_kernel<float2, false, 1> : grid == 2 ? (const void*)march_kernel<float2, false, 2> : (const void*)march_kernel<float2, false, 0>;
        else fn = grid == 1 ? (const void*)march_kernel<double2, false, 1> : grid == 2 ? (const void*)march_kernel<double2, false, 2> : (const void*)march_kernel<double2, false, 0>;
    }
    hipFuncAttributes a;
    HIPCHECK(c, hipFuncGetAttributes(&a, fn));
    if (vgprs) *vgprs = a.numRegs;
    if (static_lds) *static_lds = (int32_t)a.sharedSizeBytes;
    if (dynamic_lds) *dynamic_lds = (int32_t)ray_smem(q);
    if (scratch) *scratch = (int32_t)a.localSizeBytes;
    if (max_threads) *max_threads = a.maxThreadsPerBlock;
    return RDR_OK;
}

// pass 2 for the whole batch when no valid records are around: chunked (pass-1-store, pass-2) pairs
static int march_chunked(rdr_ctx* c, const rdr_cube* q, const RayParams& P0, int K) {
    const int64_t chunk = std::min<int64_t>(P0.ntiles, ws_chunk_tiles(c, K));
    RayParams Pw = P0;
    int rc = ws_reserve(c, chunk, K, Pw); if (rc) return rc;
    for (int64_t tb = 0; tb < P0.ntiles; tb += chunk) {
        const int64_t tc = std::min<int64_t>(chunk, P0.ntiles - tb);
        RayParams P = Pw;
        unsigned long long* keep = P.maxlen_bits;
        P.maxlen_bits = nullptr;                      // store only, no reduction
        rc = launch_crossings(c, q, P, tb, tc); if (rc) return rc;
        P.maxlen_bits = keep;
        rc = launch_march(c, q, P, tb, tc); if (rc) return rc;
    }
    return RDR_OK;
}

// Host-buffer ray tracing with the PCIe transfers overlapped: the look vectors (24 B per ray, the bulk of the input) go up in
// row chunks on the copy stream while pass 1 runs on the chunks that have arrived; after the last chunk the slice-level
// partition is complete, pass 2 runs chunk by chunk and each chunk's outputs come down while the next is integrated.
// (hipMemcpyAsync from pageable memory blocks the HOST thread, not the device: kernels launched before it keep running.)
static int raytrace_pipelined(rdr_ctx* c, const rdr_cube* q, const rdr_rays* r, RayParams P, int K, double* d_los, double* d_hts, double* dw, double* dh,
                              double* wet, double* hydro) {
    const int64_t tile_rows = P.ntiles / P.tiles_x;
    const int nchunk = (int)std::min<int64_t>(8, tile_rows);
    int rc = ws_reserve(c, P.ntiles, K, P); if (rc) return rc;
    double* const ws = P.ws;
    const int64_t nslots_total = P.ntiles * BLOCK;
    if (d_los) P.los = d_los;                  // (else the look vectors come from inc / heading or zenith: nothing to upload)
    if (d_hts) P.ht_ray = d_hts;               // per-pixel heights travel with the look vectors, chunk by chunk
    std::vector<hipEvent_t> ev(2 * nchunk, nullptr);
    auto range = [&](int k, int64_t& tb, int64_t& tc, int64_t& r0, int64_t& cnt) {
        const int64_t t0 = tile_rows * k / nchunk, t1 = tile_rows * (k + 1) / nchunk;
        tb = t0 * P.tiles_x; tc = (t1 - t0) * P.tiles_x;
        if (r->origin_mode == RDR_ORIGIN_GRID) { r0 = t0 * TILE * r->nx; cnt = std::min<int64_t>(t1 * TILE, r->ny) * r->nx - r0; }
        else { r0 = tb * BLOCK; cnt = std::min<int64_t>((tb + tc) * BLOCK, r->n) - r0; }
    };
    int status = RDR_OK;
    auto cleanup = [&]() { for (auto& e : ev) if (e) (void)hipEventDestroy(e); };
    for (auto& e : ev)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { cleanup(); return fail(c, RDR_ERR_HIP, "pipelined ray tracing: event creation failed"); }
    for (int k = 0; k < nchunk && status == RDR_OK; ++k) {
        int64_t tb, tc, r0, cnt; range(k, tb, tc, r0, cnt);
        if ((d_los && hipMemcpyAsync(d_los + 3 * r0, r->los + 3 * r0, (size_t)cnt * 24, hipMemcpyHostToDevice, c->copy_stream) != hipSuccess) ||
            (d_hts && hipMemcpyAsync(d_hts + r0, r->hts + r0, (size_t)cnt * 8, hipMemcpyHostToDevice, c->copy_stream) != hipSuccess) ||
            ((d_los || d_hts) && (hipEventRecord(ev[k], c->copy_stream) != hipSuccess || hipStreamWaitEvent(c->stream, ev[k], 0) != hipSuccess))) {
            status = fail(c, RDR_ERR_HIP, "pipelined ray tracing: look-vector / height upload failed"); break;
        }
        RayParams Pk = P; Pk.ws = ws + tb * BLOCK;
        status = launch_crossings(c, q, Pk, tb, tc, nslots_total, k == 0);
    }
    for (int k = 0; k < nchunk && status == RDR_OK; ++k) {
        int64_t tb, tc, r0, cnt; range(k, tb, tc, r0, cnt);
        RayParams Pk = P; Pk.ws = ws + tb * BLOCK;
        status = launch_march(c, q, Pk, tb, tc, nslots_total);
        if (status == RDR_OK && hipEventRecord(ev[nchunk + k], c->stream) != hipSuccess) status = fail(c, RDR_ERR_HIP, "pipelined ray tracing: event");
    }
    for (int k = 0; k < nchunk && status == RDR_OK; ++k) {
        int64_t tb, tc, r0, cnt; range(k, tb, tc, r0, cnt);
        if (hipStreamWaitEvent(c->copy_stream, ev[nchunk + k], 0) != hipSuccess ||
            hipMemcpyAsync(wet + r0, dw + r0, (size_t)cnt * 8, hipMemcpyDeviceToHost, c->copy_stream) != hipSuccess ||
            hipMemcpyAsync(hydro + r0, dh + r0, (size_t)cnt * 8, hipMemcpyDeviceToHost, c->copy_stream) != hipSuccess)
            status = fail(c, RDR_ERR_HIP, "pipelined ray tracing: output download failed");
    }
    if (hipStreamSynchronize(c->copy_stream) != hipSuccess && status == RDR_OK) status = fail(c, RDR_ERR_HIP, "pipelined ray tracing: sync");
    if (hipStreamSynchronize(c->stream) != hipSuccess && status == RDR_OK) status = fail(c, RDR_ERR_HIP, "pipelined ray tracing: sync");
    cleanup();
    return status;
}

static int flags_to_status(rdr_ctx* c, int flags) {
    if (!(flags & RDR_FLAG_ANY_FINITE)) return fail(c, RDR_ERR_ALL_NAN, "geo2rdr did not converge. Check orbit coverage");
    if (flags & RDR_FLAG_ANY_NAN) return fail(c, RDR_ERR_NAN_LENGTH, "some ray lengths are NaN: the number of integration parts (delay.py:283) is undefined");
    if (flags & RDR_FLAG_BAD_HEIGHT) return fail(c, RDR_ERR_INVALID, "per-ray heights: a ray starts below the height the batch's level table was built for (pass ht <= min(rays->hts))");
    if (flags & RDR_FLAG_DIVERGED) return fail(c, RDR_ERR_INVALID, "ray lengths diverged: a model level asks for fewer than 2 or more than 65536 integration parts (are the look vectors unit vectors?)");
    return RDR_OK;
}

int rdr_ray_prepass(rdr_ctx* c, const rdr_cube* q, const rdr_rays* r, double ht, double zref, double* maxlen, int32_t* flags) {
    if (!c || !q || !maxlen) return fail(c, RDR_ERR_INVALID, "rdr_ray_prepass: NULL argument");
    note_use(c, q);
    int rc = check_rays(c, r); if (rc) return rc;
    std::vector<double> lo, hi; std::vector<int> kz;
    const int K = levels_host(q->zs, ht, zref, lo, hi, kz);
    if (K == 0) return fail(c, RDR_ERR_NO_LEVELS, "no weather-model interval contributes to the ray integral (build_ray -> None)");
    HIPCHECK(c, hipSetDevice(c->device));
    RayParams P;
    rc = stage_rays(c, r, P); if (rc) return rc;
    P.ht = ht; P.zref = zref; P.max_seg = 1000.0;
    HIPCHECK(c, hipMemsetAsync(c->d_maxlen, 0, MAX_LEVELS * sizeof(unsigned long long), c->stream));
    HIPCHECK(c, hipMemsetAsync(c->d_flags, 0, sizeof(int), c->stream));
    c->wsig.valid = false;
    if (r->n > 0) {
        // keep the ray records for the rdr_ray_march that normally follows (same device arrays, whole batch fits)
        const bool keep = r->loc == RDR_DEVICE && P.ntiles <= ws_chunk_tiles(c, K);
        if (keep) { rc = ws_reserve(c, P.ntiles, K, P); if (rc) return rc; }
        rc = launch_crossings(c, q, P, 0, P.ntiles); if (rc) return rc;
        if (keep) wsig_set(c, q, r, ht, zref, K, true);
    }
    int f = 0, nslow = 0;
    HIPCHECK(c, hipMemcpyAsync(maxlen, c->d_maxlen, (size_t)K * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipMemcpyAsync(&f, c->d_flags, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipMemcpyAsync(&nslow, c->d_nslow, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    c->last_nslow = nslow;
    if (flags) *flags = f;
    return RDR_OK;
}

int rdr_ray_prepass_device(rdr_ctx* c, const rdr_cube* q, const rdr_rays* r, double ht, double zref, double* partition) {
    if (!c || !q || !partition) return fail(c, RDR_ERR_INVALID, "rdr_ray_prepass_device: NULL argument");
    note_use(c, q);
    int rc = check_rays(c, r); if (rc) return rc;
    if (r->loc != RDR_DEVICE) return fail(c, RDR_ERR_INVALID, "rdr_ray_prepass_device: rays must be device arrays");
    std::vector<double> lo, hi; std::vector<int> kz;
    const int K = levels_host(q->zs, ht, zref, lo, hi, kz);
    if (K == 0) return fail(c, RDR_ERR_NO_LEVELS, "no weather-model interval contributes to the ray integral (build_ray -> None)");
    HIPCHECK(c, hipSetDevice(c->device));
    RayParams P;
    rc = stage_rays(c, r, P); if (rc) return rc;
    P.ht = ht; P.zref = zref; P.max_seg = 1000.0;
    HIPCHECK(c, hipMemsetAsync(c->d_maxlen, 0, MAX_LEVELS * sizeof(unsigned long long), c->stream));
    HIPCHECK(c, hipMemsetAsync(c->d_flags, 0, sizeof(int), c->stream));
    c->wsig.valid = false;
    if (r->n > 0) {
        const bool keep = P.ntiles <= ws_chunk_tiles(c, K);
        if (keep) { rc = ws_reserve(c, P.ntiles, K, P); if (rc) return rc; }
        rc = launch_crossings(c, q, P, 0, P.ntiles); if (rc) return rc;
        if (keep) wsig_set(c, q, r, ht, zref, K, true);
    }
    hipLaunchKernelGGL(pack_partition_kernel, dim3(1), dim3(256), 0, c->stream, c->d_maxlen, c->d_flags, K, partition);
    HIPCHECK(c, hipGetLastError());
    return RDR_OK;
}

int rdr_ray_march_device(rdr_ctx* c, const rdr_cube* q, const rdr_rays* r, double ht, double zref, double max_seg, const double* partition,
                         double* wet, double* hydro) {
    if (!c || !q || !partition || !wet || !hydro) return fail(c, RDR_ERR_INVALID, "rdr_ray_march_device: NULL argument");
    note_use(c, q);
    if (!(max_seg > 0)) return fail(c, RDR_ERR_INVALID, "rdr_ray_march_device: MAX_SEGMENT_LENGTH must be positive");
    int rc = check_rays(c, r); if (rc) return rc;
    if (r->loc != RDR_DEVICE) return fail(c, RDR_ERR_INVALID, "rdr_ray_march_device: rays and outputs must be device arrays");
    std::vector<double> lo, hi; std::vector<int> kz;
    const int K = levels_host(q->zs, ht, zref, lo, hi, kz);
    if (K == 0) return fail(c, RDR_ERR_NO_LEVELS, "no weather-model interval contributes to the ray integral (build_ray -> None)");
    if (r->n == 0) return RDR_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    const bool reuse = wsig_match(c, q, r, ht, zref, K);
    RayParams P;
    rc = stage_rays(c, r, P); if (rc) return rc;
    P.ht = ht; P.zref = zref; P.max_seg = max_seg;
    P.wet = wet; P.hyd = hydro;
    hipLaunchKernelGGL(unpack_partition_kernel, dim3(1), dim3(256), 0, c->stream, partition, K, c->d_maxlen, c->d_flags);
    HIPCHECK(c, hipGetLastError());
    if (reuse) { ws_attach(c, P); rc = launch_march(c, q, P, 0, P.ntiles); }
    else rc = march_chunked(c, q, P, K);
    c->wsig.valid = false;
    return rc;
}

int rdr_ray_march(rdr_ctx* c, const rdr_cube* q, const rdr_rays* r, double ht, double zref, const int32_t* nparts, int32_t flags,
                  double* wet, double* hydro) {
    if (!c || !q || !nparts || !wet || !hydro) return fail(c, RDR_ERR_INVALID, "rdr_ray_march: NULL argument");
    note_use(c, q);
    int rc = check_rays(c, r); if (rc) return rc;
    std::vector<double> lo, hi; std::vector<int> kz;
    const int K = levels_host(q->zs, ht, zref, lo, hi, kz);
    if (K == 0) return fail(c, RDR_ERR_NO_LEVELS, "no weather-model interval contributes to the ray integral (build_ray -> None)");
    for (int k = 0; k < K; ++k) if (nparts[k] < 2 || nparts[k] > MAX_NPARTS) return fail(c, RDR_ERR_INVALID, "rdr_ray_march: nparts out of range (2..65536)");
    if (r->n == 0) return RDR_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    const bool reuse = wsig_match(c, q, r, ht, zref, K);
    RayParams P;
    rc = stage_rays(c, r, P); if (rc) return rc;
    P.ht = ht; P.zref = zref; P.max_seg = 1000.0;
    HIPCHECK(c, hipMemcpyAsync(c->d_nparts, nparts, (size_t)K * sizeof(int), hipMemcpyHostToDevice, c->stream));
    int f = flags;
    HIPCHECK(c, hipMemcpyAsync(c->d_flags, &f, sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));   // nparts / f are caller / stack memory
    P.nparts_override = c->d_nparts;
    void *dw, *dh;
    rc = stage_out(c, SLOT_OUT0, wet, (size_t)r->n * 8, r->loc, &dw); if (rc) return rc;
    rc = stage_out(c, SLOT_OUT1, hydro, (size_t)r->n * 8, r->loc, &dh); if (rc) return rc;
    P.wet = (double*)dw; P.hyd = (double*)dh;
    if (reuse) { ws_attach(c, P); rc = launch_march(c, q, P, 0, P.ntiles); }
    else rc = march_chunked(c, q, P, K);
    c->wsig.valid = false;
    if (rc) return rc;
    rc = finish_out(c, wet, dw, (size_t)r->n * 8, r->loc); if (rc) return rc;
    rc = finish_out(c, hydro, dh, (size_t)r->n * 8, r->loc); if (rc) return rc;
    if (r->loc == RDR_HOST) HIPCHECK(c, hipStreamSynchronize(c->stream));
    return RDR_OK;
}

int rdr_raytrace(rdr_ctx* c, const rdr_cube* q, const rdr_rays* r, double ht, double zref, double max_seg, double* wet,
                 double* hydro, int32_t* nparts_out, int32_t* flags_out) {
    if (!c || !q || !wet || !hydro) return fail(c, RDR_ERR_INVALID, "rdr_raytrace: NULL argument");
    note_use(c, q);
    if (!(max_seg > 0)) return fail(c, RDR_ERR_INVALID, "rdr_raytrace: MAX_SEGMENT_LENGTH must be positive");
    int rc = check_rays(c, r); if (rc) return rc;
    std::vector<double> lo, hi; std::vector<int> kz;
    const int K = levels_host(q->zs, ht, zref, lo, hi, kz);
    if (K == 0) return fail(c, RDR_ERR_NO_LEVELS, "no weather-model interval contributes to the ray integral (build_ray -> None)");
    if (r->n == 0) return RDR_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    RayParams P;
    // large host-buffer batches with per-ray look vectors: overlap the PCIe transfers with the two passes (raytrace_pipelined)
    static const bool no_pipeline = std::getenv("RAIDER_HIP_NO_PIPELINE") != nullptr;
    bool pipelined = r->loc == RDR_HOST && r->n >= (1 << 21) && !no_pipeline;
    const bool los_chunks = pipelined && r->los_mode == RDR_LOS_VEC;
    const bool hts_chunks = pipelined && r->hts != nullptr;
    rdr_rays rr = *r;
    if (los_chunks) rr.los = nullptr;                      // the look vectors are uploaded chunk by chunk below
    if (hts_chunks) rr.hts = nullptr;                      // ... and so are per-pixel heights
    rc = stage_rays(c, &rr, P); if (rc) return rc;
    if (pipelined && P.ntiles > ws_chunk_tiles(c, K)) {   // records do not fit: ordinary (chunked-march) path
        const void* d;
        if (los_chunks) {
            rc = stage_in(c, SLOT_IN3, r->los, (size_t)r->n * 24, r->loc, &d); if (rc) return rc;
            P.los = (const double*)d;
        }
        if (hts_chunks) {
            rc = stage_in(c, SLOT_IN6, r->hts, (size_t)r->n * 8, r->loc, &d); if (rc) return rc;
            P.ht_ray = (const double*)d;
        }
        pipelined = false;
    }
    P.ht = ht; P.zref = zref; P.max_seg = max_seg;
    void *dw, *dh;
    rc = stage_out(c, SLOT_OUT0, wet, (size_t)r->n * 8, r->loc, &dw); if (rc) return rc;
    rc = stage_out(c, SLOT_OUT1, hydro, (size_t)r->n * 8, r->loc, &dh); if (rc) return rc;
    P.wet = (double*)dw; P.hyd = (double*)dh;
    HIPCHECK(c, hipMemsetAsync(c->d_maxlen, 0, MAX_LEVELS * sizeof(unsigned long long), c->stream));
    HIPCHECK(c, hipMemsetAsync(c->d_flags, 0, sizeof(int), c->stream));
    c->wsig.valid = false;
    if (pipelined) {
        void *dl = nullptr, *dhts = nullptr;
        if (los_chunks) { rc = ensure(c, SLOT_IN3, (size_t)r->n * 24, &dl); if (rc) return rc; }
        if (hts_chunks) { rc = ensure(c, SLOT_IN6, (size_t)r->n * 8, &dhts); if (rc) return rc; }
        rc = raytrace_pipelined(c, q, r, P, K, (double*)dl, (double*)dhts, (double*)dw, (double*)dh, wet, hydro); if (rc) return rc;
    } else if (P.ntiles <= ws_chunk_tiles(c, K)) {
        // whole batch fits: pass 1 reduces AND stores the ray records, pass 2 streams them back
        rc = ws_reserve(c, P.ntiles, K, P); if (rc) return rc;
        rc = launch_crossings(c, q, P, 0, P.ntiles); if (rc) return rc;
        rc = launch_march(c, q, P, 0, P.ntiles); if (rc) return rc;
    } else {
        // pass 1 (reduction only) over everything, then chunked (store, march) pairs
        P.ws = nullptr;
        rc = launch_crossings(c, q, P, 0, P.ntiles); if (rc) return rc;
        rc = march_chunked(c, q, P, K); if (rc) return rc;
    }
    if (!pipelined) {
        rc = finish_out(c, wet, dw, (size_t)r->n * 8, r->loc); if (rc) return rc;
        rc = finish_out(c, hydro, dh, (size_t)r->n * 8, r->loc); if (rc) return rc;
    }
    const bool need_sync = r->loc == RDR_HOST || nparts_out || flags_out;
    if (need_sync) {
        std::vector<double> ml(K);
        int f = 0, nslow = 0;
        HIPCHECK(c, hipMemcpyAsync(ml.data(), c->d_maxlen, (size_t)K * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHECK(c, hipMemcpyAsync(&f, c->d_flags, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHECK(c, hipMemcpyAsync(&nslow, c->d_nslow, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHECK(c, hipStreamSynchronize(c->stream));
        c->last_nslow = nslow;
        if (nparts_out) rdr_nparts(ml.data(), K, max_seg, nparts_out);
        if (flags_out) *flags_out = f;
        return flags_to_status(c, f);
    }
    return RDR_OK;
}

// Several height slices of _build_cube_ray in ONE pass-1 / pass-2 launch pair (delay.py:256-323 loops over them): a production
// job is ~20 heights x 1e4-1e5 rays (aria/prepFromGUNW.py:173,180), and one such slice fills a fraction of the chip.  Tiles are
// numbered slice-major; per-level maxima / flags / nParts stay per slice (RayParams), so the result is what slice-by-slice
// calls give, bit for bit.
// keep == NULL: the public entry; keep != NULL: the delays stay in the context's scratch (planar [nslices][n], keep[0] = wet, keep[1] = hydro)
// for rdr_raytrace_slices_to_cube - nothing is downloaded, the partition outputs are still read back.
static int raytrace_slices_impl(rdr_ctx* c, const rdr_cube* q, const rdr_rays* r, const double* hts, int32_t nslices, int32_t los_per_slice,
                                double zref, double max_seg, double* wet, double* hydro, int32_t* K_out, int32_t* nparts_out, int32_t ld,
                                int32_t* flags_out, double** keep) {
    if (!c || !q || !hts || (!keep && (!wet || !hydro))) return fail(c, RDR_ERR_INVALID, "rdr_raytrace_slices: NULL argument");
    note_use(c, q);
    if (nslices < 1 || nslices > MAX_SLICES) return fail(c, RDR_ERR_INVALID, "rdr_raytrace_slices: 1..512 slices per call");
    if (!(max_seg > 0)) return fail(c, RDR_ERR_INVALID, "rdr_raytrace_slices: MAX_SEGMENT_LENGTH must be positive");
    if (nparts_out && ld < (int32_t)q->nz - 1) return fail(c, RDR_ERR_INVALID, "rdr_raytrace_slices: nparts_out needs a row length of at least nz-1");
    int rc = check_rays(c, r); if (rc) return rc;
    if (r->origin_mode == RDR_ORIGIN_XYZ && nslices > 1) return fail(c, RDR_ERR_INVALID, "rdr_raytrace_slices: XYZ origins belong to one height; use GRID or LLH origins");
    if (r->hts) return fail(c, RDR_ERR_INVALID, "rdr_raytrace_slices: per-ray heights (rays->hts) describe ONE batch; use rdr_raytrace");
    std::vector<int> Ks(nslices);
    int Kmax = 0;
    for (int s = 0; s < nslices; ++s) {
        std::vector<double> lo, hi; std::vector<int> kz;
        Ks[s] = levels_host(q->zs, hts[s], zref, lo, hi, kz);
        Kmax = std::max(Kmax, Ks[s]);
        if (K_out) K_out[s] = Ks[s];
    }
    if (r->n == 0) return RDR_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    RayParams P;
    rc = stage_rays(c, r, P, los_per_slice ? nslices : 1); if (rc) return rc;
    const int64_t per = P.ntiles;                                   // tiles of one slice
    const size_t nout = (size_t)r->n * nslices;
    void *dw, *dh;
    const int out_loc = keep ? RDR_DEVICE : r->loc;                 // where the delays end up
    rc = stage_out(c, SLOT_OUT0, wet, nout * 8, keep ? RDR_HOST : r->loc, &dw); if (rc) return rc;
    rc = stage_out(c, SLOT_OUT1, hydro, nout * 8, keep ? RDR_HOST : r->loc, &dh); if (rc) return rc;
    if (keep) { keep[0] = (double*)dw; keep[1] = (double*)dh; }
    const void* dht;
    rc = stage_in(c, SLOT_AUX, hts, (size_t)nslices * 8, RDR_HOST, &dht); if (rc) return rc;
    HIPCHECK(c, hipStreamSynchronize(c->stream));                   // (hts is the caller's memory)
    P.nslices = nslices; P.tiles_per_slice = per; P.hts = (const double*)dht; P.los_stride = los_per_slice ? r->n : 0;
    P.ntiles = per * nslices;
    P.ht = hts[0]; P.zref = zref; P.max_seg = max_seg;
    P.wet = (double*)dw; P.hyd = (double*)dh;
    HIPCHECK(c, hipMemsetAsync(c->d_maxlen, 0, (size_t)nslices * MAX_LEVELS * sizeof(unsigned long long), c->stream));
    HIPCHECK(c, hipMemsetAsync(c->d_flags, 0, (size_t)nslices * sizeof(int), c->stream));
    c->wsig.valid = false;
    const int64_t fit = ws_chunk_tiles(c, std::max(Kmax, 1));
    // Host outputs of a large batch come down slice group by slice group on the copy stream while the next groups are integrated
    // (all kernels are enqueued first: a copy into pageable memory blocks the host thread, not the device).
    static const bool no_pipeline = std::getenv("RAIDER_HIP_NO_PIPELINE") != nullptr;
    const bool pipe = out_loc == RDR_HOST && nslices >= 2 && nout * 16 >= ((size_t)32 << 20) && per <= fit && !no_pipeline;
    struct EventList {                                   // (destroyed on every exit path)
        std::vector<hipEvent_t> v;
        ~EventList() { for (auto& e : v) if (e) (void)hipEventDestroy(e); }
    } gev_owner;
    std::vector<hipEvent_t>& gev = gev_owner.v;
    std::vector<std::pair<int64_t, int64_t>> groups;
    bool downloaded = false;
    if (per <= fit) {
        // groups of whole slices whose records fit the workspace: one launch pair per group (usually one group; up to 8 when the
        // outputs are downloaded group by group)
        int64_t g = std::max<int64_t>(1, fit / per);
        if (pipe) g = std::min<int64_t>(g, std::max<int64_t>(1, (nslices + 7) / 8));
        for (int64_t s0 = 0; s0 < nslices; s0 += g) {
            const int64_t ns = std::min<int64_t>(g, nslices - s0);
            RayParams Pg = P;
            rc = ws_reserve(c, ns * per, std::max(Kmax, 1), Pg); if (rc) return rc;
            rc = launch_crossings(c, q, Pg, s0 * per, ns * per); if (rc) return rc;
            rc = launch_march(c, q, Pg, s0 * per, ns * per); if (rc) return rc;
            // a NaN anywhere in the slice's outputs -> RDR_FLAG_NAN_OUTPUT of that slice (the caller's np.isnan(...).any(), delay.py:187)
            hipLaunchKernelGGL(nan_scan_kernel, dim3(grid_for(ns * r->n, 256, c->num_cus * 8)), dim3(256), 0, c->stream,
                               (const double*)dw + (size_t)s0 * r->n, (const double*)dh + (size_t)s0 * r->n, (int64_t)ns * r->n, (int64_t)r->n,
                               c->d_flags + s0);
            HIPCHECK(c, hipGetLastError());
            if (pipe) {
                hipEvent_t e = nullptr;
                if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess || hipEventRecord(e, c->stream) != hipSuccess) {
                    if (e) (void)hipEventDestroy(e);
                    return fail(c, RDR_ERR_HIP, "rdr_raytrace_slices: event");
                }
                gev.push_back(e); groups.emplace_back(s0, ns);
            }
        }
        if (pipe) {
            int status = RDR_OK;
            for (size_t k = 0; k < groups.size() && status == RDR_OK; ++k) {
                const size_t off = (size_t)groups[k].first * r->n, cnt = (size_t)groups[k].second * r->n;
                if (hipStreamWaitEvent(c->copy_stream, gev[k], 0) != hipSuccess ||
                    hipMemcpyAsync(wet + off, (const double*)dw + off, cnt * 8, hipMemcpyDeviceToHost, c->copy_stream) != hipSuccess ||
                    hipMemcpyAsync(hydro + off, (const double*)dh + off, cnt * 8, hipMemcpyDeviceToHost, c->copy_stream) != hipSuccess)
                    status = fail(c, RDR_ERR_HIP, "rdr_raytrace_slices: output download failed");
            }
            if (hipStreamSynchronize(c->copy_stream) != hipSuccess && status == RDR_OK) status = fail(c, RDR_ERR_HIP, "rdr_raytrace_slices: sync");
            if (status != RDR_OK) return status;
            downloaded = true;
        }
    } else {
        // a single slice exceeds the workspace: per slice, pass 1 (reduction only) then chunked (store, march) pairs
        for (int s = 0; s < nslices; ++s) {
            RayParams Ps = P;
            Ps.ws = nullptr;
            rc = launch_crossings(c, q, Ps, (int64_t)s * per, per); if (rc) return rc;
            const int64_t chunk = std::min<int64_t>(per, fit);
            RayParams Pw = P;
            rc = ws_reserve(c, chunk, std::max(Kmax, 1), Pw); if (rc) return rc;
            for (int64_t tb = 0; tb < per; tb += chunk) {
                const int64_t tc = std::min<int64_t>(chunk, per - tb);
                RayParams Pc = Pw;
                Pc.maxlen_bits = nullptr;
                rc = launch_crossings(c, q, Pc, (int64_t)s * per + tb, tc); if (rc) return rc;
                Pc.maxlen_bits = P.maxlen_bits;
                rc = launch_march(c, q, Pc, (int64_t)s * per + tb, tc); if (rc) return rc;
            }
        }
    }
    if (!downloaded) {
        if (per > fit) {        // (the chunked branch: scan the finished outputs before they leave)
            hipLaunchKernelGGL(nan_scan_kernel, dim3(grid_for((int64_t)nout, 256, c->num_cus * 8)), dim3(256), 0, c->stream, (const double*)dw,
                               (const double*)dh, (int64_t)nout, (int64_t)r->n, c->d_flags);
            HIPCHECK(c, hipGetLastError());
        }
        rc = finish_out(c, wet, dw, nout * 8, out_loc); if (rc) return rc;
        rc = finish_out(c, hydro, dh, nout * 8, out_loc); if (rc) return rc;
    }
    const bool need_sync = r->loc == RDR_HOST || nparts_out || flags_out;
    if (need_sync) {
        std::vector<double> ml((size_t)nslices * MAX_LEVELS);
        std::vector<int> f(nslices);
        int nslow = 0;
        HIPCHECK(c, hipMemcpyAsync(ml.data(), c->d_maxlen, ml.size() * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHECK(c, hipMemcpyAsync(f.data(), c->d_flags, (size_t)nslices * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHECK(c, hipMemcpyAsync(&nslow, c->d_nslow, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHECK(c, hipStreamSynchronize(c->stream));
        c->last_nslow = nslow;
        for (int s = 0; s < nslices; ++s) {
            if (nparts_out) rdr_nparts(ml.data() + (size_t)s * MAX_LEVELS, Ks[s], max_seg, nparts_out + (size_t)s * ld);
            if (flags_out) flags_out[s] = f[s];
        }
    }
    return RDR_OK;
}

int rdr_raytrace_slices(rdr_ctx* c, const rdr_cube* q, const rdr_rays* r, const double* hts, int32_t nslices, int32_t los_per_slice,
                        double zref, double max_seg, double* wet, double* hydro, int32_t* K_out, int32_t* nparts_out, int32_t ld,
                        int32_t* flags_out) {
    return raytrace_slices_impl(c, q, r, hts, nslices, los_per_slice, zref, max_seg, wet, hydro, K_out, nparts_out, ld, flags_out, nullptr);
}

int rdr_raytrace_slices_to_cube(rdr_ctx* c, const rdr_cube* q, const rdr_rays* r, const double* hts, int32_t nslices, int32_t los_per_slice,
                                double zref, double max_seg, int32_t* K_out, int32_t* nparts_out, int32_t ld, int32_t* flags_out,
                                rdr_cube** out) {
    if (!c || !q || !r || !hts || !out) return fail(c, RDR_ERR_INVALID, "rdr_raytrace_slices_to_cube: NULL argument");
    if (r->origin_mode != RDR_ORIGIN_GRID) return fail(c, RDR_ERR_INVALID, "rdr_raytrace_slices_to_cube: the delay cube is a GRID batch (xpts, ypts) x heights");
    if (r->nx < 2 || r->ny < 2 || nslices < 2) return fail(c, RDR_ERR_INVALID, "rdr_raytrace_slices_to_cube: the delay cube needs two nodes per axis");
    HIPCHECK(c, hipSetDevice(c->device));
    std::vector<double> hx, hy;
    int rc = axis_to_host(c, r->xpts, r->nx, r->loc, hx); if (rc) return rc;
    rc = axis_to_host(c, r->ypts, r->ny, r->loc, hy); if (rc) return rc;
    double* planar[2] = {nullptr, nullptr};
    rc = raytrace_slices_impl(c, q, r, hts, nslices, los_per_slice, zref, max_seg, nullptr, nullptr, K_out, nparts_out, ld, flags_out, planar);
    if (rc) return rc;
    return rdr_cube_create(c, hy.data(), r->ny, hx.data(), r->nx, hts, nslices, planar[0], planar[1], RDR_F64, r->nx, 1, r->ny * r->nx, RDR_DEVICE, out);
}

int rdr_top_of_atmosphere(rdr_ctx* c, const double* xyz, const double* los, int64_t n, double h, const double* factor, double* pos, int loc) {
    if (!c || !xyz || !los || !pos) return fail(c, RDR_ERR_INVALID, "rdr_top_of_atmosphere: NULL argument");
    if (n < 0) return fail(c, RDR_ERR_INVALID, "rdr_top_of_atmosphere: negative count");
    if (n == 0) return RDR_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    const void *dx, *dl, *df; void* dp;
    int rc = stage_in(c, SLOT_IN0, xyz, (size_t)n * 24, loc, &dx); if (rc) return rc;
    rc = stage_in(c, SLOT_IN1, los, (size_t)n * 24, loc, &dl); if (rc) return rc;
    rc = stage_in(c, SLOT_IN2, factor, (size_t)n * 8, loc, &df); if (rc) return rc;
    rc = stage_out(c, SLOT_OUT0, pos, (size_t)n * 24, loc, &dp); if (rc) return rc;
    hipLaunchKernelGGL(toa_kernel, dim3(grid_for(n, 256, c->num_cus * 8)), dim3(256), 0, c->stream, (const double*)dx, (const double*)dl, n, h,
                       (const double*)df, (double*)dp);
    HIPCHECK(c, hipGetLastError());
    rc = finish_out(c, pos, dp, (size_t)n * 24, loc); if (rc) return rc;
    if (loc == RDR_HOST) HIPCHECK(c, hipStreamSynchronize(c->stream));
    return RDR_OK;
}

int rdr_build_ray(rdr_ctx* c, const double* model_zs, int64_t nz, double ht, const double* xyz, const double* los, int64_t n,
                  double zref, int32_t* K_out, double* lengths, double* low, double* high, int loc) {
    if (!c || !model_zs || !xyz || !los || !K_out) return fail(c, RDR_ERR_INVALID, "rdr_build_ray: NULL argument");
    if (nz < 2) return fail(c, RDR_ERR_INVALID, "rdr_build_ray: need at least 2 model levels");
    std::vector<double> zs(model_zs, model_zs + nz), lo, hi; std::vector<int> kz;
    const int K = levels_host(zs, ht, zref, lo, hi, kz);
    *K_out = K;
    if (K == 0) return fail(c, RDR_ERR_NO_LEVELS, "no weather-model interval contributes to the ray integral (build_ray -> None)");
    if (!lengths || !low || !high) return RDR_OK;   // size query
    if (n < 0) return fail(c, RDR_ERR_INVALID, "rdr_build_ray: negative count");
    if (n == 0) return RDR_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    const void *dx, *dl; void *dlen, *dlo, *dhi, *dtab;
    int rc = stage_in(c, SLOT_IN0, xyz, (size_t)n * 24, loc, &dx); if (rc) return rc;
    rc = stage_in(c, SLOT_IN1, los, (size_t)n * 24, loc, &dl); if (rc) return rc;
    rc = stage_out(c, SLOT_OUT0, lengths, (size_t)K * n * 8, loc, &dlen); if (rc) return rc;
    rc = stage_out(c, SLOT_OUT1, low, (size_t)K * n * 24, loc, &dlo); if (rc) return rc;
    rc = stage_out(c, SLOT_OUT2, high, (size_t)K * n * 24, loc, &dhi); if (rc) return rc;
    rc = ensure(c, SLOT_AUX, (size_t)2 * K * 8, &dtab); if (rc) return rc;
    std::vector<double> tab(lo); tab.insert(tab.end(), hi.begin(), hi.end());
    HIPCHECK(c, hipMemcpyAsync(dtab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    hipLaunchKernelGGL(build_ray_kernel, dim3(grid_for(n, 256, c->num_cus * 8)), dim3(256), 0, c->stream, (const double*)dx, (const double*)dl, n, K,
                       (const double*)dtab, (double*)dlen, (double*)dlo, (double*)dhi);
    HIPCHECK(c, hipGetLastError());
    rc = finish_out(c, lengths, dlen, (size_t)K * n * 8, loc); if (rc) return rc;
    rc = finish_out(c, low, dlo, (size_t)K * n * 24, loc); if (rc) return rc;
    rc = finish_out(c, high, dhi, (size_t)K * n * 24, loc); if (rc) return rc;
    if (loc == RDR_HOST) HIPCHECK(c, hipStreamSynchronize(c->stream));
    return RDR_OK;
}

// ---- geodesy ------------------------------------------------------------------------------------------
int rdr_lla2ecef(rdr_ctx* c, const double* lat, const double* lon, const double* h, int64_t n, double* xyz, int loc) {
    if (!c || !lat || !lon || !h || !xyz) return fail(c, RDR_ERR_INVALID, "rdr_lla2ecef: NULL argument");
    if (n < 0) return fail(c, RDR_ERR_INVALID, "rdr_lla2ecef: negative count");
    if (n == 0) return RDR_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    const void *a, *b, *d; void* o;
    int rc = stage_in(c, SLOT_IN0, lat, (size_t)n * 8, loc, &a); if (rc) return rc;
    rc = stage_in(c, SLOT_IN1, lon, (size_t)n * 8, loc, &b); if (rc) return rc;
    rc = stage_in(c, SLOT_IN2, h, (size_t)n * 8, loc, &d); if (rc) return rc;
    rc = stage_out(c, SLOT_OUT0, xyz, (size_t)n * 24, loc, &o); if (rc) return rc;
    hipLaunchKernelGGL(lla2ecef_kernel, dim3(grid_for(n, 256, c->num_cus * 8)), dim3(256), 0, c->stream, (const double*)a, (const double*)b,
                       (const double*)d, n, (double*)o);
    HIPCHECK(c, hipGetLastError());
    rc = finish_out(c, xyz, o, (size_t)n * 24, loc); if (rc) return rc;
    if (loc == RDR_HOST) HIPCHECK(c, hipStreamSynchronize(c->stream));
    return RDR_OK;
}

int rdr_ecef2lla(rdr_ctx* c, const double* xyz, int64_t n, double* lon, double* lat, double* h, int loc) {
    if (!c || !xyz || !lon || !lat || !h) return fail(c, RDR_ERR_INVALID, "rdr_ecef2lla: NULL argument");
    if (n < 0) return fail(c, RDR_ERR_INVALID, "rdr_ecef2lla: negative count");
    if (n == 0) return RDR_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    const void* a; void *o0, *o1, *o2;
    int rc = stage_in(c, SLOT_IN0, xyz, (size_t)n * 24, loc, &a); if (rc) return rc;
    rc = stage_out(c, SLOT_OUT0, lon, (size_t)n * 8, loc, &o0); if (rc) return rc;
    rc = stage_out(c, SLOT_OUT1, lat, (size_t)n * 8, loc, &o1); if (rc) return rc;
    rc = stage_out(c, SLOT_OUT2, h, (size_t)n * 8, loc, &o2); if (rc) return rc;
    hipLaunchKernelGGL(ecef2lla_kernel, dim3(grid_for(n, 256, c->num_cus * 8)), dim3(256), 0, c->stream, (const double*)a, n, (double*)o0,
                       (double*)o1, (double*)o2);
    HIPCHECK(c, hipGetLastError());
    rc = finish_out(c, lon, o0, (size_t)n * 8, loc); if (rc) return rc;
    rc = finish_out(c, lat, o1, (size_t)n * 8, loc); if (rc) return rc;
    rc = finish_out(c, h, o2, (size_t)n * 8, loc); if (rc) return rc;
    if (loc == RDR_HOST) HIPCHECK(c, hipStreamSynchronize(c->stream));
    return RDR_OK;
}

int rdr_look_vectors(rdr_ctx* c, const rdr_rays* r, double ht, double* los) {
    (void)ht;
    if (!c || !los) return fail(c, RDR_ERR_INVALID, "rdr_look_vectors: NULL argument");
    int rc = check_rays(c, r); if (rc) return rc;
    if (r->origin_mode == RDR_ORIGIN_XYZ && (!r->lat || !r->lon) && r->los_mode != RDR_LOS_VEC)
        return fail(c, RDR_ERR_INVALID, "rdr_look_vectors: lat/lon required");
    if (r->n == 0) return RDR_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    RayParams P;
    rc = stage_rays(c, r, P); if (rc) return rc;
    void* o;
    rc = stage_out(c, SLOT_OUT0, los, (size_t)r->n * 24, r->loc, &o); if (rc) return rc;
    hipLaunchKernelGGL(look_kernel, dim3(grid_for(r->n, 256, c->num_cus * 8)), dim3(256), 0, c->stream, P, (double*)o);
    HIPCHECK(c, hipGetLastError());
    rc = finish_out(c, los, o, (size_t)r->n * 24, r->loc); if (rc) return rc;
    if (r->loc == RDR_HOST) HIPCHECK(c, hipStreamSynchronize(c->stream));
    return RDR_OK;
}

int rdr_cubes_from_model_levels(rdr_ctx* c, const double* ys, int64_t ny, const double* xs, int64_t nx, const double* zs3, const double* p,
                                const double* t, const double* hum, int humidity_type, int64_t nlev, const double* new_z, int64_t nz,
                                double k1, double k2, double k3, double zmin, int loc, rdr_cube** pointwise, rdr_cube** total,
                                float* t_out, float* p_out, float* e_out) {
    if (!c || !ys || !xs || !zs3 || !p || !t || !hum || !new_z || !pointwise || !total) return fail(c, RDR_ERR_INVALID, "rdr_cubes_from_model_levels: NULL argument");
    if (humidity_type != 0 && humidity_type != 1) return fail(c, RDR_ERR_INVALID, "Not a valid humidity type");
    if (nlev < 2 || nlev > 1024 || nz < 2 || nz + 1 > MAX_LEVELS) return fail(c, RDR_ERR_INVALID, "rdr_cubes_from_model_levels: unsupported number of levels");
    int fy, fx, fz;
    if (axis_check(ys, ny, &fy) || axis_check(xs, nx, &fx) || axis_check(new_z, nz, &fz) || fy || fx || fz)
        return fail(c, RDR_ERR_INVALID, "rdr_cubes_from_model_levels: x, y and the new z levels must be strictly ascending");
    HIPCHECK(c, hipSetDevice(c->device));
    double zlow = new_z[0];
    const int pad = zmin < zlow ? 1 : 0;                              // _adjust_grid, weatherModel.py:376-378
    const int64_t nzo = nz + pad, ncol = ny * nx;
    rdr_cube* q[2] = {new rdr_cube(), new rdr_cube()};
    for (int i = 0; i < 2; ++i) {
        q[i]->ctx = c; q[i]->ny = ny; q[i]->nx = nx; q[i]->nz = nzo; q[i]->dtype = i == 0 ? RDR_F32 : RDR_F64;
        q[i]->ys.assign(ys, ys + ny); q[i]->xs.assign(xs, xs + nx);
        if (pad) q[i]->zs.push_back(zmin);
        q[i]->zs.insert(q[i]->zs.end(), new_z, new_z + nz);
        int rc = cube_alloc(c, q[i]);
        if (rc) { rdr_cube_destroy(q[0]); rdr_cube_destroy(q[1]); return rc; }
    }
    auto bail = [&](int rc) { rdr_cube_destroy(q[0]); rdr_cube_destroy(q[1]); return rc; };
    ProducerParams P;
    const void* d;
    const size_t nb = (size_t)ncol * nlev * 8;
    int rc;
    rc = stage_in(c, SLOT_IN0, zs3, nb, loc, &d); if (rc) return bail(rc); P.zs = (const double*)d;
    rc = stage_in(c, SLOT_IN1, p, nb, loc, &d); if (rc) return bail(rc); P.p = (const double*)d;
    rc = stage_in(c, SLOT_IN2, t, nb, loc, &d); if (rc) return bail(rc); P.t = (const double*)d;
    rc = stage_in(c, SLOT_IN3, hum, nb, loc, &d); if (rc) return bail(rc); P.hum = (const double*)d;
    void* dz;
    rc = ensure(c, SLOT_AUX, (size_t)nz * 8, &dz); if (rc) return bail(rc);
    if (hipMemcpyAsync(dz, new_z, (size_t)nz * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)
        return bail(fail(c, RDR_ERR_HIP, "rdr_cubes_from_model_levels: copy of the level table failed"));
    P.new_z = (const double*)dz;
    P.ncol = ncol; P.nlev = (int)nlev; P.hum_type = humidity_type; P.nz = (int)nz; P.pad = pad;
    P.k1 = (float)k1; P.k2 = (float)k2; P.k3 = (float)k3; P.zmin = zmin; P.R_v = 461.524; P.R_d = 287.06;   // weatherModel.py:78-79
    P.pw = (float2*)q[0]->d_vals; P.tot = (double2*)q[1]->d_vals;
    void *dt_ = nullptr, *dp_ = nullptr, *de_ = nullptr;
    const size_t ob = (size_t)ncol * nzo * 4;
    if (t_out && p_out && e_out) {
        rc = stage_out(c, SLOT_OUT0, t_out, ob, loc, &dt_); if (rc) return bail(rc);
        rc = stage_out(c, SLOT_OUT1, p_out, ob, loc, &dp_); if (rc) return bail(rc);
        rc = stage_out(c, SLOT_OUT2, e_out, ob, loc, &de_); if (rc) return bail(rc);
    }
    P.t_out = (float*)dt_; P.p_out = (float*)dp_; P.e_out = (float*)de_;
    const size_t per_wave = ((size_t)4 * nlev + 3 * nzo + (5 * nzo + 1) / 2 + 1) * 8;
    const int g = (int)std::max<int64_t>(1, std::min<int64_t>((ncol + 3) / 4, (int64_t)c->num_cus * 8));
    if (per_wave * 4 > c->lds_max)
        return bail(fail(c, RDR_ERR_INVALID, "rdr_cubes_from_model_levels: " + std::to_string(nlev) + " model levels -> " + std::to_string(nzo) +
                         " output levels need " + std::to_string(per_wave * 4) + " B of LDS per workgroup, the device offers " + std::to_string(c->lds_max)));
    hipError_t e;
    {
        KTimer tm(c, 3);
        e = launch_lds(producer_kernel, dim3(g), dim3(256), per_wave * 4, c->stream, P);
    }
    if (e == hipSuccess && dt_) {
        if (finish_out(c, t_out, dt_, ob, loc) || finish_out(c, p_out, dp_, ob, loc) || finish_out(c, e_out, de_, ob, loc)) return bail(RDR_ERR_HIP);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return bail(fail(c, RDR_ERR_HIP, std::string("producer_kernel: ") + hipGetErrorString(e)));
    *pointwise = q[0]; *total = q[1];
    return RDR_OK;
}

int rdr_ecmwf_model_levels(rdr_ctx* c, const float* z_surf, const float* lnsp, const float* t, const float* q, const float* lats,
                           const double* a, const double* b, int32_t nlev, int64_t ny, int64_t nx, double R_d, double* p_out, double* zs_out, int loc) {
    if (!c || !z_surf || !lnsp || !t || !q || !lats || !a || !b || !p_out || !zs_out) return fail(c, RDR_ERR_INVALID, "rdr_ecmwf_model_levels: NULL argument");
    if (nlev < 1 || ny < 1 || nx < 1) return fail(c, RDR_ERR_INVALID, "rdr_ecmwf_model_levels: empty grid");
    HIPCHECK(c, hipSetDevice(c->device));
    const int64_t ncol = ny * nx;
    // the small tables (latitudes, a, b) always come from the host; the fields and the outputs follow `loc`
    void* aux;
    int rc = ensure(c, SLOT_AUX, (size_t)(2 * (nlev + 1)) * 8 + (size_t)ny * 4, &aux); if (rc) return rc;
    double* da = (double*)aux; double* db = da + (nlev + 1); float* dl = (float*)(db + (nlev + 1));
    HIPCHECK(c, hipMemcpyAsync(da, a, (size_t)(nlev + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, hipMemcpyAsync(db, b, (size_t)(nlev + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, hipMemcpyAsync(dl, lats, (size_t)ny * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    const void *dz, *ds, *dt_, *dq; void *dp, *dh;
    rc = stage_in(c, SLOT_IN0, z_surf, (size_t)ncol * 4, loc, &dz); if (rc) return rc;
    rc = stage_in(c, SLOT_IN1, lnsp, (size_t)ncol * 4, loc, &ds); if (rc) return rc;
    rc = stage_in(c, SLOT_IN2, t, (size_t)ncol * nlev * 4, loc, &dt_); if (rc) return rc;
    rc = stage_in(c, SLOT_IN3, q, (size_t)ncol * nlev * 4, loc, &dq); if (rc) return rc;
    rc = stage_out(c, SLOT_OUT0, p_out, (size_t)ncol * nlev * 8, loc, &dp); if (rc) return rc;
    rc = stage_out(c, SLOT_OUT1, zs_out, (size_t)ncol * nlev * 8, loc, &dh); if (rc) return rc;
    hipLaunchKernelGGL(ecmwf_levels_kernel, dim3(grid_for(ncol, 256, c->num_cus * 8)), dim3(256), 0, c->stream, (const float*)dz, (const float*)ds,
                       (const float*)dt_, (const float*)dq, dl, da, db, (int)nlev, ny, nx, R_d, (double*)dp, (double*)dh);
    HIPCHECK(c, hipGetLastError());
    rc = finish_out(c, p_out, dp, (size_t)ncol * nlev * 8, loc); if (rc) return rc;
    rc = finish_out(c, zs_out, dh, (size_t)ncol * nlev * 8, loc); if (rc) return rc;
    HIPCHECK(c, hipStreamSynchronize(c->stream));       // the tables in SLOT_AUX must outlive the kernel
    return RDR_OK;
}

int rdr_orbit_look_vectors(rdr_ctx* c, const double* sv_t, const double* sv_pos, const double* sv_vel, int64_t nsv, const double* xyz,
                           int64_t n, double threshold, int maxiter, double* los, double* aztime, double* srange, int loc) {
    if (!c || !sv_t || !sv_pos || !sv_vel || !xyz || !los) return fail(c, RDR_ERR_INVALID, "rdr_orbit_look_vectors: NULL argument");
    if (nsv < 4) return fail(c, RDR_ERR_INVALID, "state_to_los: At least 4 state vectors are required for orbit interpolation");
    for (int64_t i = 1; i < nsv; ++i) if (!(sv_t[i] > sv_t[i - 1])) return fail(c, RDR_ERR_INVALID, "rdr_orbit_look_vectors: state-vector times must be strictly increasing");
    if (n < 0) return fail(c, RDR_ERR_INVALID, "rdr_orbit_look_vectors: negative count");
    if (n == 0) return RDR_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    // the (small) state-vector table always comes from the host; targets / outputs follow `loc`
    void* dtab;
    int rc = ensure(c, SLOT_AUX, (size_t)nsv * 7 * 8, &dtab); if (rc) return rc;
    double* dt_ = (double*)dtab; double* dp_ = dt_ + nsv; double* dv_ = dp_ + 3 * nsv;
    HIPCHECK(c, hipMemcpyAsync(dt_, sv_t, (size_t)nsv * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, hipMemcpyAsync(dp_, sv_pos, (size_t)nsv * 24, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, hipMemcpyAsync(dv_, sv_vel, (size_t)nsv * 24, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    const void* dx; void *dl, *da = nullptr, *dr = nullptr;
    rc = stage_in(c, SLOT_IN0, xyz, (size_t)n * 24, loc, &dx); if (rc) return rc;
    rc = stage_out(c, SLOT_OUT0, los, (size_t)n * 24, loc, &dl); if (rc) return rc;
    if (aztime) { rc = stage_out(c, SLOT_OUT1, aztime, (size_t)n * 8, loc, &da); if (rc) return rc; }
    if (srange) { rc = stage_out(c, SLOT_OUT2, srange, (size_t)n * 8, loc, &dr); if (rc) return rc; }
    if (nsv <= ORBIT_LDS_MAX_SV) {
        // state vectors + per-segment reciprocals in LDS, division-free Hermite evaluation (orbit_los_fast_kernel)
        const size_t smem = ((size_t)nsv * 7 + (size_t)(nsv - 3) * 16) * sizeof(double);
        hipLaunchKernelGGL(orbit_los_fast_kernel, dim3(grid_for(n, 256, c->num_cus * 8)), dim3(256), smem, c->stream, dt_, dp_, dv_, (int)nsv,
                           (const double*)dx, n, threshold, maxiter, (double*)dl, (double*)da, (double*)dr);
    } else {
        hipLaunchKernelGGL(orbit_los_kernel, dim3(grid_for(n, 256, c->num_cus * 8)), dim3(256), 0, c->stream, dt_, dp_, dv_, (int)nsv,
                           (const double*)dx, n, threshold, maxiter, (double*)dl, (double*)da, (double*)dr);
    }
    HIPCHECK(c, hipGetLastError());
    rc = finish_out(c, los, dl, (size_t)n * 24, loc); if (rc) return rc;
    if (aztime) { rc = finish_out(c, aztime, da, (size_t)n * 8, loc); if (rc) return rc; }
    if (srange) { rc = finish_out(c, srange, dr, (size_t)n * 8, loc); if (rc) return rc; }
    if (loc == RDR_HOST) HIPCHECK(c, hipStreamSynchronize(c->stream));
    return RDR_OK;
}

// ---- native extensions ------------------------------------------------------------------------------------
int rdr_interp_nd(rdr_ctx* c, int32_t ndim, const double* const* axes, const int64_t* axis_len, const double* values,
                  const double* q, int64_t n, int has_fill, double fill, double* out, int loc) {
    if (!c || !axes || !axis_len || !values || (n > 0 && (!q || !out))) return fail(c, RDR_ERR_INVALID, "rdr_interp_nd: NULL argument");
    if (ndim < 1 || ndim > 8) return fail(c, RDR_ERR_INVALID, "rdr_interp_nd: 1 <= ndim <= 8 supported on device");
    if (n < 0) return fail(c, RDR_ERR_INVALID, "rdr_interp_nd: negative count");
    if (n == 0) return RDR_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    NdParams P; std::memset(&P, 0, sizeof(P));
    P.ndim = ndim;
    int64_t tot_axes = 0, nvals = 1;
    for (int d = 0; d < ndim; ++d) {
        if (axis_len[d] < 2) return fail(c, RDR_ERR_INVALID, "rdr_interp_nd: every axis needs >= 2 points");
        P.len[d] = axis_len[d]; P.off[d] = tot_axes; tot_axes += axis_len[d]; nvals *= axis_len[d];
    }
    int64_t s = 1;
    for (int d = ndim - 1; d >= 0; --d) { P.stride[d] = s; s *= axis_len[d]; }
    // axes are always host-side tuples of small 1-D arrays in the reference API; accept device too
    void* dax;
    int rc = ensure(c, SLOT_AUX, (size_t)tot_axes * 8, &dax); if (rc) return rc;
    for (int d = 0; d < ndim; ++d)
        HIPCHECK(c, hipMemcpyAsync((double*)dax + P.off[d], axes[d], (size_t)axis_len[d] * 8,
                                   loc == RDR_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, c->stream));
    const void *dv, *dq; void* dout;
    rc = stage_in(c, SLOT_IN0, values, (size_t)nvals * 8, loc, &dv); if (rc) return rc;
    rc = stage_in(c, SLOT_IN1, q, (size_t)n * ndim * 8, loc, &dq); if (rc) return rc;
    rc = stage_out(c, SLOT_OUT0, out, (size_t)n * 8, loc, &dout); if (rc) return rc;
    {
        KTimer t(c, 2);
        hipLaunchKernelGGL(interp_nd_kernel, dim3(grid_for(n, 256, c->num_cus * 8)), dim3(256), 0, c->stream, P, (const double*)dax,
                           (const double*)dv, (const double*)dq, n, has_fill, fill, (double*)dout);
    }
    HIPCHECK(c, hipGetLastError());
    rc = finish_out(c, out, dout, (size_t)n * 8, loc); if (rc) return rc;
    if (loc == RDR_HOST) HIPCHECK(c, hipStreamSynchronize(c->stream));
    return RDR_OK;
}

int rdr_interp_along_axis(rdr_ctx* c, const double* points, const double* values, int64_t ncol, int64_t m, const double* q,
                          int64_t mq, int has_fill, double fill, double* out, int loc) {
    if (!c || !points || !values || !q || !out) return fail(c, RDR_ERR_INVALID, "rdr_interp_along_axis: NULL argument");
    if (m < 2) return fail(c, RDR_ERR_INVALID, "rdr_interp_along_axis: axis needs >= 2 points");
    if (ncol < 0 || mq < 0) return fail(c, RDR_ERR_INVALID, "rdr_interp_along_axis: negative count");
    if (ncol * mq == 0) return RDR_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    const void *dp, *dv, *dq; void* dout;
    int rc = stage_in(c, SLOT_IN0, points, (size_t)ncol * m * 8, loc, &dp); if (rc) return rc;
    rc = stage_in(c, SLOT_IN1, values, (size_t)ncol * m * 8, loc, &dv); if (rc) return rc;
    rc = stage_in(c, SLOT_IN2, q, (size_t)ncol * mq * 8, loc, &dq); if (rc) return rc;
    rc = stage_out(c, SLOT_OUT0, out, (size_t)ncol * mq * 8, loc, &dout); if (rc) return rc;
    hipLaunchKernelGGL(along_axis_kernel, dim3(grid_for(ncol * mq, 256, c->num_cus * 8)), dim3(256), 0, c->stream, (const double*)dp,
                       (const double*)dv, ncol, m, (const double*)dq, mq, has_fill, fill, (double*)dout);
    HIPCHECK(c, hipGetLastError());
    rc = finish_out(c, out, dout, (size_t)ncol * mq * 8, loc); if (rc) return rc;
    if (loc == RDR_HOST) HIPCHECK(c, hipStreamSynchronize(c->stream));
    return RDR_OK;
}

int64_t rdr_make_points_count(double max_len, double step) {
    // makePoints.pyx:30-33: Npts = int(max_len//step) (+1 if max_len % step != 0); python float // and %
    if (!(step > 0) || !(max_len >= 0)) return -1;
    double mod = std::fmod(max_len, step);
    double div = (max_len - mod) / step;
    double fl = std::floor(div);
    if (div - fl > 0.5) fl += 1.0;
    return (int64_t)fl + (mod != 0.0 ? 1 : 0);
}

int rdr_make_points(rdr_ctx* c, double max_len, const double* sp, const double* slv, int64_t nrays, double step, double* out, int loc) {
    if (!c || !sp || !slv || !out) return fail(c, RDR_ERR_INVALID, "rdr_make_points: NULL argument");
    const int64_t npts = rdr_make_points_count(max_len, step);
    if (npts < 0) return fail(c, RDR_ERR_INVALID, "rdr_make_points: need max_len >= 0 and step > 0");
    if (nrays < 0) return fail(c, RDR_ERR_INVALID, "rdr_make_points: negative count");
    if (nrays * npts == 0) return RDR_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    const void *ds, *dl; void* dout;
    int rc = stage_in(c, SLOT_IN0, sp, (size_t)nrays * 24, loc, &ds); if (rc) return rc;
    rc = stage_in(c, SLOT_IN1, slv, (size_t)nrays * 24, loc, &dl); if (rc) return rc;
    rc = stage_out(c, SLOT_OUT0, out, (size_t)nrays * 3 * npts * 8, loc, &dout); if (rc) return rc;
    {
        KTimer t(c, 3);
        hipLaunchKernelGGL(make_points_kernel, dim3(grid_for(nrays * 3 * npts, 256, c->num_cus * 8)), dim3(256), 0, c->stream,
                           (const double*)ds, (const double*)dl, nrays, npts, step, (double*)dout);
    }
    HIPCHECK(c, hipGetLastError());
    rc = finish_out(c, out, dout, (size_t)nrays * 3 * npts * 8, loc); if (rc) return rc;
    if (loc == RDR_HOST) HIPCHECK(c, hipStreamSynchronize(c->stream));
    return RDR_OK;
}

