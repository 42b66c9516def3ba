// api.hip -- host side of the C ABI declared in include/x265_hip.h.
//
// A context owns one HIP stream, the plane / cost-table directory the kernels read
// (DeviceTables) and grow-on-demand device + pinned-host staging for the host-array entry points.
// The primitive-level thunks share one process-wide context and serialise on its mutex.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <math.h>
#include <mutex>
#include <new>
#include <shared_mutex>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

#include "kernels.h"
#include "me_device.h"

using namespace x265hip;

namespace {

thread_local char t_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err, sizeof(t_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do                                                                                                  \
    {                                                                                                   \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(e_ == hipErrorOutOfMemory ? X265_HIP_ENOMEM : X265_HIP_ERUNTIME, "%s: %s", #expr, \
                        hipGetErrorString(e_));                                                         \
    } while (0)

// A kernel launch reports failure only through hipGetLastError(), which also returns whatever an earlier,
// unrelated runtime call of this thread (the caller's, a framework's) left behind: drop that first.
#define LAUNCH(call)                    \
    do                                  \
    {                                   \
        (void)hipGetLastError();        \
        call;                           \
        HIP_TRY(hipGetLastError());     \
    } while (0)

const uint8_t k_part_w[X265_HIP_NUM_LUMA_PARTITIONS] = { 4, 8, 8, 4, 16, 16, 8, 16, 12, 16, 4, 32, 32, 16, 32,
                                                         24, 32, 8, 64, 64, 32, 64, 48, 64, 16 };
const uint8_t k_part_h[X265_HIP_NUM_LUMA_PARTITIONS] = { 4, 8, 4, 8, 16, 8, 16, 12, 16, 4, 16, 32, 16, 32, 24,
                                                         32, 8, 32, 64, 32, 64, 48, 64, 16, 64 };

struct Buf
{
    void *d = nullptr;
    void *h = nullptr; // pinned mirror
    size_t cap = 0;
};

} // namespace

struct x265_hip_ctx
{
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;            // me_big_kernel runs here, beside me_kernel on `stream`
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t copy = nullptr;            // asynchronous plane uploads / result downloads (made at their first use)
    hipEvent_t ev_copy = nullptr, ev_compute = nullptr;
    std::mutex mu;
    DeviceTables host_tabs;
    DeviceTables *dev_tabs = nullptr;
    bool tabs_dirty = true;
    uint32_t *batch_counter = nullptr; // batch counters of the cost kernels (two sets, see launch_cost)
    int batch_parity = 0;
    int route_parity = 0; // counter set of the next motionEstimate launch (me_kernel.hip)
    int noalloc_slot = -1; // this context's part of the GPU's `scratch` array (taken at the first noalloc call)
    int noalloc_parity = 0;
    int check_records = 0; // record-check mode of the device-array entry points: 0 off, 1 copy-back, 2 device guard
    uint32_t *guard_words = nullptr; // the guard's two status words (device)
    void *owned_plane[X265_HIP_MAX_PLANES] = {};
    size_t plane_bytes[X265_HIP_MAX_PLANES] = {};
    int plane_origin[X265_HIP_MAX_PLANES] = {};
    uint16_t *owned_tab[X265_HIP_MAX_QP] = {};
    Buf in, out;       // staging of the host-array batch calls
    Buf mejobs;        // motionEstimate jobs made on the device by x265_hip_inter_search_batch (device side only is used)
    Buf route;         // motionEstimate launches: the list of large-block jobs me_route_kernel makes for me_big_kernel (device side only)
    Buf guard;         // record guard of the request lists: the list the kernel runs and a byte per record (device side only)
    Buf bidir;         // x265_hip_inter_decide_batch: two bidir requests and two costs per unit (device side only)
    float *bitsizes = nullptr; // BitCost::s_bitsizes (device)
    Buf look;          // x265_hip_lookahead_batch: row counters (64 bytes per search) behind one status word
    Buf prim_a, prim_b; // staging of the primitive thunks
};

static const uint32_t k_guard_clear[2] = { 0u, 0xffffffffu };

// ---- scratch of the entry point that allocates nothing (x265_hip_cost_batch_device_noalloc) --------------
// The cost kernels hand their batches out through counters in device memory.  Contexts keep theirs in a buffer
// from hipMalloc; callers that cannot allocate get a slot of this array instead: one copy of the array per GPU, held
// until the process ends, X265HIP_NOALLOC_SLOTS slots of X265HIP_COST_COUNTER_BYTES (512 KiB in all).  A context
// takes a slot of its GPU's copy at its first noalloc call and keeps it until it is destroyed, so contexts never
// share counters: calls of different contexts may overlap on one GPU (with one shared set a memset from one context
// zeroed the hand-out counters of another's kernel in flight, which then skipped batches without any error).  In host
// code `scratch` is only a handle for hipGetSymbolAddress; the device address it yields (one per device) is what
// host code passes on.
#define X265HIP_MAX_DEVICES 64
#define X265HIP_NOALLOC_SLOTS 64
static __device__ __align__(256) uint32_t scratch[X265HIP_NOALLOC_SLOTS * (X265HIP_COST_COUNTER_BYTES / sizeof(uint32_t))];

namespace {

uint32_t *g_scratch_ptr[X265HIP_MAX_DEVICES] = {};
uint64_t g_scratch_used[X265HIP_MAX_DEVICES] = {}; // a bit per slot
std::mutex g_scratch_mu;
static_assert(X265HIP_NOALLOC_SLOTS <= 64, "one bit per slot");

// once per device, with that device current
int scratch_setup(int device)
{
    if (device < 0 || device >= X265HIP_MAX_DEVICES)
        return 0; // such a device gets no scratch: the entry point refuses it
    std::lock_guard<std::mutex> g(g_scratch_mu);
    if (!g_scratch_ptr[device])
    {
        void *p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p)
            return fail(X265_HIP_ERUNTIME, "hipGetSymbolAddress(scratch) returned a null address on device %d", device);
        g_scratch_ptr[device] = (uint32_t *)p;
    }
    return 0;
}

// the context's slot of its GPU's array (device current), taken and cleared at the first call; null when none is left
uint32_t *scratch_slot(x265_hip_ctx *c)
{
    if (c->device < 0 || c->device >= X265HIP_MAX_DEVICES || !g_scratch_ptr[c->device])
        return nullptr;
    if (c->noalloc_slot < 0)
    {
        std::lock_guard<std::mutex> g(g_scratch_mu);
        for (int k = 0; k < X265HIP_NOALLOC_SLOTS && c->noalloc_slot < 0; k++)
            if (!(g_scratch_used[c->device] >> k & 1))
            {
                g_scratch_used[c->device] |= 1ull << k;
                c->noalloc_slot = k;
            }
        if (c->noalloc_slot < 0)
            return nullptr;
        // the slot's last owner left its counts behind: both sets start at zero (on the stream, in front of the launch;
        // from then on every launch clears the set of the next one)
        uint32_t *slot = g_scratch_ptr[c->device] + (size_t)c->noalloc_slot * (X265HIP_COST_COUNTER_BYTES / sizeof(uint32_t));
        if (hipMemsetAsync(slot, 0, X265HIP_COST_COUNTER_BYTES, c->stream) != hipSuccess)
        {
            g_scratch_used[c->device] &= ~(1ull << c->noalloc_slot);
            c->noalloc_slot = -1;
            return nullptr;
        }
        c->noalloc_parity = 0;
    }
    return g_scratch_ptr[c->device] + (size_t)c->noalloc_slot * (X265HIP_COST_COUNTER_BYTES / sizeof(uint32_t));
}

void scratch_release(x265_hip_ctx *c)
{
    if (c->noalloc_slot < 0)
        return;
    std::lock_guard<std::mutex> g(g_scratch_mu);
    g_scratch_used[c->device] &= ~(1ull << c->noalloc_slot);
    c->noalloc_slot = -1;
}

// `pinned`: also a pinned host mirror of the same size (the staging of the host-array calls); buffers that only the
// device reads and writes (routing lists, jobs made on the device, lookahead counters) go without
int ensure(x265_hip_ctx *c, Buf &b, size_t bytes, bool pinned = true)
{
    if (bytes <= b.cap)
        return 0;
    size_t cap = bytes + bytes / 2 + 4096;
    if (b.d)
        (void)hipFree(b.d);
    if (b.h)
        (void)hipHostFree(b.h);
    b.d = b.h = nullptr;
    b.cap = 0;
    HIP_TRY(hipMalloc(&b.d, cap));
    if (pinned)
        HIP_TRY(hipHostMalloc(&b.h, cap, hipHostMallocDefault));
    b.cap = cap;
    (void)c;
    return 0;
}

// `launches`: whether kernels will run (a call without jobs must not take a counter set: the set is cleared for the
// launch after next by the launch that uses it)
MeRoute me_route(x265_hip_ctx *c, bool launches, bool any_routed = true, bool guarded = false)
{
    if (!launches || !any_routed)
        return MeRoute{ nullptr, 0, c->stream, nullptr, nullptr, nullptr, nullptr };
    return MeRoute{ (uint32_t *)c->route.d, c->route_parity++, c->stream, c->side, c->ev_fork, c->ev_join, guarded ? c->guard_words : nullptr };
}

template <typename Job>
bool any_routed_job(const Job *jobs, int n)
{
    for (int i = 0; i < n; i++)
        if (me_job_is_routed(jobs[i].pu_w, jobs[i].pu_h, jobs[i].method))
            return true;
    return false;
}

// room for the routing list of an n-job motionEstimate launch
int ensure_route(x265_hip_ctx *c, int n)
{
    const size_t bytes = me_route_words(n) * sizeof(uint32_t);
    if (c->route.cap >= bytes)
        return 0;
    HIP_TRY(hipStreamSynchronize(c->stream)); // an earlier launch may still read the buffer that is about to be replaced
    int r = ensure(c, c->route, bytes, false);
    if (r)
        return r;
    HIP_TRY(hipMemsetAsync(c->route.d, 0, me_route_counter_words() * sizeof(uint32_t), c->stream)); // both counter sets
    c->route_parity = 0;
    return 0;
}

int push_tabs(x265_hip_ctx *c)
{
    if (!c->tabs_dirty)
        return 0;
    HIP_TRY(hipMemcpyAsync(c->dev_tabs, &c->host_tabs, sizeof(DeviceTables), hipMemcpyHostToDevice, c->stream));
    // host_tabs may change right after this call returns
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->tabs_dirty = false;
    return 0;
}

bool is_gfx950(int device)
{
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess)
        return false;
    return strncmp(p.gcnArchName, "gfx950", 6) == 0;
}

} // namespace

extern "C" {

const char *x265_hip_last_error(void) { return t_err; }

int x265_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

int x265_hip_partition_from_sizes(int width, int height)
{
    for (int p = 0; p < X265_HIP_NUM_LUMA_PARTITIONS; p++)
        if (k_part_w[p] == width && k_part_h[p] == height)
            return p;
    return -1;
}

int x265_hip_ctx_create(int device, x265_hip_ctx **out)
{
    if (!out)
        return fail(X265_HIP_EINVAL, "null out pointer");
    *out = nullptr;
    int n = x265_hip_device_count();
    if (device < 0 || device >= n)
        return fail(X265_HIP_ENODEV, "device %d not present (%d HIP devices)", device, n);
    if (!is_gfx950(device))
        return fail(X265_HIP_ENODEV, "device %d is not gfx950; this library carries gfx950 code only", device);
    HIP_TRY(hipSetDevice(device));
    int sr = scratch_setup(device);
    if (sr)
        return sr;
    x265_hip_ctx *c = new (std::nothrow) x265_hip_ctx;
    if (!c)
        return fail(X265_HIP_ENOMEM, "out of host memory");
    c->device = device;
    memset(&c->host_tabs, 0, sizeof(c->host_tabs));
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess && !(getenv("X265_HIP_ME_ONE_STREAM") && atoi(getenv("X265_HIP_ME_ONE_STREAM"))))
    {
        // X265_HIP_ME_SIDE_PRIORITY=1: the side stream at the highest priority, an experiment of round 2 that hung once
        // in the GPU suite and is kept only to examine that; never set in production
        if (getenv("X265_HIP_ME_SIDE_PRIORITY") && atoi(getenv("X265_HIP_ME_SIDE_PRIORITY")))
        {
            int least = 0, greatest = 0;
            e = hipDeviceGetStreamPriorityRange(&least, &greatest);
            if (e == hipSuccess)
                e = hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, greatest);
        }
        else
            e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
    }
    if (e == hipSuccess)
        e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess)
        e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
    if (e == hipSuccess)
        e = hipMalloc((void **)&c->dev_tabs, sizeof(DeviceTables));
    if (e == hipSuccess)
        e = hipMalloc((void **)&c->batch_counter, X265HIP_COST_COUNTER_BYTES);
    if (e == hipSuccess)
        e = hipMemset(c->batch_counter, 0, X265HIP_COST_COUNTER_BYTES);
    if (e == hipSuccess)
        e = hipMalloc((void **)&c->bitsizes, 32769 * sizeof(float));
    if (e == hipSuccess)
    {
        // BitCost::CalculateLogs (bitcost.cpp:68-80), in the reference's own precision: see x265_hip_costtab_build
        std::vector<float> bs(32769);
        const float scale = (float)(2.0f / log((double)2.0f));
        bs[0] = 0.718f;
        for (int i = 1; i <= 32768; i++)
            bs[i] = (float)(log((double)(float)(i + 1)) * scale + 1.718f);
        e = hipMemcpy(c->bitsizes, bs.data(), bs.size() * sizeof(float), hipMemcpyHostToDevice);
        c->host_tabs.bitsizes = c->bitsizes;
        c->tabs_dirty = true;
    }
    if (e == hipSuccess)
        e = hipMalloc((void **)&c->guard_words, 2 * sizeof(uint32_t));
    if (e == hipSuccess)
        e = hipMemcpy(c->guard_words, k_guard_clear, sizeof(k_guard_clear), hipMemcpyHostToDevice);
    if (e != hipSuccess)
    {
        x265_hip_ctx_destroy(c);
        return fail(X265_HIP_ERUNTIME, "context setup: %s", hipGetErrorString(e));
    }
    c->check_records = getenv("X265_HIP_CHECK_RECORDS") ? atoi(getenv("X265_HIP_CHECK_RECORDS")) : 0;
    if (c->check_records < 0 || c->check_records > 2)
        c->check_records = 1;
    if (fullsearch2_setup() != 0)
    {
        x265_hip_ctx_destroy(c);
        return fail(X265_HIP_ERUNTIME, "fullsearch kernel attribute setup failed");
    }
    *out = c;
    return X265_HIP_OK;
}

void x265_hip_ctx_destroy(x265_hip_ctx *c)
{
    if (!c)
        return;
    (void)hipSetDevice(c->device);
    if (c->stream)
        (void)hipStreamSynchronize(c->stream);
    if (c->side)
        (void)hipStreamSynchronize(c->side);
    if (c->copy)
        (void)hipStreamSynchronize(c->copy);
    scratch_release(c); // (its last kernel is over)
    for (int i = 0; i < X265_HIP_MAX_PLANES; i++)
        if (c->owned_plane[i])
            (void)hipFree(c->owned_plane[i]);
    for (int i = 0; i < X265_HIP_MAX_QP; i++)
        if (c->owned_tab[i])
            (void)hipFree(c->owned_tab[i]);
    Buf *bufs[9] = { &c->in, &c->out, &c->prim_a, &c->prim_b, &c->mejobs, &c->look, &c->route, &c->bidir, &c->guard };
    for (Buf *b : bufs)
    {
        if (b->d)
            (void)hipFree(b->d);
        if (b->h)
            (void)hipHostFree(b->h);
    }
    if (c->dev_tabs)
        (void)hipFree(c->dev_tabs);
    if (c->batch_counter)
        (void)hipFree(c->batch_counter);
    if (c->guard_words)
        (void)hipFree(c->guard_words);
    if (c->bitsizes)
        (void)hipFree(c->bitsizes);
    if (c->ev_fork)
        (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join)
        (void)hipEventDestroy(c->ev_join);
    if (c->ev_copy)
        (void)hipEventDestroy(c->ev_copy);
    if (c->ev_compute)
        (void)hipEventDestroy(c->ev_compute);
    if (c->copy)
        (void)hipStreamDestroy(c->copy);
    if (c->side)
        (void)hipStreamDestroy(c->side);
    if (c->stream)
        (void)hipStreamDestroy(c->stream);
    delete c;
}

void *x265_hip_ctx_stream(x265_hip_ctx *c) { return c ? (void *)c->stream : nullptr; }

int x265_hip_ctx_check_records(x265_hip_ctx *c, int on)
{
    if (!c)
        return fail(X265_HIP_EINVAL, "null context");
    if (on < 0 || on > 2)
        return fail(X265_HIP_EINVAL, "check_records: mode %d (0 off, 1 copy-back, 2 device guard)", on);
    std::lock_guard<std::mutex> g(c->mu);
    c->check_records = on;
    return X265_HIP_OK;
}

// the guard's words, read and cleared (the stream is idle)
static int guard_take(x265_hip_ctx *c, uint32_t w[2])
{
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpy(w, c->guard_words, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (w[0])
        HIP_TRY(hipMemcpy(c->guard_words, k_guard_clear, sizeof(k_guard_clear), hipMemcpyHostToDevice));
    return 0;
}

int x265_hip_ctx_guard_status(x265_hip_ctx *c, uint32_t *refused, uint32_t *first_index)
{
    if (!c || !refused || !first_index)
        return fail(X265_HIP_EINVAL, "guard_status: null argument");
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipStreamSynchronize(c->stream));
    uint32_t w[2];
    int r = guard_take(c, w);
    if (r)
        return r;
    *refused = w[0];
    *first_index = w[1];
    return X265_HIP_OK;
}

int x265_hip_ctx_sync(x265_hip_ctx *c)
{
    if (!c)
        return fail(X265_HIP_EINVAL, "null context");
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->check_records == 2)
    {
        // what the guard refused since the words were last read: reported once, here
        std::lock_guard<std::mutex> g(c->mu);
        uint32_t w[2];
        int r = guard_take(c, w);
        if (r)
            return r;
        if (w[0])
            return fail(X265_HIP_ERANGE, "the record guard refused %u record(s), the first at index %u of its list: their results are -1",
                        w[0], w[1]);
    }
    return X265_HIP_OK;
}

// ---- planes and cost tables ---------------------------------------------------------------------

// drop plane `id` from the directory and free its buffer if the context owns it: after this no validated request
// can name the plane, whatever happens to the allocation that usually follows
static void plane_clear(x265_hip_ctx *c, int id)
{
    if (c->owned_plane[id])
        (void)hipFree(c->owned_plane[id]);
    c->owned_plane[id] = nullptr;
    c->host_tabs.planes[id].origin = nullptr;
    c->host_tabs.planes[id].stride = 0;
    c->plane_bytes[id] = 0;
    c->plane_origin[id] = 0;
    c->host_tabs.span_first[id] = c->host_tabs.span_end[id] = 0;
    c->tabs_dirty = true;
}

static int plane_set(x265_hip_ctx *c, int id, const void *dev, size_t bytes, int stride, int origin)
{
    if (stride <= 0 || origin < 0 || (size_t)origin >= bytes)
        return fail(X265_HIP_EINVAL, "plane %d: bad stride/origin (%d, %d, %zu bytes)", id, stride, origin, bytes);
    c->host_tabs.planes[id].origin = (const uint8_t *)dev + origin;
    c->host_tabs.planes[id].stride = stride;
    c->plane_bytes[id] = bytes;
    c->plane_origin[id] = origin;
    c->host_tabs.span_first[id] = -(long)origin;
    c->host_tabs.span_end[id] = (long)bytes - origin;
    c->tabs_dirty = true;
    return X265_HIP_OK;
}

int x265_hip_plane_release(x265_hip_ctx *c, int id)
{
    if (!c || id < 0 || id >= X265_HIP_MAX_PLANES)
        return fail(X265_HIP_EINVAL, "bad plane id %d", id);
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    plane_clear(c, id);
    return X265_HIP_OK;
}

int x265_hip_plane_upload(x265_hip_ctx *c, int id, const uint8_t *host_buf, size_t bytes, int stride, int origin)
{
    if (!c || !host_buf || id < 0 || id >= X265_HIP_MAX_PLANES || bytes == 0)
        return fail(X265_HIP_EINVAL, "plane_upload: bad argument (id %d)", id);
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!c->owned_plane[id] || c->plane_bytes[id] != bytes)
    {
        plane_clear(c, id);
        // 64 bytes of slack: dword loads at unaligned byte addresses may run past the last sample
        HIP_TRY(hipMalloc(&c->owned_plane[id], bytes + 64));
        HIP_TRY(hipMemsetAsync((uint8_t *)c->owned_plane[id] + bytes, 0, 64, c->stream));
    }
    HIP_TRY(hipMemcpyAsync(c->owned_plane[id], host_buf, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return plane_set(c, id, c->owned_plane[id], bytes, stride, origin);
}

int x265_hip_plane_bind(x265_hip_ctx *c, int id, const void *device_buf, size_t bytes, int stride, int origin)
{
    if (!c || !device_buf || id < 0 || id >= X265_HIP_MAX_PLANES || bytes == 0)
        return fail(X265_HIP_EINVAL, "plane_bind: bad argument (id %d)", id);
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    plane_clear(c, id);
    return plane_set(c, id, device_buf, bytes, stride, origin);
}

int x265_hip_plane_download(x265_hip_ctx *c, int id, uint8_t *host_buf, size_t bytes)
{
    if (!c || !host_buf || id < 0 || id >= X265_HIP_MAX_PLANES)
        return fail(X265_HIP_EINVAL, "plane_download: bad argument (id %d)", id);
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->host_tabs.planes[id].origin || bytes > c->plane_bytes[id])
        return fail(X265_HIP_EINVAL, "plane_download: plane %d is not set or smaller than %zu bytes", id, bytes);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(host_buf, c->host_tabs.planes[id].origin - c->plane_origin[id], bytes, hipMemcpyDeviceToHost,
                           c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return X265_HIP_OK;
}

// ---- asynchronous transfers: planes in and results out beside the kernels ----------------------------------------
// A sequence runner keeps two sets of planes: while the searches of one closed GOP segment run on the context's
// stream, the pictures of the next segment arrive in the other set over a copy stream of the context, and finished
// results leave over it.  (The reference's analogue is the frame-parallel pipeline, encoder/frameencoder.cpp:1039-1099:
// a frame encoder starts on a row as soon as the rows it references are there.)  Ordering is the caller's, with two
// fences: x265_hip_copy_wait_compute (transfers queued from now on start after every kernel queued so far) and
// x265_hip_compute_wait_copy (kernels queued from now on start after every transfer queued so far).

static int copy_stream(x265_hip_ctx *c)
{
    if (c->copy)
        return 0;
    HIP_TRY(hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_compute, hipEventDisableTiming));
    return 0;
}

int x265_hip_host_alloc(size_t bytes, void **out)
{
    if (!out || bytes == 0)
        return fail(X265_HIP_EINVAL, "host_alloc: bad argument");
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess)
        return fail(X265_HIP_ENOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return X265_HIP_OK;
}

void x265_hip_host_free(void *p)
{
    if (p)
        (void)hipHostFree(p);
}

int x265_hip_plane_alloc(x265_hip_ctx *c, int id, size_t bytes, int stride, int origin)
{
    if (!c || id < 0 || id >= X265_HIP_MAX_PLANES || bytes == 0)
        return fail(X265_HIP_EINVAL, "plane_alloc: bad argument (id %d)", id);
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!c->owned_plane[id] || c->plane_bytes[id] != bytes)
    {
        plane_clear(c, id);
        HIP_TRY(hipMalloc(&c->owned_plane[id], bytes + 64));
        HIP_TRY(hipMemsetAsync((uint8_t *)c->owned_plane[id] + bytes, 0, 64, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return plane_set(c, id, c->owned_plane[id], bytes, stride, origin);
}

int x265_hip_plane_upload_async(x265_hip_ctx *c, int id, const uint8_t *host_buf, size_t bytes)
{
    if (!c || !host_buf || id < 0 || id >= X265_HIP_MAX_PLANES)
        return fail(X265_HIP_EINVAL, "plane_upload_async: bad argument (id %d)", id);
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->owned_plane[id] || c->plane_bytes[id] != bytes)
        return fail(X265_HIP_EINVAL, "plane_upload_async: plane %d is not an owned plane of %zu bytes (x265_hip_plane_alloc / x265_hip_plane_upload first)",
                    id, bytes);
    HIP_TRY(hipSetDevice(c->device));
    int r = copy_stream(c);
    if (r)
        return r;
    HIP_TRY(hipMemcpyAsync(c->owned_plane[id], host_buf, bytes, hipMemcpyHostToDevice, c->copy));
    return X265_HIP_OK;
}

int x265_hip_download_async(x265_hip_ctx *c, const void *d_src, void *host_dst, size_t bytes)
{
    if (!c || !d_src || !host_dst)
        return fail(X265_HIP_EINVAL, "download_async: bad argument");
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    int r = copy_stream(c);
    if (r)
        return r;
    HIP_TRY(hipMemcpyAsync(host_dst, d_src, bytes, hipMemcpyDeviceToHost, c->copy));
    return X265_HIP_OK;
}

int x265_hip_copy_wait_compute(x265_hip_ctx *c)
{
    if (!c)
        return fail(X265_HIP_EINVAL, "null context");
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    int r = copy_stream(c);
    if (r)
        return r;
    HIP_TRY(hipEventRecord(c->ev_compute, c->stream));
    HIP_TRY(hipStreamWaitEvent(c->copy, c->ev_compute, 0));
    return X265_HIP_OK;
}

int x265_hip_compute_wait_copy(x265_hip_ctx *c)
{
    if (!c)
        return fail(X265_HIP_EINVAL, "null context");
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    int r = copy_stream(c);
    if (r)
        return r;
    HIP_TRY(hipEventRecord(c->ev_copy, c->copy));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_copy, 0));
    return X265_HIP_OK;
}

int x265_hip_copy_sync(x265_hip_ctx *c)
{
    if (!c)
        return fail(X265_HIP_EINVAL, "null context");
    if (c->copy)
        HIP_TRY(hipStreamSynchronize(c->copy));
    return X265_HIP_OK;
}

int x265_hip_lowres_geometry_of(int width, int height, int margin_x, int margin_y, x265_hip_lowres_geometry *out)
{
    if (!out || width < 16 || height < 16 || margin_x < 0 || margin_y < 0)
        return fail(X265_HIP_EINVAL, "lowres_geometry_of: bad argument");
    // Lowres::create, lowres.cpp:33-48
    int w = width / 2, l = height / 2;
    int stride = w + 2 * margin_x;
    if (stride & 31)
        stride += 32 - (stride & 31);
    w = (w + 7) / 8 * 8;
    l = (l + 7) / 8 * 8;
    out->width = w;
    out->lines = l;
    out->stride = stride;
    out->origin_offset = stride * margin_y + margin_x;
    out->bytes = (size_t)stride * (l + 2 * margin_y);
    return X265_HIP_OK;
}

int x265_hip_lowres_init(x265_hip_ctx *c, int src_plane, int width, int height, int margin_x, int margin_y, int dst0)
{
    x265_hip_lowres_geometry g;
    if (!c || src_plane < 0 || src_plane >= X265_HIP_MAX_PLANES || dst0 < 0 || dst0 + 3 >= X265_HIP_MAX_PLANES ||
        (src_plane >= dst0 && src_plane <= dst0 + 3))
        return fail(X265_HIP_EINVAL, "lowres_init: bad plane ids (src %d, dst %d..%d)", src_plane, dst0, dst0 + 3);
    int r = x265_hip_lowres_geometry_of(width, height, margin_x, margin_y, &g);
    if (r)
        return r;
    if (margin_x & 1) // padded rows are written four samples at a time
        return fail(X265_HIP_EINVAL, "lowres_init: marginX must be even");
    if (g.width + 2 * margin_x > g.stride)
        return fail(X265_HIP_EINVAL, "lowres_init: rounded width %d + margins does not fit stride %d", g.width, g.stride);
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->host_tabs.planes[src_plane].origin)
        return fail(X265_HIP_EINVAL, "lowres_init: source plane %d is not set", src_plane);
    {
        // lowres_init_kernel computes every padded sample from picture positions clamped to the rounded lowres
        // size, so it reads source rows 0 .. 2 lines and, with its dword loads, columns 0 .. 2 width + 3
        const PlaneDesc &sp = c->host_tabs.planes[src_plane];
        const long last = (long)c->plane_origin[src_plane] + 2L * g.lines * sp.stride + 2L * g.width + 4;
        if (last > (long)c->plane_bytes[src_plane] + 16)
            return fail(X265_HIP_ERANGE, "lowres_init: plane %d does not cover rows 0..%d and columns 0..%d of the picture "
                        "(%d rows below and %d columns right of a %dx%d picture are read)", src_plane, 2 * g.lines,
                        2 * g.width + 3, 2 * g.lines - height + 1, 2 * g.width - width + 4, width, height);
    }
    HIP_TRY(hipSetDevice(c->device));
    LowresArgs a;
    a.src = c->host_tabs.planes[src_plane].origin;
    a.src_stride = c->host_tabs.planes[src_plane].stride;
    a.dst_stride = g.stride;
    a.width = g.width;
    a.lines = g.lines;
    a.margin_x = margin_x;
    a.margin_y = margin_y;
    for (int k = 0; k < 4; k++)
    {
        const int id = dst0 + k;
        if (!c->owned_plane[id] || c->plane_bytes[id] != g.bytes)
        {
            HIP_TRY(hipStreamSynchronize(c->stream));
            plane_clear(c, id);
            HIP_TRY(hipMalloc(&c->owned_plane[id], g.bytes + 64));
            // the reference clears the buffers (lowres.cpp:63); bytes between the padded width and the stride stay 0
            HIP_TRY(hipMemsetAsync(c->owned_plane[id], 0, g.bytes + 64, c->stream));
        }
        a.dst[k] = (uint8_t *)c->owned_plane[id];
        r = plane_set(c, id, c->owned_plane[id], g.bytes, g.stride, g.origin_offset);
        if (r)
            return r;
    }
    LAUNCH(launch_lowres_init(a, c->stream));
    return X265_HIP_OK;
}

static int border_args(x265_hip_ctx *c, const char *who, int src, int dst, int width, int height, int mx, int my,
                       PlaneBorderArgs &a)
{
    const PlaneDesc &p = c->host_tabs.planes[src];
    if (!p.origin)
        return fail(X265_HIP_EINVAL, "%s: plane %d is not set", who, src);
    if (width <= 0 || height <= 0 || (width & 3) || mx < 0 || my < 0 || (mx & 3) || (p.stride & 3) ||
        ((c->plane_origin[src] - mx) & 3))
        return fail(X265_HIP_EINVAL, "%s: width, marginX, stride and origin must be multiples of 4", who);
    // the padded picture must lie inside the buffer
    const long first = (long)c->plane_origin[src] - (long)my * p.stride - mx;
    const long last = (long)c->plane_origin[src] + (long)(height - 1 + my) * p.stride + width + mx;
    if (first < 0 || (size_t)last > c->plane_bytes[src] || width + 2 * mx > p.stride)
        return fail(X265_HIP_EINVAL, "%s: %dx%d with margins %d/%d does not fit plane %d", who, width, height, mx, my, src);
    a.src = p.origin;
    a.dst = (uint8_t *)c->host_tabs.planes[dst].origin;
    a.stride = p.stride;
    a.width = width;
    a.height = height;
    a.margin_x = mx;
    a.margin_y = my;
    a.weighted = 0;
    a.w0 = a.round = a.shift = a.offset = 0;
    a.py0 = 0;
    a.pyn = height + 2 * my;
    return 0;
}

// Picture rows row0 .. row0 + rows - 1 become available (a CTU row of a frame encoder, say): the padded rows that
// depend on them alone -- these rows with their left and right margins, the rows above the picture with row 0, the
// rows below it with the last row -- as MotionReference::applyWeight (reference.cpp:78-121) and the frame filter's
// extendRowBorder (framefilter.cpp:221, ipfilter.cpp:56-74) produce them row by row.
static int border_rows(const char *who, int height, int my, int row0, int rows, PlaneBorderArgs &a)
{
    if (row0 < 0 || rows <= 0 || row0 + rows > height)
        return fail(X265_HIP_EINVAL, "%s: rows %d .. %d outside the picture's %d", who, row0, row0 + rows - 1, height);
    const int first = row0 == 0 ? 0 : my + row0, last = row0 + rows == height ? height + 2 * my : my + row0 + rows;
    a.py0 = first;
    a.pyn = last - first;
    return 0;
}

int x265_hip_plane_extend_border(x265_hip_ctx *c, int id, int width, int height, int mx, int my)
{
    if (!c || id < 0 || id >= X265_HIP_MAX_PLANES)
        return fail(X265_HIP_EINVAL, "plane_extend_border: bad plane id %d", id);
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    PlaneBorderArgs a;
    int r = border_args(c, "plane_extend_border", id, id, width, height, mx, my, a);
    if (r)
        return r;
    LAUNCH(launch_plane_border(a, c->stream));
    return X265_HIP_OK;
}

int x265_hip_plane_extend_border_rows(x265_hip_ctx *c, int id, int width, int height, int mx, int my, int row0, int rows)
{
    if (!c || id < 0 || id >= X265_HIP_MAX_PLANES)
        return fail(X265_HIP_EINVAL, "plane_extend_border_rows: bad plane id %d", id);
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    PlaneBorderArgs a;
    int r = border_args(c, "plane_extend_border_rows", id, id, width, height, mx, my, a);
    if (r || (r = border_rows("plane_extend_border_rows", height, my, row0, rows, a)))
        return r;
    LAUNCH(launch_plane_border(a, c->stream));
    return X265_HIP_OK;
}

static int plane_weight_impl(x265_hip_ctx *c, int src, int dst, int width, int height, int mx, int my, int w0, int offset, int log2_denom,
                             int row0, int rows);

int x265_hip_plane_weight(x265_hip_ctx *c, int src, int dst, int width, int height, int mx, int my, int w0, int offset,
                          int log2_denom)
{
    return plane_weight_impl(c, src, dst, width, height, mx, my, w0, offset, log2_denom, 0, height);
}

int x265_hip_plane_weight_rows(x265_hip_ctx *c, int src, int dst, int width, int height, int mx, int my, int w0, int offset,
                               int log2_denom, int row0, int rows)
{
    return plane_weight_impl(c, src, dst, width, height, mx, my, w0, offset, log2_denom, row0, rows);
}

static int plane_weight_impl(x265_hip_ctx *c, int src, int dst, int width, int height, int mx, int my, int w0, int offset, int log2_denom,
                             int row0, int rows)
{
    if (!c || src < 0 || src >= X265_HIP_MAX_PLANES || dst < 0 || dst >= X265_HIP_MAX_PLANES || src == dst ||
        log2_denom < 0 || log2_denom > 7)
        return fail(X265_HIP_EINVAL, "plane_weight: bad argument (src %d, dst %d, log2 denominator %d)", src, dst, log2_denom);
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    if (!c->host_tabs.planes[src].origin)
        return fail(X265_HIP_EINVAL, "plane_weight: plane %d is not set", src);
    const size_t bytes = c->plane_bytes[src];
    if (!c->owned_plane[dst] || c->plane_bytes[dst] != bytes)
    {
        HIP_TRY(hipStreamSynchronize(c->stream));
        plane_clear(c, dst);
        HIP_TRY(hipMalloc(&c->owned_plane[dst], bytes + 64));
        HIP_TRY(hipMemsetAsync(c->owned_plane[dst], 0, bytes + 64, c->stream));
    }
    int r = plane_set(c, dst, c->owned_plane[dst], bytes, c->host_tabs.planes[src].stride, c->plane_origin[src]);
    PlaneBorderArgs a;
    if (r || (r = border_args(c, "plane_weight", src, dst, width, height, mx, my, a)) ||
        (r = border_rows("plane_weight", height, my, row0, rows, a)))
        return r;
    // MotionReference::init / applyWeight, reference.cpp:57-61, 95-98
    const int round = log2_denom ? 1 << (log2_denom - 1) : 0;
    a.weighted = 1;
    a.w0 = w0;
    a.round = round << 6;
    a.shift = log2_denom + 6;
    a.offset = offset;
    LAUNCH(launch_plane_border(a, c->stream));
    return X265_HIP_OK;
}

int x265_hip_costtab_upload(x265_hip_ctx *c, int qp, const uint16_t *tab)
{
    if (!c || !tab || qp < 0 || qp >= X265_HIP_MAX_QP)
        return fail(X265_HIP_EINVAL, "costtab_upload: bad argument (qp %d)", qp);
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    if (!c->owned_tab[qp])
        HIP_TRY(hipMalloc((void **)&c->owned_tab[qp], 65536 * sizeof(uint16_t)));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpyAsync(c->owned_tab[qp], tab, 65536 * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->host_tabs.costtab[qp] = c->owned_tab[qp];
    c->tabs_dirty = true;
    return X265_HIP_OK;
}

// BitCost::setQP / CalculateLogs restated (encoder/bitcost.cpp:31-80; lambda table
// Lib/TLibCommon/TComRom.cpp:127-143).  Host floating point, as in the reference: bits[i] =
// log(i+1) * (2/ln2) + 1.718 evaluated in double (the reference's unqualified log() is the double
// overload) and stored as float; cost = min(bits * lambda + 0.5, 65535) truncated to uint16.
static const double k_lambda_tab[70] = {
    0.2500, 0.2806, 0.3150, 0.3536, 0.3969, 0.4454, 0.5000, 0.5612, 0.6300, 0.7071,
    0.7937, 0.8909, 1.0000, 1.1225, 1.2599, 1.4142, 1.5874, 1.7818, 2.0000, 2.2449,
    2.5198, 2.8284, 3.1748, 3.5636, 4.0000, 4.4898, 5.0397, 5.6569, 6.3496, 7.1272,
    8.0000, 8.9797, 10.0794, 11.3137, 12.6992, 14.2544, 16.0000, 17.9594, 20.1587, 22.6274,
    25.3984, 28.5088, 32.0000, 35.9188, 40.3175, 45.2548, 50.7968, 57.0175, 64.0000, 71.8376,
    80.6349, 90.5097, 101.5937, 114.0350, 128.0000, 143.6751, 161.2699, 181.0193, 203.1873, 228.0701,
    256.0000, 287.3503, 322.5398, 362.0387, 406.3747, 456.1401, 512.0000, 574.7006, 645.0796, 724.0773
};

int x265_hip_costtab_build(int qp, uint16_t *tab)
{
    if (!tab || qp < 0 || qp >= 70)
        return fail(X265_HIP_EINVAL, "costtab_build: qp %d outside the lambda table (0..69)", qp);
    const float scale = (float)(2.0f / log((double)2.0f));
    const double lambda = k_lambda_tab[qp];
    for (int i = 0; i < 32768; i++)
    {
        const float bits = i ? (float)(log((double)(float)(i + 1)) * scale + 1.718f) : 0.718f;
        const double v = bits * lambda + 0.5f;
        const uint16_t c = (uint16_t)(v < 65535.0 ? v : 65535.0);
        tab[32768 + i] = c;
        tab[32768 - i] = c;
    }
    tab[0] = tab[1]; // difference -32768 is never written by the reference; defined here
    return X265_HIP_OK;
}

int x265_hip_costtab_set_qp(x265_hip_ctx *c, int qp)
{
    if (!c)
        return fail(X265_HIP_EINVAL, "null context");
    std::vector<uint16_t> tab(65536);
    int r = x265_hip_costtab_build(qp, tab.data());
    if (r)
        return r;
    return x265_hip_costtab_upload(c, qp, tab.data());
}

// ---- validation of batched requests (host arrays only; device arrays are the caller's contract) --

static int check_block(const x265_hip_ctx *c, int fp, int rp, int w, int h, const char *what, int i)
{
    if (fp < 0 || fp >= X265_HIP_MAX_PLANES || rp < 0 || rp >= X265_HIP_MAX_PLANES || !c->host_tabs.planes[fp].origin ||
        !c->host_tabs.planes[rp].origin)
        return fail(X265_HIP_EINVAL, "%s[%d]: plane %d or %d not uploaded", what, i, fp, rp);
    if (x265_hip_partition_from_sizes(w, h) < 0)
        return fail(X265_HIP_EINVAL, "%s[%d]: %dx%d is not a luma partition", what, i, w, h);
    return 0;
}

// Every byte a kernel may read for picture samples [x0, x1) x [y0, y1) of plane `id` (the caller has added the
// displacement; this adds the 8-tap filters' 3 before / 4 after and the aligned 16-byte loads' overhang) must lie
// inside the plane's buffer (+ the 16 readable bytes behind it, see x265_hip_plane_bind).  A request that
// reaches outside would fault the GPU, so it is refused here.
static int check_reach(const x265_hip_ctx *c, int id, long x0, long y0, long x1, long y1, const char *what, int i)
{
    const PlaneDesc &p = c->host_tabs.planes[id];
    const long first = c->plane_origin[id] + (y0 - 3) * p.stride + (x0 - 3 - 3);
    const long last = c->plane_origin[id] + (y1 + 4 - 1) * p.stride + (x1 + 4 + 12);
    if (first < 0 || last > (long)c->plane_bytes[id] + 16)
        return fail(X265_HIP_ERANGE, "%s[%d]: samples [%ld, %ld) x [%ld, %ld) of plane %d (with filter margins) lie outside its buffer",
                    what, i, x0, x1, y0, y1, id);
    return 0;
}

// the block itself, exactly (source blocks, prediction destinations)
static int check_inside(const x265_hip_ctx *c, int id, int x, int y, int w, int h, const char *what, int i)
{
    const PlaneDesc &p = c->host_tabs.planes[id];
    const long first = c->plane_origin[id] + (long)y * p.stride + x;
    const long last = c->plane_origin[id] + (long)(y + h - 1) * p.stride + x + w;
    if (first < 0 || last > (long)c->plane_bytes[id] || x + w > p.stride)
        return fail(X265_HIP_ERANGE, "%s[%d]: block %dx%d at (%d, %d) lies outside plane %d", what, i, w, h, x, y, id);
    return 0;
}

// a cost / distortion / prediction operand: the block displaced by a quarter-pel vector
static int check_displaced(const x265_hip_ctx *c, int id, int x, int y, int w, int h, int qx, int qy, const char *what, int i)
{
    const long bx = x + (qx >> 2), by = y + (qy >> 2);
    return check_reach(c, id, bx, by, bx + w, by + h, what, i);
}

static int check_job(const x265_hip_ctx *c, const x265_hip_me_job &j, int i)
{
    {
        int r = check_block(c, j.fenc_plane, j.ref_plane, j.pu_w, j.pu_h, "me_job", i);
        if (r)
            return r;
        if ((j.method & ~X265_HIP_LOWRES_REF) > X265_HIP_FULL || j.subme > 7 || j.num_mvc > X265_HIP_MAX_MVC)
            return fail(X265_HIP_EINVAL, "me_job[%d]: method %d / subme %d / num_mvc %d out of range", i, j.method, j.subme,
                        j.num_mvc);
        if (j.method & X265_HIP_LOWRES_REF)
        {
            // a Lowres reference: the lookahead's 8x8 CU, four planes of one stride
            if (j.pu_w != 8 || j.pu_h != 8)
                return fail(X265_HIP_EINVAL, "me_job[%d]: a lowres reference takes 8x8 blocks only (lowres.h:83)", i);
            if (j.ref_plane + 3 >= X265_HIP_MAX_PLANES)
                return fail(X265_HIP_EINVAL, "me_job[%d]: lowres planes %d..%d out of range", i, j.ref_plane, j.ref_plane + 3);
            for (int k = 1; k < 4; k++)
                if (!c->host_tabs.planes[j.ref_plane + k].origin ||
                    c->host_tabs.planes[j.ref_plane + k].stride != c->host_tabs.planes[j.ref_plane].stride ||
                    c->plane_bytes[j.ref_plane + k] != c->plane_bytes[j.ref_plane] ||
                    c->plane_origin[j.ref_plane + k] != c->plane_origin[j.ref_plane])
                    return fail(X265_HIP_EINVAL, "me_job[%d]: lowres plane %d missing or of another layout", i, j.ref_plane + k);
        }
        if (j.qp >= X265_HIP_MAX_QP || !c->host_tabs.costtab[j.qp])
            return fail(X265_HIP_EINVAL, "me_job[%d]: no cost table uploaded for qp %d", i, j.qp);
        if (j.mvmin_x > j.mvmax_x || j.mvmin_y > j.mvmax_y)
            return fail(X265_HIP_EINVAL, "me_job[%d]: empty search range", i);
        // Reach: the window and the zero vector (tested without a range check, motion.cpp:570-578), plus what the
        // search may step outside them, as the reference does: its range checks stop a pattern only after the
        // centre has left the window.  DIA: one diamond step (motion.cpp:608-625).  HEX: a hexagon step of 2 and
        // the square refine of 1 (:650-698).  UMH: the unguarded diamonds, octagon and hexagon around the start
        // (:707-742) take the centre up to 4 out; the crosses and rings are guarded and the closing hexagon search
        // only starts from inside (:882-883).  STAR and FULL guard every point.  The sub-pel stage then moves by
        // up to 2 hpel_iters + qpel_iters quarter samples (workload[], :47-57; a Lowres reference: one half-pel and
        // one quarter-pel ring, :1055-1080); check_reach adds the filters' margins.
        static const int k_int_over[5] = { 1, 3, 4, 0, 0 };
        static const int k_qsteps[8] = { 2, 3, 3, 5, 6, 3, 5, 6 };
        const int qsteps = (j.method & X265_HIP_LOWRES_REF) ? 3 : k_qsteps[j.subme];
        const long over = k_int_over[j.method & ~X265_HIP_LOWRES_REF] + (qsteps + 3) / 4;
        const long lox = std::min<long>(j.mvmin_x, 0) - over, hix = std::max<long>(j.mvmax_x, 0) + over;
        const long loy = std::min<long>(j.mvmin_y, 0) - over, hiy = std::max<long>(j.mvmax_y, 0) + over;
        if ((r = check_inside(c, j.fenc_plane, j.pu_x, j.pu_y, j.pu_w, j.pu_h, "me_job", i)) ||
            (r = check_reach(c, j.ref_plane, j.pu_x + lox, j.pu_y + loy, j.pu_x + hix + j.pu_w, j.pu_y + hiy + j.pu_h, "me_job", i)))
            return r;
    }
    return 0;
}

static int check_jobs(const x265_hip_ctx *c, const x265_hip_me_job *jobs, int n)
{
    for (int i = 0; i < n; i++)
    {
        int r = check_job(c, jobs[i], i);
        if (r)
            return r;
    }
    return 0;
}

// ---- validation of whole request lists (the host-array entry points; the device-array ones in record-check mode) --

static int validate_cost(const x265_hip_ctx *c, const x265_hip_cost_req *reqs, int n)
{
    for (int i = 0; i < n; i++)
    {
        int r = check_block(c, reqs[i].fenc_plane, reqs[i].ref_plane, reqs[i].pu_w, reqs[i].pu_h, "cost_req", i);
        if (r)
            return r;
        if (reqs[i].metric > X265_HIP_SATD)
            return fail(X265_HIP_EINVAL, "cost_req[%d]: unknown metric %d", i, reqs[i].metric);
        if ((r = check_inside(c, reqs[i].fenc_plane, reqs[i].pu_x, reqs[i].pu_y, reqs[i].pu_w, reqs[i].pu_h, "cost_req", i)) ||
            (r = check_displaced(c, reqs[i].ref_plane, reqs[i].pu_x, reqs[i].pu_y, reqs[i].pu_w, reqs[i].pu_h, reqs[i].qmv_x,
                                 reqs[i].qmv_y, "cost_req", i)))
            return r;
    }
    return 0;
}

static int validate_refine(const x265_hip_ctx *c, const x265_hip_refine_req *reqs, int n)
{
    for (int i = 0; i < n; i++)
    {
        const x265_hip_refine_req &q = reqs[i];
        int r = check_block(c, q.fenc_plane, q.ref_plane, q.pu_w, q.pu_h, "refine_req", i);
        if (r)
            return r;
        const int ring = q.pattern & (X265_HIP_RING_QPEL | X265_HIP_RING_HPEL);
        if (q.metric > X265_HIP_SATD || (ring != X265_HIP_RING_QPEL && ring != X265_HIP_RING_HPEL) ||
            (q.pattern & ~(X265_HIP_RING_QPEL | X265_HIP_RING_HPEL | X265_HIP_RING_8 | X265_HIP_RING_CENTRE)))
            return fail(X265_HIP_EINVAL, "refine_req[%d]: unknown metric %d / pattern 0x%x", i, q.metric, q.pattern);
        // the kernel stages rows (cy - step) >> 2 - 3 .. + h + 5 and, per horizontal position, columns ix - 3 .. ix + w + 4
        const int step = ring == X265_HIP_RING_HPEL ? 2 : 1;
        const long x0 = q.pu_x + ((q.qmv_x - step) >> 2), x1 = q.pu_x + ((q.qmv_x + step) >> 2) + q.pu_w;
        const long y0 = q.pu_y + ((q.qmv_y - step) >> 2), y1 = y0 + q.pu_h + 2;
        if ((r = check_inside(c, q.fenc_plane, q.pu_x, q.pu_y, q.pu_w, q.pu_h, "refine_req", i)) ||
            (r = check_reach(c, q.ref_plane, x0, y0, x1, y1, "refine_req", i)))
            return r;
    }
    return 0;
}

static int validate_dist(const x265_hip_ctx *c, const x265_hip_cost_req *reqs, int n)
{
    for (int i = 0; i < n; i++)
    {
        int r = check_block(c, reqs[i].fenc_plane, reqs[i].ref_plane, reqs[i].pu_w, reqs[i].pu_h, "dist_req", i);
        if (r)
            return r;
        if (reqs[i].metric > X265_HIP_SSE)
            return fail(X265_HIP_EINVAL, "dist_req[%d]: unknown metric %d", i, reqs[i].metric);
        if ((r = check_inside(c, reqs[i].fenc_plane, reqs[i].pu_x, reqs[i].pu_y, reqs[i].pu_w, reqs[i].pu_h, "dist_req", i)) ||
            (r = check_displaced(c, reqs[i].ref_plane, reqs[i].pu_x, reqs[i].pu_y, reqs[i].pu_w, reqs[i].pu_h, reqs[i].qmv_x,
                                 reqs[i].qmv_y, "dist_req", i)))
            return r;
    }
    return 0;
}

static int validate_pred(const x265_hip_ctx *c, const x265_hip_pred_req *reqs, int n, bool satd = false)
{
    for (int i = 0; i < n; i++)
    {
        const x265_hip_pred_req &q = reqs[i];
        int r = check_block(c, q.dst_plane, q.ref0_plane, q.pu_w, q.pu_h, "pred_req", i);
        if (r || (q.ref1_plane >= 0 && (r = check_block(c, q.dst_plane, q.ref1_plane, q.pu_w, q.pu_h, "pred_req", i))))
            return r;
        if (!satd && (q.dst_plane == q.ref0_plane || q.dst_plane == q.ref1_plane)) // (mc_satd writes nothing)
            return fail(X265_HIP_EINVAL, "pred_req[%d]: the destination plane is one of the references", i);
        if (q.ref1_plane == X265_HIP_PRED_WEIGHTED &&
            (q.log2_denom > 7 || q.qmv1_x < -128 || q.qmv1_x > 255 || q.qmv1_y < -128 || q.qmv1_y > 127))
            return fail(X265_HIP_EINVAL, "pred_req[%d]: weight %d / offset %d / log2 denominator %d out of range", i, q.qmv1_x,
                        q.qmv1_y, q.log2_denom);
        if (q.weighted_bi && q.ref1_plane < 0)
            return fail(X265_HIP_EINVAL, "pred_req[%d]: weighted bi-prediction takes two references", i);
        if (q.weighted_bi && (q.log2_denom > 7 || q.wp_w0 < -128 || q.wp_w0 > 255 || q.wp_w1 < -128 || q.wp_w1 > 255 || q.wp_o0 < -128 ||
                              q.wp_o0 > 127 || q.wp_o1 < -128 || q.wp_o1 > 127))
            return fail(X265_HIP_EINVAL, "pred_req[%d]: weights %d %d / offsets %d %d / log2 denominator %d out of range", i, q.wp_w0,
                        q.wp_w1, q.wp_o0, q.wp_o1, q.log2_denom);
        if ((r = check_inside(c, q.dst_plane, q.pu_x, q.pu_y, q.pu_w, q.pu_h, "pred_req", i)) ||
            (r = check_displaced(c, q.ref0_plane, q.pu_x, q.pu_y, q.pu_w, q.pu_h, q.qmv0_x, q.qmv0_y, "pred_req", i)) ||
            (q.ref1_plane >= 0 &&
             (r = check_displaced(c, q.ref1_plane, q.pu_x, q.pu_y, q.pu_w, q.pu_h, q.qmv1_x, q.qmv1_y, "pred_req", i))))
            return r;
    }
    return 0;
}

static int validate_bidir(const x265_hip_ctx *c, const x265_hip_bidir_req *reqs, int n)
{
    for (int i = 0; i < n; i++)
    {
        int r = check_block(c, reqs[i].fenc_plane, reqs[i].ref0_plane, reqs[i].pu_w, reqs[i].pu_h, "bidir_req", i);
        if (r || (r = check_block(c, reqs[i].fenc_plane, reqs[i].ref1_plane, reqs[i].pu_w, reqs[i].pu_h, "bidir_req", i)))
            return r;
        const x265_hip_bidir_req &q = reqs[i];
        if ((q.metric & ~(X265_HIP_LOWRES_REF | X265_HIP_MC_AVG)) > X265_HIP_SSE ||
            (q.metric & (X265_HIP_LOWRES_REF | X265_HIP_MC_AVG)) == (X265_HIP_LOWRES_REF | X265_HIP_MC_AVG))
            return fail(X265_HIP_EINVAL, "bidir_req[%d]: unknown metric %d", i, q.metric);
        if (q.metric & X265_HIP_LOWRES_REF)
        {
            // Lowres references: the lookahead's 8x8 CU, four planes of one stride per reference
            if (q.pu_w != 8 || q.pu_h != 8)
                return fail(X265_HIP_EINVAL, "bidir_req[%d]: lowres references take 8x8 blocks only (lowres.h:64)", i);
            for (int rp : { q.ref0_plane, q.ref1_plane })
            {
                if (rp + 3 >= X265_HIP_MAX_PLANES)
                    return fail(X265_HIP_EINVAL, "bidir_req[%d]: lowres planes %d..%d out of range", i, rp, rp + 3);
                for (int k = 1; k < 4; k++)
                    if (!c->host_tabs.planes[rp + k].origin || c->host_tabs.planes[rp + k].stride != c->host_tabs.planes[rp].stride ||
                        c->plane_bytes[rp + k] != c->plane_bytes[rp] || c->plane_origin[rp + k] != c->plane_origin[rp])
                        return fail(X265_HIP_EINVAL, "bidir_req[%d]: lowres plane %d missing or of another layout", i, rp + k);
            }
        }
        if ((r = check_inside(c, q.fenc_plane, q.pu_x, q.pu_y, q.pu_w, q.pu_h, "bidir_req", i)) ||
            (r = check_displaced(c, q.ref0_plane, q.pu_x, q.pu_y, q.pu_w, q.pu_h, q.qmv0_x, q.qmv0_y, "bidir_req", i)) ||
            (r = check_displaced(c, q.ref1_plane, q.pu_x, q.pu_y, q.pu_w, q.pu_h, q.qmv1_x, q.qmv1_y, "bidir_req", i)))
            return r;
    }
    return 0;
}

static int validate_ctu(const x265_hip_ctx *c, const x265_hip_ctu_job *jobs, int n)
{
    for (int i = 0; i < n; i++)
    {
        const x265_hip_ctu_job &j = jobs[i];
        int r = check_block(c, j.fenc_plane, j.ref_plane, 64, 64, "ctu_job", i);
        if (r)
            return r;
        if (j.qp >= X265_HIP_MAX_QP || !c->host_tabs.costtab[j.qp])
            return fail(X265_HIP_EINVAL, "ctu_job[%d]: no cost table uploaded for qp %d", i, j.qp);
        if (j.mvmin_x > j.mvmax_x || j.mvmin_y > j.mvmax_y)
            return fail(X265_HIP_EINVAL, "ctu_job[%d]: empty search range", i);
        if (j.mvmax_x - j.mvmin_x > 131 || j.mvmax_y - j.mvmin_y > 131)
            return fail(X265_HIP_ERANGE, "ctu_job[%d]: window wider than 132 positions", i);
        if ((r = check_inside(c, j.fenc_plane, j.ctu_x, j.ctu_y, 64, 64, "ctu_job", i)) ||
            (r = check_reach(c, j.ref_plane, j.ctu_x + j.mvmin_x, j.ctu_y + j.mvmin_y, j.ctu_x + j.mvmax_x + 64,
                             j.ctu_y + j.mvmax_y + 64, "ctu_job", i)))
            return r;
    }
    return 0;
}

static int validate_inter(const x265_hip_ctx *c, const x265_hip_inter_job *jobs, int n, int pic_w, int pic_h)
{
    for (int i = 0; i < n; i++)
    {
        const x265_hip_inter_job &q = jobs[i];
        if (q.num_amvp < 1 || q.num_amvp > 2)
            return fail(X265_HIP_EINVAL, "inter_job[%d]: %d AMVP candidates (1 or 2)", i, q.num_amvp);
        if (q.pred_plane < 0 || q.pred_plane >= X265_HIP_MAX_PLANES || !c->host_tabs.planes[q.pred_plane].origin)
            return fail(X265_HIP_EINVAL, "inter_job[%d]: plane %d not uploaded", i, q.pred_plane);
        // every search the job can turn into (one per candidate) must pass the checks of a motionEstimate job, and the
        // candidates' predictions must lie inside the plane they are made from
        for (int k = 0; k < q.num_amvp; k++)
        {
            x265_hip_me_job m;
            memset(&m, 0, sizeof(m));
            m.fenc_plane = q.fenc_plane, m.ref_plane = q.ref_plane;
            m.pu_x = q.pu_x, m.pu_y = q.pu_y, m.pu_w = q.pu_w, m.pu_h = q.pu_h;
            m.method = q.method, m.subme = q.subme, m.merange = q.merange, m.qp = q.qp, m.num_mvc = q.num_mvc;
            int lox, loy, hix, hiy;
            search_window(q.amvp[2 * k], q.amvp[2 * k + 1], q.merange, q.ref_lag, q.cu_x, q.cu_y, pic_w, pic_h, lox, loy, hix, hiy);
            m.mvmin_x = (int16_t)lox, m.mvmin_y = (int16_t)loy, m.mvmax_x = (int16_t)hix, m.mvmax_y = (int16_t)hiy;
            int r = check_job(c, m, i);
            if (r)
                return r;
            int qx = q.amvp[2 * k], qy = q.amvp[2 * k + 1];
            clip_mv(qx, qy, q.cu_x, q.cu_y, pic_w, pic_h);
            if ((r = check_displaced(c, q.pred_plane, q.pu_x, q.pu_y, q.pu_w, q.pu_h, qx, qy, "inter_job", i)))
                return r;
        }
    }
    return 0;
}

static int validate_lookahead(const x265_hip_ctx *c, const x265_hip_lookahead_job *jobs, int njobs, int w_cu, int h_cu)
{
    for (int i = 0; i < njobs; i++)
    {
        // every CU's window spans the same samples of the reference, [-8, 8 (W - 1) + 8] x [-8, 8 (H - 1) + 8] for the
        // block's origin (slicetype.cpp:1540-1544): the four corner CUs are put through the checks of a motionEstimate job
        for (int corner = 0; corner < 4; corner++)
        {
            const int cux = (corner & 1) ? w_cu - 1 : 0, cuy = (corner & 2) ? h_cu - 1 : 0;
            x265_hip_me_job m;
            memset(&m, 0, sizeof(m));
            m.fenc_plane = jobs[i].fenc_plane, m.ref_plane = jobs[i].ref_plane;
            m.pu_x = (int16_t)(8 * cux), m.pu_y = (int16_t)(8 * cuy), m.pu_w = m.pu_h = 8;
            m.method = (uint8_t)(jobs[i].method | X265_HIP_LOWRES_REF), m.subme = jobs[i].subme, m.merange = jobs[i].merange, m.qp = jobs[i].qp;
            m.mvmin_x = (int16_t)(-8 * cux - 8), m.mvmin_y = (int16_t)(-8 * cuy - 8);
            m.mvmax_x = (int16_t)((w_cu - cux - 1) * 8 + 8), m.mvmax_y = (int16_t)((h_cu - cuy - 1) * 8 + 8);
            if (jobs[i].method & X265_HIP_LOWRES_REF)
                return fail(X265_HIP_EINVAL, "lookahead_job[%d]: method %d (the Lowres flag is implied)", i, jobs[i].method);
            int r = check_job(c, m, i);
            if (r)
                return r;
        }
    }
    return 0;
}

// Record-check mode (x265_hip_ctx_check_records, or X265_HIP_CHECK_RECORDS=1 in the environment when the context is
// made): the device-array entry points, which otherwise take their records on trust, wait for the stream, copy the
// records back and put them through the checks of the host-array form before anything is launched.  A debugging aid:
// a bad record then gives an error code and a message instead of a GPU memory fault.
extern "C++" {
template <typename T, typename F>
static int checked_records(x265_hip_ctx *c, const void *d_recs, int n, F validate, bool has_guard = false)
{
    if (!c->check_records || n <= 0 || (c->check_records == 2 && has_guard))
        return 0;
    std::vector<T> host((size_t)n);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(host.data(), d_recs, (size_t)n * sizeof(T), hipMemcpyDeviceToHost));
    return validate(host.data());
}
} // extern "C++"

// Record guard of a request list (mode 2): a plane the replaced records can point at -- one that holds a 64x64 block at
// its origin with the filters' margins -- or -1 (then the call falls back to the copy-back of mode 1)
static int guard_safe_plane(const x265_hip_ctx *c)
{
    for (int id = 0; id < X265_HIP_MAX_PLANES; id++)
    {
        const PlaneDesc &p = c->host_tabs.planes[id];
        if (!p.origin || p.stride < 64)
            continue;
        const long first = c->plane_origin[id] + (-1 - 3) * (long)p.stride + (-1 - 3 - 3);
        const long last = c->plane_origin[id] + (67 + 4 - 1) * (long)p.stride + (65 + 4 + 12);
        if (first >= 0 && last <= (long)c->plane_bytes[id] + 16)
            return id;
    }
    return -1;
}

// room for the guarded copy of n records of `rec` bytes and their flags; *sane / *bad point into it
static int guard_room(x265_hip_ctx *c, int n, size_t rec, void **sane, uint8_t **bad)
{
    const size_t list = ((size_t)n * rec + 255) & ~(size_t)255, need = list + (size_t)n;
    if (c->guard.cap < need)
        HIP_TRY(hipStreamSynchronize(c->stream)); // an earlier launch may still read the buffer that is about to be replaced
    int r = ensure(c, c->guard, need, false);
    if (r)
        return r;
    *sane = c->guard.d;
    *bad = (uint8_t *)c->guard.d + list;
    return 0;
}

// ---- batched entry points -----------------------------------------------------------------------

int x265_hip_me_batch_device(x265_hip_ctx *c, const void *d_jobs, int n, void *d_results)
{
    if (!c || n < 0 || (n && (!d_jobs || !d_results)))
        return fail(X265_HIP_EINVAL, "me_batch_device: bad argument");
    if (n == 0)
        return X265_HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    {
        int vr = checked_records<x265_hip_me_job>(c, d_jobs, n, [&](const x265_hip_me_job *h) { return check_jobs(c, h, n); }, true);
        if (vr)
            return vr;
    }
    int r = push_tabs(c);
    if (r || (r = ensure_route(c, n)))
        return r;
    LAUNCH(launch_me(c->dev_tabs, (const x265_hip_me_job *)d_jobs, n, (x265_hip_me_result *)d_results, me_route(c, n > 0, true, c->check_records == 2)));
    return X265_HIP_OK;
}

int x265_hip_me_batch(x265_hip_ctx *c, const x265_hip_me_job *jobs, int n, x265_hip_me_result *results)
{
    if (!c || n < 0 || (n && (!jobs || !results)))
        return fail(X265_HIP_EINVAL, "me_batch: bad argument");
    if (n == 0)
        return X265_HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    int r = check_jobs(c, jobs, n);
    if (r || (r = push_tabs(c)) || (r = ensure(c, c->in, (size_t)n * sizeof(*jobs))) ||
        (r = ensure(c, c->out, (size_t)n * sizeof(*results))) || (r = ensure_route(c, n)))
        return r;
    // straight from the caller's memory: the runtime pipelines its own staging, which beats a copy into the
    // pinned mirror followed by one DMA
    HIP_TRY(hipMemcpyAsync(c->in.d, jobs, (size_t)n * sizeof(*jobs), hipMemcpyHostToDevice, c->stream));
    LAUNCH(launch_me(c->dev_tabs, (const x265_hip_me_job *)c->in.d, n, (x265_hip_me_result *)c->out.d, me_route(c, n > 0, any_routed_job(jobs, n))));
    HIP_TRY(hipMemcpyAsync(c->out.h, c->out.d, (size_t)n * sizeof(*results), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(results, c->out.h, (size_t)n * sizeof(*results));
    return X265_HIP_OK;
}

// ---- predInterSearch's (list, reference) iteration: AMVP pick, search window, motionEstimate -------------------------

int x265_hip_inter_search_batch_device(x265_hip_ctx *c, const void *d_jobs, int n, int pic_w, int pic_h, void *d_results)
{
    if (!c || n < 0 || pic_w <= 0 || pic_h <= 0 || (n && (!d_jobs || !d_results)))
        return fail(X265_HIP_EINVAL, "inter_search_batch_device: bad argument");
    if (n == 0)
        return X265_HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    {
        int vr = checked_records<x265_hip_inter_job>(c, d_jobs, n, [&](const x265_hip_inter_job *h) { return validate_inter(c, h, n, pic_w, pic_h); });
        if (vr)
            return vr;
    }
    int r;
    if (c->mejobs.cap < X265HIP_INTER_SCRATCH_BYTES(n))
        HIP_TRY(hipStreamSynchronize(c->stream)); // an earlier launch may still read the buffer that is about to be replaced
    if ((r = push_tabs(c)) || (r = ensure(c, c->mejobs, X265HIP_INTER_SCRATCH_BYTES(n), false)) || (r = ensure_route(c, n)))
        return r;
    LAUNCH(launch_inter_search(c->dev_tabs, (const x265_hip_inter_job *)d_jobs, n, pic_w, pic_h, (x265_hip_me_job *)c->mejobs.d,
                               (x265_hip_inter_result *)d_results, me_route(c, n > 0), c->batch_counter, c->batch_parity ^= 1));
    return X265_HIP_OK;
}

int x265_hip_inter_search_batch(x265_hip_ctx *c, const x265_hip_inter_job *jobs, int n, int pic_w, int pic_h, x265_hip_inter_result *results)
{
    if (!c || n < 0 || pic_w <= 0 || pic_h <= 0 || (n && (!jobs || !results)))
        return fail(X265_HIP_EINVAL, "inter_search_batch: bad argument");
    if (n == 0)
        return X265_HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    {
        int vr = validate_inter(c, jobs, n, pic_w, pic_h);
        if (vr)
            return vr;
    }
    int r;
    const size_t ob = (size_t)n * sizeof(*results);
    if ((r = push_tabs(c)) || (r = ensure(c, c->in, (size_t)n * sizeof(*jobs))) || (r = ensure(c, c->out, ob)) ||
        (r = ensure(c, c->mejobs, X265HIP_INTER_SCRATCH_BYTES(n), false)) || (r = ensure_route(c, n)))
        return r;
    HIP_TRY(hipMemcpyAsync(c->in.d, jobs, (size_t)n * sizeof(*jobs), hipMemcpyHostToDevice, c->stream));
    LAUNCH(launch_inter_search(c->dev_tabs, (const x265_hip_inter_job *)c->in.d, n, pic_w, pic_h, (x265_hip_me_job *)c->mejobs.d,
                               (x265_hip_inter_result *)c->out.d, me_route(c, n > 0, any_routed_job(jobs, n)), c->batch_counter, c->batch_parity ^= 1));
    HIP_TRY(hipMemcpyAsync(c->out.h, c->out.d, ob, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(results, c->out.h, ob);
    return X265_HIP_OK;
}

// ---- predInterSearch behind its iterations: bits, xCheckBestMVP, the bidirectional candidate, the choice ------------

static int validate_inter_pus(const x265_hip_ctx *c, const x265_hip_inter_job *jobs, int njobs, const x265_hip_inter_pu *pus, int npu)
{
    for (int u = 0; u < npu; u++)
    {
        const x265_hip_inter_pu &p = pus[u];
        const int n = p.num_ref[0] + p.num_ref[1];
        if (p.num_ref[0] > 16 || p.num_ref[1] > 16 || n < 1 || p.first_job < 0 || (long)p.first_job + n > njobs)
            return fail(X265_HIP_EINVAL, "inter_pu[%d]: iterations %d .. %d of %d jobs (%d + %d references)", u, p.first_job, p.first_job + n - 1, njobs,
                        p.num_ref[0], p.num_ref[1]);
        const x265_hip_inter_job &a = jobs[p.first_job];
        for (int k = 0; k < n; k++)
        {
            const x265_hip_inter_job &q = jobs[p.first_job + k];
            if (q.fenc_plane != a.fenc_plane || q.pu_x != a.pu_x || q.pu_y != a.pu_y || q.pu_w != a.pu_w || q.pu_h != a.pu_h)
                return fail(X265_HIP_EINVAL, "inter_pu[%d]: its iterations are not of one block", u);
            // the bidirectional candidate predicts from the reconstructed planes at the vectors the searches found in the
            // searched planes: the two must be laid out alike for the searches' checks to cover it
            if (c->host_tabs.planes[q.pred_plane].stride != c->host_tabs.planes[q.ref_plane].stride ||
                c->plane_bytes[q.pred_plane] != c->plane_bytes[q.ref_plane] || c->plane_origin[q.pred_plane] != c->plane_origin[q.ref_plane])
                return fail(X265_HIP_EINVAL, "inter_pu[%d]: planes %d and %d of iteration %d differ in layout", u, q.pred_plane, q.ref_plane, k);
        }
    }
    return 0;
}

// (the caller holds the lock, has validated, and has pushed the tables)
static int inter_decide_launch(x265_hip_ctx *c, const x265_hip_inter_job *d_jobs, int njobs, const x265_hip_inter_pu *d_pus, int npu, int pic_w, int pic_h,
                               x265_hip_inter_result *d_job_results, x265_hip_inter_pu_result *d_results, bool any_bidir, bool any_routed)
{
    int r;
    const size_t req_bytes = ((size_t)2 * npu * sizeof(x265_hip_bidir_req) + 255) & ~(size_t)255;
    if (c->mejobs.cap < X265HIP_INTER_SCRATCH_BYTES(njobs) || c->bidir.cap < req_bytes + (size_t)2 * npu * sizeof(int32_t))
        HIP_TRY(hipStreamSynchronize(c->stream)); // an earlier launch may still read a buffer that is about to be replaced
    if ((r = ensure(c, c->mejobs, X265HIP_INTER_SCRATCH_BYTES(njobs), false)) || (r = ensure_route(c, njobs)) ||
        (r = ensure(c, c->bidir, req_bytes + (size_t)2 * npu * sizeof(int32_t), false)))
        return r;
    x265_hip_bidir_req *bireq = (x265_hip_bidir_req *)c->bidir.d;
    int32_t *satd = (int32_t *)((char *)c->bidir.d + req_bytes);
    LAUNCH(launch_inter_search(c->dev_tabs, d_jobs, njobs, pic_w, pic_h, (x265_hip_me_job *)c->mejobs.d, d_job_results, me_route(c, njobs > 0, any_routed),
                               c->batch_counter, c->batch_parity ^= 1));
    LAUNCH(launch_inter_decide_lists(c->dev_tabs, d_jobs, d_job_results, d_pus, npu, pic_w, pic_h, d_results, bireq, any_bidir, c->stream));
    if (any_bidir)
    {
        LAUNCH(launch_bidir_cost(c->dev_tabs, bireq, 2 * npu, satd, c->batch_counter, c->batch_parity ^= 1, c->stream));
        LAUNCH(launch_inter_decide_bidir(c->dev_tabs, d_jobs, d_pus, npu, satd, d_results, c->stream));
    }
    return X265_HIP_OK;
}

int x265_hip_inter_decide_batch_device(x265_hip_ctx *c, const void *d_jobs, int njobs, const void *d_pus, int npu, int pic_w, int pic_h,
                                       void *d_job_results, void *d_results, int any_bidir)
{
    if (!c || njobs < 0 || npu < 0 || pic_w <= 0 || pic_h <= 0 || (npu && (!d_jobs || !d_pus || !d_job_results || !d_results || njobs <= 0)))
        return fail(X265_HIP_EINVAL, "inter_decide_batch_device: bad argument");
    if (npu == 0)
        return X265_HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    if (c->check_records)
    {
        std::vector<x265_hip_inter_job> hj((size_t)njobs);
        std::vector<x265_hip_inter_pu> hp((size_t)npu);
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpy(hj.data(), d_jobs, hj.size() * sizeof(hj[0]), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(hp.data(), d_pus, hp.size() * sizeof(hp[0]), hipMemcpyDeviceToHost));
        int vr = validate_inter(c, hj.data(), njobs, pic_w, pic_h);
        if (vr || (vr = validate_inter_pus(c, hj.data(), njobs, hp.data(), npu)))
            return vr;
    }
    int r = push_tabs(c);
    if (r)
        return r;
    return inter_decide_launch(c, (const x265_hip_inter_job *)d_jobs, njobs, (const x265_hip_inter_pu *)d_pus, npu, pic_w, pic_h,
                               (x265_hip_inter_result *)d_job_results, (x265_hip_inter_pu_result *)d_results, any_bidir != 0, true);
}

int x265_hip_inter_decide_batch(x265_hip_ctx *c, const x265_hip_inter_job *jobs, int njobs, const x265_hip_inter_pu *pus, int npu, int pic_w, int pic_h,
                                x265_hip_inter_result *job_results, x265_hip_inter_pu_result *results)
{
    if (!c || njobs < 0 || npu < 0 || pic_w <= 0 || pic_h <= 0 || (npu && (!jobs || !pus || !results || njobs <= 0)))
        return fail(X265_HIP_EINVAL, "inter_decide_batch: bad argument");
    if (npu == 0)
        return X265_HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    int r = validate_inter(c, jobs, njobs, pic_w, pic_h);
    if (r || (r = validate_inter_pus(c, jobs, njobs, pus, npu)))
        return r;
    bool any_bidir = false;
    for (int u = 0; u < npu; u++)
        any_bidir |= !pus[u].no_bidir;
    // staging: jobs then units in `in`, iteration results then unit results in `out`
    const size_t jb = ((size_t)njobs * sizeof(*jobs) + 255) & ~(size_t)255, pb = (size_t)npu * sizeof(*pus);
    const size_t rb = ((size_t)njobs * sizeof(x265_hip_inter_result) + 255) & ~(size_t)255, ob = (size_t)npu * sizeof(*results);
    if ((r = push_tabs(c)) || (r = ensure(c, c->in, jb + pb)) || (r = ensure(c, c->out, rb + ob)))
        return r;
    HIP_TRY(hipMemcpyAsync(c->in.d, jobs, (size_t)njobs * sizeof(*jobs), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync((char *)c->in.d + jb, pus, pb, hipMemcpyHostToDevice, c->stream));
    r = inter_decide_launch(c, (const x265_hip_inter_job *)c->in.d, njobs, (const x265_hip_inter_pu *)((char *)c->in.d + jb), npu, pic_w, pic_h,
                            (x265_hip_inter_result *)c->out.d, (x265_hip_inter_pu_result *)((char *)c->out.d + rb), any_bidir, any_routed_job(jobs, njobs));
    if (r)
        return r;
    HIP_TRY(hipMemcpyAsync(c->out.h, c->out.d, rb + ob, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (job_results)
        memcpy(job_results, c->out.h, (size_t)njobs * sizeof(*job_results));
    memcpy(results, (char *)c->out.h + rb, ob);
    return X265_HIP_OK;
}

// ---- whole-frame lookahead search ----------------------------------------------------------------------------------

#define LOOK_HEAD 64 // bytes in front of the row counters: the status word

static int lookahead_launch(x265_hip_ctx *c, const void *d_jobs, int njobs, int w_cu, int h_cu, void *d_results)
{
    int r;
    const size_t need = LOOK_HEAD + (size_t)njobs * 64;
    if (c->look.cap < need)
        HIP_TRY(hipStreamSynchronize(c->stream)); // an earlier launch may still count in the buffer that is about to be replaced
    if ((r = push_tabs(c)) || (r = ensure(c, c->look, need)))
        return r;
    HIP_TRY(hipMemsetAsync(c->look.d, 0, need, c->stream));
    HIP_TRY(hipMemsetAsync(d_results, 0xff, (size_t)njobs * w_cu * h_cu * sizeof(x265_hip_me_result), c->stream));
    LAUNCH(launch_lookahead(c->dev_tabs, (const x265_hip_lookahead_job *)d_jobs, njobs, w_cu, h_cu, (uint32_t *)((char *)c->look.d + LOOK_HEAD),
                            (x265_hip_me_result *)d_results, (int *)c->look.d, c->stream));
    return 0;
}

static int lookahead_args(const x265_hip_ctx *c, int njobs, int w_cu, int h_cu, const void *a, const void *b, const char *who)
{
    if (!c || njobs < 0 || w_cu < 1 || h_cu < 1 || w_cu > 4096 || h_cu > 4096 || (njobs && (!a || !b)))
        return fail(X265_HIP_EINVAL, "%s: bad argument", who);
    if ((long)njobs * h_cu > (1l << 24))
        return fail(X265_HIP_EINVAL, "%s: %d searches of %d rows are more than one launch takes (2^24 rows)", who, njobs, h_cu);
    return 0;
}

int x265_hip_lookahead_batch_device(x265_hip_ctx *c, const void *d_jobs, int njobs, int w_cu, int h_cu, void *d_results)
{
    int r = lookahead_args(c, njobs, w_cu, h_cu, d_jobs, d_results, "lookahead_batch_device");
    if (r || njobs == 0)
        return r;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    if ((r = checked_records<x265_hip_lookahead_job>(c, d_jobs, njobs, [&](const x265_hip_lookahead_job *h) { return validate_lookahead(c, h, njobs, w_cu, h_cu); })))
        return r;
    return lookahead_launch(c, d_jobs, njobs, w_cu, h_cu, d_results);
}

int x265_hip_lookahead_status(x265_hip_ctx *c, int *status)
{
    if (!c || !status)
        return fail(X265_HIP_EINVAL, "lookahead_status: null argument");
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *status = 0;
    if (c->look.d)
        HIP_TRY(hipMemcpy(status, c->look.d, sizeof(int), hipMemcpyDeviceToHost));
    return X265_HIP_OK;
}

int x265_hip_lookahead_batch(x265_hip_ctx *c, const x265_hip_lookahead_job *jobs, int njobs, int w_cu, int h_cu, x265_hip_me_result *results)
{
    int r = lookahead_args(c, njobs, w_cu, h_cu, jobs, results, "lookahead_batch");
    if (r || njobs == 0)
        return r;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    if ((r = validate_lookahead(c, jobs, njobs, w_cu, h_cu)))
        return r;
    const size_t ncu = (size_t)njobs * w_cu * h_cu, ob = ncu * sizeof(*results);
    if ((r = ensure(c, c->in, (size_t)njobs * sizeof(*jobs))) || (r = ensure(c, c->out, ob)))
        return r;
    HIP_TRY(hipMemcpyAsync(c->in.d, jobs, (size_t)njobs * sizeof(*jobs), hipMemcpyHostToDevice, c->stream));
    if ((r = lookahead_launch(c, c->in.d, njobs, w_cu, h_cu, c->out.d)))
        return r;
    int status = 0;
    HIP_TRY(hipMemcpyAsync(c->out.h, c->out.d, ob, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(c->look.h, c->look.d, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    status = *(const int *)c->look.h;
    if (status)
        return fail(X265_HIP_ERUNTIME, "lookahead_batch: a row gave up waiting for the row below it");
    memcpy(results, c->out.h, ob);
    return X265_HIP_OK;
}

int x265_hip_cost_batch_device(x265_hip_ctx *c, const void *d_reqs, int n, void *d_costs)
{
    if (!c || n < 0 || (n && (!d_reqs || !d_costs)))
        return fail(X265_HIP_EINVAL, "cost_batch_device: bad argument");
    if (n == 0) // nothing to launch; a launch takes one of the two counter sets and clears the other
        return X265_HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    const int safe = c->check_records == 2 ? guard_safe_plane(c) : -1;
    {
        int vr = checked_records<x265_hip_cost_req>(c, d_reqs, n, [&](const x265_hip_cost_req *h) { return validate_cost(c, h, n); }, safe >= 0);
        if (vr)
            return vr;
    }
    int r = push_tabs(c);
    if (r)
        return r;
    uint8_t *bad = nullptr;
    if (safe >= 0)
    {
        void *sane;
        if ((r = guard_room(c, n, sizeof(x265_hip_cost_req), &sane, &bad)))
            return r;
        LAUNCH(launch_guard_cost(c->dev_tabs, (const x265_hip_cost_req *)d_reqs, n, X265_HIP_SATD, safe, (x265_hip_cost_req *)sane, bad, c->guard_words,
                                 c->stream));
        d_reqs = sane;
    }
    LAUNCH(launch_cost(c->dev_tabs, (const x265_hip_cost_req *)d_reqs, n, (int32_t *)d_costs, c->batch_counter, c->batch_parity ^= 1, c->stream));
    if (bad)
        LAUNCH(launch_guard_fix(bad, n, 1, (int32_t *)d_costs, c->stream));
    return X265_HIP_OK;
}

// x265_hip_cost_batch_device without any allocation: the batch counters are this context's slot of the GPU's
// `scratch` array (through the address taken at context creation) instead of the context's buffer.  The slot is
// cleared once, when the context takes it; after that the two counter sets alternate as in the context's own buffer
// (a launch clears the set of the next one).  No hipMalloc / hipFree / growing staging on this path.
int x265_hip_cost_batch_device_noalloc(x265_hip_ctx *c, const void *d_reqs, int n, void *d_costs)
{
    if (!c || n < 0 || (n && (!d_reqs || !d_costs)))
        return fail(X265_HIP_EINVAL, "cost_batch_device_noalloc: bad argument");
    if (n > X265_HIP_NOALLOC_MAX_REQUESTS)
        return fail(X265_HIP_ERANGE, "cost_batch_device_noalloc: %d requests exceed the limit of %d per call", n,
                    X265_HIP_NOALLOC_MAX_REQUESTS);
    if (n == 0)
        return X265_HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    {
        int vr = checked_records<x265_hip_cost_req>(c, d_reqs, n, [&](const x265_hip_cost_req *h) { return validate_cost(c, h, n); });
        if (vr)
            return vr;
    }
    uint32_t *const ptr = scratch_slot(c);
    if (!ptr)
        return fail(X265_HIP_ERANGE, "cost_batch_device_noalloc: no scratch slot left on device %d (%d contexts at most use this entry point per GPU)",
                    c->device, X265HIP_NOALLOC_SLOTS);
    int r = push_tabs(c); // a copy into the context's directory, nothing is allocated
    if (r)
        return r;
    LAUNCH(launch_cost(c->dev_tabs, (const x265_hip_cost_req *)d_reqs, n, (int32_t *)d_costs, ptr, c->noalloc_parity ^= 1, c->stream));
    return X265_HIP_OK;
}

int x265_hip_cost_batch(x265_hip_ctx *c, const x265_hip_cost_req *reqs, int n, int32_t *costs)
{
    if (!c || n < 0 || (n && (!reqs || !costs)))
        return fail(X265_HIP_EINVAL, "cost_batch: bad argument");
    if (n == 0)
        return X265_HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    {
        int vr = validate_cost(c, reqs, n);
        if (vr)
            return vr;
    }
    int r;
    if ((r = push_tabs(c)) || (r = ensure(c, c->in, (size_t)n * sizeof(*reqs))) ||
        (r = ensure(c, c->out, (size_t)n * sizeof(int32_t))))
        return r;
    memcpy(c->in.h, reqs, (size_t)n * sizeof(*reqs));
    HIP_TRY(hipMemcpyAsync(c->in.d, c->in.h, (size_t)n * sizeof(*reqs), hipMemcpyHostToDevice, c->stream));
    LAUNCH(launch_cost(c->dev_tabs, (const x265_hip_cost_req *)c->in.d, n, (int32_t *)c->out.d, c->batch_counter, c->batch_parity ^= 1, c->stream));
    HIP_TRY(hipMemcpyAsync(c->out.h, c->out.d, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(costs, c->out.h, (size_t)n * sizeof(int32_t));
    return X265_HIP_OK;
}

int x265_hip_refine_batch_device(x265_hip_ctx *c, const void *d_reqs, int n, void *d_costs9)
{
    if (!c || n < 0 || (n && (!d_reqs || !d_costs9)))
        return fail(X265_HIP_EINVAL, "refine_batch_device: bad argument");
    if (n == 0) // nothing to launch; a launch takes one of the two counter sets and clears the other
        return X265_HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    const int safe = c->check_records == 2 ? guard_safe_plane(c) : -1;
    {
        int vr = checked_records<x265_hip_refine_req>(c, d_reqs, n, [&](const x265_hip_refine_req *h) { return validate_refine(c, h, n); }, safe >= 0);
        if (vr)
            return vr;
    }
    int r = push_tabs(c);
    if (r)
        return r;
    uint8_t *bad = nullptr;
    if (safe >= 0)
    {
        void *sane;
        if ((r = guard_room(c, n, sizeof(x265_hip_refine_req), &sane, &bad)))
            return r;
        LAUNCH(launch_guard_refine(c->dev_tabs, (const x265_hip_refine_req *)d_reqs, n, safe, (x265_hip_refine_req *)sane, bad, c->guard_words, c->stream));
        d_reqs = sane;
    }
    LAUNCH(launch_refine(c->dev_tabs, (const x265_hip_refine_req *)d_reqs, n, (int32_t *)d_costs9, c->batch_counter, c->batch_parity ^= 1, c->stream));
    if (bad)
        LAUNCH(launch_guard_fix(bad, n, X265_HIP_RING_COSTS, (int32_t *)d_costs9, c->stream));
    return X265_HIP_OK;
}

int x265_hip_refine_sections_device(x265_hip_ctx *c, const void *d_reqs, const x265_hip_refine_section *sections, int nsections, void *d_costs9)
{
    if (!c || !sections || nsections < 1 || nsections > X265_HIP_MAX_SECTIONS || !d_reqs || !d_costs9)
        return fail(X265_HIP_EINVAL, "refine_sections_device: bad argument");
    long n = 0;
    for (int s = 0; s < nsections; s++)
    {
        if (sections[s].count < 0 || x265_hip_partition_from_sizes(sections[s].pu_w, sections[s].pu_h) < 0)
            return fail(X265_HIP_EINVAL, "refine_sections_device: section %d: %d requests of %dx%d", s, sections[s].count, sections[s].pu_w,
                        sections[s].pu_h);
        n += sections[s].count;
    }
    if (n > 0x7fffffff)
        return fail(X265_HIP_EINVAL, "refine_sections_device: too many requests");
    if (n == 0)
        return X265_HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    const int safe = c->check_records == 2 ? guard_safe_plane(c) : -1;
    {
        int vr = checked_records<x265_hip_refine_req>(c, d_reqs, (int)n, [&](const x265_hip_refine_req *h) {
            int r = validate_refine(c, h, (int)n);
            long i = 0;
            for (int s = 0; s < nsections && !r; s++)
                for (int k = 0; k < sections[s].count && !r; k++, i++)
                    if (h[i].pu_w != sections[s].pu_w || h[i].pu_h != sections[s].pu_h)
                        r = fail(X265_HIP_EINVAL, "refine_req[%ld]: %dx%d in a section of %dx%d", i, h[i].pu_w, h[i].pu_h, sections[s].pu_w, sections[s].pu_h);
            return r;
        }, safe >= 0);
        if (vr)
            return vr;
    }
    int r = push_tabs(c);
    if (r)
        return r;
    uint8_t *bad = nullptr;
    if (safe >= 0)
    {
        // (a record of another size than its run's only costs time: the runs size the batches, the kernel takes a
        // request's strips from the request)
        void *sane;
        if ((r = guard_room(c, (int)n, sizeof(x265_hip_refine_req), &sane, &bad)))
            return r;
        LAUNCH(launch_guard_refine(c->dev_tabs, (const x265_hip_refine_req *)d_reqs, (int)n, safe, (x265_hip_refine_req *)sane, bad, c->guard_words,
                                   c->stream));
        d_reqs = sane;
    }
    LAUNCH(launch_refine_sections(c->dev_tabs, (const x265_hip_refine_req *)d_reqs, sections, nsections, (int32_t *)d_costs9, c->batch_counter,
                                  c->batch_parity ^= 1, c->stream));
    if (bad)
        LAUNCH(launch_guard_fix(bad, (int)n, X265_HIP_RING_COSTS, (int32_t *)d_costs9, c->stream));
    return X265_HIP_OK;
}

int x265_hip_refine_batch(x265_hip_ctx *c, const x265_hip_refine_req *reqs, int n, int32_t *costs9)
{
    if (!c || n < 0 || (n && (!reqs || !costs9)))
        return fail(X265_HIP_EINVAL, "refine_batch: bad argument");
    if (n == 0)
        return X265_HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    {
        int vr = validate_refine(c, reqs, n);
        if (vr)
            return vr;
    }
    const size_t ob = (size_t)n * X265_HIP_RING_COSTS * sizeof(int32_t);
    int r;
    if ((r = push_tabs(c)) || (r = ensure(c, c->in, (size_t)n * sizeof(*reqs))) || (r = ensure(c, c->out, ob)))
        return r;
    HIP_TRY(hipMemcpyAsync(c->in.d, reqs, (size_t)n * sizeof(*reqs), hipMemcpyHostToDevice, c->stream));
    // a list made of a few runs of one PU size each goes out with its batches cut per run
    x265_hip_refine_section runs[X265_HIP_MAX_SECTIONS];
    int nruns = 0;
    for (int i = 0; i < n && nruns <= X265_HIP_MAX_SECTIONS; i++)
    {
        if (nruns && reqs[i].pu_w == runs[nruns - 1].pu_w && reqs[i].pu_h == runs[nruns - 1].pu_h)
            runs[nruns - 1].count++;
        else if (nruns++ < X265_HIP_MAX_SECTIONS)
            runs[nruns - 1] = x265_hip_refine_section{ 1, reqs[i].pu_w, reqs[i].pu_h, { 0, 0 } };
    }
    if (nruns <= X265_HIP_MAX_SECTIONS)
        LAUNCH(launch_refine_sections(c->dev_tabs, (const x265_hip_refine_req *)c->in.d, runs, nruns, (int32_t *)c->out.d, c->batch_counter,
                                      c->batch_parity ^= 1, c->stream));
    else
        LAUNCH(launch_refine(c->dev_tabs, (const x265_hip_refine_req *)c->in.d, n, (int32_t *)c->out.d, c->batch_counter, c->batch_parity ^= 1, c->stream));
    HIP_TRY(hipMemcpyAsync(c->out.h, c->out.d, ob, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(costs9, c->out.h, ob);
    return X265_HIP_OK;
}

int x265_hip_dist_batch_device(x265_hip_ctx *c, const void *d_reqs, int n, void *d_costs)
{
    if (!c || n < 0 || (n && (!d_reqs || !d_costs)))
        return fail(X265_HIP_EINVAL, "dist_batch_device: bad argument");
    if (n == 0) // nothing to launch; a launch takes one of the two counter sets and clears the other
        return X265_HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    const int safe = c->check_records == 2 ? guard_safe_plane(c) : -1;
    {
        int vr = checked_records<x265_hip_cost_req>(c, d_reqs, n, [&](const x265_hip_cost_req *h) { return validate_dist(c, h, n); }, safe >= 0);
        if (vr)
            return vr;
    }
    int r = push_tabs(c);
    if (r)
        return r;
    uint8_t *bad = nullptr;
    if (safe >= 0)
    {
        void *sane;
        if ((r = guard_room(c, n, sizeof(x265_hip_cost_req), &sane, &bad)))
            return r;
        LAUNCH(launch_guard_cost(c->dev_tabs, (const x265_hip_cost_req *)d_reqs, n, X265_HIP_SSE, safe, (x265_hip_cost_req *)sane, bad, c->guard_words,
                                 c->stream));
        d_reqs = sane;
    }
    LAUNCH(launch_dist(c->dev_tabs, (const x265_hip_cost_req *)d_reqs, n, (int32_t *)d_costs, c->batch_counter, c->batch_parity ^= 1, c->stream));
    if (bad)
        LAUNCH(launch_guard_fix(bad, n, 1, (int32_t *)d_costs, c->stream));
    return X265_HIP_OK;
}

int x265_hip_dist_batch(x265_hip_ctx *c, const x265_hip_cost_req *reqs, int n, int32_t *costs)
{
    if (!c || n < 0 || (n && (!reqs || !costs)))
        return fail(X265_HIP_EINVAL, "dist_batch: bad argument");
    if (n == 0)
        return X265_HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    {
        int vr = validate_dist(c, reqs, n);
        if (vr)
            return vr;
    }
    int r;
    if ((r = push_tabs(c)) || (r = ensure(c, c->in, (size_t)n * sizeof(*reqs))) ||
        (r = ensure(c, c->out, (size_t)n * sizeof(int32_t))))
        return r;
    HIP_TRY(hipMemcpyAsync(c->in.d, reqs, (size_t)n * sizeof(*reqs), hipMemcpyHostToDevice, c->stream));
    LAUNCH(launch_dist(c->dev_tabs, (const x265_hip_cost_req *)c->in.d, n, (int32_t *)c->out.d, c->batch_counter, c->batch_parity ^= 1, c->stream));
    HIP_TRY(hipMemcpyAsync(c->out.h, c->out.d, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(costs, c->out.h, (size_t)n * sizeof(int32_t));
    return X265_HIP_OK;
}

int x265_hip_predict_batch_device(x265_hip_ctx *c, const void *d_reqs, int n)
{
    if (!c || n < 0 || (n && !d_reqs))
        return fail(X265_HIP_EINVAL, "predict_batch_device: bad argument");
    if (n == 0) // nothing to launch; a launch takes one of the two counter sets and clears the other
        return X265_HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    {
        int vr = checked_records<x265_hip_pred_req>(c, d_reqs, n, [&](const x265_hip_pred_req *h) { return validate_pred(c, h, n); });
        if (vr)
            return vr;
    }
    int r = push_tabs(c);
    if (r)
        return r;
    LAUNCH(launch_predict(c->dev_tabs, (const x265_hip_pred_req *)d_reqs, n, c->batch_counter, c->batch_parity ^= 1, c->stream));
    return X265_HIP_OK;
}

int x265_hip_predict_batch(x265_hip_ctx *c, const x265_hip_pred_req *reqs, int n)
{
    if (!c || n < 0 || (n && !reqs))
        return fail(X265_HIP_EINVAL, "predict_batch: bad argument");
    if (n == 0)
        return X265_HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    {
        int vr = validate_pred(c, reqs, n);
        if (vr)
            return vr;
    }
    int r;
    if ((r = push_tabs(c)) || (r = ensure(c, c->in, (size_t)n * sizeof(*reqs))))
        return r;
    HIP_TRY(hipMemcpyAsync(c->in.d, reqs, (size_t)n * sizeof(*reqs), hipMemcpyHostToDevice, c->stream));
    LAUNCH(launch_predict(c->dev_tabs, (const x265_hip_pred_req *)c->in.d, n, c->batch_counter, c->batch_parity ^= 1, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream)); // `reqs` is read by the copy until here
    return X265_HIP_OK;
}

int x265_hip_mc_satd_batch_device(x265_hip_ctx *c, const void *d_reqs, int n, void *d_costs)
{
    if (!c || n < 0 || (n && (!d_reqs || !d_costs)))
        return fail(X265_HIP_EINVAL, "mc_satd_batch_device: bad argument");
    if (n == 0)
        return X265_HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    const int safe = c->check_records == 2 ? guard_safe_plane(c) : -1;
    {
        int vr = checked_records<x265_hip_pred_req>(c, d_reqs, n, [&](const x265_hip_pred_req *h) { return validate_pred(c, h, n, true); }, safe >= 0);
        if (vr)
            return vr;
    }
    int r = push_tabs(c);
    if (r)
        return r;
    uint8_t *bad = nullptr;
    if (safe >= 0)
    {
        void *sane;
        if ((r = guard_room(c, n, sizeof(x265_hip_pred_req), &sane, &bad)))
            return r;
        LAUNCH(launch_guard_pred(c->dev_tabs, (const x265_hip_pred_req *)d_reqs, n, safe, (x265_hip_pred_req *)sane, bad, c->guard_words, c->stream));
        d_reqs = sane;
    }
    LAUNCH(launch_mc_satd(c->dev_tabs, (const x265_hip_pred_req *)d_reqs, n, (int32_t *)d_costs, c->batch_counter, c->batch_parity ^= 1, c->stream));
    if (bad)
        LAUNCH(launch_guard_fix(bad, n, 1, (int32_t *)d_costs, c->stream));
    return X265_HIP_OK;
}

int x265_hip_mc_satd_batch(x265_hip_ctx *c, const x265_hip_pred_req *reqs, int n, int32_t *costs)
{
    if (!c || n < 0 || (n && (!reqs || !costs)))
        return fail(X265_HIP_EINVAL, "mc_satd_batch: bad argument");
    if (n == 0)
        return X265_HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    {
        int vr = validate_pred(c, reqs, n, true);
        if (vr)
            return vr;
    }
    int r;
    const size_t ob = (size_t)n * sizeof(int32_t);
    if ((r = push_tabs(c)) || (r = ensure(c, c->in, (size_t)n * sizeof(*reqs))) || (r = ensure(c, c->out, ob)))
        return r;
    HIP_TRY(hipMemcpyAsync(c->in.d, reqs, (size_t)n * sizeof(*reqs), hipMemcpyHostToDevice, c->stream));
    LAUNCH(launch_mc_satd(c->dev_tabs, (const x265_hip_pred_req *)c->in.d, n, (int32_t *)c->out.d, c->batch_counter, c->batch_parity ^= 1, c->stream));
    HIP_TRY(hipMemcpyAsync(c->out.h, c->out.d, ob, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(costs, c->out.h, ob);
    return X265_HIP_OK;
}

// ---- xMergeEstimation -----------------------------------------------------------------------------------------------

// the requests the device will make of the units' candidates, put through the checks of a prediction request
static int validate_merge(const x265_hip_ctx *c, const x265_hip_merge_pu *pus, int n, int pic_w, int pic_h)
{
    for (int i = 0; i < n; i++)
    {
        const x265_hip_merge_pu &p = pus[i];
        if (p.num_cand < 1 || p.num_cand > X265_HIP_MRG_MAX_NUM_CANDS || p.weighted > 2)
            return fail(X265_HIP_EINVAL, "merge_pu[%d]: %d candidates / weighted %d out of range", i, p.num_cand, p.weighted);
        int r = check_block(c, p.fenc_plane, p.fenc_plane, p.pu_w, p.pu_h, "merge_pu", i);
        if (r || (r = check_inside(c, p.fenc_plane, p.pu_x, p.pu_y, p.pu_w, p.pu_h, "merge_pu", i)))
            return r;
        for (int k = 0; k < p.num_cand; k++)
        {
            if (p.cand[k].ref_plane[0] < 0 && (p.cand[k].ref_plane[1] < 0 || p.weighted == 1))
                return fail(X265_HIP_EINVAL, "merge_pu[%d]: candidate %d has no reference", i, k);
            x265_hip_pred_req q;
            if (merge_make_req(p, k, pic_w, pic_h, q) && (r = validate_pred(c, &q, 1, true)))
            {
                const std::string why = x265_hip_last_error();
                return fail(r, "merge_pu[%d] candidate %d: %s", i, k, why.c_str());
            }
        }
    }
    return 0;
}

#define X265HIP_MERGE_SCRATCH_BYTES(n) ((size_t)(n) * X265_HIP_MRG_MAX_NUM_CANDS * (sizeof(x265_hip_pred_req) + sizeof(int32_t)))

static int merge_launch(x265_hip_ctx *c, const x265_hip_merge_pu *d_pus, int n, int pic_w, int pic_h, x265_hip_merge_result *d_out)
{
    int r;
    if (c->mejobs.cap < X265HIP_MERGE_SCRATCH_BYTES(n))
        HIP_TRY(hipStreamSynchronize(c->stream)); // an earlier launch may still read the buffer that is about to be replaced
    if ((r = push_tabs(c)) || (r = ensure(c, c->mejobs, X265HIP_MERGE_SCRATCH_BYTES(n), false)))
        return r;
    x265_hip_pred_req *reqs = (x265_hip_pred_req *)c->mejobs.d;
    int32_t *satd = (int32_t *)(reqs + (size_t)n * X265_HIP_MRG_MAX_NUM_CANDS);
    LAUNCH(launch_merge_estimate(c->dev_tabs, d_pus, n, pic_w, pic_h, reqs, satd, d_out, c->batch_counter, c->batch_parity ^= 1, c->stream));
    return X265_HIP_OK;
}

int x265_hip_merge_estimate_batch_device(x265_hip_ctx *c, const void *d_pus, int n, int pic_w, int pic_h, void *d_results)
{
    if (!c || n < 0 || pic_w <= 0 || pic_h <= 0 || (n && (!d_pus || !d_results)))
        return fail(X265_HIP_EINVAL, "merge_estimate_batch_device: bad argument");
    if (n == 0)
        return X265_HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    {
        int vr = checked_records<x265_hip_merge_pu>(c, d_pus, n, [&](const x265_hip_merge_pu *h) { return validate_merge(c, h, n, pic_w, pic_h); });
        if (vr)
            return vr;
    }
    return merge_launch(c, (const x265_hip_merge_pu *)d_pus, n, pic_w, pic_h, (x265_hip_merge_result *)d_results);
}

int x265_hip_merge_estimate_batch(x265_hip_ctx *c, const x265_hip_merge_pu *pus, int n, int pic_w, int pic_h, x265_hip_merge_result *results)
{
    if (!c || n < 0 || pic_w <= 0 || pic_h <= 0 || (n && (!pus || !results)))
        return fail(X265_HIP_EINVAL, "merge_estimate_batch: bad argument");
    if (n == 0)
        return X265_HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    {
        int vr = validate_merge(c, pus, n, pic_w, pic_h);
        if (vr)
            return vr;
    }
    int r;
    const size_t ob = (size_t)n * sizeof(*results);
    if ((r = ensure(c, c->in, (size_t)n * sizeof(*pus))) || (r = ensure(c, c->out, ob)))
        return r;
    HIP_TRY(hipMemcpyAsync(c->in.d, pus, (size_t)n * sizeof(*pus), hipMemcpyHostToDevice, c->stream));
    if ((r = merge_launch(c, (const x265_hip_merge_pu *)c->in.d, n, pic_w, pic_h, (x265_hip_merge_result *)c->out.d)))
        return r;
    HIP_TRY(hipMemcpyAsync(c->out.h, c->out.d, ob, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(results, c->out.h, ob);
    return X265_HIP_OK;
}

int x265_hip_bidir_cost_batch_device(x265_hip_ctx *c, const void *d_reqs, int n, void *d_costs)
{
    if (!c || n < 0 || (n && (!d_reqs || !d_costs)))
        return fail(X265_HIP_EINVAL, "bidir_cost_batch_device: bad argument");
    if (n == 0) // nothing to launch; a launch takes one of the two counter sets and clears the other
        return X265_HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    const int safe = c->check_records == 2 ? guard_safe_plane(c) : -1;
    {
        int vr = checked_records<x265_hip_bidir_req>(c, d_reqs, n, [&](const x265_hip_bidir_req *h) { return validate_bidir(c, h, n); }, safe >= 0);
        if (vr)
            return vr;
    }
    int r = push_tabs(c);
    if (r)
        return r;
    uint8_t *bad = nullptr;
    if (safe >= 0)
    {
        void *sane;
        if ((r = guard_room(c, n, sizeof(x265_hip_bidir_req), &sane, &bad)))
            return r;
        LAUNCH(launch_guard_bidir(c->dev_tabs, (const x265_hip_bidir_req *)d_reqs, n, safe, (x265_hip_bidir_req *)sane, bad, c->guard_words, c->stream));
        d_reqs = sane;
    }
    LAUNCH(launch_bidir_cost(c->dev_tabs, (const x265_hip_bidir_req *)d_reqs, n, (int32_t *)d_costs, c->batch_counter, c->batch_parity ^= 1, c->stream));
    if (bad)
        LAUNCH(launch_guard_fix(bad, n, 1, (int32_t *)d_costs, c->stream));
    return X265_HIP_OK;
}

int x265_hip_bidir_cost_batch(x265_hip_ctx *c, const x265_hip_bidir_req *reqs, int n, int32_t *costs)
{
    if (!c || n < 0 || (n && (!reqs || !costs)))
        return fail(X265_HIP_EINVAL, "bidir_cost_batch: bad argument");
    if (n == 0)
        return X265_HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    {
        int vr = validate_bidir(c, reqs, n);
        if (vr)
            return vr;
    }
    int r;
    if ((r = push_tabs(c)) || (r = ensure(c, c->in, (size_t)n * sizeof(*reqs))) ||
        (r = ensure(c, c->out, (size_t)n * sizeof(int32_t))))
        return r;
    HIP_TRY(hipMemcpyAsync(c->in.d, reqs, (size_t)n * sizeof(*reqs), hipMemcpyHostToDevice, c->stream));
    LAUNCH(launch_bidir_cost(c->dev_tabs, (const x265_hip_bidir_req *)c->in.d, n, (int32_t *)c->out.d, c->batch_counter, c->batch_parity ^= 1, c->stream));
    HIP_TRY(hipMemcpyAsync(c->out.h, c->out.d, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(costs, c->out.h, (size_t)n * sizeof(int32_t));
    return X265_HIP_OK;
}

int x265_hip_fullsearch_device(x265_hip_ctx *c, const void *d_jobs, int n, void *d_results)
{
    if (!c || n < 0 || (n && (!d_jobs || !d_results)))
        return fail(X265_HIP_EINVAL, "fullsearch_device: bad argument");
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    if (n == 0)
        return X265_HIP_OK;
    int safe = c->check_records == 2 ? guard_safe_plane(c) : -1, safe_qp = -1;
    for (int q = 0; q < X265_HIP_MAX_QP && safe_qp < 0; q++)
        if (c->host_tabs.costtab[q])
            safe_qp = q;
    if (safe_qp < 0)
        safe = -1;
    {
        int vr = checked_records<x265_hip_ctu_job>(c, d_jobs, n, [&](const x265_hip_ctu_job *h) { return validate_ctu(c, h, n); }, safe >= 0);
        if (vr)
            return vr;
    }
    int r = push_tabs(c);
    if (r)
        return r;
    uint8_t *bad = nullptr;
    if (safe >= 0)
    {
        void *sane;
        if ((r = guard_room(c, n, sizeof(x265_hip_ctu_job), &sane, &bad)))
            return r;
        LAUNCH(launch_guard_ctu(c->dev_tabs, (const x265_hip_ctu_job *)d_jobs, n, safe, safe_qp, (x265_hip_ctu_job *)sane, bad, c->guard_words, c->stream));
        d_jobs = sane;
    }
    LAUNCH(launch_fullsearch2(c->dev_tabs, (const x265_hip_ctu_job *)d_jobs, n, (x265_hip_me_result *)d_results, c->stream));
    if (bad) // 85 results of two words each per CTU: cost -1, vector zero
        LAUNCH(launch_guard_fix(bad, n, 2 * X265_HIP_PUS_PER_CTU, (int32_t *)d_results, c->stream, true));
    return X265_HIP_OK;
}

int x265_hip_fullsearch(x265_hip_ctx *c, const x265_hip_ctu_job *jobs, int n, x265_hip_me_result *results)
{
    if (!c || n < 0 || (n && (!jobs || !results)))
        return fail(X265_HIP_EINVAL, "fullsearch: bad argument");
    if (n == 0)
        return X265_HIP_OK;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    {
        int vr = validate_ctu(c, jobs, n);
        if (vr)
            return vr;
    }
    const size_t ob = (size_t)n * X265_HIP_PUS_PER_CTU * sizeof(*results);
    int r;
    if ((r = push_tabs(c)) || (r = ensure(c, c->in, (size_t)n * sizeof(*jobs))) || (r = ensure(c, c->out, ob)))
        return r;
    memcpy(c->in.h, jobs, (size_t)n * sizeof(*jobs));
    HIP_TRY(hipMemcpyAsync(c->in.d, c->in.h, (size_t)n * sizeof(*jobs), hipMemcpyHostToDevice, c->stream));
    LAUNCH(launch_fullsearch2(c->dev_tabs, (const x265_hip_ctu_job *)c->in.d, n, (x265_hip_me_result *)c->out.d, c->stream));
    HIP_TRY(hipMemcpyAsync(c->out.h, c->out.d, ob, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(results, c->out.h, ob);
    return X265_HIP_OK;
}

} // extern "C"

// =================================================================================================
// primitive level: host-pointer thunks with the reference's signatures

namespace {

x265_hip_ctx *g_prim_ctx = nullptr; // made by x265_hip_setup_primitives; owns the host-plane registry's copies
int g_prim_device = 0;
std::mutex g_prim_init_mu;

// The table's entries are called concurrently by every pool thread of the encoder (one MotionEstimate per CTU row,
// SURVEY.md 8(b) "Threading").  Each calling thread gets a context of its own (stream, staging buffers) the first time
// it calls an entry, so calls from different threads neither wait for each other's mutex nor share a stream; the
// context of a thread is destroyed when the thread ends (an encoder's pool threads are created and joined per encoder
// instance).
struct PrimCtxHolder
{
    x265_hip_ctx *c = nullptr;
    ~PrimCtxHolder()
    {
        if (c)
            x265_hip_ctx_destroy(c);
    }
};

x265_hip_ctx *prim_ctx()
{
    static thread_local PrimCtxHolder t_ctx;
    if (t_ctx.c)
        return t_ctx.c;
    if (!g_prim_ctx)
        return nullptr;
    x265_hip_ctx *c = nullptr;
    if (x265_hip_ctx_create(g_prim_device, &c) != X265_HIP_OK)
        return nullptr;
    return t_ctx.c = c;
}

struct HostPlane
{
    const uint8_t *base;
    size_t bytes;
    uint8_t *dev;
};
std::vector<HostPlane> g_host_planes;
std::shared_mutex g_host_planes_mu;

// device address serving host address p (+- the bytes a primitive may touch), or null.  The caller holds
// g_host_planes_mu shared from here until its kernel has finished: x265_hip_unregister_host_plane takes the lock
// exclusively before it frees a copy, so no primitive is between resolve and the end of its launch at that point.
const uint8_t *resolve(const uint8_t *p, long before, long after)
{
    for (const HostPlane &hp : g_host_planes)
        if (p - before >= hp.base && p + after <= hp.base + hp.bytes)
            return hp.dev + (p - hp.base);
    return nullptr;
}

[[noreturn]] void prim_fatal(const char *what)
{
    // the reference's primitives have no error channel; a silent wrong cost would be worse than a stop
    fprintf(stderr, "x265_hip primitive %s failed: %s\n", what, t_err);
    abort();
}

// copy a w x rows region (elements of `es` bytes) starting `pre_cols` left and `pre_rows` above p
// into packed staging; returns the packed stride in elements
int pack_region(uint8_t *dst, const uint8_t *p, long stride, int es, int w, int rows, int pre_cols, int pre_rows)
{
    const uint8_t *s = p - ((long)pre_rows * stride + pre_cols) * es;
    for (int y = 0; y < rows; y++)
        memcpy(dst + (size_t)y * w * es, s + (long)y * stride * es, (size_t)w * es);
    return w;
}

int prim_cmp(int part, int metric, const uint8_t *fenc, intptr_t fstride, const uint8_t *const *fref, int nref,
             intptr_t rstride, int32_t *res)
{
    x265_hip_ctx *c = prim_ctx();
    if (!c)
        return fail(X265_HIP_ENODEV, "x265_hip_setup_primitives has not succeeded");
    const int w = k_part_w[part], h = k_part_h[part];
    std::lock_guard<std::mutex> g(c->mu);
    std::shared_lock<std::shared_mutex> planes(g_host_planes_mu);
    HIP_TRY(hipSetDevice(c->device));
    const size_t blk = (size_t)w * h;
    int r;
    if ((r = ensure(c, c->prim_a, 5 * blk + 64)) || (r = ensure(c, c->prim_b, 64)))
        return r;
    uint8_t *hs = (uint8_t *)c->prim_a.h, *ds = (uint8_t *)c->prim_a.d;
    PrimCmpArgs a;
    memset(&a, 0, sizeof(a));
    a.w = w;
    a.h = h;
    a.nref = nref;
    a.metric = metric;
    size_t used = 0;
    pack_region(hs, fenc, fstride, 1, w, h, 0, 0);
    a.fenc = ds;
    a.fstride = w;
    used = blk;
    bool all_resolved = true;
    const uint8_t *dev[4];
    for (int i = 0; i < nref; i++)
    {
        dev[i] = resolve(fref[i], 0, (long)(h - 1) * rstride + w);
        all_resolved = all_resolved && dev[i];
    }
    if (all_resolved)
    {
        for (int i = 0; i < nref; i++)
            a.fref[i] = dev[i];
        a.rstride = (int)rstride;
    }
    else
    {
        for (int i = 0; i < nref; i++)
        {
            pack_region(hs + used, fref[i], rstride, 1, w, h, 0, 0);
            a.fref[i] = ds + used;
            used += blk;
        }
        a.rstride = w;
    }
    HIP_TRY(hipMemcpyAsync(ds, hs, used, hipMemcpyHostToDevice, c->stream));
    LAUNCH(launch_prim_cmp(a, (int32_t *)c->prim_b.d, c->stream));
    HIP_TRY(hipMemcpyAsync(c->prim_b.h, c->prim_b.d, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(res, c->prim_b.h, nref * sizeof(int32_t));
    return 0;
}

// kind: PRIM_HPP / PRIM_VPP / PRIM_HPS / PRIM_VSP.  Writes exactly w x rows elements of dst.
int prim_filter_wh(int w, int rows, int kind, const void *src, intptr_t sstride, void *dst, intptr_t dstride, int coeff, int row_ext)
{
    x265_hip_ctx *c = prim_ctx();
    if (!c)
        return fail(X265_HIP_ENODEV, "x265_hip_setup_primitives has not succeeded");
    if (coeff < 0 || coeff > 3)
        return fail(X265_HIP_EINVAL, "coeffIdx %d out of range", coeff);
    if (w <= 0 || rows <= 0)
        return 0;
    const int ses = kind == PRIM_VSP || kind == PRIM_VSS ? 2 : 1;
    const int des = kind == PRIM_HPS || kind == PRIM_VPS || kind == PRIM_VSS || kind == PRIM_P2S ? 2 : 1;
    const uint8_t *sp = (const uint8_t *)src;
    if (kind == PRIM_HPS && row_ext)
    {
        sp -= 3 * sstride * ses; // luma_hps with isRowExt starts three rows up (ipfilter.cpp:128-132)
        rows += 7;
    }
    const bool horiz = kind == PRIM_HPP || kind == PRIM_HPS, vert = !horiz && kind != PRIM_P2S;
    const int pre_c = horiz ? 3 : 0, post_c = horiz ? 4 : 0, pre_r = vert ? 3 : 0, post_r = vert ? 4 : 0;
    const int sw = w + pre_c + post_c, sr = rows + pre_r + post_r;
    std::lock_guard<std::mutex> g(c->mu);
    std::shared_lock<std::shared_mutex> planes(g_host_planes_mu);
    HIP_TRY(hipSetDevice(c->device));
    int r;
    const size_t sbytes = (size_t)sw * sr * ses, dbytes = (size_t)w * rows * des;
    if ((r = ensure(c, c->prim_a, sbytes + 64)) || (r = ensure(c, c->prim_b, dbytes + 64)))
        return r;
    PrimFilterArgs a;
    a.kind = kind;
    a.coeff = coeff;
    a.w = w;
    a.rows = rows;
    a.dst = c->prim_b.d;
    a.dstride = w;
    const uint8_t *dev = ses == 1 ? resolve(sp, (long)pre_r * sstride + pre_c, (long)(rows - 1 + post_r) * sstride + w + post_c)
                                  : nullptr;
    if (dev)
    {
        a.src = dev;
        a.sstride = (int)sstride;
    }
    else
    {
        pack_region((uint8_t *)c->prim_a.h, sp, sstride, ses, sw, sr, pre_c, pre_r);
        HIP_TRY(hipMemcpyAsync(c->prim_a.d, c->prim_a.h, sbytes, hipMemcpyHostToDevice, c->stream));
        a.src = (const uint8_t *)c->prim_a.d + ((size_t)pre_r * sw + pre_c) * ses;
        a.sstride = sw;
    }
    LAUNCH(launch_prim_filter(a, c->stream));
    HIP_TRY(hipMemcpyAsync(c->prim_b.h, c->prim_b.d, dbytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int y = 0; y < rows; y++)
        memcpy((uint8_t *)dst + (size_t)y * dstride * des, (uint8_t *)c->prim_b.h + (size_t)y * w * des, (size_t)w * des);
    return 0;
}

int prim_filter(int part, int kind, const void *src, intptr_t sstride, void *dst, intptr_t dstride, int coeff, int row_ext)
{
    return prim_filter_wh(k_part_w[part], k_part_h[part], kind, src, sstride, dst, dstride, coeff, row_ext);
}

template <int P>
int t_sad(uint8_t *fenc, intptr_t fs, uint8_t *fref, intptr_t rs)
{
    int32_t res[4];
    const uint8_t *r[1] = { fref };
    if (prim_cmp(P, X265_HIP_SAD, fenc, fs, r, 1, rs, res))
        prim_fatal("sad");
    return res[0];
}

template <int P>
int t_satd(uint8_t *fenc, intptr_t fs, uint8_t *fref, intptr_t rs)
{
    int32_t res[4];
    const uint8_t *r[1] = { fref };
    if (prim_cmp(P, X265_HIP_SATD, fenc, fs, r, 1, rs, res))
        prim_fatal("satd");
    return res[0];
}

template <int P>
int t_sa8d_inter(uint8_t *fenc, intptr_t fs, uint8_t *fref, intptr_t rs)
{
    int32_t res[4];
    const uint8_t *r[1] = { fref };
    if (prim_cmp(P, X265_HIP_SA8D, fenc, fs, r, 1, rs, res))
        prim_fatal("sa8d_inter");
    return res[0];
}

template <int P>
int t_sse_pp(uint8_t *fenc, intptr_t fs, uint8_t *fref, intptr_t rs)
{
    int32_t res[4];
    const uint8_t *r[1] = { fref };
    if (prim_cmp(P, X265_HIP_SSE, fenc, fs, r, 1, rs, res))
        prim_fatal("sse_pp");
    return res[0];
}

template <int P>
void t_sad_x3(uint8_t *fenc, uint8_t *r0, uint8_t *r1, uint8_t *r2, intptr_t rs, int32_t *res)
{
    const uint8_t *r[3] = { r0, r1, r2 };
    if (prim_cmp(P, X265_HIP_SAD, fenc, X265_HIP_FENC_STRIDE, r, 3, rs, res))
        prim_fatal("sad_x3");
}

template <int P>
void t_sad_x4(uint8_t *fenc, uint8_t *r0, uint8_t *r1, uint8_t *r2, uint8_t *r3, intptr_t rs, int32_t *res)
{
    const uint8_t *r[4] = { r0, r1, r2, r3 };
    if (prim_cmp(P, X265_HIP_SAD, fenc, X265_HIP_FENC_STRIDE, r, 4, rs, res))
        prim_fatal("sad_x4");
}

template <int P>
void t_hpp(uint8_t *src, intptr_t ss, uint8_t *dst, intptr_t ds, int coeff)
{
    if (prim_filter(P, PRIM_HPP, src, ss, dst, ds, coeff, 0))
        prim_fatal("luma_hpp");
}

template <int P>
void t_vpp(uint8_t *src, intptr_t ss, uint8_t *dst, intptr_t ds, int coeff)
{
    if (prim_filter(P, PRIM_VPP, src, ss, dst, ds, coeff, 0))
        prim_fatal("luma_vpp");
}

template <int P>
void t_hps(uint8_t *src, intptr_t ss, int16_t *dst, intptr_t ds, int coeff, int ext)
{
    if (prim_filter(P, PRIM_HPS, src, ss, dst, ds, coeff, ext))
        prim_fatal("luma_hps");
}

template <int P>
void t_vsp(int16_t *src, intptr_t ss, uint8_t *dst, intptr_t ds, int coeff)
{
    if (prim_filter(P, PRIM_VSP, src, ss, dst, ds, coeff, 0))
        prim_fatal("luma_vsp");
}

template <int P>
void t_vps(uint8_t *src, intptr_t ss, int16_t *dst, intptr_t ds, int coeff)
{
    if (prim_filter(P, PRIM_VPS, src, ss, dst, ds, coeff, 0))
        prim_fatal("luma_vps");
}

template <int P>
void t_vss(int16_t *src, intptr_t ss, int16_t *dst, intptr_t ds, int coeff)
{
    if (prim_filter(P, PRIM_VSS, src, ss, dst, ds, coeff, 0))
        prim_fatal("luma_vss");
}

// luma_hvpp = luma_hps with the row extension into a block-wide buffer, then luma_vsp (ipfilter.cpp:362-369)
template <int P>
void t_hvpp(uint8_t *src, intptr_t ss, uint8_t *dst, intptr_t ds, int idx_x, int idx_y)
{
    static thread_local int16_t immed[64 * (64 + 8)];
    const int w = k_part_w[P];
    if (prim_filter(P, PRIM_HPS, src, ss, immed, w, idx_x, 1) || prim_filter(P, PRIM_VSP, immed + 3 * w, w, dst, ds, idx_y, 0))
        prim_fatal("luma_hvpp");
}

// luma_p2s: the destination stride is MAX_CU_SIZE (filterConvertPelToShort_c<MAX_CU_SIZE>, ipfilter.cpp:510)
void t_p2s(uint8_t *src, intptr_t ss, int16_t *dst, int width, int height)
{
    if (prim_filter_wh(width, height, PRIM_P2S, src, ss, dst, 64, 0, 0))
        prim_fatal("luma_p2s");
}

// rows of `rowbytes` from host memory into the pinned mirror at `pitch`, then to the device
int stage_rows(x265_hip_ctx *c, Buf &b, size_t off, const uint8_t *src, long sstride, int rowbytes, int rows, int pitch)
{
    for (int y = 0; y < rows; y++)
        memcpy((uint8_t *)b.h + off + (size_t)y * pitch, src + (long)y * sstride, rowbytes);
    HIP_TRY(hipMemcpyAsync((uint8_t *)b.d + off, (uint8_t *)b.h + off, (size_t)rows * pitch, hipMemcpyHostToDevice, c->stream));
    return 0;
}

int prim_avg(int part, uint8_t *dst, intptr_t ds, const uint8_t *s0, intptr_t ss0, const uint8_t *s1, intptr_t ss1)
{
    x265_hip_ctx *c = prim_ctx();
    if (!c)
        return fail(X265_HIP_ENODEV, "x265_hip_setup_primitives has not succeeded");
    const int w = k_part_w[part], h = k_part_h[part];
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    int r;
    const size_t blk = (size_t)w * h;
    if ((r = ensure(c, c->prim_a, 2 * blk + 64)) || (r = ensure(c, c->prim_b, blk + 64)) ||
        (r = stage_rows(c, c->prim_a, 0, s0, ss0, w, h, w)) || (r = stage_rows(c, c->prim_a, blk, s1, ss1, w, h, w)))
        return r;
    PrimAvgArgs a = { (const uint8_t *)c->prim_a.d, (const uint8_t *)c->prim_a.d + blk, (uint8_t *)c->prim_b.d, w, w, w, w, h };
    LAUNCH(launch_prim_avg(a, c->stream));
    HIP_TRY(hipMemcpyAsync(c->prim_b.h, c->prim_b.d, blk, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int y = 0; y < h; y++)
        memcpy(dst + (long)y * ds, (uint8_t *)c->prim_b.h + (size_t)y * w, w);
    return 0;
}

int prim_addavg(int part, const int16_t *s0, const int16_t *s1, uint8_t *dst, intptr_t ss0, intptr_t ss1, intptr_t ds)
{
    x265_hip_ctx *c = prim_ctx();
    if (!c)
        return fail(X265_HIP_ENODEV, "x265_hip_setup_primitives has not succeeded");
    const int w = k_part_w[part], h = k_part_h[part];
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    int r;
    const size_t blk = (size_t)w * h;
    if ((r = ensure(c, c->prim_a, 4 * blk + 64)) || (r = ensure(c, c->prim_b, blk + 64)) ||
        (r = stage_rows(c, c->prim_a, 0, (const uint8_t *)s0, 2 * ss0, 2 * w, h, 2 * w)) ||
        (r = stage_rows(c, c->prim_a, 2 * blk, (const uint8_t *)s1, 2 * ss1, 2 * w, h, 2 * w)))
        return r;
    PrimAddAvgArgs a = { (const int16_t *)c->prim_a.d, (const int16_t *)c->prim_a.d + blk, (uint8_t *)c->prim_b.d, w, w, w, w, h };
    LAUNCH(launch_prim_addavg(a, c->stream));
    HIP_TRY(hipMemcpyAsync(c->prim_b.h, c->prim_b.d, blk, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int y = 0; y < h; y++)
        memcpy(dst + (long)y * ds, (uint8_t *)c->prim_b.h + (size_t)y * w, w);
    return 0;
}

template <int P>
void t_addavg(int16_t *s0, int16_t *s1, uint8_t *dst, intptr_t ss0, intptr_t ss1, intptr_t ds)
{
    if (prim_addavg(P, s0, s1, dst, ss0, ss1, ds))
        prim_fatal("luma_addAvg");
}

int prim_weight(const uint8_t *src, uint8_t *dst, intptr_t ss, intptr_t ds, int width, int height, int w0, int round, int shift,
                int offset)
{
    x265_hip_ctx *c = prim_ctx();
    if (!c)
        return fail(X265_HIP_ENODEV, "x265_hip_setup_primitives has not succeeded");
    if (width <= 0 || height <= 0)
        return 0;
    const int pitch = (width + 3) & ~3;
    const size_t blk = (size_t)pitch * height;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    int r;
    if ((r = ensure(c, c->prim_a, blk + 64)) || (r = ensure(c, c->prim_b, blk + 64)) ||
        (r = stage_rows(c, c->prim_a, 0, src, ss, width, height, pitch)))
        return r;
    PlaneBorderArgs a;
    a.src = (const uint8_t *)c->prim_a.d;
    a.dst = (uint8_t *)c->prim_b.d;
    a.stride = pitch;
    a.width = width;
    a.height = height;
    a.margin_x = a.margin_y = 0;
    a.py0 = 0;
    a.pyn = height;
    a.weighted = 1;
    a.w0 = w0;
    a.round = round;
    a.shift = shift;
    a.offset = offset;
    LAUNCH(launch_plane_border(a, c->stream));
    HIP_TRY(hipMemcpyAsync(c->prim_b.h, c->prim_b.d, blk, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int y = 0; y < height; y++)
        memcpy(dst + (long)y * ds, (uint8_t *)c->prim_b.h + (size_t)y * pitch, width);
    return 0;
}

// extendCURowColBorder (ipfilter.cpp:56-74): the rows' first and last samples into their left and right margins
int prim_extend_rows(uint8_t *txt, intptr_t stride, int width, int height, int mx)
{
    x265_hip_ctx *c = prim_ctx();
    if (!c)
        return fail(X265_HIP_ENODEV, "x265_hip_setup_primitives has not succeeded");
    if (width <= 0 || height <= 0 || mx <= 0)
        return 0;
    const int pitch = (width + 2 * mx + 3) & ~3; // staged rows start at their left margin, 4-byte aligned
    const size_t blk = (size_t)pitch * height;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    int r;
    if ((r = ensure(c, c->prim_a, blk + 64)) || (r = stage_rows(c, c->prim_a, mx, txt, stride, width, height, pitch)))
        return r;
    PlaneBorderArgs a = {};
    a.src = (const uint8_t *)c->prim_a.d + mx;
    a.dst = (uint8_t *)c->prim_a.d + mx;
    a.stride = pitch;
    a.width = width;
    a.height = height;
    a.margin_x = mx;
    a.margin_y = 0;
    a.py0 = 0;
    a.pyn = height;
    LAUNCH(launch_plane_border(a, c->stream));
    HIP_TRY(hipMemcpyAsync(c->prim_a.h, c->prim_a.d, blk, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int y = 0; y < height; y++)
    {
        const uint8_t *row = (const uint8_t *)c->prim_a.h + (size_t)y * pitch;
        memcpy(txt + (long)y * stride - mx, row, mx);
        memcpy(txt + (long)y * stride + width, row + mx + width, mx);
    }
    return 0;
}

int prim_lowres(const uint8_t *src0, uint8_t *const dst[4], intptr_t ss, intptr_t ds, int width, int height)
{
    x265_hip_ctx *c = prim_ctx();
    if (!c)
        return fail(X265_HIP_ENODEV, "x265_hip_setup_primitives has not succeeded");
    if (width <= 0 || height <= 0)
        return 0;
    // the reference reads 2 height + 1 rows of 2 width + 1 samples; the kernel's dword loads may run up to
    // seven bytes further inside the padded staging rows
    const int srows = 2 * height + 1, scols = 2 * width + 1, spitch = (scols + 8 + 3) & ~3;
    const int dpitch = (width + 3) & ~3;
    const size_t sblk = (size_t)spitch * srows, dblk = (size_t)dpitch * height;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    int r;
    if ((r = ensure(c, c->prim_a, sblk + 64)) || (r = ensure(c, c->prim_b, 4 * dblk + 64)))
        return r;
    memset(c->prim_a.h, 0, sblk);
    if ((r = stage_rows(c, c->prim_a, 0, src0, ss, scols, srows, spitch)))
        return r;
    LowresArgs a;
    a.src = (const uint8_t *)c->prim_a.d;
    a.src_stride = spitch;
    a.dst_stride = dpitch;
    a.width = width;
    a.lines = height;
    a.margin_x = a.margin_y = 0;
    for (int k = 0; k < 4; k++)
        a.dst[k] = (uint8_t *)c->prim_b.d + k * dblk;
    LAUNCH(launch_lowres_init(a, c->stream));
    HIP_TRY(hipMemcpyAsync(c->prim_b.h, c->prim_b.d, 4 * dblk, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int k = 0; k < 4; k++)
        for (int y = 0; y < height; y++)
            memcpy(dst[k] + (long)y * ds, (uint8_t *)c->prim_b.h + k * dblk + (size_t)y * dpitch, width);
    return 0;
}

template <int P>
void t_avg(uint8_t *dst, intptr_t ds, uint8_t *s0, intptr_t ss0, uint8_t *s1, intptr_t ss1, int)
{
    if (prim_avg(P, dst, ds, s0, ss0, s1, ss1))
        prim_fatal("pixelavg_pp");
}

void t_weight(uint8_t *src, uint8_t *dst, intptr_t ss, intptr_t ds, int width, int height, int w0, int round, int shift, int offset)
{
    if (prim_weight(src, dst, ss, ds, width, height, w0, round, shift, offset))
        prim_fatal("weight_pp");
}

void t_extend(uint8_t *txt, intptr_t stride, int width, int height, int mx)
{
    if (prim_extend_rows(txt, stride, width, height, mx))
        prim_fatal("extendRowBorder");
}

void t_lowres(uint8_t *src0, uint8_t *dstf, uint8_t *dsth, uint8_t *dstv, uint8_t *dstc, intptr_t ss, intptr_t ds, int width,
              int height)
{
    uint8_t *const dst[4] = { dstf, dsth, dstv, dstc };
    if (prim_lowres(src0, dst, ss, ds, width, height))
        prim_fatal("frame_init_lowres_core");
}

template <int P>
void fill(x265_hip_primitives *t)
{
    t->pixelavg_pp[P] = t_avg<P>;
    t->sad[P] = t_sad<P>;
    t->sad_x3[P] = t_sad_x3<P>;
    t->sad_x4[P] = t_sad_x4<P>;
    t->satd[P] = t_satd<P>;
    t->sa8d_inter[P] = t_sa8d_inter<P>;
    t->sse_pp[P] = t_sse_pp<P>;
    t->luma_vps[P] = t_vps<P>;
    t->luma_vss[P] = t_vss<P>;
    t->luma_hvpp[P] = t_hvpp<P>;
    t->luma_addAvg[P] = t_addavg<P>;
    t->luma_hpp[P] = t_hpp<P>;
    t->luma_hps[P] = t_hps<P>;
    t->luma_vpp[P] = t_vpp<P>;
    t->luma_vsp[P] = t_vsp<P>;
    if constexpr (P + 1 < X265_HIP_NUM_LUMA_PARTITIONS)
        fill<P + 1>(t);
}

} // namespace

extern "C" {

int x265_hip_setup_primitives(x265_hip_primitives *table, int device)
{
    if (!table)
        return fail(X265_HIP_EINVAL, "null table");
    std::lock_guard<std::mutex> g(g_prim_init_mu);
    if (!g_prim_ctx)
    {
        x265_hip_ctx *c = nullptr;
        int r = x265_hip_ctx_create(device, &c);
        if (r)
            return r;
        g_prim_ctx = c;
        g_prim_device = device;
    }
    fill<0>(table);
    table->luma_p2s = t_p2s;
    table->weight_pp = t_weight;
    table->frame_init_lowres_core = t_lowres;
    table->extendRowBorder = t_extend;
    return X265_HIP_OK;
}

int x265_hip_register_host_plane(const uint8_t *host_base, size_t bytes)
{
    x265_hip_ctx *c = g_prim_ctx;
    if (!c)
        return fail(X265_HIP_ENODEV, "x265_hip_setup_primitives has not succeeded");
    if (!host_base || !bytes)
        return fail(X265_HIP_EINVAL, "register_host_plane: bad argument");
    uint8_t *dev = nullptr;
    {
        std::lock_guard<std::mutex> g(c->mu);
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipMalloc((void **)&dev, bytes + 64));
        HIP_TRY(hipMemcpy(dev, host_base, bytes, hipMemcpyHostToDevice));
    }
    std::unique_lock<std::shared_mutex> g(g_host_planes_mu);
    g_host_planes.push_back({ host_base, bytes, dev });
    return X265_HIP_OK;
}

int x265_hip_unregister_host_plane(const uint8_t *host_base)
{
    x265_hip_ctx *c = g_prim_ctx;
    if (!c)
        return fail(X265_HIP_ENODEV, "x265_hip_setup_primitives has not succeeded");
    uint8_t *dev = nullptr;
    {
        // exclusive: every primitive that resolved an address inside the copy has finished its kernel (they hold the lock
        // shared until then), and none can find the copy afterwards
        std::unique_lock<std::shared_mutex> g(g_host_planes_mu);
        for (size_t i = 0; i < g_host_planes.size(); i++)
            if (g_host_planes[i].base == host_base)
            {
                dev = g_host_planes[i].dev;
                g_host_planes.erase(g_host_planes.begin() + i);
                break;
            }
    }
    if (!dev)
        return fail(X265_HIP_EINVAL, "unregister_host_plane: not registered");
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipFree(dev));
    return X265_HIP_OK;
}

} // extern "C"
