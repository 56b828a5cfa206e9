// See engine.hpp.  Handle life cycle, inputs, ordering + symbolic analysis and launch plans, the gen_feast driver loop.
// Reference semantics: src/feast.jl:10-80, 89-156; src/utils.jl:69-77, 104-116, 151-179.
#include "engine_impl.hpp"
#include "ordering.hpp"
#include <memory>
#include <unistd.h>

namespace feasthip {

// Large device blocks (factor arena, block vectors, contribution stacks) are kept in a small process-wide cache when a handle
// releases them and handed to the next handle that asks for a block of about that size: FEAST is typically run several times
// per process (other contours / spectrum slices), and a fresh multi-GB hipMalloc costs 1 ms to 0.8 s depending on what the
// driver has to map.  What stays cached is bounded by ONE handle's footprint: when a handle is destroyed, whatever earlier
// handles left on its device is returned to the driver first (pool_drop_device), then its own blocks are cached -- other
// allocators of the process (torch, RCCL, the caller's hipMalloc) never face more idle memory than the last solver run used.
// feasthip_trim_memory() returns the cached blocks to the driver; so does an allocation that fails.  FEASTHIP_POOL_GB caps the
// cache (default 96, a third of the 288 GB part: dual_gen_feast on C3 holds 55 GB; 0 switches it off), FEASTHIP_POOL_POISON=1 fills a recycled block with 0xFF bytes before it is handed out
// (NaN patterns: a dependence on what a previous problem left in a block then shows in the tests).
namespace {
struct CachedBlock { int dev; void* p; size_t bytes; };
std::mutex g_pool_mu;
std::vector<CachedBlock> g_pool;
constexpr size_t POOL_MIN_BYTES = (size_t)64 << 20;
size_t pool_limit_bytes() {
    static const size_t lim = [] { const char* e = getenv("FEASTHIP_POOL_GB"); return (size_t)((e ? atof(e) : 96.0) * 1073741824.0); }();
    return lim;
}
bool pool_poison() {
    static const bool on = [] { const char* e = getenv("FEASTHIP_POOL_POISON"); return e && atoi(e) != 0; }();
    return on;
}
void* pool_take(int dev, size_t bytes, size_t& got) {       // smallest cached block of this device with bytes <= size <= 1.25 bytes
    void* q = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        int best = -1;
        for (int i = 0; i < (int)g_pool.size(); ++i)
            if (g_pool[i].dev == dev && g_pool[i].bytes >= bytes && g_pool[i].bytes <= bytes + bytes / 4 && (best < 0 || g_pool[i].bytes < g_pool[best].bytes)) best = i;
        if (best < 0) return nullptr;
        q = g_pool[best].p;
        got = g_pool[best].bytes;
        g_pool.erase(g_pool.begin() + best);
    }
    if (pool_poison()) { (void)hipMemset(q, 0xFF, got); (void)hipDeviceSynchronize(); }
    return q;
}
// `dev` is the device the block was allocated on (DevBuf records it), not whatever device is current in the releasing thread
bool pool_put(int dev, void* q, size_t bytes) {
    if (bytes < POOL_MIN_BYTES || pool_limit_bytes() == 0) return false;
    // hipFree waits for the device before it returns the memory; a block that goes to another handle instead must not have
    // work in flight either
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) return false;
    if (cur != dev && hipSetDevice(dev) != hipSuccess) return false;
    const bool idle = hipDeviceSynchronize() == hipSuccess;
    if (cur != dev) (void)hipSetDevice(cur);
    if (!idle) return false;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    size_t held = 0;
    for (const CachedBlock& b : g_pool) held += b.bytes;
    if (held + bytes > pool_limit_bytes() || g_pool.size() >= 32) return false;
    g_pool.push_back({dev, q, bytes});
    return true;
}
void pool_free_blocks(std::vector<CachedBlock>& blocks) {
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (const CachedBlock& b : blocks) { (void)hipSetDevice(b.dev); (void)hipFree(b.p); }
    (void)hipSetDevice(cur);
}
}  // namespace

static void trim_handle_resources();     // streams / events / pinned slots kept from closed handles (below)

void trim_device_memory() {
    std::vector<CachedBlock> blocks;
    { std::lock_guard<std::mutex> lk(g_pool_mu); blocks.swap(g_pool); }
    pool_free_blocks(blocks);
    trim_handle_resources();
}

// what earlier handles left cached on this device goes back to the driver (called when a handle is destroyed, before its own
// blocks are released into the cache)
void pool_drop_device(int dev) {
    std::vector<CachedBlock> blocks;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (size_t i = 0; i < g_pool.size();)
            if (g_pool[i].dev == dev) { blocks.push_back(g_pool[i]); g_pool.erase(g_pool.begin() + i); } else ++i;
    }
    pool_free_blocks(blocks);
}

template <class T> void DevBuf<T>::release() {
    if (p) { if (!pool_put(dev, (void*)p, cap_bytes)) { int cur = 0; (void)hipGetDevice(&cur); if (cur != dev) (void)hipSetDevice(dev); (void)hipFree((void*)p); if (cur != dev) (void)hipSetDevice(cur); } p = nullptr; n = 0; cap_bytes = 0; }
}
template <class T> bool DevBuf<T>::alloc(size_t count) {
    if (count <= n && p) return true;
    release();
    if (count == 0) count = 1;
    const size_t bytes = count * sizeof(T);
    size_t got = bytes;
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); return false; }
    void* q = bytes >= POOL_MIN_BYTES ? pool_take(cur, bytes, got) : nullptr;
    if (!q) {
        got = bytes;
        if (hipMalloc(&q, bytes) != hipSuccess) {
            (void)hipGetLastError();
            trim_device_memory();          // give the cached blocks back and try once more
            if (hipMalloc(&q, bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
        }
    }
    p = (T*)q;
    n = count;
    cap_bytes = got;
    dev = cur;
    return true;
}
template <class T> bool DevBuf<T>::upload(const T* h, size_t count, hipStream_t st) {
    if (!alloc(count)) return false;
    if (count == 0) return true;
    if (hipMemcpyAsync((void*)p, h, count * sizeof(T), hipMemcpyHostToDevice, st) != hipSuccess) return false;
    return hipStreamSynchronize(st) == hipSuccess;
}
template struct DevBuf<cz>;
template struct DevBuf<cf>;
template struct DevBuf<int>;
template struct DevBuf<int64_t>;
template struct DevBuf<double>;
template struct DevBuf<FrontD>;
template struct DevBuf<Tile>;
template struct DevBuf<const cz*>;


// Streams, events and the pinned landing slots of a handle are kept for the next handle of the same device and configuration
// instead of being destroyed: creating them (six streams whose hardware queues come with their first operation, a pinned
// allocation, two dozen events) costs 10-17 ms and destroying them 10 ms -- nothing beside a 0.1 s factorisation, but the
// reference's plug-in seam creates and finalizes one factor object per contour node and iteration (src/feast.jl:29-36, 61-66
// with store=false), and for a 64 x 64 matrix that was 20 ms of set-up around 0.3 ms of work.
struct HandleResources {
    int device = 0;
    std::string sig;                       // the environment switches that shape the streams
    hipStream_t stream = nullptr, la = nullptr, fin = nullptr, rest = nullptr;
    std::vector<hipStream_t> aux, up;
    std::vector<hipEvent_t> ev_join, ev_chunk, ev_la;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_p0 = nullptr, ev_p1 = nullptr, ev_fork = nullptr, ev_fin = nullptr, ev_fin_fork = nullptr;
    char* pin = nullptr;
    bool warm = false;
};
static std::mutex g_res_mu;
static long g_res_pid = 0;
static std::vector<HandleResources>& res_cache() { static std::vector<HandleResources>* v = new std::vector<HandleResources>(); return *v; }   // (never destroyed: the HIP runtime may be gone by then)
static std::string res_signature() {
    std::string s;
    for (const char* k : {"FEASTHIP_STREAMS", "FEASTHIP_RESERVE_CUS", "FEASTHIP_FIN_MASK"}) { const char* v = getenv(k); s += v ? v : "-"; s += '|'; }
    return s;
}
static const bool g_res_reuse = [] { const char* e = getenv("FEASTHIP_REUSE_STREAMS"); return e ? atoi(e) != 0 : true; }();

static void trim_handle_resources() {
    std::vector<HandleResources> sets;
    { std::lock_guard<std::mutex> lk(g_res_mu); sets.swap(res_cache()); }
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (HandleResources& r : sets) {
        (void)hipSetDevice(r.device);
        for (hipStream_t s2 : r.up) if (s2) (void)hipStreamDestroy(s2);
        for (hipStream_t s2 : r.aux) (void)hipStreamDestroy(s2);
        for (hipStream_t s2 : {r.la, r.rest, r.fin, r.stream}) if (s2) (void)hipStreamDestroy(s2);
        for (auto* v : {&r.ev_join, &r.ev_chunk, &r.ev_la}) for (hipEvent_t e2 : *v) if (e2) (void)hipEventDestroy(e2);
        for (hipEvent_t e2 : {r.ev0, r.ev1, r.ev_p0, r.ev_p1, r.ev_fork, r.ev_fin, r.ev_fin_fork}) if (e2) (void)hipEventDestroy(e2);
        if (r.pin) (void)hipHostFree(r.pin);
    }
    (void)hipSetDevice(cur);
}

bool Engine::take_resources() {
    if (!g_res_reuse) return false;
    HandleResources r;
    {
        std::lock_guard<std::mutex> lk(g_res_mu);
        auto& c = res_cache();
        const std::string& sig = res_sig_;
        size_t i = 0;
        for (; i < c.size(); ++i) if (c[i].device == device_ && c[i].sig == sig) break;
        if (i == c.size()) return false;
        r = std::move(c[i]);
        c.erase(c.begin() + i);
    }
    stream_ = r.stream; own_stream_ = true; la_stream_ = r.la; fin_stream_ = r.fin; rest_stream_ = r.rest;
    aux_streams_ = std::move(r.aux); up_streams_ = std::move(r.up);
    ev_join_ = std::move(r.ev_join); ev_chunk_ = std::move(r.ev_chunk); ev_la_ = std::move(r.ev_la);
    ev0_ = r.ev0; ev1_ = r.ev1; ev_p0_ = r.ev_p0; ev_p1_ = r.ev_p1; ev_fork_ = r.ev_fork; ev_fin_ = r.ev_fin; ev_fin_fork_ = r.ev_fin_fork;
    pin_ = r.pin; streams_warm_ = r.warm;
    return true;
}

bool Engine::give_resources() {
    if (!g_res_reuse || !own_stream_ || !stream_) return false;
    {
        std::lock_guard<std::mutex> lk(g_res_mu);
        if (res_cache().size() >= 64) return false;          // (handles alive at once beyond that are not worth keeping warm)
    }
    HandleResources r;
    r.device = device_; r.sig = res_sig_;          // the configuration the streams were CREATED under (the environment may have changed since)
    r.stream = stream_; r.la = la_stream_; r.fin = fin_stream_; r.rest = rest_stream_;
    r.aux = std::move(aux_streams_); r.up = std::move(up_streams_);
    r.ev_join = std::move(ev_join_); r.ev_chunk = std::move(ev_chunk_); r.ev_la = std::move(ev_la_);
    r.ev0 = ev0_; r.ev1 = ev1_; r.ev_p0 = ev_p0_; r.ev_p1 = ev_p1_; r.ev_fork = ev_fork_; r.ev_fin = ev_fin_; r.ev_fin_fork = ev_fin_fork_;
    r.pin = pin_; r.warm = streams_warm_;
    // the kept sets are destroyed when the process exits normally, while the HIP runtime (loaded before this library, so torn down
    // after its atexit handlers) is still there: tools that finalise at exit (rocprofv3 counter passes) otherwise find live queues
    static std::once_flag at_exit_once;
    std::call_once(at_exit_once, [] { g_res_pid = (long)getpid(); std::atexit([] { if ((long)getpid() == g_res_pid) trim_handle_resources(); }); });      // (not in a forked child: its HIP state is not its own)
    std::lock_guard<std::mutex> lk(g_res_mu);
    res_cache().push_back(std::move(r));
    stream_ = la_stream_ = fin_stream_ = rest_stream_ = nullptr;
    ev0_ = ev1_ = ev_p0_ = ev_p1_ = ev_fork_ = ev_fin_ = ev_fin_fork_ = nullptr;
    pin_ = nullptr;
    return true;
}

Engine::Engine(int device) : device_(device) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) { (void)hipGetLastError(); fail(FEASTHIP_ENOGPU, "no HIP device visible: feasthip has no CPU fallback"); return; }
    if (device < 0 || device >= cnt) { fail(FEASTHIP_EARG, "device id out of range"); return; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { fail(FEASTHIP_EHIP, "hipGetDeviceProperties failed"); return; }
    if (std::strstr(prop.gcnArchName, "gfx950") == nullptr) { fail(FEASTHIP_ENOGPU, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only"); return; }
    if (hipSetDevice(device) != hipSuccess) { fail(FEASTHIP_EHIP, "hipSetDevice failed"); return; }
    res_sig_ = res_signature();
    const bool recycled = take_resources();
    if (!recycled) {
        if (hipStreamCreate(&stream_) != hipSuccess) { fail(FEASTHIP_EHIP, "hipStreamCreate failed"); return; }
        own_stream_ = true;
        (void)hipEventCreate(&ev0_);
        (void)hipEventCreate(&ev1_);
        (void)hipEventCreate(&ev_p0_);
        (void)hipEventCreate(&ev_p1_);
        { void* q = nullptr; if (hipHostMalloc(&q, PIN_SLOTS * PIN_SLOT_BYTES, hipHostMallocDefault) == hipSuccess) pin_ = (char*)q; else (void)hipGetLastError(); }
    }
    if (const char* v = getenv("FEASTHIP_LOOKAHEAD")) look_ahead_ = atoi(v) != 0;
    if (const char* v = getenv("FEASTHIP_GROUP")) group_panels_ = std::max(1, std::min(8, atoi(v)));
    if (const char* v = getenv("FEASTHIP_FUSED_UTRSM")) fused_utrsm_ = atoi(v) != 0;
    if (const char* v = getenv("FEASTHIP_FUSE_EXTEND")) { fuse_extend_ = atoi(v) != 0; if (atoi(v) > 0) fuse_min_p_ = atoi(v); }
    if (const char* v = getenv("FEASTHIP_FIN_ASYNC")) finalize_async_ = atoi(v) != 0;
    if (const char* v = getenv("FEASTHIP_MIXED")) mixed_ = atoi(v) != 0;      // tuning / test knob; the API is feasthip_set_precision
    if (const char* v = getenv("FEASTHIP_NB256_MINP")) nb256_min_p_ = std::max(129, atoi(v));      // tuning knobs: 1 << 30 keeps the 128-column blocks
    if (const char* v = getenv("FEASTHIP_FUSE_B3")) fuse_b3_ = atoi(v) != 0;
    if (const char* v = getenv("FEASTHIP_NB512_MINP")) nb512_min_p_ = std::max(257, atoi(v));
    if (const char* v = getenv("FEASTHIP_NB1024_MINP")) nb1024_min_p_ = std::max(513, atoi(v));
    if (const char* v = getenv("FEASTHIP_XCD_BALANCE")) xcd_balance_ = atoi(v) != 0;
    if (const char* v = getenv("FEASTHIP_SHORTK_MAXP")) short_k_max_p_ = atoi(v);
    if (const char* v = getenv("FEASTHIP_SLAB64_MIN")) slab64_min_wgs_ = atoi(v);
    if (const char* v = getenv("FEASTHIP_FWD_SPLITK")) fwd_splitk_ = atoi(v) != 0;
    if (const char* v = getenv("FEASTHIP_NOPIV")) nopiv_enabled_ = atoi(v) != 0;
    if (const char* v = getenv("FEASTHIP_NOPIV_GROWTH")) nopiv_max_growth_ = atof(v);
    if (const char* v = getenv("FEASTHIP_NOPIV_GROWTH32")) nopiv_max_growth32_ = atof(v);
    if (const char* v = getenv("FEASTHIP_NOPIV_GEN")) nopiv_gen_enabled_ = atoi(v) != 0;
    if (const char* v = getenv("FEASTHIP_ADJ_FOLD")) adj_fold_enabled_ = atoi(v) != 0;
    if (const char* v = getenv("FEASTHIP_RR_DEVICE")) rr_device_ = atoi(v) != 0;
    if (const char* v = getenv("FEASTHIP_RR_COND")) rr_cond_max_ = atof(v);
    if (const char* v = getenv("FEASTHIP_FOLD")) fold_enabled_ = atoi(v) != 0;          // initial state of feasthip_set_conjugate_fold
    if (const char* v = getenv("FEASTHIP_EARLY_SWEEP")) early_sweep_enabled_ = atoi(v) != 0;
    int nst = 2;
    if (const char* v = getenv("FEASTHIP_STREAMS")) { nst = std::max(1, std::min(8, atoi(v))); streams_forced_ = true; }
    if (const char* v = getenv("FEASTHIP_RESERVE_CUS")) reserve_cus_ = std::max(0, std::min(16, atoi(v)));
    if (recycled) return;
    if (nst > 1) {
        (void)hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming);
        for (int i = 0; i < nst; ++i) {
            hipStream_t s2 = nullptr; hipEvent_t e2 = nullptr;
            if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess) break;
            (void)hipEventCreateWithFlags(&e2, hipEventDisableTiming);
            aux_streams_.push_back(s2); ev_join_.push_back(e2);
        }
    }
    {   // look-ahead stream of the factorisation: its small panel kernels must be dispatched ahead of the trailing updates
        int plo = 0, phi = 0;
        (void)hipDeviceGetStreamPriorityRange(&plo, &phi);
        if (hipStreamCreateWithPriority(&la_stream_, hipStreamNonBlocking, phi) != hipSuccess) la_stream_ = nullptr;
        // the solve-ready conversion of a finished level fills the gaps of the levels above: lowest priority
        if (hipStreamCreateWithPriority(&fin_stream_, hipStreamNonBlocking, plo) != hipSuccess) fin_stream_ = nullptr;
        (void)hipEventCreateWithFlags(&ev_fin_, hipEventDisableTiming);
        (void)hipEventCreateWithFlags(&ev_fin_fork_, hipEventDisableTiming);
        for (int i = 0; i < 7; ++i) { hipEvent_t e3 = nullptr; (void)hipEventCreateWithFlags(&e3, hipEventDisableTiming); ev_la_.push_back(e3); }
        // A panel workgroup (1024 threads, 126 registers) needs an empty CU: behind a trailing update that fills every CU it
        // would only start when that update drains.  The trailing updates of the look-ahead levels therefore run on a stream
        // whose CU mask leaves reserve_cus_ CUs of every XCD alone (MI355X: 8 XCDs x 32 CUs).  The cleared bits are 32x + 8m + x
        // (x = XCD, m < reserve_cus_): two free CUs per XCD whether the mask bits enumerate the CUs XCD by XCD or round-robin.
        hipDeviceProp_t prop;
        if (la_stream_ && reserve_cus_ > 0 && hipGetDeviceProperties(&prop, device_) == hipSuccess && prop.multiProcessorCount == 256) {
            uint32_t mask[8];
            for (int x = 0; x < 8; ++x) {
                mask[x] = 0xffffffffu;
                for (int m = 0; m < reserve_cus_; ++m) mask[x] &= ~(1u << ((x + 8 * (m % 4) + 2 * (m / 4)) % 32));      // (distinct bits for up to 16 per XCD; m < 4: 8 m + x)
            }
            if (hipExtStreamCreateWithCUMask(&rest_stream_, 8, mask) != hipSuccess) { rest_stream_ = nullptr; (void)hipGetLastError(); }
            // The solve-ready conversion of the finished levels (long-running workgroups on a low-priority stream) must not sit on the
            // reserved CUs either: a chain kernel's workgroup would wait for one of those to drain.  FEASTHIP_FIN_MASK=1 trades the
            // stream's low priority for the same CU mask.
            if (const char* v = getenv("FEASTHIP_FIN_MASK")) if (atoi(v) != 0) {
                hipStream_t fm = nullptr;
                if (hipExtStreamCreateWithCUMask(&fm, 8, mask) == hipSuccess) { if (fin_stream_) (void)hipStreamDestroy(fin_stream_); fin_stream_ = fm; } else (void)hipGetLastError();
            }
        }
    }
}

Engine::~Engine() {
    (void)hipSetDevice(device_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    (void)comm_destroy();
    if (comm_stream_) { (void)hipStreamSynchronize(comm_stream_); (void)hipStreamDestroy(comm_stream_); comm_stream_ = nullptr; }
    if (ev_comm0_) { (void)hipEventDestroy(ev_comm0_); ev_comm0_ = nullptr; }
    if (ev_comm1_) { (void)hipEventDestroy(ev_comm1_); ev_comm1_ = nullptr; }
    pool_drop_device(device_);       // the cache holds at most what ONE handle used: earlier leftovers go back to the driver now
    for (hipStream_t s2 : up_streams_) if (s2) (void)hipStreamSynchronize(s2);
    for (hipStream_t s2 : aux_streams_) (void)hipStreamSynchronize(s2);
    for (hipStream_t s2 : {la_stream_, rest_stream_, fin_stream_}) if (s2) (void)hipStreamSynchronize(s2);
    if (give_resources()) return;
    for (hipStream_t s2 : up_streams_) if (s2) { (void)hipStreamSynchronize(s2); (void)hipStreamDestroy(s2); }
    for (hipStream_t s2 : aux_streams_) { (void)hipStreamSynchronize(s2); (void)hipStreamDestroy(s2); }
    for (hipEvent_t e2 : ev_join_) (void)hipEventDestroy(e2);
    for (hipEvent_t e2 : ev_chunk_) (void)hipEventDestroy(e2);
    if (ev_fork_) (void)hipEventDestroy(ev_fork_);
    for (hipEvent_t e3 : ev_la_) if (e3) (void)hipEventDestroy(e3);
    if (la_stream_) { (void)hipStreamSynchronize(la_stream_); (void)hipStreamDestroy(la_stream_); }
    if (rest_stream_) { (void)hipStreamSynchronize(rest_stream_); (void)hipStreamDestroy(rest_stream_); }
    if (fin_stream_) { (void)hipStreamSynchronize(fin_stream_); (void)hipStreamDestroy(fin_stream_); }
    if (ev_fin_) (void)hipEventDestroy(ev_fin_);
    if (ev_fin_fork_) (void)hipEventDestroy(ev_fin_fork_);
    if (ev0_) (void)hipEventDestroy(ev0_);
    if (ev1_) (void)hipEventDestroy(ev1_);
    if (ev_p0_) (void)hipEventDestroy(ev_p0_);
    if (ev_p1_) (void)hipEventDestroy(ev_p1_);
    if (own_stream_ && stream_) (void)hipStreamDestroy(stream_);
    if (pin_) (void)hipHostFree(pin_);
}

// device -> host through a pinned slot: the bytes land in *landed (the slot, or host_dst itself when no slot fits);
// after the stream synchronisation the caller copies them out if landed != host_dst
hipError_t Engine::d2h_async(int slot, void* host_dst, const void* dev, size_t bytes, void** landed) {
    void* q = pin_slot(slot, bytes);
    *landed = q ? q : host_dst;
    return hipMemcpyAsync(*landed, dev, bytes, hipMemcpyDeviceToHost, stream_);
}

// host -> device through a pinned slot (the slot must not be rewritten before the stream has passed the copy: every user
// synchronises the stream before its next use of the same slot)
hipError_t Engine::h2d_async(int slot, void* dev, const void* host_src, size_t bytes) {
    void* q = pin_slot(slot, bytes);
    if (q) std::memcpy(q, host_src, bytes);
    return hipMemcpyAsync(dev, q ? q : host_src, bytes, hipMemcpyHostToDevice, stream_);
}

int Engine::hip_fail(hipError_t e, const char* what) {
    (void)hipGetLastError();
    return fail(FEASTHIP_EHIP, std::string("HIP error '") + hipGetErrorString(e) + "' in " + what);
}

int Engine::set_stream(void* s) {
    if (own_stream_ && stream_) { (void)hipStreamSynchronize(stream_); (void)hipStreamDestroy(stream_); }
    stream_ = (hipStream_t)s;
    own_stream_ = false;
    return 0;
}

static zc read_val(const void* p, int type, int64_t idx) {
    switch (type) {
        case FEASTHIP_F64: return zc(((const double*)p)[idx], 0.0);
        case FEASTHIP_C128: return zc(((const double*)p)[2 * idx], ((const double*)p)[2 * idx + 1]);
        default: return zc((double)((const int64_t*)p)[idx], 0.0);
    }
}

int Engine::set_dense(int64_t n_, int nterms_, const void* const* mats, const int* types, const double* scale) {
    if (n_ <= 0 || n_ > 65536 || nterms_ < 1 || nterms_ > 16 || !mats) return fail(FEASTHIP_EARG, "set_dense: bad n or nterms");
    HIPCHK(hipSetDevice(device_));
    n = (int)n_; nterms = nterms_; dense = true;
    analyzed_ = factored_ = have_x_ = false;
    terms_.clear();
    terms_.resize(nterms);
    terms_real_ = false;
    bool real = true, sym = true, herm = true, hdiag = true;
    // The caller's bytes go to the device as they are; element type, storage order (FEASTHIP_ROW_MAJOR), max |a_ij|, realness and
    // M == M^T (exactly) are dealt with there (k_dense_import): the host touches the matrix once, for the copy.
    DevBuf<int64_t> d_props;
    DevBuf<cz> d_stage;
    if (!d_props.alloc(5)) return fail(FEASTHIP_EHIP, "set_dense: device allocation failed");
    for (int t = 0; t < nterms; ++t) {
        SparseTerm& T = terms_[t];
        T.scale = scale ? zc(scale[2 * t], scale[2 * t + 1]) : zc(1.0);
        if (!mats[t]) { T.identity = true; T.maxabs = 1.0; continue; }
        const int tyf = types ? types[t] : FEASTHIP_C128;
        const int ty = tyf & 0xff, row_major = (tyf & FEASTHIP_ROW_MAJOR) ? 1 : 0;
        if (ty < 0 || ty > 3 || (tyf & ~(0xff | FEASTHIP_ROW_MAJOR))) return fail(FEASTHIP_EARG, "set_dense: bad element type");
        const size_t nn = (size_t)n * n;
        const size_t esz = ty == FEASTHIP_C128 ? 16 : 8;
        if (!T.d_dense.alloc(nn)) return fail(FEASTHIP_EHIP, "set_dense: device allocation failed");
        const bool in_place = ty == FEASTHIP_C128 && !row_major;
        void* raw = T.d_dense.p;
        if (!in_place) {
            if (!d_stage.alloc((nn * esz + 15) / 16)) return fail(FEASTHIP_EHIP, "set_dense: device allocation failed");
            raw = d_stage.p;
        }
        HIPCHK(hipMemsetAsync(d_props.p, 0, 5 * sizeof(int64_t), stream_));
        HIPCHK(hipMemcpyAsync(raw, mats[t], nn * esz, hipMemcpyHostToDevice, stream_));
        launch_dense_import(stream_, raw, ty, row_major, n, T.d_dense.p, reinterpret_cast<unsigned long long*>(d_props.p));
        int64_t props[5] = {0, 0, 0, 0, 0};
        HIPCHK(hipMemcpyAsync(props, d_props.p, sizeof(props), hipMemcpyDeviceToHost, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        HIPCHK(hipGetLastError());
        double mx;
        std::memcpy(&mx, &props[0], sizeof(mx));
        T.maxabs = mx;
        if (props[1]) real = false;
        if (props[2]) sym = false;
        if (props[3]) herm = false;
        if (props[4]) hdiag = false;
    }
    for (auto& T : terms_) if (T.scale.imag() != 0.0) real = false;
    terms_real_ = real;
    terms_symmetric_ = sym;
    terms_hermitian_ = herm;
    terms_herm_exact_ = herm && hdiag;
    nopiv_failed_ = false;
    rr_dev_state_ = 0;
    nopiv_gen_failed_ = false;
    rebuild_shard();
    return 0;
}

int Engine::set_csc(int64_t n_, int nterms_, const int64_t* const* colptr, const int64_t* const* rowval,
                    const void* const* nzval, const int* types, const double* scale, int base) {
    if (n_ <= 0 || n_ > 0x7fffffff / 2 || nterms_ < 1 || nterms_ > 16 || !colptr || !rowval || !nzval)
        return fail(FEASTHIP_EARG, "set_csc: bad n or nterms");
    HIPCHK(hipSetDevice(device_));
    n = (int)n_; nterms = nterms_; dense = false;
    analyzed_ = factored_ = have_x_ = false;
    terms_.clear();
    terms_.resize(nterms);
    terms_real_ = false;
    bool real = true;
    for (int t = 0; t < nterms; ++t) {
        SparseTerm& T = terms_[t];
        T.scale = scale ? zc(scale[2 * t], scale[2 * t + 1]) : zc(1.0);
        if (T.scale.imag() != 0.0) real = false;
        T.colptr.resize(n + 1);
        if (!colptr[t]) {   // identity, stored as an explicit diagonal
            T.identity = true;
            T.rowval.resize(n);
            T.val.assign(n, zc(1.0));
            for (int j = 0; j <= n; ++j) T.colptr[j] = j;
            for (int j = 0; j < n; ++j) T.rowval[j] = j;
            T.maxabs = 1.0;
            continue;
        }
        int ty = types ? types[t] : FEASTHIP_C128;
        if (ty < 0 || ty > 2) return fail(FEASTHIP_EARG, "set_csc: bad element type");
        for (int j = 0; j <= n; ++j) T.colptr[j] = colptr[t][j] - base;
        int64_t nnz = T.colptr[n];
        if (T.colptr[0] != 0 || nnz < 0) return fail(FEASTHIP_EARG, "set_csc: colptr[0] != index_base");
        T.rowval.resize(nnz);
        T.val.resize(nnz);
        double mx = 0;
        for (int j = 0; j < n; ++j) {
            if (T.colptr[j + 1] < T.colptr[j]) return fail(FEASTHIP_EARG, "set_csc: colptr not monotone");
            for (int64_t e = T.colptr[j]; e < T.colptr[j + 1]; ++e) {
                int64_t r = rowval[t][e] - base;
                if (r < 0 || r >= n) return fail(FEASTHIP_EDIM_A, "Incorrect dimensions of A, must be square");
                T.rowval[e] = (int)r;
                const zc v = read_val(nzval[t], ty, e);
                T.val[e] = v;
                if (v.imag() != 0.0) real = false;
                // max |a_ij|: the modulus (a hypot call, 20 ns) only for entries that can raise the maximum
                const double m1 = std::max(std::fabs(v.real()), std::fabs(v.imag()));
                if (!(m1 * 1.4142135623730951 <= mx)) mx = std::max(mx, std::abs(v));
            }
        }
        T.maxabs = mx;
    }
    terms_real_ = real;
    terms_symmetric_ = false;          // decided in analyze(), where the transposes of the terms are built anyway
    terms_hermitian_ = false;
    terms_herm_exact_ = false;
    nopiv_failed_ = false;
    rr_dev_state_ = 0;
    nopiv_gen_failed_ = false;
    rebuild_shard();
    return 0;
}

int Engine::set_q_buffer(void* dptr, size_t bytes) {
    q_ext_ = (cz*)dptr;
    q_ext_bytes_ = dptr ? bytes : 0;
    have_x_ = false;
    return 0;
}

int Engine::set_contour(int K_, const double* nodes, const double* weights, int kind, const double* region) {
    if (K_ <= 0 || !nodes || !weights) return fail(FEASTHIP_EARG, "set_contour: bad arguments");
    K = K_;
    nodes_.resize(K); weights_.resize(K);
    for (int k = 0; k < K; ++k) { nodes_[k] = zc(nodes[2 * k], nodes[2 * k + 1]); weights_[k] = zc(weights[2 * k], weights[2 * k + 1]); }
    region_kind_ = kind;
    if (region) std::memcpy(region_, region, sizeof(region_));
    factored_ = false;
    return set_shard(shard_first_, shard_stride_);
}

int Engine::set_shard(int first, int stride) {
    if (first < 0 || stride < 1) return fail(FEASTHIP_EARG, "set_node_shard: bad shard");
    shard_first_ = first; shard_stride_ = stride;
    rebuild_shard();
    factored_ = false;
    return 0;
}

int Engine::set_fold(int on) {
    fold_enabled_ = on != 0;
    rebuild_shard();
    return 0;
}

// Factors that will serve one sweep (the reference's store = false: every iteration factorises again) skip the scaling of their update
// rows; see engine.hpp.  Takes effect at the next factorisation; the current factors stay what they are (lazy_factors_).
int Engine::set_lazy_conversion(int on) {
    lazy_conversion_ = on != 0;
    return 0;
}

bool plan_node_shard(const std::vector<zc>& nodes, int first, int stride, bool fold_allowed, std::vector<int>& loc, std::vector<int>& par) {
    const int K = (int)nodes.size();
    loc.clear(); par.clear();
    bool can = fold_allowed && K >= 2 && K % 2 == 0;
    std::vector<int> mate(K, -1);
    // mirror images agree to the rounding of the contour constructors (z = c + r e^{i theta}: a few ulps of the contour's size)
    double zmax = 0;
    for (const zc& z : nodes) zmax = std::max(zmax, std::abs(z));
    const double tol = 16.0 * std::numeric_limits<double>::epsilon() * zmax;
    for (int k = 0; k < K && can; ++k) {
        if (mate[k] >= 0) continue;
        const zc zk = nodes[k];
        if (std::abs(zk.imag()) <= tol) { can = false; break; }        // a node on the real axis is its own mirror image
        for (int k2 = k + 1; k2 < K; ++k2)
            if (mate[k2] < 0 && std::abs(nodes[k2] - std::conj(zk)) <= tol) { mate[k] = k2; mate[k2] = k; break; }
        if (mate[k] < 0) can = false;
    }
    if (can && K / 2 < stride) can = false;          // fewer pairs than ranks: deal out single nodes instead (same decision on every rank)
    if (can) {
        int u = 0;
        for (int k = 0; k < K; ++k) {
            if (mate[k] < k) continue;                      // second member of its pair
            if (u >= first && (u - first) % stride == 0) {
                const int rep = nodes[k].imag() > 0 ? k : mate[k];
                loc.push_back(rep); par.push_back(rep == k ? mate[k] : k);
            }
            ++u;
        }
    } else {
        for (int k = first; k < K; k += stride) { loc.push_back(k); par.push_back(-1); }
    }
    return can;
}

// local_ / partner_ / fold_ / Kloc from the contour, the shard, the terms and the per-node overrides (plan_node_shard, engine.hpp)
void Engine::rebuild_shard() {
    std::vector<int> loc, par;
    const bool base = fold_enabled_ && !terms_.empty() && node_coef_.empty() && coef_fn_ == nullptr;
    bool scales_real = true;
    for (const SparseTerm& T : terms_) if (T.scale.imag() != 0.0) scales_real = false;
    const bool allow_conj = base && terms_real_;
    // the adjoint fold: Hermitian coefficient matrices that are not real (real ones take the conjugate fold: one sweep of twice the width
    // instead of two); complex128 sweeps only
    const bool allow_adj = base && !terms_real_ && adj_fold_enabled_ && terms_herm_exact_ && scales_real && !mixed_ && !fwd_splitk_;   // (the adjoint sweeps have no forward split-K form)
    const bool can = plan_node_shard(nodes_, shard_first_, shard_stride_, allow_conj || allow_adj, loc, par);
    const bool nf = can && allow_conj, na = can && !allow_conj && allow_adj;
    if (loc.size() != local_.size() || nf != fold_ || na != adj_fold_) { analyzed_ = false; have_x_ = false; }
    if (loc != local_ || par != partner_ || nf != fold_ || na != adj_fold_) factored_ = false;
    local_.swap(loc); partner_.swap(par); fold_ = nf; adj_fold_ = na;
    Kloc = (int)local_.size();
}

// ---------------------------------------------------------------------------
bool Engine::build_plans() {
    // Every tree level builds its tiles, lists and XCD permutations into arrays of its own (levels are independent: host threads take
    // them, deepest -- most tiles -- first); the arrays are concatenated afterwards and the levels' offsets shifted accordingly.
    struct LevelOut { std::vector<Tile> tiles; std::vector<int> lists, xperm; int64_t sk_rows = 0; bool ok = true; };
    auto cdiv = [](int a, int b) { return (a + b - 1) / b; };
    int nd = S.maxdepth + 1;
    fplan_.assign(nd, FactorLevel());
    splan_.assign(nd, SolveLevel());
    sk_off_.assign(S.nsuper, -1);
    skf_off_.assign(S.nsuper, -1);
    sk_rows_ = 0;
    std::vector<LevelOut> outs(nd);
    std::vector<int> nchild(S.nsuper);
    for (int s = 0; s < S.nsuper; ++s) nchild[s] = (int)(S.child_ptr[s + 1] - S.child_ptr[s]);
    std::vector<char> fused_b3(S.nsuper, 0);
    const double xcd_c0 = getenv("FEASTHIP_XCD_C0") ? atof(getenv("FEASTHIP_XCD_C0")) : 64.0, xcd_thr = getenv("FEASTHIP_XCD_THR") ? atof(getenv("FEASTHIP_XCD_THR")) : 64.0;
    auto build_level = [&](int d) {
        LevelOut& O = outs[d];
        std::vector<Tile>& tiles = O.tiles;
        std::vector<int>& lists = O.lists;
        std::vector<int>& xperm = O.xperm;
        // product tiles of one front in 8 x 8 super-tiles (64 consecutive tiles share 8 row panels and 8 column panels; k_gemm_nn
        // hands such runs to one XCD)
        auto push_supertiles = [&](int s, int kind, int na, int nc) {
            for (int A = 0; A < na; A += 8)
                for (int C = 0; C < nc; C += 8)
                    for (int a = A; a < std::min(na, A + 8); ++a)
                        for (int c = C; c < std::min(nc, C + 8); ++c) tiles.push_back({s, kind, a, c});
        };
        const std::vector<int>& lv = S.levels[d];
        FactorLevel& F = fplan_[d];
        SolveLevel& V = splan_[d];
        int pmax = 0;
        for (int s : lv) pmax = std::max(pmax, S.p(s));
        // extend-add / solve assembly
        F.extend.off = (int64_t)tiles.size();
        // phase 0 touches the L / U12 panels (tiles with a row or a column inside the pivot block), phase 1 the contribution block
        for (int s : lv) if (nchild[s] > 0) { int nt = cdiv(S.p(s) + S.u(s), 64), p = S.p(s); for (int a = 0; a < nt; ++a) for (int b = 0; b < nt; ++b) if (a * 64 < p || b * 64 < p) tiles.push_back({s, a, b, 0}); }
        F.extend.count = (int)(tiles.size() - F.extend.off);
        F.extend1.off = (int64_t)tiles.size();
        for (int s : lv) if (nchild[s] > 0 && S.u(s) > 0) { int nt = cdiv(S.p(s) + S.u(s), 64), p = S.p(s); for (int a = 0; a < nt; ++a) for (int b = 0; b < nt; ++b) if (a * 64 + 64 > p && b * 64 + 64 > p) tiles.push_back({s, a, b, 0}); }
        F.extend1.count = (int)(tiles.size() - F.extend1.off);
        V.assemble.off = (int64_t)tiles.size();
        for (int s : lv) if (nchild[s] > 0) { int nt = cdiv(S.p(s), 64); for (int a = 0; a < nt; ++a) tiles.push_back({s, a, 0, 0}); }      // pivot-block rows only
        V.assemble.count = (int)(tiles.size() - V.assemble.off);
        for (int s : lv)
            for (int64_t c = S.child_ptr[s]; c < S.child_ptr[s + 1]; ++c) {
                int gch = S.child_idx[c];
                for (int i = 0; i < S.u(gch); ++i) { if (S.rel[S.struct_ptr[gch] + i] < S.p(s)) V.asm_rows += 1; else V.pull_rows += 1; }
            }
        // factor steps
        for (int k0 = 0; k0 < pmax; k0 += FACT_NB) {
            FactorStep st;
            st.k0 = k0;
            st.panel.off = (int64_t)lists.size();
            for (int s : lv) if (S.p(s) > k0) lists.push_back(s);
            st.panel.count = (int)(lists.size() - st.panel.off);
            st.swap.off = (int64_t)tiles.size();
            for (int s : lv) if (S.p(s) > k0) { int p = S.p(s), u = S.u(s), b = std::min(FACT_NB, p - k0); int nt = cdiv(p - b + u, 256); for (int a = 0; a < nt; ++a) tiles.push_back({s, a, 0, 0}); }
            st.swap.count = (int)(tiles.size() - st.swap.off);
            st.trsm.off = (int64_t)tiles.size();
            for (int s : lv) if (S.p(s) > k0) { int p = S.p(s), u = S.u(s), b = std::min(FACT_NB, p - k0); int nc = (p - k0 - b) + u; int nt = cdiv(nc, TRSM_COLS); for (int a = 0; a < nt; ++a) tiles.push_back({s, a, 0, 0}); }
            st.trsm.count = (int)(tiles.size() - st.trsm.off);
            st.gemm.off = (int64_t)tiles.size();
            for (int s : lv) if (S.p(s) > k0) {
                int p = S.p(s), u = S.u(s), b = std::min(FACT_NB, p - k0), ld = p + u;
                int M0 = ld - k0 - b, N0 = p - k0 - b;
                if (M0 > 0 && N0 > 0) push_supertiles(s, 0, cdiv(M0, 64), cdiv(N0, 64));
                int M1 = p - k0 - b, N1 = u;
                if (M1 > 0 && N1 > 0) push_supertiles(s, 1, cdiv(M1, 64), cdiv(N1, 64));
            }
            st.gemm.count = (int)(tiles.size() - st.gemm.off);
            st.gemm0.off = (int64_t)tiles.size();        // first row of product tiles only: the 32-row products of the grouped U-row solves
            for (int s : lv) if (S.p(s) > k0) {
                int p = S.p(s), u = S.u(s), b = std::min(FACT_NB, p - k0);
                int N0 = p - k0 - b;
                for (int c = 0; c < cdiv(N0, 64); ++c) tiles.push_back({s, 0, 0, c});
                if (N0 > 0) for (int c = 0; c < cdiv(u, 64); ++c) tiles.push_back({s, 1, 0, c});
            }
            st.gemm0.count = (int)(tiles.size() - st.gemm0.off);
            st.gemm_in.off = (int64_t)tiles.size();      // product tiles of the first two tile columns: updates inside a panel group and of the next group's columns
            for (int s : lv) if (S.p(s) > k0) {
                int p = S.p(s), u = S.u(s), b = std::min(FACT_NB, p - k0), ld = p + u;
                int M0 = ld - k0 - b, N0 = p - k0 - b;
                if (M0 > 0 && N0 > 0) for (int a = 0; a < cdiv(M0, 64); ++a) for (int c = 0; c < std::min(2, cdiv(N0, 64)); ++c) tiles.push_back({s, 0, a, c});
            }
            st.gemm_in.count = (int)(tiles.size() - st.gemm_in.off);
            st.grp.off = (int64_t)tiles.size();          // 32-column tiles right of the panel group that starts here (k_group_utrsm)
            if ((int)F.steps.size() % group_panels_ == 0 && group_panels_ * FACT_NB <= 128)
                for (int s : lv) if (S.p(s) > k0) {
                    int p = S.p(s), u = S.u(s), b = std::min(group_panels_ * FACT_NB, p - k0);
                    for (int a = 0; a < cdiv(p - k0 - b + u, 32); ++a) tiles.push_back({s, a, 0, 0});
                }
            st.grp.count = (int)(tiles.size() - st.grp.off);
            st.lsg.off = (int64_t)tiles.size();
            if ((int)F.steps.size() % 4 == 0)
                for (int s : lv) if (S.p(s) > k0) { int p = S.p(s), u = S.u(s), gw = std::min(128, p - k0); for (int a = 0; a < cdiv(p + u - k0 - gw, 64); ++a) tiles.push_back({s, a, 0, 0}); }
            st.lsg.count = (int)(tiles.size() - st.lsg.off);
            F.steps.push_back(st);
        }
        F.cbgemm.off = (int64_t)tiles.size();
        for (int s : lv) if (S.u(s) > 0) { int nt = cdiv(S.u(s), 64); push_supertiles(s, 2, nt, nt); }
        F.cbgemm.count = (int)(tiles.size() - F.cbgemm.off);
        F.cbgemm_sym.off = (int64_t)tiles.size();
        for (int s : lv) if (S.u(s) > 0) {      // lower triangle of tiles, still in 8 x 8 super-tile runs for the XCD-local operand reuse
            int nt = cdiv(S.u(s), 64);
            for (int A = 0; A < nt; A += 8) for (int C = 0; C <= A; C += 8)
                for (int a = A; a < std::min(nt, A + 8); ++a) for (int c = C; c < std::min(nt, C + 8); ++c) if (c <= a) tiles.push_back({s, 2, a, c});
        }
        F.cbgemm_sym.count = (int)(tiles.size() - F.cbgemm_sym.off);
        F.fronts.off = (int64_t)lists.size();
        for (int s : lv) lists.push_back(s);
        F.fronts.count = (int)lv.size();
        // solve-ready finalisation (block inverses, coupling into super blocks, panel scaling)
        F.binv.off = (int64_t)tiles.size();
        for (int s : lv) for (int t = 0; t < cdiv(S.p(s), INV_NB); ++t) tiles.push_back({s, t, 0, 0});
        F.binv.count = (int)(tiles.size() - F.binv.off);
        F.couple.off = (int64_t)tiles.size();
        for (int s : lv) for (int t = 0; t < cdiv(S.p(s), SOLVE_NB); ++t) if (S.p(s) - t * SOLVE_NB > INV_NB) tiles.push_back({s, t, 0, 0});
        F.couple.count = (int)(tiles.size() - F.couple.off);
        F.scale.off = (int64_t)tiles.size();
        for (int s : lv) for (int t = 0; t < cdiv(S.p(s), SOLVE_NB); ++t) {
            int p = S.p(s), ld = p + S.u(s), k0 = t * SOLVE_NB, b = std::min(SOLVE_NB, p - k0);
            for (int a = 0; a < cdiv(k0, 64 * SCALE_ROWTILES); ++a) tiles.push_back({s, t, a, 0});
            for (int a = 0; a < cdiv(ld - k0 - b, 64 * SCALE_ROWTILES); ++a) tiles.push_back({s, t, a, 1});
        }
        F.scale.count = (int)(tiles.size() - F.scale.off);
        // coupling into 256- and 512-column super blocks (kernels.hpp: launch_couple_outer / launch_couple_level)
        for (int lvl = 0; lvl < 3; ++lvl) {
            const int half = SOLVE_NB << lvl;
            F.outer[lvl].off = (int64_t)tiles.size();
            for (int s : lv) if (solve_nb(s) >= 2 * half) {
                int p = S.p(s), ld = p + S.u(s);
                for (int pr = 0; pr * 2 * half + half < p; ++pr) {
                    int kp = pr * 2 * half, bb = std::min(half, p - kp - half);
                    int mb = ld - (kp + half + bb);
                    if (mb > 0) push_supertiles(s, 3 | (pr << 8), cdiv(mb, 64), cdiv(half, 64));
                    if (kp > 0) push_supertiles(s, 4 | (pr << 8), cdiv(kp, 64), cdiv(bb, 64));
                }
            }
            F.outer[lvl].count = (int)(tiles.size() - F.outer[lvl].off);
            F.inner[lvl].off = (int64_t)tiles.size();
            for (int s : lv) if (solve_nb(s) >= 2 * half) {
                int p = S.p(s);
                for (int pr = 0; pr * 2 * half + half < p; ++pr) {
                    int bb = std::min(half, p - pr * 2 * half - half);
                    for (int c = 0; c < cdiv(half, 16); ++c) tiles.push_back({s, pr, 0, c});       // Pc: bb x half
                    for (int c = 0; c < cdiv(bb, 16); ++c) tiles.push_back({s, pr, 1, c});         // Qd: half x bb
                }
            }
            F.inner[lvl].count = (int)(tiles.size() - F.inner[lvl].off);
        }
        // block substitution launches (kernels.hpp: launch_apply).  Tile height (64/32/16 rows) is chosen per
        // launch so that even the top of the tree (few fronts, long K) spreads over the whole chip.
        auto pick_slabs = [&](auto&& rows_of) {     // rows_of(callback(rows)) enumerates the row extents of the launch
            int64_t wgs = 0;
            rows_of([&](int rows) { wgs += cdiv(rows, 64); });
            if (wgs * nodes_local() < 128) return 1;      // latency-bound launch (few nodes left on this GPU): 16-row tiles, four waves share the columns
            if (pmax <= short_k_max_p_) return 2;         // levels of small fronts (short K loops): 32-row tiles run the k_apply2 instance with four (gathered: three) workgroups per CU
            return (wgs * nodes_local() >= slab64_min_wgs_) ? 4 : 2;
        };
        double rowb = 1.0;   // bytes are scaled by 16*ldw at run time: count block-vector ROWS here
        (void)rowb;
        int nsteps = 0;
        for (int s : lv) nsteps = std::max(nsteps, cdiv(S.p(s), solve_nb(s)));
        for (int c = 0; c + 1 < nsteps; ++c) {
            Launch L;
            L.off = (int64_t)tiles.size();
            int sl = pick_slabs([&](auto&& cb) { for (int s : lv) { int NB = solve_nb(s); if (S.p(s) > c * NB + NB) cb(S.p(s) - c * NB - NB); } });
            for (int s : lv) {
                int NB = solve_nb(s), p = S.p(s), k0 = c * NB;
                if (p <= k0 + NB) continue;
                int rows = p - k0 - NB;
                for (int a = 0; a < cdiv(rows, 16 * sl); ++a) tiles.push_back({s, 0 | (sl << 8), a, c});
                L.bytes += (double)rows * NB * 16.0; L.flops += (double)rows * NB; L.vrows += 2.0 * rows + NB;
            }
            L.count = (int)(tiles.size() - L.off);
            L.sl = sl;
            V.f1.push_back(L);
        }
        V.f2.off = (int64_t)tiles.size();
        int sl2 = pick_slabs([&](auto&& cb) { for (int s : lv) if (S.u(s) > 0) cb(S.u(s)); });
        {   // few, long tiles (top of the tree, few nodes on this GPU): split the K = p loop of the L21 products over several workgroups
            int64_t wgs = 0;
            for (int s : lv) if (S.u(s) > 0) wgs += cdiv(S.u(s), 64);
            const int keff = std::max(1, nodes_local() / stream_groups());     // nodes per launch
            int ns = 0;
            if (fwd_splitk_ && wgs > 0 && wgs * keff < 1024 && pmax >= 256) ns = (int)std::min<int64_t>(8, std::min<int64_t>(pmax / 128, cdiv(2048, (int)(wgs * keff))));
            V.f2split = ns >= 2 ? ns : 0;
            if (V.f2split) sl2 = 4;
        }
        int64_t skf_level = 0;
        for (int s : lv) {
            int p = S.p(s), u = S.u(s), NB = solve_nb(s);
            if (V.f2split && u > 0) {
                const int Kp = (cdiv(p, V.f2split) + 31) & ~31, nparts = cdiv(p, Kp);
                skf_off_[s] = skf_level; skf_level += (int64_t)nparts * u;
                for (int q = 0; q < nparts; ++q) for (int a = 0; a < cdiv(u, 16 * sl2); ++a) tiles.push_back({s, 1 | (sl2 << 8) | (V.f2split << 16), a, q});
            } else {
                for (int a = 0; a < cdiv(u, 16 * sl2); ++a) tiles.push_back({s, 1 | (sl2 << 8), a, 0});
            }
            for (int t = 0; t < cdiv(p, NB); ++t)
                for (int rt = 0; rt < cdiv(std::min(NB, p - t * NB), 16 * sl2); ++rt) tiles.push_back({s, 2 | (sl2 << 8), t, rt});
            V.f2.bytes += ((double)u * p + 0.5 * p * std::min(p, NB)) * 16.0; V.f2.flops += (double)u * p + 0.5 * p * std::min(p, NB);
        }
        V.f2.count = (int)(tiles.size() - V.f2.off);
        V.f2.sl = sl2;
        O.sk_rows = std::max(O.sk_rows, skf_level);
        // longest tiles first (LPT): the K loop of a tile is p (kind 1; a part of it when split) or at most the super-block size (kind 2); fronts stay together
        {
            const int nsp = V.f2split;
            std::stable_sort(tiles.begin() + V.f2.off, tiles.end(), [&](const Tile& a, const Tile& b) {
                auto klen = [&](const Tile& t) { return (t.a & 0xff) == 1 ? (nsp ? (cdiv(S.p(t.f), nsp) + 31) & ~31 : S.p(t.f)) : std::min(solve_nb(t.f), S.p(t.f)); };
                return klen(a) > klen(b);
            });
        }
        V.f2red.off = (int64_t)tiles.size();
        if (V.f2split) for (int s : lv) if (S.u(s) > 0) for (int a = 0; a < cdiv(S.u(s), 64); ++a) tiles.push_back({s, a, 0, 0});
        V.f2red.count = (int)(tiles.size() - V.f2red.off);
        // the same tiles as two lists, for the sweeps over lazily converted factors (engine.hpp): first every Y_t = inv(L_tt) w_t
        // (kind 2), then the contribution rows C_I = .. - L21 Y_J (kind 1, raw L21) -- in the order f2 has them (longest K first)
        for (int pass = 0; pass < 2; ++pass) {
            Launch& L = pass == 0 ? V.f2y : V.f2c;
            L.off = (int64_t)tiles.size(); L.sl = sl2;
            if (!V.f2split)
                for (int64_t q = V.f2.off, qe = V.f2.off + V.f2.count; q < qe; ++q) { const Tile tq = tiles[(size_t)q]; if (((tq.a & 0xff) == 2) == (pass == 0)) tiles.push_back(tq); }
            L.count = (int)(tiles.size() - L.off);
        }
        V.b1.off = (int64_t)tiles.size();
        int sl3 = pick_slabs([&](auto&& cb) { for (int s : lv) if (S.u(s) > 0) cb(S.p(s)); });
        // 64-row tiles where 32 rows were only chosen for the short K loops of small fronts (the gather's K loop is long): fronts of up
        // to 64 pivot columns then fuse their B3.  With few nodes per GPU the smaller tiles' extra workgroups are worth more.
        if (fuse_b3_ && sl3 == 2 && pmax <= short_k_max_p_ && nodes_local() >= 12) sl3 = 4;
        {   // few, long tiles (top of the tree): split the K = u loop over several workgroups
            int64_t wgs = 0; int umax = 0;
            for (int s : lv) if (S.u(s) > 0) { wgs += cdiv(S.p(s), 64); umax = std::max(umax, S.u(s)); }
            int ns = 0;
            int keff = std::max(1, nodes_local() / stream_groups());     // nodes per launch
            if (wgs > 0 && wgs * keff < 1024 && umax >= 512) ns = (int)std::min<int64_t>(8, std::min<int64_t>(umax / 256, cdiv(2048, (int)(wgs * keff))));
            V.b1split = ns >= 2 ? ns : 0;
            if (V.b1split) sl3 = 4;
        }
        int64_t sk_level = 0;
        for (int s : lv) if (S.u(s) > 0) {
            int p = S.p(s), u = S.u(s);
            if (V.b1split) {
                int Kp = ((cdiv(u, V.b1split)) + 31) & ~31, nparts = cdiv(u, Kp);
                sk_off_[s] = sk_level; sk_level += (int64_t)nparts * p;
                for (int q = 0; q < nparts; ++q) for (int a = 0; a < cdiv(p, 64); ++a) tiles.push_back({s, 3 | (sl3 << 8) | (V.b1split << 16), a, q});
            } else {
                const bool fuse = fuse_b3_ && p <= 16 * sl3 && p <= solve_nb(s);     // one row tile, one diagonal block: B1 + B3 in one tile
                fused_b3[s] = fuse ? 1 : 0;
                for (int a = 0; a < cdiv(p, 16 * sl3); ++a) tiles.push_back({s, (fuse ? 6 : 3) | (sl3 << 8), a, 0});
            }
            V.b1.bytes += (double)u * p * 16.0; V.b1.flops += (double)u * p;
        }
        V.b1.count = (int)(tiles.size() - V.b1.off);
        V.b1.sl = sl3;
        std::stable_sort(tiles.begin() + V.b1.off, tiles.end(), [&](const Tile& a, const Tile& b) { return S.u(a.f) > S.u(b.f); });   // K = u
        O.sk_rows = std::max(O.sk_rows, sk_level);
        V.b1red.off = (int64_t)tiles.size();
        if (V.b1split) for (int s : lv) if (S.u(s) > 0) for (int a = 0; a < cdiv(S.p(s), 64); ++a) tiles.push_back({s, a, 0, 0});
        V.b1red.count = (int)(tiles.size() - V.b1red.off);
        V.b2.assign(nsteps, Launch());
        for (int c = 1; c < nsteps; ++c) {
            Launch& L = V.b2[c];
            L.off = (int64_t)tiles.size();
            int sl = pick_slabs([&](auto&& cb) { for (int s : lv) { int NB = solve_nb(s); if (S.p(s) > c * NB) cb(c * NB); } });
            for (int s : lv) {
                int NB = solve_nb(s), p = S.p(s), k0 = c * NB;
                if (p <= k0) continue;
                int b = std::min(NB, p - k0);
                for (int a = 0; a < cdiv(k0, 16 * sl); ++a) tiles.push_back({s, 4 | (sl << 8), a, c});
                L.bytes += (double)k0 * b * 16.0; L.flops += (double)k0 * b; L.vrows += 2.0 * k0 + b;
            }
            L.count = (int)(tiles.size() - L.off);
            L.sl = sl;
        }
        // ---- adjoint sweeps (kernels.hpp, kinds 8 .. 13): the same blocks, conjugate-transposed, in the reverse order
        {
            const int sld = std::max(2, pick_slabs([&](auto&& cb) { for (int s : lv) cb(S.p(s)); }));
            V.afd.off = (int64_t)tiles.size(); V.afd.sl = sld;
            for (int s : lv) {
                int p = S.p(s), NB = solve_nb(s);
                for (int t = 0; t < cdiv(p, NB); ++t)
                    for (int rt = 0; rt < cdiv(std::min(NB, p - t * NB), 16 * sld); ++rt) tiles.push_back({s, 8 | (sld << 8), t, rt});
            }
            V.afd.count = (int)(tiles.size() - V.afd.off);
            V.afe.assign(nsteps, Launch());
            V.abe.assign(nsteps, Launch());
            for (int c = 1; c < nsteps; ++c) {         // forward step c: rows of block c, K = c NB
                Launch& L = V.afe[c];
                L.off = (int64_t)tiles.size();
                int sl = pick_slabs([&](auto&& cb) { for (int s : lv) { int NB = solve_nb(s); if (S.p(s) > c * NB) cb(std::min(NB, S.p(s) - c * NB)); } });
                for (int s : lv) {
                    int NB = solve_nb(s), p = S.p(s), k0 = c * NB;
                    if (p <= k0) continue;
                    for (int a = 0; a < cdiv(std::min(NB, p - k0), 16 * sl); ++a) tiles.push_back({s, 9 | (sl << 8), a, c});
                }
                L.count = (int)(tiles.size() - L.off); L.sl = sl;
            }
            for (int c = 0; c + 1 < nsteps; ++c) {     // backward step c: rows of block c, K = p - (c + 1) NB
                Launch& L = V.abe[c];
                L.off = (int64_t)tiles.size();
                int sl = pick_slabs([&](auto&& cb) { for (int s : lv) { int NB = solve_nb(s); if (S.p(s) > c * NB + NB) cb(NB); } });
                for (int s : lv) {
                    int NB = solve_nb(s), p = S.p(s), k0 = c * NB;
                    if (p <= k0 + NB) continue;
                    for (int a = 0; a < cdiv(NB, 16 * sl); ++a) tiles.push_back({s, 13 | (sl << 8), a, c});
                }
                L.count = (int)(tiles.size() - L.off); L.sl = sl;
            }
            V.af2.off = (int64_t)tiles.size();
            const int slc = V.f2split ? pick_slabs([&](auto&& cb) { for (int s : lv) if (S.u(s) > 0) cb(S.u(s)); }) : sl2;
            for (int s : lv) {
                int p = S.p(s), u = S.u(s), NB = solve_nb(s);
                for (int a = 0; a < cdiv(u, 16 * slc); ++a) tiles.push_back({s, 10 | (slc << 8), a, 0});
                for (int t = 0; t < cdiv(p, NB); ++t)
                    for (int rt = 0; rt < cdiv(std::min(NB, p - t * NB), 16 * slc); ++rt) tiles.push_back({s, 11 | (slc << 8), t, rt});
            }
            V.af2.count = (int)(tiles.size() - V.af2.off); V.af2.sl = slc;
            std::stable_sort(tiles.begin() + V.af2.off, tiles.end(), [&](const Tile& a, const Tile& b) {
                auto klen = [&](const Tile& t) { return (t.a & 0xff) == 10 ? S.p(t.f) : std::min(solve_nb(t.f), S.p(t.f)); };
                return klen(a) > klen(b);
            });
            V.ab1.off = (int64_t)tiles.size();
            const int slg = V.b1split ? 4 : pick_slabs([&](auto&& cb) { for (int s : lv) if (S.u(s) > 0) cb(S.p(s)); });
            for (int s : lv) if (S.u(s) > 0) {
                int p = S.p(s), u = S.u(s);
                if (V.b1split) {
                    int Kp = ((cdiv(u, V.b1split)) + 31) & ~31, nparts = cdiv(u, Kp);      // the scratch rows of the primal split (sk_off_) serve here too
                    for (int q = 0; q < nparts; ++q) for (int a = 0; a < cdiv(p, 64); ++a) tiles.push_back({s, 12 | (slg << 8) | (V.b1split << 16), a, q});
                } else {
                    for (int a = 0; a < cdiv(p, 16 * slg); ++a) tiles.push_back({s, 12 | (slg << 8), a, 0});
                }
            }
            V.ab1.count = (int)(tiles.size() - V.ab1.off); V.ab1.sl = slg;
            std::stable_sort(tiles.begin() + V.ab1.off, tiles.end(), [&](const Tile& a, const Tile& b) { return S.u(a.f) > S.u(b.f); });   // K = u
        }
        const int sl5 = std::max(2, pick_slabs([&](auto&& cb) { for (int s : lv) cb(S.p(s)); }));
        for (int pass = 0; pass < 2; ++pass) {       // the fronts whose B3 rides with their B1 tile (kind 6) go to their own launch
            Launch& L = pass == 0 ? V.b3 : V.b3s;
            L.off = (int64_t)tiles.size();
            L.sl = sl5;
            for (int s : lv) {
                if ((fused_b3[s] != 0) != (pass == 1)) continue;
                int p = S.p(s), NB = solve_nb(s);
                for (int t = 0; t < cdiv(p, NB); ++t)
                    for (int rt = 0; rt < cdiv(std::min(NB, p - t * NB), 16 * sl5); ++rt) tiles.push_back({s, 5 | (sl5 << 8), t, rt});
                L.bytes += 0.5 * p * std::min(p, NB) * 16.0; L.flops += 0.5 * p * std::min(p, NB); L.vrows += 2.0 * p;
            }
            L.count = (int)(tiles.size() - L.off);
        }
    // ---- split of every block-substitution launch over the 8 XCDs (dev_common.hpp: XcdParts).  A workgroup's XCD is fixed by its index,
    // so the parts are cut here for equal estimated time: a tile costs a start-up plus its K extent times the 16-row slabs a wave
    // walks (host mirror of decode_tile).  Launches of many small fronts are cut contiguously at front boundaries (neighbouring fronts,
    // and a front's tiles in consecutive launches, stay on one XCD's L2); launches with few, long tiles (top of the tree, or the sorted
    // lists of the L21 / U12 products when few nodes are left on this GPU) are dealt longest-first to the least loaded XCD.
    {
        auto tile_cost = [&](const Tile& t) -> double {
            const int kind = t.a & 0xff, slv = (t.a >> 8) & 0xff, th = slv ? 16 * slv : 64, nsplit = t.a >> 16;
            const int p = S.p(t.f), u = S.u(t.f), NB = solve_nb(t.f);
            int Kd = 0, rows = th;
            auto part = [&](int len) { if (!nsplit) return len; int Kp = (cdiv(len, nsplit) + 31) & ~31; return std::max(0, std::min(Kp, len - t.c * Kp)); };
            switch (kind) {
                case 0: { int k0 = t.c * NB, b = std::min(NB, p - k0); Kd = b; rows = p - (k0 + b + t.b * th); } break;
                case 1: Kd = part(p); rows = u - t.b * th; break;
                case 2: case 8: { int k0 = t.b * NB, b = std::min(NB, p - k0); Kd = std::min(b, t.c * th + th); rows = b - t.c * th; } break;
                case 3: case 6: case 12: Kd = part(u); rows = p - t.b * th; break;
                case 4: { int k0 = t.c * NB; Kd = std::min(NB, p - k0); rows = k0 - t.b * th; } break;
                case 5: case 11: { int kb = t.b * NB, b = std::min(NB, p - kb), ks = t.c * th; Kd = b - ks; rows = b - ks; } break;
                case 9: { int kc0 = t.c * NB, b = std::min(NB, p - kc0); Kd = kc0; rows = b - t.b * th; } break;
                case 10: Kd = p; rows = u - t.b * th; break;
                default: { int kc0 = t.c * NB, b = std::min(NB, p - kc0); Kd = p - kc0 - b; rows = b - t.b * th; } break;     // 13
            }
            rows = std::max(1, std::min(rows, th));
            const int ns = rows <= 16 ? 1 : rows <= 32 ? 2 : 4;
            return xcd_c0 + (double)std::max(0, Kd) * ns + (kind == 6 ? 2.0 * p : 0.0);
        };
        std::vector<double> w;
        std::vector<int> order, bin;
        bool plans_ok = true;
        auto balance = [&](Launch& L) {
            const int nt = L.count;
            for (int x = 0; x <= 8; ++x) L.xp.off[x] = (int)((int64_t)nt * x / 8);
            L.xp.cmax = cdiv(nt, 8); L.xp.perm = -1;
            if (nt <= 0 || !xcd_balance_) return;
            const Tile* tl = tiles.data() + L.off;
            w.resize(nt);
            double total = 0, gmax = 0, gcur = 0;
            for (int i = 0; i < nt; ++i) {
                w[i] = tile_cost(tl[i]); total += w[i];
                const bool same = i > 0 && tl[i].f == tl[i - 1].f && (tl[i].a & 0xff) == (tl[i - 1].a & 0xff);
                gcur = same ? gcur + w[i] : w[i];
                gmax = std::max(gmax, gcur);
            }
            const double target = total / 8;
            if (gmax * xcd_thr <= target) {          // many small groups: contiguous parts, cut where the running cost passes x / 8 of the total
                double cum = 0; int x = 0, i = 0;
                L.xp.off[0] = 0;
                while (i < nt) {
                    int e = i + 1; double g = w[i];
                    while (e < nt && tl[e].f == tl[i].f && (tl[e].a & 0xff) == (tl[i].a & 0xff)) g += w[e++];
                    while (x < 7 && cum + 0.5 * g > (x + 1) * target) L.xp.off[++x] = i;
                    cum += g; i = e;
                }
                while (x < 8) L.xp.off[++x] = nt;
            } else {                            // longest first to the least loaded XCD
                order.resize(nt); bin.assign(nt, 0);
                for (int i = 0; i < nt; ++i) order[i] = i;
                std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return w[a] > w[b]; });
                double load[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int i : order) {
                    int best = 0;
                    for (int x = 1; x < 8; ++x) if (load[x] < load[best]) best = x;
                    bin[i] = best; load[best] += w[i]; ++cnt[best];
                }
                L.xp.off[0] = 0;
                for (int x = 0; x < 8; ++x) L.xp.off[x + 1] = L.xp.off[x] + cnt[x];
                int pos[8];
                for (int x = 0; x < 8; ++x) pos[x] = L.xp.off[x];
                L.xp.perm = (int)xperm.size();
                xperm.resize(xperm.size() + nt);
                for (int i : order) xperm[L.xp.perm + pos[bin[i]]++] = i;
            }
            L.xp.cmax = 0;
            for (int x = 0; x < 8; ++x) L.xp.cmax = std::max(L.xp.cmax, L.xp.off[x + 1] - L.xp.off[x]);
            // every tile in exactly one part (a wrong split would silently skip or repeat work): checked here, once per analysis
            bool fine = L.xp.off[0] == 0 && L.xp.off[8] == nt;
            for (int x = 0; x < 8 && fine; ++x) fine = L.xp.off[x] <= L.xp.off[x + 1];
            if (fine && L.xp.perm >= 0) {
                bin.assign(nt, 0);
                for (int i = 0; i < nt && fine; ++i) { const int t2 = xperm[L.xp.perm + i]; fine = t2 >= 0 && t2 < nt && !bin[t2]; if (fine) bin[t2] = 1; }
            }
            if (!fine) plans_ok = false;
        };
        {
            for (Launch& L : V.f1) balance(L);
            for (Launch& L : V.b2) balance(L);
            for (Launch& L : V.afe) balance(L);
            for (Launch& L : V.abe) balance(L);
            balance(V.f2); balance(V.b1); balance(V.b3); balance(V.b3s); balance(V.afd); balance(V.af2); balance(V.ab1);
            balance(V.f2y); balance(V.f2c);
        }
        O.ok = plans_ok;
    }
    };      // build_level
    {
        // deepest levels first: they hold most of the tiles
        std::atomic<int> next_level{nd - 1};
        run_parts(std::min(nd, 8), [&](int) { for (;;) { const int d = next_level.fetch_sub(1); if (d < 0) break; build_level(d); } });
    }
    std::vector<Tile> tiles;
    std::vector<int> lists, xperm;
    {
        size_t nt = 0, nl = 0, nx = 0;
        for (LevelOut& O : outs) { nt += O.tiles.size(); nl += O.lists.size(); nx += O.xperm.size(); }
        tiles.reserve(nt + 1); lists.reserve(nl + 1); xperm.reserve(nx + 1);
        for (int d = 0; d < nd; ++d) {
            LevelOut& O = outs[d];
            if (!O.ok) { err = "build_plans: internal, a launch list is not split into disjoint parts"; return false; }
            sk_rows_ = std::max(sk_rows_, O.sk_rows);
            const int64_t tb = (int64_t)tiles.size(), lb = (int64_t)lists.size();
            const int xb = (int)xperm.size();
            auto shift = [&](Launch& L) { L.off += tb; if (L.xp.perm >= 0) L.xp.perm += xb; };
            auto shift_list = [&](Launch& L) { L.off += lb; };
            FactorLevel& F = fplan_[d];
            SolveLevel& V = splan_[d];
            shift(F.extend); shift(F.extend1); shift(F.cbgemm); shift(F.cbgemm_sym); shift_list(F.fronts); shift(F.binv); shift(F.couple); shift(F.scale);
            for (int lvl = 0; lvl < 3; ++lvl) { shift(F.outer[lvl]); shift(F.inner[lvl]); }
            for (FactorStep& st : F.steps) { shift_list(st.panel); shift(st.swap); shift(st.trsm); shift(st.gemm); shift(st.gemm0); shift(st.gemm_in); shift(st.grp); shift(st.lsg); }
            shift(V.assemble); shift(V.f2); shift(V.b1); shift(V.b3); shift(V.b1red); shift(V.f2red); shift(V.b3s); shift(V.afd); shift(V.af2); shift(V.ab1);
            shift(V.f2y); shift(V.f2c);
            for (Launch& L : V.f1) shift(L);
            for (Launch& L : V.b2) shift(L);
            for (Launch& L : V.afe) shift(L);
            for (Launch& L : V.abe) shift(L);
            tiles.insert(tiles.end(), O.tiles.begin(), O.tiles.end());
            lists.insert(lists.end(), O.lists.begin(), O.lists.end());
            xperm.insert(xperm.end(), O.xperm.begin(), O.xperm.end());
            O = LevelOut();
        }
        if (xperm.empty()) xperm.push_back(0);
        if (!d_xperm_.upload(xperm.data(), xperm.size(), stream_)) return false;
        dsym_.xperm = d_xperm_.p;
    }
    if (tiles.empty()) tiles.push_back({0, 0, 0, 0});
    if (lists.empty()) lists.push_back(0);
    if (getenv("FEASTHIP_ANALYZE_TRACE")) fprintf(stderr, "[feasthip analyze]   launch plans: %zu tiles (%.1f MB)\n", tiles.size(), tiles.size() * sizeof(Tile) / 1048576.0);
    return d_tiles_.upload(tiles.data(), tiles.size(), stream_) && d_lists_.upload(lists.data(), lists.size(), stream_);
}

static std::mutex g_sym_mu;
// Structure of the CSR copy of a term (the transpose of the caller's CSC arrays) and where every CSC entry lands in it: a later analysis of
// the same patterns uploads these and lets the device permute the values (k_permute_vals) instead of transposing on the host again.
struct CsrPattern {
    std::vector<int64_t> rp;
    std::vector<int> col, q;        // q[e]: position of CSC entry e in the CSR arrays
    bool same = false;              // the CSR arrays equal the CSC arrays: a symmetric pattern with sorted indices
};
struct SymCacheEntry {       // most recent symbolic analyses (see Engine::analyze), with the arena destinations of the terms' entries
    uint64_t key = 0;
    Symbolic S;
    std::vector<std::shared_ptr<const std::vector<int64_t>>> dest;      // per term, filled by the first handle that computes them
    std::vector<std::shared_ptr<const CsrPattern>> csr;                 // per term: the CSR copy's structure (SpMM), likewise
};
static std::vector<std::shared_ptr<SymCacheEntry>> g_sym_cache;

int Engine::analyze() {
    if (n <= 0 || terms_.empty()) return fail(FEASTHIP_ESTATE, "analyze: no matrices set");
    if (Kloc <= 0) return fail(FEASTHIP_ESTATE, "analyze: no contour nodes on this shard");
    HIPCHK(hipSetDevice(device_));
    auto t0 = std::chrono::steady_clock::now();
    // A HIP stream gets its hardware queue with the first operation submitted to it (25-30 ms each, measured): every stream of
    // the handle is touched from a helper thread now, beside the host-side analysis, instead of in the middle of the first
    // upload / factorisation / contour step.
    std::thread warm;
    if (!streams_warm_) {
        streams_warm_ = true;
        warm = std::thread([this]() {
            if (hipSetDevice(device_) != hipSuccess) return;
            void* q = nullptr;
            if (hipMalloc(&q, 256) != hipSuccess) { (void)hipGetLastError(); return; }
            std::vector<hipStream_t> all{stream_, la_stream_, fin_stream_, rest_stream_};
            all.insert(all.end(), aux_streams_.begin(), aux_streams_.end());
            for (hipStream_t st : all) if (st) (void)hipMemsetAsync(q, 0, 256, st);
            for (hipStream_t st : all) if (st) (void)hipStreamSynchronize(st);
            (void)hipFree(q);
        });
    }
    struct JoinWarm { std::thread& t; ~JoinWarm() { if (t.joinable()) t.join(); } } join_warm{warm};
    // per-term host work that runs beside the ordering (CSR copies + their uploads) and beside the launch plans (arena destinations)
    struct TermHost { std::vector<int64_t> dest; std::shared_ptr<const CsrPattern> pat; char bad = 0, bad_dest = 0, sym = 0, herm = 0, diag = 0, hreal = 0; };
    std::vector<TermHost> th;
    std::shared_ptr<SymCacheEntry> sym_entry;      // the cache entry this analysis came from / went to (sparse terms)
    std::vector<std::shared_ptr<const std::vector<int64_t>>> dest_ready;    // arena destinations found in it, per term
    std::vector<std::thread> csr_pool, dest_pool;
    struct JoinAll { std::vector<std::thread>& p; ~JoinAll() { for (auto& t : p) if (t.joinable()) t.join(); } } csr_join{csr_pool}, dest_join{dest_pool};
    if (dense) {
        analyze_dense(n, S);
    } else {
        SymbolicOptions opt;
        if (const char* v = getenv("FEASTHIP_RELAX_SMALL")) opt.relax_small = atoi(v);      // tuning knobs
        if (const char* v = getenv("FEASTHIP_RELAX_MID")) opt.relax_mid = atoi(v);
        if (const char* v = getenv("FEASTHIP_ZMID")) opt.zmid = atof(v);
        if (const char* v = getenv("FEASTHIP_ZBIG")) opt.zbig = atof(v);
        if (const char* v = getenv("FEASTHIP_LEAF")) opt.leaf_size = atoi(v);
        const bool trace = getenv("FEASTHIP_ANALYZE_TRACE") != nullptr;
        // Ordering + symbolic analysis (and the arena destination of every entry of every term) depend only on the patterns: FEAST is
        // typically run several times on the same matrices (other contours / spectrum slices, README.md:12 of the reference; one
        // factor object per node at the plug-in seam), so they are kept in a small process-wide cache keyed by a hash of the terms'
        // patterns and the options.
        uint64_t key = 1469598103934665603ull;
        auto mix = [&](uint64_t v) { key = (key ^ v) * 1099511628211ull; };
        mix((uint64_t)n); mix((uint64_t)terms_.size());
        for (auto& T : terms_) {
            for (int j = 0; j <= n; ++j) mix((uint64_t)T.colptr[j]);
            const int64_t nnz = T.colptr[n];
            for (int64_t t = 0; t < nnz; ++t) mix((uint64_t)T.rowval[t]);
        }
        mix((uint64_t)opt.leaf_size); mix((uint64_t)opt.relax_small); mix((uint64_t)opt.relax_mid); mix((uint64_t)(opt.zmid * 1e6)); mix((uint64_t)(opt.zbig * 1e6)); mix((uint64_t)opt.ordering);
        if (trace) fprintf(stderr, "[feasthip analyze] pattern hash                 %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        // CSR copies of the terms for SpMM (a transpose of the caller's CSC arrays) do not depend on the ordering: one host thread
        // per term builds them while the nested dissection leaves most cores idle at its top levels
        if (trace) fprintf(stderr, "[feasthip analyze] + pattern hash               %.1f ms (cumulative)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        th.resize(terms_.size());
        for (auto& T : terms_) if (T.colptr[n] >= ((int64_t)1 << 31)) return fail(FEASTHIP_EARG, "analyze: more than 2^31 - 1 entries in one term");
        {
            std::lock_guard<std::mutex> lk(g_sym_mu);
            for (auto& c : g_sym_cache) if (c->key == key && c->S.n == n) { sym_entry = c; break; }
            if (sym_entry)
                for (size_t ti = 0; ti < terms_.size() && ti < sym_entry->csr.size(); ++ti)
                    if (sym_entry->csr[ti] && (int64_t)sym_entry->csr[ti]->q.size() == terms_[ti].colptr[n]) th[ti].pat = sym_entry->csr[ti];
        }
        // The helper threads overwrite device buffers a previous analysis of this handle may still be using (DevBuf::alloc keeps a
        // buffer that is large enough): everything queued on the handle's stream has to be through first.  One upload stream per
        // term, created once per handle (a stream gets its hardware queue with its first operation: 25-30 ms).
        if (analyzed_once_) HIPCHK(hipStreamSynchronize(stream_));      // (the first analysis of a handle has nothing to wait for)
        analyzed_once_ = true;
        if (up_streams_.size() < terms_.size()) up_streams_.resize(terms_.size(), nullptr);      // created by the helper threads themselves (3 ms each)
        for (size_t ti = 0; ti < terms_.size(); ++ti)
            csr_pool.emplace_back([&, ti]() {
                const SparseTerm& T = terms_[ti];
                TermHost& H = th[ti];
                const int64_t nnz = T.colptr[n];
                // the structure of the CSR copy and the position q[e] of every CSC entry in it: from the cache entry of these patterns
                // (H.pat, set above) or built here and published after the analysis
                std::shared_ptr<const CsrPattern> pat = H.pat;
                if (!pat) {
                    auto np = std::make_shared<CsrPattern>();
                    np->rp.assign(n + 1, 0);
                    for (int64_t e2 = 0; e2 < nnz; ++e2) np->rp[T.rowval[e2] + 1]++;
                    for (int i = 0; i < n; ++i) np->rp[i + 1] += np->rp[i];
                    np->col.resize(nnz);
                    np->q.resize(nnz);
                    std::vector<int64_t> pos(np->rp.begin(), np->rp.end() - 1);
                    for (int j = 0; j < n; ++j)
                        for (int64_t e2 = T.colptr[j]; e2 < T.colptr[j + 1]; ++e2) { int64_t q = pos[T.rowval[e2]]++; np->col[q] = j; np->q[e2] = (int)q; }
                    bool same = true;
                    for (int j = 0; j <= n && same; ++j) same = np->rp[j] == T.colptr[j];
                    for (int64_t e2 = 0; e2 < nnz && same; ++e2) same = np->col[e2] == T.rowval[e2];
                    np->same = same;
                    pat = np;
                    H.pat = np;
                }
                const int* const q = pat->q.data();
                // The CSR arrays of M are the CSC arrays of M^T with sorted indices: M == M^T iff they equal the caller's CSC arrays
                // (unsorted caller indices compare unequal and merely keep the pivoted factorisation): the same structure, and the value
                // of every entry equal to the one at its transposed position
                {
                    bool eq = pat->same;
                    for (int64_t e2 = 0; e2 < nnz && eq; ++e2) eq = T.val[q[e2]] == T.val[e2];
                    H.sym = eq ? 1 : 0;
                    // ... and M == M^H off the diagonal iff they equal them up to conjugation of the values (the diagonal is left out:
                    // an assembled shifted matrix z I - A, z D - A of a Hermitian pencil has a complex diagonal and is as good a
                    // candidate for the elimination without interchanges as the pencil's terms; the growth check decides)
                    bool eqh = pat->same;
                    for (int j = 0; j < n && eqh; ++j)
                        for (int64_t e2 = T.colptr[j]; e2 < T.colptr[j + 1] && eqh; ++e2)
                            eqh = T.rowval[e2] == j || T.val[q[e2]] == std::conj(T.val[e2]);
                    H.herm = eqh ? 1 : 0;
                    bool dreal = true;      // a real diagonal on top of that: the matrix IS Hermitian (adjoint fold)
                    for (int j = 0; j < n && dreal; ++j)
                        for (int64_t e2 = T.colptr[j]; e2 < T.colptr[j + 1]; ++e2) if (T.rowval[e2] == j && T.val[e2].imag() != 0.0) { dreal = false; break; }
                    H.hreal = dreal ? 1 : 0;
                }
                {   // entries on the diagonal only: the Rayleigh-Ritz step then scales rows instead of running SpMM products
                    bool dg = !dense;
                    for (int j = 0; j < n && dg; ++j)
                        for (int64_t e2 = T.colptr[j]; e2 < T.colptr[j + 1]; ++e2) if (T.rowval[e2] != j) { dg = false; break; }
                    H.diag = dg ? 1 : 0;
                }
                // ... and uploads them (and the CSC values the factorisation scatters) from this thread, on its own stream
                SparseTerm& Tw = terms_[ti];
                bool ok = hipSetDevice(device_) == hipSuccess;
                if (ok && !up_streams_[ti] && hipStreamCreateWithFlags(&up_streams_[ti], hipStreamNonBlocking) != hipSuccess) { up_streams_[ti] = nullptr; ok = false; }
                hipStream_t us = up_streams_[ti];
                // (the values in CSR order are a permutation of the CSC values the factorisation scatters: made on the device)
                ok = ok && Tw.d_cscval.upload((const cz*)T.val.data(), nnz, us) && Tw.d_rowptr.upload(pat->rp.data(), n + 1, us) &&
                     Tw.d_col.upload(pat->col.data(), nnz, us) && Tw.d_csrq.upload(q, nnz, us) && Tw.d_val.alloc(nnz);
                if (ok && nnz > 0) { launch_permute_vals(us, nnz, Tw.d_csrq.p, Tw.d_cscval.p, Tw.d_val.p); ok = hipGetLastError() == hipSuccess; }
                ok = ok && hipStreamSynchronize(us) == hipSuccess;      // the handle's own streams read these buffers after the join
                if (!ok) H.bad = 2;
            });
        if (getenv("FEASTHIP_PROBE_JOIN_HELPERS")) {      // probe: how much the helper threads (stream warm-up, CSR copies + uploads) slow the ordering beside them
            if (warm.joinable()) warm.join();
            for (auto& t2 : csr_pool) if (t2.joinable()) t2.join();
        }
        if (trace) fprintf(stderr, "[feasthip analyze] + stream sync, CSR threads   %.1f ms (cumulative)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        if (sym_entry) S = sym_entry->S;
        else {
            // union of the terms' patterns (+ the diagonal), column by column, every column sorted and free of duplicates (a symmetric
            // union then is its own transpose: analyze_pattern's fast path), column ranges on host threads
            std::vector<int64_t> cp(n + 1, 0), ri;
            // The usual pencil: one term carries the whole pattern (ascending columns with their diagonal entries) and the others -- a
            // diagonal or identity B, a mass matrix on the same mesh -- lie inside it.  Then the union IS that term's pattern: checked
            // column range by column range on host threads, no sort, no merge (8 -> 1.5 ms of C3's analysis).
            bool dominant = false;
            {
                size_t big = 0;
                for (size_t ti = 1; ti < terms_.size(); ++ti) if (terms_[ti].colptr[n] > terms_[big].colptr[n]) big = ti;
                const SparseTerm& D = terms_[big];
                const int nth = n >= 20000 ? 8 : 1;
                std::vector<char> bad(nth, 0);
                run_parts(nth, [&](int ti) {
                    const int j0 = (int)((int64_t)n * ti / nth), j1 = (int)((int64_t)n * (ti + 1) / nth);
                    for (int j = j0; j < j1 && !bad[ti]; ++j) {
                        const int* d0 = D.rowval.data() + D.colptr[j]; const int* d1 = D.rowval.data() + D.colptr[j + 1];
                        bool okc = std::binary_search(d0, d1, j);
                        for (const int* q = d0; q + 1 < d1 && okc; ++q) okc = q[0] < q[1];
                        for (size_t t2 = 0; t2 < terms_.size() && okc; ++t2) {
                            if (t2 == big) continue;
                            const SparseTerm& T = terms_[t2];
                            for (int64_t e2 = T.colptr[j]; e2 < T.colptr[j + 1] && okc; ++e2) okc = std::binary_search(d0, d1, T.rowval[e2]);
                        }
                        if (!okc) bad[ti] = 1;
                    }
                });
                dominant = true;
                for (char b : bad) if (b) dominant = false;
                if (dominant) {
                    cp.assign(D.colptr.begin(), D.colptr.end());
                    ri.resize(cp[n]);
                    run_parts(nth, [&](int ti) {
                        const int64_t e0 = cp[n] * ti / nth, e1 = cp[n] * (ti + 1) / nth;
                        for (int64_t e2 = e0; e2 < e1; ++e2) ri[e2] = D.rowval[e2];
                    });
                }
            }
            if (!dominant) {
                const int nth = n >= 20000 ? std::max(1, std::min(8, host_cpu_budget() / 2)) : 1;
                std::vector<std::vector<int64_t>> part(nth);
                auto build = [&](int ti) {
                    const int j0 = (int)((int64_t)n * ti / nth), j1 = (int)((int64_t)n * (ti + 1) / nth);
                    std::vector<int64_t>& out = part[ti];
                    std::vector<int64_t> colrows;
                    for (int j = j0; j < j1; ++j) {
                        colrows.clear();
                        colrows.push_back(j);
                        for (auto& T : terms_) for (int64_t e2 = T.colptr[j]; e2 < T.colptr[j + 1]; ++e2) colrows.push_back(T.rowval[e2]);
                        std::sort(colrows.begin(), colrows.end());
                        colrows.erase(std::unique(colrows.begin(), colrows.end()), colrows.end());
                        cp[j + 1] = (int64_t)colrows.size();
                        out.insert(out.end(), colrows.begin(), colrows.end());
                    }
                };
                std::vector<std::thread> pool;
                for (int ti = 1; ti < nth; ++ti) pool.emplace_back(build, ti);
                build(0);
                for (auto& t2 : pool) t2.join();
                for (int j = 0; j < n; ++j) cp[j + 1] += cp[j];
                ri.resize(cp[n]);
                int64_t off = 0;
                for (int ti = 0; ti < nth; ++ti) { std::copy(part[ti].begin(), part[ti].end(), ri.begin() + off); off += (int64_t)part[ti].size(); }
            }
            std::string e = analyze_pattern(n, cp.data(), ri.data(), opt, S);
            if (!e.empty()) return fail(FEASTHIP_EARG, "analyze: " + e);
            sym_entry = std::make_shared<SymCacheEntry>();
            sym_entry->key = key; sym_entry->S = S; sym_entry->dest.resize(terms_.size());
            std::lock_guard<std::mutex> lk(g_sym_mu);
            if (g_sym_cache.size() >= 4) g_sym_cache.erase(g_sym_cache.begin());
            g_sym_cache.push_back(sym_entry);
        }
        if (trace) fprintf(stderr, "[feasthip analyze] analyze_pattern total        %.1f ms (cumulative)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        // arena destination of every entry (a binary search in the supernode's row structure): independent per column, fanned
        // out over host threads per term, beside the launch plans below; joined and uploaded after those
        dest_ready.assign(terms_.size(), nullptr);
        {
            std::lock_guard<std::mutex> lk(g_sym_mu);
            for (size_t ti = 0; ti < terms_.size() && ti < sym_entry->dest.size(); ++ti)
                if (sym_entry->dest[ti] && (int64_t)sym_entry->dest[ti]->size() == terms_[ti].colptr[n]) dest_ready[ti] = sym_entry->dest[ti];
        }
        for (size_t ti = 0; ti < terms_.size(); ++ti)
            if (!dest_ready[ti]) dest_pool.emplace_back([&, ti]() {
                SparseTerm& T = terms_[ti];
                TermHost& H = th[ti];
                int64_t nnz = T.colptr[n];
                H.dest.resize(nnz);
                int nth = (int)std::min<int64_t>(std::max(1, host_cpu_budget() / 2), std::max<int64_t>(1, nnz / 100000));
                std::vector<std::thread> pool;
                std::vector<char> bad(nth, 0);
                for (int tix = 1; tix <= nth; ++tix) {
                    auto work = [&, tix]() {
                        int j0 = (int)((int64_t)n * (tix - 1) / nth), j1 = (int)((int64_t)n * tix / nth);
                        for (int j = j0; j < j1; ++j)
                            for (int64_t e2 = T.colptr[j]; e2 < T.colptr[j + 1]; ++e2) {
                                H.dest[e2] = entry_dest(S, T.rowval[e2], j);
                                if (H.dest[e2] < 0) bad[tix - 1] = 1;
                            }
                    };
                    if (tix < nth) pool.emplace_back(work); else work();
                }
                for (auto& t2 : pool) t2.join();
                for (char b2 : bad) if (b2) H.bad_dest = 1;
            });
    }
    // device symbolic
    std::vector<FrontD> fr(S.nsuper);
    for (int s = 0; s < S.nsuper; ++s) {
        FrontD& f = fr[s];
        f.c0 = S.sn_start[s]; f.p = S.p(s); f.u = S.u(s); f.parent = S.sn_parent[s];
        f.lp = S.lp_off[s]; f.f12 = S.f12_off[s]; f.cb = S.cb_off[s];
        f.sptr = S.struct_ptr[s]; f.crow = S.crow_off[s];
        f.child_ptr = S.child_ptr[s]; f.nchild = (int)(S.child_ptr[s + 1] - S.child_ptr[s]); f.depth = S.depth[s];
        f.sk_off = -1;
        f.skf_off = -1;
        f.cmap = -1;
        f.ebp = -1;
        f.nb = solve_nb(s); f.wslot = 0;
    }
    wstride_ = 1;
    for (const std::vector<int>& lv : S.levels) {
        for (size_t i = 0; i < lv.size(); ++i) fr[lv[i]].wslot = (int)i;
        wstride_ = std::max<int64_t>(wstride_, (int64_t)lv.size());
    }
    // pull map of the forward sweep: for every contribution row of a front and every child slot, the row of the child's
    // contribution block that is added to it (rel >= p), as an absolute row of the children's arena; -1 if none
    std::vector<int> cmap;
    for (int s = 0; s < S.nsuper; ++s) {
        int nch = fr[s].nchild, u = fr[s].u, p = fr[s].p;
        if (nch == 0 || u == 0) continue;
        fr[s].cmap = (int64_t)cmap.size();
        size_t base = cmap.size();
        cmap.resize(base + (size_t)u * nch, -1);
        for (int c = 0; c < nch; ++c) {
            int gch = S.child_idx[S.child_ptr[s] + c];
            int ug = S.u(gch);
            for (int i = 0; i < ug; ++i) {
                int dst = S.rel[S.struct_ptr[gch] + i];
                if (dst >= p) cmap[base + (size_t)(dst - p) * nch + c] = (int)(S.crow_off[gch] + i);
            }
        }
    }
    if (cmap.empty()) cmap.push_back(-1);
    // ... and the same for the pivot rows (rel < p): the forward sweep's assembly adds, row by row, the children's rows in child order
    std::vector<int> pmap;
    for (int s = 0; s < S.nsuper; ++s) {
        int nch = fr[s].nchild, p = fr[s].p;
        fr[s].pmap = -1;
        if (nch == 0) continue;
        fr[s].pmap = (int64_t)pmap.size();
        size_t base = pmap.size();
        pmap.resize(base + (size_t)p * nch, -1);
        for (int c = 0; c < nch; ++c) {
            int gch = S.child_idx[S.child_ptr[s] + c];
            int ug = S.u(gch);
            for (int i = 0; i < ug; ++i) {
                int dst = S.rel[S.struct_ptr[gch] + i];
                if (dst < p) pmap[base + (size_t)dst * nch + c] = (int)(S.crow_off[gch] + i);
            }
        }
    }
    if (pmap.empty()) pmap.push_back(-1);
    // extend-add break points: for a child g of parent s, ebp[b] = number of contribution rows of g that land before parent
    // row 64 b (rel is ascending), so a 64 x 64 parent tile finds its slice of the child's block without searching
    std::vector<int> ebp;
    for (int g2 = 0; g2 < S.nsuper; ++g2) {
        int par = S.sn_parent[g2];
        if (par < 0 || S.u(g2) == 0) continue;
        int nb = (S.p(par) + S.u(par) + 63) / 64, ug = S.u(g2);
        fr[g2].ebp = (int64_t)ebp.size();
        const int* rc = S.rel.data() + S.struct_ptr[g2];
        int i = 0;
        for (int b = 0; b <= nb; ++b) {
            while (i < ug && rc[i] < 64 * b) ++i;
            ebp.push_back(i);
        }
    }
    if (ebp.empty()) ebp.push_back(0);
    if (!dense) {
        // the CSR threads (long finished: they ran beside the ordering) also compared every term with its transposes: whether the
        // adjoint fold applies, and with it the number of units of this handle, is known before anything is sized by it
        for (auto& t2 : csr_pool) if (t2.joinable()) t2.join();
        terms_symmetric_ = true;
        terms_hermitian_ = true;
        terms_herm_exact_ = true;
        for (TermHost& H : th) { if (!H.sym) terms_symmetric_ = false; if (!H.herm) terms_hermitian_ = false; if (!H.herm || !H.hreal) terms_herm_exact_ = false; }
        rebuild_shard();
        if (Kloc <= 0) return fail(FEASTHIP_ESTATE, "analyze: no contour nodes on this shard");
    }
    std::vector<int> c0row(n);
    for (int i = 0; i < n; ++i) c0row[i] = S.sn_start[S.sn_of[i]];
    std::vector<int> drow(n);
    for (int i = 0; i < n; ++i) drow[i] = S.depth[S.sn_of[i]];
    std::vector<int> rid((size_t)Kloc * n);          // (function scope: the asynchronous upload reads it until the stream is synchronised below)
    for (int k = 0; k < Kloc; ++k) for (int i = 0; i < n; ++i) rid[(size_t)k * n + i] = i - c0row[i];
    if (!d_rowid_.upload(rid.data(), rid.size(), stream_)) return fail(FEASTHIP_EHIP, "analyze: device upload failed");
    std::vector<int> one(1, 0);
    bool ok = d_fr_.upload(fr.data(), fr.size(), stream_) &&
              d_struct_.upload(S.struct_idx.empty() ? one.data() : S.struct_idx.data(), std::max<size_t>(1, S.struct_idx.size()), stream_) &&
              d_rel_.upload(S.rel.empty() ? one.data() : S.rel.data(), std::max<size_t>(1, S.rel.size()), stream_) &&
              d_child_.upload(S.child_idx.empty() ? one.data() : S.child_idx.data(), std::max<size_t>(1, S.child_idx.size()), stream_) &&
              d_perm_.upload(S.perm.data(), n, stream_) && d_iperm_.upload(S.iperm.data(), n, stream_) && d_c0row_.upload(c0row.data(), n, stream_) && d_drow_.upload(drow.data(), n, stream_) &&
              d_cmap_.upload(cmap.data(), cmap.size(), stream_) && d_ebp_.upload(ebp.data(), ebp.size(), stream_) &&
              d_pmap_.upload(pmap.data(), pmap.size(), stream_);
    if (getenv("FEASTHIP_ANALYZE_TRACE")) fprintf(stderr, "[feasthip analyze] + front records, index uploads  %.1f ms (cumulative)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    err.clear();
    if (!ok || !build_plans()) return fail(FEASTHIP_EHIP, err.empty() ? std::string("analyze: device upload of the symbolic structure failed") : "analyze: " + err);
    if (getenv("FEASTHIP_ANALYZE_TRACE")) fprintf(stderr, "[feasthip analyze] + launch plans (build + upload) %.1f ms (cumulative)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    for (int s = 0; s < S.nsuper; ++s) { fr[s].sk_off = sk_off_[s]; fr[s].skf_off = skf_off_[s]; }
    if (!d_fr_.upload(fr.data(), fr.size(), stream_)) return fail(FEASTHIP_EHIP, "analyze: device upload failed");
    dsym_.fr = d_fr_.p; dsym_.struct_idx = d_struct_.p; dsym_.rel = d_rel_.p; dsym_.child_idx = d_child_.p;
    dsym_.perm = d_perm_.p; dsym_.iperm = d_iperm_.p; dsym_.c0_of_row = d_c0row_.p; dsym_.depth_of_row = d_drow_.p; dsym_.cmap = d_cmap_.p; dsym_.ebp = d_ebp_.p; dsym_.pmap = d_pmap_.p; dsym_.n = n;
    dsym_.asmflag = nullptr;
    {
        // fused right-hand-side gather: the pivot rows that receive contributions of children are written by k_fwd_assemble
        static const bool fuse_gather = [] { const char* e = getenv("FEASTHIP_FUSE_GATHER"); return e ? atoi(e) != 0 : true; }();
        if (fuse_gather && !dense && !S.rel.empty()) {
            std::vector<int> asmflag(n, 0);
            for (int s2 = 0; s2 < S.nsuper; ++s2) {
                const int p = S.p(s2), c0 = S.sn_start[s2];
                for (int64_t c = S.child_ptr[s2]; c < S.child_ptr[s2 + 1]; ++c) {
                    const int g2 = S.child_idx[c], ug = S.u(g2);
                    for (int i = 0; i < ug; ++i) {
                        const int dst = S.rel[S.struct_ptr[g2] + i];
                        if (dst < p) asmflag[c0 + dst] = 1;
                    }
                }
            }
            if (!d_asmflag_.upload(asmflag.data(), asmflag.size(), stream_))
                return fail(FEASTHIP_EHIP, "analyze: device upload failed");
            dsym_.asmflag = d_asmflag_.p;
        }
    }
    if (getenv("FEASTHIP_ANALYZE_TRACE")) fprintf(stderr, "[feasthip analyze] + plans, uploads             %.1f ms (cumulative)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    if (!dense) {
        for (auto& t2 : dest_pool) if (t2.joinable()) t2.join();
        for (auto& t2 : csr_pool) if (t2.joinable()) t2.join();
        for (size_t ti = 0; ti < terms_.size(); ++ti) {
            SparseTerm& T = terms_[ti];
            T.diagonal = th[ti].diag != 0;
            if (T.diagonal) {
                std::vector<zc> dg(n, zc(0.0));
                for (int j = 0; j < n; ++j) for (int64_t e2 = T.colptr[j]; e2 < T.colptr[j + 1]; ++e2) dg[j] += T.val[e2];
                if (!T.d_diag.upload((const cz*)dg.data(), n, stream_)) return fail(FEASTHIP_EHIP, "analyze: device upload failed");
            }
        }
        for (TermHost& H : th) if (H.bad_dest) H.bad = 1;
        for (size_t ti = 0; ti < terms_.size(); ++ti) {
            SparseTerm& T = terms_[ti];
            TermHost& H = th[ti];
            if (H.bad == 1) return fail(FEASTHIP_EARG, "analyze: internal, entry outside the symbolic structure");
            int64_t nnz = T.colptr[n];
            if (H.pat && sym_entry) {   // the CSR structure, likewise
                std::lock_guard<std::mutex> lk(g_sym_mu);
                if (sym_entry->csr.size() < terms_.size()) sym_entry->csr.resize(terms_.size());
                if (!sym_entry->csr[ti]) sym_entry->csr[ti] = H.pat;
            }
            if (!dest_ready[ti]) {      // computed here: published for the next analysis of these patterns
                auto sp2 = std::make_shared<const std::vector<int64_t>>(std::move(H.dest));
                dest_ready[ti] = sp2;
                std::lock_guard<std::mutex> lk(g_sym_mu);
                if (sym_entry && ti < sym_entry->dest.size()) sym_entry->dest[ti] = sp2;
            }
            if (H.bad == 2 || !T.d_dest.upload(dest_ready[ti]->data(), nnz, stream_)) return fail(FEASTHIP_EHIP, "analyze: device upload failed");
        }
    }
    // numeric arenas
    analyze_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    auto t_alloc = std::chrono::steady_clock::now();
    ok = d_fact_.alloc((size_t)Kloc * S.fact_size) && d_cb_[0].alloc((size_t)Kloc * std::max<int64_t>(1, S.cb_size[0])) &&
         d_cb_[1].alloc((size_t)Kloc * std::max<int64_t>(1, S.cb_size[1])) && d_piv_.alloc((size_t)Kloc * n) &&
         d_rowperm_.alloc((size_t)Kloc * n) && d_rowinv_.alloc((size_t)Kloc * n) && d_nperturb_.alloc(Kloc) &&
         d_tiny_.alloc(Kloc) && d_growth_.alloc((size_t)3 * Kloc) && d_coef_.alloc((size_t)Kloc * nterms);
    // (d_wscr_, the scratch of the eliminations without interchanges -- 67.6 KB per unit and front of the widest tree level: 0.35 GB on
    // C3, tens of GB on a wide 2-D mesh -- is allocated by factorize() when one of those paths is taken, not here)
    if (!ok) return fail(FEASTHIP_EHIP, "analyze: out of device memory for the factor arenas");
    alloc_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_alloc).count();
    if (getenv("FEASTHIP_ANALYZE_TRACE")) fprintf(stderr, "[feasthip analyze] + arena destinations, CSR    %.1f ms (cumulative)\n[feasthip analyze] arena allocation (hipMalloc)  %.1f ms\n", analyze_seconds * 1e3, alloc_seconds * 1e3);
    analyzed_ = true;
    factored_ = false;
    return 0;
}

// Rayleigh-Ritz only (feasthip_set_rr_only): the device copies of the terms that the products A Q, B Q need; no ordering, no
// symbolic analysis, no factor arenas -- the shifted solves happen outside the library (a caller-supplied factorizer).
int Engine::analyze_rr_only() {
    if (n <= 0 || terms_.empty()) return fail(FEASTHIP_ESTATE, "analyze: no matrices set");
    HIPCHK(hipSetDevice(device_));
    if (!dense) {
        for (SparseTerm& T : terms_) {
            const int64_t nnz = T.colptr[n];
            std::vector<int64_t> rp(n + 1, 0);
            for (int64_t e = 0; e < nnz; ++e) rp[T.rowval[e] + 1]++;
            for (int i = 0; i < n; ++i) rp[i + 1] += rp[i];
            std::vector<int> col(std::max<int64_t>(1, nnz));
            std::vector<zc> v(std::max<int64_t>(1, nnz));
            std::vector<int64_t> pos(rp.begin(), rp.end() - 1);
            for (int j = 0; j < n; ++j)
                for (int64_t e = T.colptr[j]; e < T.colptr[j + 1]; ++e) { int64_t q = pos[T.rowval[e]]++; col[q] = j; v[q] = T.val[e]; }
            if (!T.d_rowptr.upload(rp.data(), n + 1, stream_) || !T.d_col.upload(col.data(), col.size(), stream_) ||
                !T.d_val.upload((const cz*)v.data(), v.size(), stream_)) return fail(FEASTHIP_EHIP, "analyze: device upload failed");
        }
    }
    S = Symbolic();
    S.n = n;
    dsym_ = DevSym();
    dsym_.n = n;
    analyzed_ = true;
    factored_ = false;
    return 0;
}

// f_t(z) of T(z) = sum_t f_t(z) scale_t M_t at the given points: powers z^t by default (polynomial NEP, linear pencil),
// or the caller's functions (split form, feasthip_set_coef_functions).  out is point-major: out[i*nterms + t].
int Engine::alloc_iteration_buffers() {
    size_t blk = (size_t)n * ldw;
    const size_t lw = (size_t)solve_width(), sblk = (size_t)n * lw;      // per-unit solve blocks (twice as wide with the conjugate fold)
    bool ok = d_X_.alloc(blk) && d_R_.alloc(blk) && (q_ext_ ? (q_ext_bytes_ >= 2 * blk * sizeof(cz)) : d_Q_.alloc(2 * blk)) && d_T1_.alloc(blk) && d_T2_.alloc(blk) &&
              d_MX_.alloc(blk * std::max(2, nterms)) &&
              (rr_only_ || (d_W_.alloc((size_t)ninst() * sblk) && d_Y_.alloc((size_t)ninst() * sblk) && d_sk_.alloc((size_t)ninst() * std::max<int64_t>(1, sk_rows_) * lw) &&
                            d_crow_[0].alloc((size_t)ninst() * std::max<int64_t>(1, S.crow_size[0]) * lw) &&
                            d_crow_[1].alloc((size_t)ninst() * std::max<int64_t>(1, S.crow_size[1]) * lw))) &&
              d_e0_.alloc((size_t)std::max(1, ninst()) * lw) && d_e1_.alloc((size_t)std::max(1, ninst()) * lw) && d_small_.alloc((size_t)8 * ldw * ldw) &&
              d_lam_.alloc(ldw) && d_gram_scratch_.alloc(gram_scratch_elems(n, ldw)) && d_io_.alloc((size_t)n * std::max(1, m0)) &&
              d_nrm_.alloc(ldw) && d_nrm_scratch_.alloc(colnorm_scratch_elems(n, ldw)) && d_rrscr_.alloc(64 * 64) && d_rrstat_.alloc(16 + (size_t)ldw);
    if (ok && mixed_)
        ok = d_W32_.alloc((size_t)Kloc * sblk) && d_Y32_.alloc((size_t)Kloc * sblk) && d_sk32_.alloc((size_t)Kloc * std::max<int64_t>(1, sk_rows_) * lw) &&
             d_crow32_[0].alloc((size_t)Kloc * std::max<int64_t>(1, S.crow_size[0]) * lw) &&
             d_crow32_[1].alloc((size_t)Kloc * std::max<int64_t>(1, S.crow_size[1]) * lw);
    if (!ok) return fail(FEASTHIP_EHIP, "out of device memory for the iteration buffers");
    return 0;
}

int Engine::set_precision(int mixed) {
    if ((mixed != 0) != mixed_) { mixed_ = mixed != 0; factored_ = false; have_x_ = false; rebuild_shard(); }
    return 0;
}

int Engine::set_x(const void* X, int64_t n_, int m0_) {
    if (!X || m0_ <= 0) return fail(FEASTHIP_EARG, "set_x: bad arguments");
    if (n_ != n) return fail(FEASTHIP_EDIM_X, "Incorrect dimensions of X, must match A");
    if (m0_ > n) return fail(FEASTHIP_EARG, "set_x: m0 must not exceed n");
    if (!analyzed_) { int r = rr_only_ ? analyze_rr_only() : analyze(); if (r) return r; }
    HIPCHK(hipSetDevice(device_));
    m0 = m0_;
    ldw = ((m0 + 15) / 16) * 16;
    int r = alloc_iteration_buffers();
    if (r) return r;
    HIPCHK(hipMemcpyAsync(d_io_.p, X, (size_t)n * m0 * sizeof(cz), hipMemcpyHostToDevice, stream_));
    launch_cm_to_rm(stream_, n, m0, ldw, d_io_.p, n, d_X_.p);
    HIPCHK(hipMemcpyAsync(Qp(), d_X_.p, (size_t)n * ldw * sizeof(cz), hipMemcpyDeviceToDevice, stream_));
    HIPCHK(hipMemsetAsync(d_R_.p, 0, (size_t)n * ldw * sizeof(cz), stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    lambda_.assign(m0, zc(0.0));
    have_x_ = true;
    return 0;
}

int Engine::get_block(int which, void* out) {
    if (!have_x_ || !out) return fail(FEASTHIP_ESTATE, "get_block: no iteration state");
    HIPCHK(hipSetDevice(device_));
    const cz* src = which == 0 ? d_X_.p : which == 1 ? d_R_.p : which == 2 ? Qp() : Qp() + (size_t)n * ldw;
    launch_rm_to_cm(stream_, n, m0, ldw, src, d_io_.p, n);
    HIPCHK(hipMemcpyAsync(out, d_io_.p, (size_t)n * m0 * sizeof(cz), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    return 0;
}

int Engine::q_device(void** p, int64_t* n_, int64_t* ldq, int* nblocks) {
    if (!have_x_) return fail(FEASTHIP_ESTATE, "q_device: call set_x first");
    if (p) *p = (void*)Qp();
    if (n_) *n_ = n;
    if (ldq) *ldq = ldw;
    if (nblocks) *nblocks = 2;
    return 0;
}

int Engine::feast_run(void* X, int64_t n_, int m0_, int iter, double eps, int store, double* lambda, double* res,
                      uint8_t* inside, int* nit_out) {
    int r = set_x(X, n_, m0_);
    if (r) return r;
    int warn = 0;
    // store = false (the reference's default, src/feast.jl:5,64-65): every iteration factorises again, so a factorisation serves one
    // contour step -- the lazy conversion; the handle's setting comes back at the end.  With store = true the factors serve at most
    // `iter` sweeps: the lazy conversion saves 6 ms of C3's 76 per round of factorisations and costs 0.08 ms per sweep, so it pays
    // for itself up to ~75 iterations -- taken for iter <= 64 (the reference's default is 10).
    const bool lazy_before = lazy_conversion_;
    static const bool lazy_auto = [] { const char* e = getenv("FEASTHIP_LAZY"); return e ? atoi(e) != 0 : true; }();      // FEASTHIP_LAZY=0: feast_run never switches it on
    if (lazy_auto && (!store || iter <= 64)) lazy_conversion_ = true;
    struct RestoreLazy { bool& f; bool v; ~RestoreLazy() { f = v; } } restore_lazy{lazy_conversion_, lazy_before};
    // store = true: src/feast.jl:28-38 factorises before the loop.  Nothing of the first Rayleigh-Ritz step reads the factors, so here the
    // first contour step factorises -- its forward sweep then starts under the factorisation (early forward sweep, engine.hpp); a start
    // block that already meets the stopping criterion costs no factorisation at all.
    bool first_step_factorises = store != 0;
    int nit = 0;
    for (nit = 0; nit <= iter; ++nit) {                                             // :40
        r = rr_step(lambda, res, inside);
        if (r) return r;
        bool any = false;
        double mx = 0;
        for (int j = 0; j < m0; ++j) if (inside[j]) { any = true; mx = std::max(mx, res[j]); }
        if (any && mx < eps) break;                                                 // :53-56
        if (nit < iter) { r = contour_step((!store || first_step_factorises) ? 1 : 0); first_step_factorises = false; if (r < 0) return r; if (r > 0) warn = r; }   // :57-72
    }
    if (nit_out) *nit_out = std::min(nit, iter);
    r = get_block(0, X);
    if (r) return r;
    return warn;
}

int Engine::ldiv(void* Y, int64_t ldy, const void* R, int64_t ldr, int m0_, int adjoint) {
    if (!Y || !R || m0_ <= 0 || ldy < n || ldr < n) return fail(FEASTHIP_EARG, "ldiv: bad arguments");
    if (!factored_) return fail(FEASTHIP_ESTATE, "ldiv: not factorised");
    // unit 0 of the coefficient table must be node 0 itself: under the conjugate fold it would be the upper-half-plane member of
    // the first pair (possibly the mirror image of node 0), and the solve would silently be the one at conj(z_0)
    if (fold_ || adj_fold_ || dual_adj_ || Kloc != 1) return fail(FEASTHIP_ESTATE, "ldiv: factor handles hold exactly one unfolded node");
    HIPCHK(hipSetDevice(device_));
    if (m0_ != m0 || !have_x_) {
        m0 = m0_;
        ldw = ((m0 + 15) / 16) * 16;
        int r = alloc_iteration_buffers();
        if (r) return r;
        have_x_ = true;
        lambda_.assign(m0, zc(0.0));
    }
    // stage R column by column when ldr != n
    if (ldr == n) HIPCHK(hipMemcpyAsync(d_io_.p, R, (size_t)n * m0 * sizeof(cz), hipMemcpyHostToDevice, stream_));
    else HIPCHK(hipMemcpy2DAsync(d_io_.p, (size_t)n * sizeof(cz), R, (size_t)ldr * sizeof(cz), (size_t)n * sizeof(cz), m0, hipMemcpyHostToDevice, stream_));
    launch_cm_to_rm(stream_, n, m0, ldw, d_io_.p, n, d_R_.p);
    if (adjoint && mixed_) return fail(FEASTHIP_ESTATE, "ldiv: the adjoint solve runs in complex128 only");
    int r = adjoint ? solve_range_adj(stream_, 0, 1, 0, d_R_.p, ldw) : solve_all(d_R_.p);
    if (r) return r;
    std::vector<zc> e0((size_t)Kloc * solve_width(), zc(0.0));
    for (int j = 0; j < ldw; ++j) e0[j] = 1.0;        // node 0 only
    HIPCHK(hipMemcpyAsync(d_e0_.p, e0.data(), e0.size() * sizeof(zc), hipMemcpyHostToDevice, stream_));
    if (adjoint) launch_accumulate(stream_, dsym_, d_X_.p, adj_solution(), (int64_t)n * solve_width(), ldw, solve_width(), 1, d_e0_.p, nullptr, 0, d_T1_.p, nullptr);
    else accumulate_solutions(1, d_e0_.p, nullptr, 0, d_T1_.p, nullptr);
    launch_rm_to_cm(stream_, n, m0, ldw, d_T1_.p, d_io_.p, n);
    if (ldy == n) HIPCHK(hipMemcpyAsync(Y, d_io_.p, (size_t)n * m0 * sizeof(cz), hipMemcpyDeviceToHost, stream_));
    else HIPCHK(hipMemcpy2DAsync(Y, (size_t)ldy * sizeof(cz), d_io_.p, (size_t)n * sizeof(cz), (size_t)n * sizeof(cz), m0, hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipGetLastError());
    return 0;
}

}  // namespace feasthip
