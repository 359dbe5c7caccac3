// hostcopy.hip - page-locked bounce buffers for every copy between pageable host memory and the device (hostcopy.hpp says why).
//
// The pool: per device (the device of the stream a copy is ordered on), slots of two sizes - SMALL (64 KiB, for the many counters / offsets / short arrays of a call) and BIG (8 MiB, larger arrays go through
// them in pieces, two in flight).  A slot carries an event: a host -> device copy leaves the slot "pending" until the event is over, and whoever takes the slot next
// waits for it (or takes another slot whose event is already over).  Slots are allocated on demand with hipHostMalloc and live until the process ends - page-locked
// memory that is never unmapped is the one kind of host memory whose registration with the GPU cannot go stale.
#include "common.hpp"
#include "hostcopy.hpp"
#include <mutex>
#include <condition_variable>
#include <chrono>
#include <thread>
#include <string>
#include <cstdlib>

struct BounceSlot { void* p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool pending = false, busy = false; int dev = 0; bool big = false; };

namespace {
const size_t SMALL_CAP = (size_t)64 << 10, BIG_CAP = (size_t)8 << 20;
const size_t MAX_BIG = 16, MAX_SMALL = 256;              // 128 MiB + 16 MiB of page-locked memory per device at most
struct Pool { std::mutex m; std::condition_variable freed; std::vector<BounceSlot*> small, big; };
// one pool per device of the node (sized once from hipGetDeviceCount: no fixed limit on the number of GPUs)
Pool* pool_of(int dev) {
    static std::vector<Pool>* pools = []() { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess || n < 1) { (void)hipGetLastError(); n = 1; } return new std::vector<Pool>((size_t)n); }();
    return dev >= 0 && (size_t)dev < pools->size() ? &(*pools)[(size_t)dev] : nullptr;
}
// the device a stream belongs to (the null stream: the calling thread's current device); slots, events and copies all live on that device
int device_of(hipStream_t st) {
    int dev = -1;
    if (st && hipStreamGetDevice(st, &dev) == hipSuccess) return dev;
    (void)hipGetLastError();
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return dev;
}

BounceSlot* acquire(size_t bytes, int dev) {
    Pool* pp = pool_of(dev);
    if (!pp) return nullptr;
    Pool& P = *pp;
    const bool big = bytes > SMALL_CAP;
    BounceSlot* wait_for = nullptr;
    {
        std::unique_lock<std::mutex> g(P.m);
        std::vector<BounceSlot*>& v = big ? P.big : P.small;
        for (int attempt = 0; ; attempt++) {
            wait_for = nullptr;
            for (BounceSlot* s : v) {
                if (s->busy) continue;
                if (s->pending) { if (hipEventQuery(s->ev) != hipSuccess) { (void)hipGetLastError(); if (!wait_for) wait_for = s; continue; } s->pending = false; }
                s->busy = true;
                return s;
            }
            // nothing free without waiting: a new slot while the pool is below its cap, else the oldest pending one, else (every slot is in some thread's hands)
            // wait for a release - the cap is a bound.  A thread that waited 2 s without anybody releasing gets a slot beyond the cap rather than a deadlock
            // (several HostCopy objects of ONE thread holding the whole pool: not a pattern of the library, but not excluded by its interface either).
            if (v.size() < (big ? MAX_BIG : MAX_SMALL) || (!wait_for && attempt > 0)) {
                BounceSlot* s = new BounceSlot();
                s->cap = big ? BIG_CAP : SMALL_CAP; s->dev = dev; s->big = big;
                if (hipHostMalloc(&s->p, s->cap, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&s->ev, hipEventDisableTiming) != hipSuccess) {
                    (void)hipGetLastError();
                    if (s->p) (void)hipHostFree(s->p);
                    delete s;
                    if (!wait_for) return nullptr;
                } else { s->busy = true; v.push_back(s); return s; }
            }
            if (wait_for) break;
            P.freed.wait_for(g, std::chrono::seconds(2));
        }
        wait_for->busy = true;
    }
    if (hipEventSynchronize(wait_for->ev) != hipSuccess) (void)hipGetLastError();      // (a sticky device error shows at the caller's next call)
    wait_for->pending = false;
    return wait_for;
}
void release(BounceSlot* s, bool pending) {
    Pool& P = *pool_of(s->dev);
    { std::lock_guard<std::mutex> g(P.m); s->pending = pending; s->busy = false; }
    P.freed.notify_one();
}
}  // namespace

// ---- library-owned page-locked arrays (svx_host_alloc / svx_host_free, include/svx.h) ---------------------------------------------------------------------------
// A caller that builds its batch arrays in memory obtained here spares the bounce pass: the copy engine reads the array itself.  The memory is hipHostMalloc memory
// and stays mapped for the life of the process - svx_host_free puts a block on a free list, a later svx_host_alloc of at most that size takes it again - so the one
// kind of registration the runtime ever holds for it never goes stale (hostcopy.hpp).
#include <map>
namespace {
struct HostBlock { size_t cap; bool in_use; };
std::mutex g_host_m;
std::map<uintptr_t, HostBlock> g_host_blocks;          // by start address
}
extern "C" void* svx_host_alloc(uint64_t bytes) {
    const size_t want = bytes ? (size_t)bytes : 1;
    {
        std::lock_guard<std::mutex> g(g_host_m);
        uintptr_t best = 0; size_t best_cap = ~(size_t)0;
        for (auto& kv : g_host_blocks) if (!kv.second.in_use && kv.second.cap >= want && kv.second.cap < best_cap && kv.second.cap <= 2 * want + (1u << 20)) { best = kv.first; best_cap = kv.second.cap; }
        if (best) { g_host_blocks[best].in_use = true; return (void*)best; }
    }
    void* p = nullptr;
    const size_t cap = (want + 4095) & ~(size_t)4095;
    if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); (void)svx_fail(SVX_E_HIP, "svx_host_alloc: hipHostMalloc", __FILE__, __LINE__, hipSuccess); return nullptr; }
    std::lock_guard<std::mutex> g(g_host_m);
    g_host_blocks[(uintptr_t)p] = HostBlock{cap, true};
    return p;
}
extern "C" void svx_host_free(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> g(g_host_m);
    auto it = g_host_blocks.find((uintptr_t)p);
    if (it != g_host_blocks.end()) it->second.in_use = false;          // (kept mapped: see above)
}
// [p, p + bytes) lies inside one block handed out by svx_host_alloc
bool svx_host_owned(const void* p, size_t bytes) {
    std::lock_guard<std::mutex> g(g_host_m);
    if (g_host_blocks.empty()) return false;
    auto it = g_host_blocks.upper_bound((uintptr_t)p);
    if (it == g_host_blocks.begin()) return false;
    --it;
    return it->second.in_use && (uintptr_t)p >= it->first && (uintptr_t)p + bytes <= it->first + it->second.cap;
}

bool svx_is_device_pointer(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }     // memory the runtime does not know: pageable host memory
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeArray || a.type == hipMemoryTypeManaged;
}

// (a HostCopy dropped with device -> host pieces still in flight - an error return of its caller: the slots go back PENDING, their events were recorded behind the
// copies, so whoever takes them next waits for the copy that still writes into them)
HostCopy::~HostCopy() {
    for (auto& q : pend_) release(q.slot, true);
    if (direct_pending_) (void)hipStreamSynchronize(st_);            // the copy engine may still be reading the caller's page-locked array
}

// one piece of a large upload: its own slot, its own copy, ordered on the caller's stream like every other piece
static int h2d_piece(char* dst, const char* src, size_t n, hipStream_t st, int dev) {
    BounceSlot* s = acquire(n, dev);
    if (!s) return svx_fail(SVX_E_HIP, "no page-locked bounce buffer (hipHostMalloc)", __FILE__, __LINE__, hipSuccess);
    memcpy(s->p, src, n);
    hipError_t e = hipMemcpyAsync(dst, s->p, n, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipEventRecord(s->ev, st);
    release(s, e == hipSuccess);
    if (e != hipSuccess) return svx_fail(SVX_E_HIP, "host -> device copy through a bounce buffer", __FILE__, __LINE__, e);
    return SVX_OK;
}

int HostCopy::h2d(void* dev_dst, const void* host_src, size_t bytes) {
    if (!bytes) return SVX_OK;
    const char* src = (const char*)host_src; char* dst = (char*)dev_dst;
    if (svx_host_owned(host_src, bytes)) {
        // the library's own page-locked memory: the copy engine reads it in place; finish() (or the destructor) waits for the stream before the caller may touch it
        HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st_));
        direct_pending_ = true;
        return SVX_OK;
    }
    const int dev = device_of(st_);
    if (bytes >= ((size_t)16 << 20)) {
        // a large array (the CIGAR words / packed bases of a host batch): one thread's memcpy into the bounce buffers (5-6 GB/s) would be the bottleneck of the
        // upload, so several threads take the 8 MiB pieces in turn - each piece has a slot of its own, the device copies queue on the caller's stream
        const size_t pieces = (bytes + BIG_CAP - 1) / BIG_CAP;
        // (round 6: up to eight threads and sixteen slots - with four, the memcpy side (4 x 5-6 GB/s) was what the 6 GB of CIGAR words of a configs[1] batch waited for)
        static const int max_t = []() { const char* e = getenv("SVX_UPLOAD_THREADS"); const int v = e ? atoi(e) : 8; return v < 1 ? 1 : (v > 16 ? 16 : v); }();
        const int T = pieces < (size_t)max_t ? (int)pieces : max_t;
        std::vector<int> rc((size_t)T, SVX_OK); std::vector<std::string> msg((size_t)T);
        std::vector<std::thread> th;
        hipStream_t st = st_;
        for (int t = 0; t < T; t++) th.emplace_back([=, &rc, &msg]() {
            if (hipSetDevice(dev) != hipSuccess) { (void)hipGetLastError(); rc[(size_t)t] = SVX_E_HIP; msg[(size_t)t] = "hipSetDevice in an upload thread"; return; }
            for (size_t k = (size_t)t; k < pieces; k += (size_t)T) {
                const size_t off = k * BIG_CAP, n = bytes - off < BIG_CAP ? bytes - off : BIG_CAP;
                const int r = h2d_piece(dst + off, src + off, n, st, dev);
                if (r != SVX_OK) { rc[(size_t)t] = r; msg[(size_t)t] = g_svx_err; return; }
            }
        });
        for (auto& x : th) x.join();
        for (int t = 0; t < T; t++) if (rc[(size_t)t] != SVX_OK) { g_svx_err = msg[(size_t)t]; return rc[(size_t)t]; }
        return SVX_OK;
    }
    while (bytes) {
        const size_t n = bytes < BIG_CAP ? bytes : BIG_CAP;
        BounceSlot* s = acquire(n, dev);
        if (!s) return svx_fail(SVX_E_HIP, "no page-locked bounce buffer (hipHostMalloc)", __FILE__, __LINE__, hipSuccess);
        memcpy(s->p, src, n);
        hipError_t e = hipMemcpyAsync(dst, s->p, n, hipMemcpyHostToDevice, st_);
        if (e == hipSuccess) e = hipEventRecord(s->ev, st_);
        release(s, e == hipSuccess);
        if (e != hipSuccess) return svx_fail(SVX_E_HIP, "host -> device copy through a bounce buffer", __FILE__, __LINE__, e);
        src += n; dst += n; bytes -= n;
    }
    return SVX_OK;
}

// hand out the oldest pending device -> host pieces until at most `keep_big` BIG ones and `keep_all` pieces altogether are left in flight
int HostCopy::drain(size_t keep_big, size_t keep_all) {
    size_t k = 0;
    while (k < pend_.size() && (pend_big_ > keep_big || pend_.size() - k > keep_all)) {
        Pending& q = pend_[k];
        const hipError_t e = hipEventSynchronize(q.slot->ev);
        if (e != hipSuccess) { pend_.erase(pend_.begin(), pend_.begin() + (long)k); return svx_fail(SVX_E_HIP, "device -> host copy through a bounce buffer", __FILE__, __LINE__, e); }
        memcpy(q.dst, q.slot->p, q.bytes);
        if (q.slot->big) pend_big_--;
        release(q.slot, false);
        k++;
    }
    pend_.erase(pend_.begin(), pend_.begin() + (long)k);
    return SVX_OK;
}

int HostCopy::d2h(void* host_dst, const void* dev_src, size_t bytes) {
    if (!bytes) return SVX_OK;
    char* dst = (char*)host_dst; const char* src = (const char*)dev_src;
    const int dev = device_of(st_);
    while (bytes) {
        const size_t n = bytes < BIG_CAP ? bytes : BIG_CAP;
        if (n > SMALL_CAP && pend_big_ >= 2) SVXCHK(drain(1, (size_t)-1));              // two big pieces in flight: the copy of one overlaps the memcpy of the other
        if (pend_.size() >= 192) SVXCHK(drain((size_t)-1, 96));                        // (a call with very many small arrays must not hold the whole pool)
        BounceSlot* s = acquire(n, dev);
        if (!s) return svx_fail(SVX_E_HIP, "no page-locked bounce buffer (hipHostMalloc)", __FILE__, __LINE__, hipSuccess);
        hipError_t e = hipMemcpyAsync(s->p, src, n, hipMemcpyDeviceToHost, st_);
        if (e == hipSuccess) e = hipEventRecord(s->ev, st_);
        if (e != hipSuccess) { release(s, false); return svx_fail(SVX_E_HIP, "device -> host copy through a bounce buffer", __FILE__, __LINE__, e); }
        pend_.push_back(Pending{s, dst, n});
        if (s->big) pend_big_++;
        src += n; dst += n; bytes -= n;
    }
    return SVX_OK;
}

int HostCopy::out(void* dst, const void* dev_src, size_t bytes) {
    if (!bytes) return SVX_OK;
    if (svx_is_device_pointer(dst)) { HIPCHK(hipMemcpyAsync(dst, dev_src, bytes, hipMemcpyDeviceToDevice, st_)); return SVX_OK; }
    return d2h(dst, dev_src, bytes);
}

int HostCopy::finish() {
    if (pend_.empty() && !direct_pending_) return SVX_OK;
    const hipError_t e = hipStreamSynchronize(st_);
    direct_pending_ = false;
    if (e != hipSuccess) return svx_fail(SVX_E_HIP, "device -> host copies through bounce buffers", __FILE__, __LINE__, e);
    for (auto& q : pend_) { memcpy(q.dst, q.slot->p, q.bytes); release(q.slot, false); }
    pend_.clear(); pend_big_ = 0;
    return SVX_OK;
}

int svx_h2d(void* dev_dst, const void* host_src, size_t bytes, hipStream_t st) { HostCopy hc(st); SVXCHK(hc.h2d(dev_dst, host_src, bytes)); return hc.finish(); }
int svx_d2h(void* host_dst, const void* dev_src, size_t bytes, hipStream_t st) { HostCopy hc(st); SVXCHK(hc.d2h(host_dst, dev_src, bytes)); return hc.finish(); }
