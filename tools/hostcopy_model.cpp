// hostcopy_model.cpp - svim_amd/csrc/hostcopy.hip (the bounce-buffer copy layer every host <-> device copy of the library goes through since round 5) on the CPU,
// over a model of the dozen HIP calls it makes in which copies are REALLY asynchronous: hipMemcpyAsync only queues the copy on its stream; the bytes move when the
// stream is drained - by hipStreamSynchronize, by hipEventSynchronize on an event recorded behind them, or a little at a time by hipEventQuery (which first answers
// "not ready" a few times, like a copy in flight).  So a bounce slot that were handed out again before the copy out of it has run, or a destination handed to the
// caller before its bytes have arrived, shows up as wrong data; a slot written by two threads at once as a data race (ThreadSanitizer) or a heap error
// (AddressSanitizer).  "Device memory" is plain host memory here.
//
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -x c++ svim_amd/csrc/hostcopy.hip tools/hostcopy_model.cpp -lpthread -o hostcopy_model
//   (or -fsanitize=thread);  ./hostcopy_model [threads] [rounds]
// tests/test_host_cpu.py::test_bounce_buffer_copy_layer_over_an_asynchronous_model builds and runs it (both sanitizer sets).
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>
#include "../include/svx.h"
#include "../svim_amd/csrc/hostcopy.hpp"

// ---- what hostcopy.hip needs from the library -----------------------------------------------------------------------------------------------------------
thread_local std::string g_svx_err;
int svx_fail(int code, const char* what, const char* file, int line, hipError_t e) { char b[400]; snprintf(b, sizeof b, "%s at %s:%d (%d)", what, file, line, (int)e); g_svx_err = b; return code; }
bool svx_guard_mode() { return false; }
bool svx_guard_slack() { return false; }
int svx_guard_alloc(void**, size_t) { return SVX_E_HIP; }
void svx_guard_free(void*) {}

// ---- the model of the HIP runtime -----------------------------------------------------------------------------------------------------------------------
namespace {
struct Op { int kind; void* dst; const void* src; size_t n; long long ev_id; };          // kind 0: copy, 1: event marker
struct Stream { std::mutex m; std::deque<Op> q; };
struct Event { std::atomic<long long> recorded{0}, done{0}; std::atomic<int> polls{0}; Stream* on = nullptr; };
std::mutex g_reg;
std::map<hipStream_t, Stream*> g_streams;
Stream g_null;
std::atomic<long long> g_copied{0}, g_pinned_allocs{0};
Stream* stream_of(hipStream_t s) { if (!s) return &g_null; std::lock_guard<std::mutex> g(g_reg); auto it = g_streams.find(s); return it == g_streams.end() ? &g_null : it->second; }
// run the queue of a stream up to (and including) the marker of `until` (0: everything); at most `budget` operations
void drain(Stream* st, Event* until, long long until_id, int budget = 1 << 30) {
    std::lock_guard<std::mutex> g(st->m);
    while (!st->q.empty() && budget-- > 0) {
        Op op = st->q.front(); st->q.pop_front();
        if (op.kind == 0) { memcpy(op.dst, op.src, op.n); g_copied += (long long)op.n; }
        else { Event* e = (Event*)op.dst; e->done.store(op.ev_id); if (e == until && op.ev_id >= until_id) return; }
    }
}
}  // namespace

extern "C" {
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipStreamGetDevice(hipStream_t, hipDevice_t* d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "model"; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned int) { *p = malloc(n); g_pinned_allocs++; return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t) new Event(); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t ev, hipStream_t s) {
    Event* e = (Event*)ev; Stream* st = stream_of(s);
    const long long id = e->recorded.fetch_add(1) + 1;
    e->on = st; e->polls.store(0);
    std::lock_guard<std::mutex> g(st->m);
    st->q.push_back(Op{1, e, nullptr, 0, id});
    return hipSuccess;
}
hipError_t hipEventQuery(hipEvent_t ev) {
    Event* e = (Event*)ev;
    if (e->done.load() >= e->recorded.load()) return hipSuccess;
    if (e->polls.fetch_add(1) < 2) return hipErrorNotReady;            // still in flight the first times somebody asks
    if (e->on) drain(e->on, nullptr, 0, 3);                             // the "hardware" makes a little progress
    return e->done.load() >= e->recorded.load() ? hipSuccess : hipErrorNotReady;
}
hipError_t hipEventSynchronize(hipEvent_t ev) { Event* e = (Event*)ev; const long long want = e->recorded.load(); while (e->done.load() < want && e->on) drain(e->on, e, want); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { drain(stream_of(s), nullptr, 0); return hipSuccess; }
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t s) {
    Stream* st = stream_of(s);
    std::lock_guard<std::mutex> g(st->m);
    st->q.push_back(Op{0, dst, src, n, 0});
    return hipSuccess;
}
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void*) { memset(a, 0, sizeof *a); return hipErrorInvalidValue; }      // everything the tests hand in is "pageable host memory"
}

static hipStream_t make_stream() { Stream* s = new Stream(); hipStream_t h = (hipStream_t)s; std::lock_guard<std::mutex> g(g_reg); g_streams[h] = s; return h; }

static void fill(std::vector<uint8_t>& v, uint32_t seed) { uint32_t x = seed * 2654435761u + 1; for (auto& b : v) { x = x * 1664525u + 1013904223u; b = (uint8_t)(x >> 24); } }

int main(int argc, char** argv) {
    const int n_threads = argc > 1 ? atoi(argv[1]) : 4, rounds = argc > 2 ? atoi(argv[2]) : 60;
    std::atomic<int> bad{0};
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++) th.emplace_back([&, t]() {
        std::mt19937 rng(1000 + t);
        hipStream_t st = make_stream();
        auto pick = [&]() -> size_t {                                     // sizes around every boundary of the layer: 0, tiny, 64 KiB, 8 MiB pieces, the 64 MB thread split
            static const size_t s[] = {0, 1, 7, 4096, 65535, 65536, 65537, 300000, (8u << 20) - 1, 8u << 20, (8u << 20) + 1, 20000000, 70000000};
            const size_t base = s[rng() % (sizeof s / sizeof s[0])];
            return base > 100 && rng() % 2 ? base - rng() % 97 : base;
        };
        for (int r = 0; r < rounds && !bad; r++) {
            // a "call" of the library: several uploads, then several fetches of what was uploaded, one finish
            const int k = 1 + (int)(rng() % 5);
            std::vector<std::vector<uint8_t>> host((size_t)k), dev((size_t)k), back((size_t)k);
            HostCopy hc(st);
            for (int i = 0; i < k; i++) {
                size_t n = pick(); if (r % 7 && n > 30000000) n = 1 + n % 3000000;      // the 70 MB case now and then only
                host[(size_t)i].resize(n); dev[(size_t)i].assign(n + 8, 0xEE); back[(size_t)i].assign(n + 8, 0xDD);
                fill(host[(size_t)i], (uint32_t)(t * 100000 + r * 10 + i));
                if (hc.h2d(dev[(size_t)i].data(), host[(size_t)i].data(), n) != SVX_OK) { bad++; fprintf(stderr, "h2d failed: %s\n", g_svx_err.c_str()); return; }
                std::vector<uint8_t> keep = host[(size_t)i];
                if (n) memset(host[(size_t)i].data(), 0x11, n);          // the caller's array is its own again as soon as h2d returns
                host[(size_t)i].swap(keep);
            }
            for (int i = 0; i < k; i++) {
                const size_t n = host[(size_t)i].size();
                const int rc = (i & 1) ? hc.out(back[(size_t)i].data(), dev[(size_t)i].data(), n) : hc.d2h(back[(size_t)i].data(), dev[(size_t)i].data(), n);
                if (rc != SVX_OK) { bad++; fprintf(stderr, "d2h failed: %s\n", g_svx_err.c_str()); return; }
            }
            if (hc.finish() != SVX_OK) { bad++; return; }
            for (int i = 0; i < k; i++) {
                const size_t n = host[(size_t)i].size();
                if (n && memcmp(back[(size_t)i].data(), host[(size_t)i].data(), n) != 0) { bad++; fprintf(stderr, "thread %d round %d array %d (%zu bytes): fetched bytes differ from the uploaded ones\n", t, r, i, n); return; }
                for (int g = 0; g < 8; g++) if (back[(size_t)i][n + (size_t)g] != 0xDD || dev[(size_t)i][n + (size_t)g] != 0xEE) { bad++; fprintf(stderr, "a copy wrote beyond its %zu bytes\n", n); return; }
            }
            // the one-shot forms, and a HostCopy that is dropped with copies pending (an error path of a caller): its slots must come back
            std::vector<uint8_t> a(pick() % 5000000 + 1), d(a.size()), b2(a.size());
            fill(a, (uint32_t)(7 + r));
            if (svx_h2d(d.data(), a.data(), a.size(), st) != SVX_OK || svx_d2h(b2.data(), d.data(), a.size(), st) != SVX_OK || a != b2) { bad++; fprintf(stderr, "one-shot round trip failed\n"); return; }
            { HostCopy drop(st); std::vector<uint8_t> sink(100000); (void)drop.d2h(sink.data(), d.data(), sink.size() < d.size() ? sink.size() : d.size()); }      // (no wait: the copy into the slot is still queued)
            (void)hipStreamSynchronize(st);                               // the "device arrays" of this round die below: like hipFree, wait for what still reads them
        }
    });
    for (auto& x : th) x.join();
    printf("hostcopy model: %d threads x %d rounds, %lld MB moved by the modelled copies, %lld pinned allocations, %s\n", n_threads, rounds, g_copied.load() >> 20, g_pinned_allocs.load(),
           bad ? "FAILED" : "every byte arrived, no slot reused early");
    return bad ? 1 : 0;
}
