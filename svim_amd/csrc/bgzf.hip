// bgzf.hip - BGZF payloads inflated on the GPU: one 64-lane wavefront per block (inflate_core.hpp), blocks independent by construction of
// the format.  First piece of a device-resident BAM front-end: today the reader of bamio.cpp inflates on the host's cores,
// which bounds the end-to-end rate (section 7); svx_bgzf_inflate is the measured, zlib-identical replacement for that stage.
//
// Replaces: zlib inflate() under htslib's bgzf_read_block (the reference reads BAM through pysam.AlignmentFile, SVIM_COLLECT.py:132-137).
#include "common.hpp"
#include "hostcopy.hpp"
#include <atomic>
#include "inflate_core.hpp"
#include "inflate_lanes.hpp"
#include <mutex>
#include <thread>
#include <cstdlib>

struct BgzfJob { unsigned long long in_off; unsigned long long out_off; uint32_t in_bytes; uint32_t out_bytes; };

// one wavefront = one workgroup = one BGZF block (the waves share nothing, so they are scheduled one by one; ~8 KB of LDS each: 20 per CU).
// only_if != nullptr: just the blocks the lane-per-block decoder below has given up (stored blocks, code tables beyond its LDS budget, damaged streams)
__global__ __launch_bounds__(64) void k_bgzf_inflate(const uint8_t* comp, const BgzfJob* jobs, long long n_jobs, uint8_t* out, int* status, const uint8_t* only_if) {
    __shared__ InfScratch scratch;
    const long long j = (long long)blockIdx.x;
    if (j >= n_jobs) return;
    if (only_if) { if (!only_if[j]) return; if (lane_id() == 0) atomicAdd(&status[3], 1); }
    const BgzfJob job = jobs[j];
    const int rc = inflate_raw(comp + job.in_off, job.in_bytes, out + job.out_off, job.out_bytes, scratch);
    if (lane_id() == 0 && rc != (int)job.out_bytes) { status[0] = 1; status[1] = (int)j; status[2] = rc; }
}

// one LANE = one BGZF block (inflate_lanes.hpp): 64 blocks per wavefront, the lane's code tables in its own INFL_STRIDE bytes of LDS (636: four workgroups per CU, one per SIMD).
// A lane that reaches a block header parks until INFL_HDR_BATCH lanes of the wave wait at one (or nobody is decoding): the serial header code then runs for all of
// them at once.  redo[j] = 1: the block is left to k_bgzf_inflate.  The trip bound ends a wave whatever its input is (a sound block of 64 KiB takes ~45 k trips)
#define INFL_TRIP_BOUND 600000u
#define HDR_T0
#define HDR_T1
#define INFL_TRIP_LIMIT INFL_TRIP_BOUND
static_assert(6 % INFL_DEPTH == 0, "the loop body holds six trips");
__global__ __launch_bounds__(64) void k_bgzf_inflate_lanes(const uint8_t* comp, const BgzfJob* jobs, long long n_jobs, uint8_t* out, uint8_t* redo, uint32_t* lens_scratch) {
    __shared__ uint32_t lds[64 * INFL_STRIDE / 4];
    const int lane = (int)threadIdx.x;
    const long long j = (long long)blockIdx.x * 64 + lane;
    InflLane L;
    BgzfJob job{0ull, 0ull, 0u, 0u};
    if (j < n_jobs) job = jobs[j];
    infl_init(L, comp + job.in_off, job.in_bytes, out + job.out_off, job.out_bytes, reinterpret_cast<uint8_t*>(lds) + lane * INFL_STRIDE,
              lens_scratch + (size_t)blockIdx.x * (64 * INFL_LENS_WORDS) + lane, 64u);       // word k of the wave's lanes side by side
    if (j >= n_jobs) L.state = INFL_ST_DONE;
    uint32_t trips = 0;
    // the loop, unrolled INFL_DEPTH times (inflate_lanes.hpp: the chunk slot a trip stores from and loads into is a constant of the trip's position in the body)
#define INFL_TRIP(PAR_) \
        {                                                                                                   \
            infl_step<(PAR_) % INFL_DEPTH>(L);                                                              \
            if (__ballot(infl_running(L)) == 0ull) break;                                                   \
            if (++trips > INFL_TRIP_LIMIT) { if (L.state != INFL_ST_DONE) L.state = INFL_ST_FAIL; break; }  \
        }
    for (;;) {
        // block headers: once per six trips (ONE copy of the header code in the kernel), for the lanes that wait at one - when enough of them do, or nobody decodes
        const uint64_t hm = __ballot(L.state == INFL_ST_HEADER);
        if (hm && (__popcll(hm) >= INFL_HDR_BATCH || __ballot(L.state == INFL_ST_DECODE) == 0ull)) {
            HDR_T0 if (L.state == INFL_ST_HEADER) infl_header(L); HDR_T1
        }
        INFL_TRIP(0) INFL_TRIP(1) INFL_TRIP(2) INFL_TRIP(3) INFL_TRIP(4) INFL_TRIP(5)
    }
    if (j < n_jobs) redo[j] = L.state == INFL_ST_DONE ? (uint8_t)0 : (uint8_t)1;
}

#ifdef INF_PROFILE
// cycle accounting of the decode loop, summed over every block inflated since the last reset (experiment builds only; tools/inflate_profile.py)
extern "C" int svx_inflate_profile(unsigned long long* out16, int reset) {
    HIPCHK(hipDeviceSynchronize());
    if (out16) HIPCHK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_inf_prof), 16 * sizeof(unsigned long long)));
    if (reset) { unsigned long long z[16] = {0}; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_inf_prof), z, sizeof(z))); }
    return SVX_OK;
}
#endif

#define INF_SLOTS 8            /* sub-batches in flight (the readers use three with the wave-per-block decoder, six with the lane-per-block one) */
struct InflaterSlot {
    hipStream_t stream = nullptr;
    hipEvent_t ev[2];
    DevBuf comp, out, jobs, status, redo, lens;
    void* staging = nullptr; size_t staging_cap = 0;      // pinned host memory the caller packs this slot's payloads into
    int* host_status = nullptr;                           // 4 ints of the inflater's pinned block
    std::vector<BgzfJob> host_jobs;
    // inflated data on its way to a HOST window: device -> this slot's own page-locked buffer (a DMA), then a memcpy into the caller's window at the wait
    // (the window itself is pageable memory of the reader: the runtime must not page-lock it in place - hostcopy.hpp)
    void* out_stage = nullptr; size_t out_stage_cap = 0; uint8_t* out_host = nullptr; size_t out_host_bytes = 0;
    bool busy = false;
};
struct svx_inflater {
    int device = 0;
    InflaterSlot slot[INF_SLOTS];
    int* status_block = nullptr;                          // pinned: 4 ints per slot
};

extern "C" int svx_inflater_create(int device, svx_inflater** out) {
    if (!out) return svx_fail(SVX_E_ARG, "null argument", __FILE__, __LINE__, hipSuccess);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= device) return svx_fail(SVX_E_NODEVICE, "no such GPU (svx_inflater has no CPU fallback)", __FILE__, __LINE__, hipSuccess);
    HIPCHK(hipSetDevice(device));
    svx_inflater* f = new svx_inflater();
    f->device = device;
    { void* p = nullptr; HIPCHK(hipHostMalloc(&p, 256, hipHostMallocDefault)); f->status_block = (int*)p; memset(p, 0, 256); }
    for (int k = 0; k < INF_SLOTS; k++) {
        InflaterSlot& sl = f->slot[k];
        sl.host_status = f->status_block + 4 * k;
        HIPCHK(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
        for (auto& e : sl.ev) HIPCHK(hipEventCreate(&e));
    }
    *out = f;
    return SVX_OK;
}

// Nothing of the reader's memory is registered with the GPU (hipHostRegister): a registration describes an address range the library later frees or unmaps, and any
// trace of it that the runtime keeps beyond hipHostUnregister makes a later copy into whatever the allocator places there fault (hostcopy.hpp).  All data takes the
// library's own page-locked buffers.  (Round 5 kept the registrations of the reader's windows and of the memory-mapped file behind SVX_READER_REGISTER /
// SVX_BAM_DEV_MAPFILE; nothing tested them and the registered file measured no faster than the staged input, profiles/r05_device_reader_rate*.txt: removed in round 6.)
extern "C" void svx_inflater_destroy(svx_inflater* f) {
    if (!f) return;
    (void)hipSetDevice(f->device);
    for (auto& sl : f->slot) {
        (void)hipStreamSynchronize(sl.stream);
        sl.comp.release(); sl.out.release(); sl.jobs.release(); sl.status.release(); sl.redo.release(); sl.lens.release();
        if (sl.staging) (void)hipHostFree(sl.staging);
        if (sl.out_stage) (void)hipHostFree(sl.out_stage);
        for (auto& e : sl.ev) (void)hipEventDestroy(e);
        (void)hipStreamDestroy(sl.stream);
    }
    if (f->status_block) (void)hipHostFree(f->status_block);
    delete f;
}

// pinned host buffer of at least `bytes` for the payloads of one slot (a slice of the file as it is, or payloads packed one by one); valid until the next larger
// request for the same slot.  Three slots: while one sub-batch is inflated and copied back, the caller packs the next.
extern "C" void* svx_inflater_staging(svx_inflater* f, int slot, uint64_t bytes) {
    if (!f || slot < 0 || slot >= INF_SLOTS) return nullptr;
    InflaterSlot& sl = f->slot[slot];
    if (bytes > sl.staging_cap) {
        (void)hipSetDevice(f->device);
        if (sl.busy) { (void)svx_fail(SVX_E_STATE, "slot still busy: svx_inflater_wait before asking for a larger staging buffer", __FILE__, __LINE__, hipSuccess); return nullptr; }
        if (sl.staging) (void)hipHostFree(sl.staging);
        sl.staging = nullptr; sl.staging_cap = 0;
        const size_t want = (size_t)bytes + (size_t)bytes / 4 + 4096;
        if (hipHostMalloc(&sl.staging, want, hipHostMallocDefault) != hipSuccess) return nullptr;
        sl.staging_cap = want;
    }
    return sl.staging;
}

// n payloads in the slot's staging buffer (in_off[i], any alignment, clen[i] bytes of raw DEFLATE) -> out + out_at[i] (isize[i] bytes each),
// asynchronously on the slot's stream: H2D, inflate, copy back.  out_on_device != 0: `out` is device memory (the inflated stream stays in HBM).
extern "C" int svx_inflater_enqueue(svx_inflater* f, int slot, int64_t n, const uint64_t* in_off, const uint32_t* clen, const uint32_t* isize, const uint64_t* out_at,
                                    uint64_t staged_bytes, uint8_t* out, uint64_t out_bytes, int out_on_device) {
    if (!f || slot < 0 || slot >= INF_SLOTS || n < 0 || (n && (!in_off || !clen || !isize || !out_at || !out))) return svx_fail(SVX_E_ARG, "bad argument", __FILE__, __LINE__, hipSuccess);
    InflaterSlot& sl = f->slot[slot];
    if (sl.busy) return svx_fail(SVX_E_STATE, "slot still busy: svx_inflater_wait first", __FILE__, __LINE__, hipSuccess);
    if (staged_bytes > sl.staging_cap) return svx_fail(SVX_E_ARG, "payloads are not in the staging buffer", __FILE__, __LINE__, hipSuccess);
    if (n == 0) return SVX_OK;
    HIPCHK(hipSetDevice(f->device));
    std::vector<BgzfJob>& jobs = sl.host_jobs;
    jobs.resize((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        if (in_off[i] + clen[i] > staged_bytes || out_at[i] + isize[i] > out_bytes)
            return svx_fail(SVX_E_ARG, "payload offset / size out of range", __FILE__, __LINE__, hipSuccess);
        jobs[(size_t)i] = BgzfJob{in_off[i], out_at[i], clen[i], isize[i]};
    }
    // the compressed payloads: read by the kernel straight from the pinned staging buffer over PCIe (each byte is read once, 256 B at a time per wave and far
    // ahead of its use - and the copy engine's H2D would not overlap the kernel of the sub-batch before: it starts when that kernel ends), or copied to HBM first
    // (SVX_INFLATE_ZEROCOPY=0)
    // SVX_INFLATE_LANES=1: lane per block (round 6), the blocks it gives up redone by the wave-per-block decoder.  Measured and NOT the default: a lane's code tables
    // take 1276 bytes of LDS, so a CU holds two such waves - one wave on a SIMD issues a dependent instruction every ~4 cycles, a trip of ~365 instructions takes 1500
    // cycles before any memory wait and 2900 with the copy engine's far loads: 29 GB/s on the file with base qualities against 31-35 (profiles/r06_inflate_lanes_probe.txt)
    const bool lanes = []() { const char* e = getenv("SVX_INFLATE_LANES"); return e && e[0] == '1'; }();      // (read per call: the GPU tests run both decoders in one process)
    // the lane decoder reads 8 bytes per lane and load: its input has to be in HBM (a lane's loads over PCIe would fetch a line for every 8 bytes)
    static const bool zero_copy_env = []() { const char* e = getenv("SVX_INFLATE_ZEROCOPY"); return !(e && e[0] == '0'); }();
    const bool zero_copy = zero_copy_env && !lanes;
    const uint8_t* comp_dev = nullptr;
    if (zero_copy) { void* dp = nullptr; if (hipHostGetDevicePointer(&dp, sl.staging, 0) == hipSuccess) comp_dev = (const uint8_t*)dp; else (void)hipGetLastError(); }
    if (!comp_dev) SVXCHK(sl.comp.reserve((size_t)staged_bytes + 64));
    SVXCHK(sl.jobs.reserve((size_t)n * sizeof(BgzfJob)));
    SVXCHK(sl.status.reserve(16));
    uint8_t* out_dev = out;
    if (!out_on_device) { SVXCHK(sl.out.reserve((size_t)out_bytes + 64)); out_dev = sl.out.as<uint8_t>(); }
    hipStream_t st = sl.stream;
    // a host window receives the data through the slot's page-locked buffer
    if (!out_on_device && out_bytes > sl.out_stage_cap) {
        if (sl.out_stage) (void)hipHostFree(sl.out_stage);
        sl.out_stage = nullptr; sl.out_stage_cap = 0;
        const size_t want = (size_t)out_bytes + (size_t)out_bytes / 4 + 4096;
        HIPCHK(hipHostMalloc(&sl.out_stage, want, hipHostMallocDefault));
        sl.out_stage_cap = want;
    }
    if (!comp_dev) { HIPCHK(hipMemcpyAsync(sl.comp.p, sl.staging, (size_t)staged_bytes, hipMemcpyHostToDevice, st)); comp_dev = sl.comp.as<uint8_t>(); }
    SVXCHK(svx_h2d(sl.jobs.p, jobs.data(), (size_t)n * sizeof(BgzfJob), st));
    HIPCHK(hipMemsetAsync(sl.status.p, 0, 16, st));
    HIPCHK(hipEventRecord(sl.ev[0], st));
    static const unsigned lds_pad = []() { const char* e = getenv("SVX_INFLATE_LDS_PAD"); return e ? (unsigned)atoi(e) : 0u; }();      // experiment: fewer resident waves per CU
    if (lanes) {
        SVXCHK(sl.redo.reserve((size_t)n + 64));
        SVXCHK(sl.lens.reserve((size_t)((n + 63) / 64) * 64 * INFL_LENS_WORDS * 4));
        k_bgzf_inflate_lanes<<<(unsigned)((n + 63) / 64), 64, 0, st>>>(comp_dev, sl.jobs.as<BgzfJob>(), (long long)n, out_dev, sl.redo.as<uint8_t>(), sl.lens.as<uint32_t>());
        HIPCHK(hipGetLastError());
    }
    k_bgzf_inflate<<<(unsigned)n, 64, lds_pad, st>>>(comp_dev, sl.jobs.as<BgzfJob>(), (long long)n, out_dev, sl.status.as<int>(), lanes ? sl.redo.as<uint8_t>() : nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(sl.ev[1], st));
    HIPCHK(hipMemcpyAsync(sl.host_status, sl.status.p, 16, hipMemcpyDeviceToHost, st));
    sl.out_host = nullptr; sl.out_host_bytes = 0;
    if (!out_on_device) {
        HIPCHK(hipMemcpyAsync(sl.out_stage, out_dev, (size_t)out_bytes, hipMemcpyDeviceToHost, st)); sl.out_host = out; sl.out_host_bytes = (size_t)out_bytes;
    }
    sl.busy = true;
    return SVX_OK;
}

extern "C" int svx_inflater_wait(svx_inflater* f, int slot, float* kernel_ms) {
    if (!f || slot < 0 || slot >= INF_SLOTS) return svx_fail(SVX_E_ARG, "bad argument", __FILE__, __LINE__, hipSuccess);
    InflaterSlot& sl = f->slot[slot];
    if (kernel_ms) *kernel_ms = 0;
    if (!sl.busy) return SVX_OK;
    HIPCHK(hipSetDevice(f->device));
    HIPCHK(hipStreamSynchronize(sl.stream));
    sl.busy = false;
    if (sl.out_host && sl.out_host_bytes) {                // the inflated data of a host window: out of the slot's page-locked buffer, a few threads for large pieces
        const size_t nb = sl.out_host_bytes;
        const int parts = nb > ((size_t)8 << 20) ? 4 : 1;
        std::vector<std::thread> cp;
        uint8_t* dst = sl.out_host; const uint8_t* src = (const uint8_t*)sl.out_stage;
        for (int q = 1; q < parts; q++) { const size_t lo = nb * (size_t)q / parts, hi = nb * (size_t)(q + 1) / parts; cp.emplace_back([=]() { memcpy(dst + lo, src + lo, hi - lo); }); }
        memcpy(dst, src, nb / parts);
        for (auto& t : cp) t.join();
        sl.out_host = nullptr; sl.out_host_bytes = 0;
    }
    if (kernel_ms) { float ms = 0; HIPCHK(hipEventElapsedTime(&ms, sl.ev[0], sl.ev[1])); *kernel_ms = ms; }
    if (sl.host_status[0]) {
        char msg[128];
        snprintf(msg, sizeof msg, "BGZF inflate failed on block %d of the sub-batch (code %d)", sl.host_status[1], sl.host_status[2]);
        return svx_fail(SVX_E_ARG, msg, __FILE__, __LINE__, hipSuccess);
    }
    return SVX_OK;
}

// one sub-batch, synchronously (slot 0)
extern "C" int svx_inflater_run(svx_inflater* f, int64_t n, const uint64_t* in_off, const uint32_t* clen, const uint32_t* isize, const uint64_t* out_at,
                                uint64_t staged_bytes, uint8_t* out, uint64_t out_bytes, int out_on_device, float* kernel_ms) {
    SVXCHK(svx_inflater_enqueue(f, 0, n, in_off, clen, isize, out_at, staged_bytes, out, out_bytes, out_on_device));
    return svx_inflater_wait(f, 0, kernel_ms);
}
