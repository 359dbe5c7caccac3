// bgzf.hip - BGZF payloads inflated on the GPU: one 64-lane wavefront per block (inflate_core.hpp), blocks independent by construction of
// the format.  First piece of a device-resident BAM front-end: today the reader of bamio.cpp inflates on the host's cores,
// which bounds the end-to-end rate (section 7); svx_bgzf_inflate is the measured, zlib-identical replacement for that stage.
//
// Replaces: zlib inflate() under htslib's bgzf_read_block (the reference reads BAM through pysam.AlignmentFile, SVIM_COLLECT.py:132-137).
#include "common.hpp"
#include "inflate_core.hpp"

struct BgzfJob { unsigned long long in_off; unsigned long long out_off; uint32_t in_bytes; uint32_t out_bytes; };

__global__ __launch_bounds__(256) void k_bgzf_inflate(const uint8_t* comp, const BgzfJob* jobs, long long n_jobs, uint8_t* out, int* status) {
    __shared__ InfScratch scratch[4];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long j = (long long)blockIdx.x * 4 + wave;
    if (j >= n_jobs) return;
    const BgzfJob job = jobs[j];
    const int rc = inflate_raw(reinterpret_cast<const uint32_t*>(comp + job.in_off), job.in_bytes, out + job.out_off, job.out_bytes, scratch[wave]);
    if (lane_id() == 0 && rc != (int)job.out_bytes) { status[0] = 1; status[1] = (int)j; status[2] = rc; }
}

struct svx_inflater {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[2];
    DevBuf comp, out, jobs, status;
    void* staging = nullptr; size_t staging_cap = 0;      // pinned host memory the caller packs the payloads into
    float last_kernel_ms = 0;
};

extern "C" int svx_inflater_create(int device, svx_inflater** out) {
    if (!out) return svx_fail(SVX_E_ARG, "null argument", __FILE__, __LINE__, hipSuccess);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= device) return svx_fail(SVX_E_NODEVICE, "no such GPU (svx_inflater has no CPU fallback)", __FILE__, __LINE__, hipSuccess);
    HIPCHK(hipSetDevice(device));
    svx_inflater* f = new svx_inflater();
    f->device = device;
    HIPCHK(hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking));
    for (auto& e : f->ev) HIPCHK(hipEventCreate(&e));
    *out = f;
    return SVX_OK;
}

extern "C" void svx_inflater_destroy(svx_inflater* f) {
    if (!f) return;
    (void)hipSetDevice(f->device);
    (void)hipStreamSynchronize(f->stream);
    f->comp.release(); f->out.release(); f->jobs.release(); f->status.release();
    if (f->staging) (void)hipHostFree(f->staging);
    for (auto& e : f->ev) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(f->stream);
    delete f;
}

// pinned host buffer of at least `bytes` for the packed payloads (8-byte aligned start of every payload); valid until the next larger request
extern "C" void* svx_inflater_staging(svx_inflater* f, uint64_t bytes) {
    if (!f) return nullptr;
    if (bytes > f->staging_cap) {
        (void)hipSetDevice(f->device);
        if (f->staging) (void)hipHostFree(f->staging);
        f->staging = nullptr; f->staging_cap = 0;
        const size_t want = (size_t)bytes + (size_t)bytes / 4 + 4096;
        if (hipHostMalloc(&f->staging, want, hipHostMallocDefault) != hipSuccess) return nullptr;
        f->staging_cap = want;
    }
    return f->staging;
}

// n payloads packed in the staging buffer (in_off[i], 8-byte aligned, clen[i] bytes of raw DEFLATE) -> out_host + out_at[i] (isize[i] bytes each).
// out_on_device != 0: out_host is device memory (the inflated stream stays in HBM).  kernel_ms (optional): duration of the inflate launch.
extern "C" int svx_inflater_run(svx_inflater* f, int64_t n, const uint64_t* in_off, const uint32_t* clen, const uint32_t* isize, const uint64_t* out_at,
                                uint64_t staged_bytes, uint8_t* out_host, uint64_t out_bytes, int out_on_device, float* kernel_ms) {
    if (!f || n < 0 || (n && (!in_off || !clen || !isize || !out_at || !out_host))) return svx_fail(SVX_E_ARG, "null argument", __FILE__, __LINE__, hipSuccess);
    if (staged_bytes > f->staging_cap) return svx_fail(SVX_E_ARG, "payloads are not in the staging buffer", __FILE__, __LINE__, hipSuccess);
    if (kernel_ms) *kernel_ms = 0;
    if (n == 0) return SVX_OK;
    HIPCHK(hipSetDevice(f->device));
    std::vector<BgzfJob> jobs((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        if ((in_off[i] & 7ull) || in_off[i] + clen[i] > staged_bytes || out_at[i] + isize[i] > out_bytes)
            return svx_fail(SVX_E_ARG, "payload offset / size out of range", __FILE__, __LINE__, hipSuccess);
        jobs[(size_t)i] = BgzfJob{in_off[i], out_at[i], clen[i], isize[i]};
    }
    SVXCHK(f->comp.reserve((size_t)staged_bytes + 64));
    SVXCHK(f->jobs.reserve((size_t)n * sizeof(BgzfJob)));
    SVXCHK(f->status.reserve(16));
    uint8_t* out_dev = out_host;
    if (!out_on_device) { SVXCHK(f->out.reserve((size_t)out_bytes + 64)); out_dev = f->out.as<uint8_t>(); }
    hipStream_t st = f->stream;
    HIPCHK(hipMemcpyAsync(f->comp.p, f->staging, (size_t)staged_bytes, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(f->jobs.p, jobs.data(), (size_t)n * sizeof(BgzfJob), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(f->status.p, 0, 16, st));
    HIPCHK(hipEventRecord(f->ev[0], st));
    k_bgzf_inflate<<<(unsigned)((n + 3) / 4), 256, 0, st>>>(f->comp.as<uint8_t>(), f->jobs.as<BgzfJob>(), (long long)n, out_dev, f->status.as<int>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(f->ev[1], st));
    int status[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(status, f->status.p, 16, hipMemcpyDeviceToHost, st));
    if (!out_on_device) HIPCHK(hipMemcpyAsync(out_host, out_dev, (size_t)out_bytes, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, f->ev[0], f->ev[1]));
    f->last_kernel_ms = ms;
    if (kernel_ms) *kernel_ms = ms;
    if (status[0]) {
        char msg[128];
        snprintf(msg, sizeof msg, "BGZF inflate failed on block %d (code %d)", status[1], status[2]);
        return svx_fail(SVX_E_ARG, msg, __FILE__, __LINE__, hipSuccess);
    }
    return SVX_OK;
}
