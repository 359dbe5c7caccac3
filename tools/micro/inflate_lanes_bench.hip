// The lane-per-block inflater (svim_amd/csrc/inflate_lanes.hpp) alone, on the BGZF blocks of a file replicated to the launch size asked for: kernel time,
// GB/s of inflated output, and per wave (s_memtime): cycles, trips, cycles inside the batched block headers.  Probe builds (-DINFL_PROBE_...) switch parts of a
// trip off to see what a trip waits for (their output is wrong by construction: only the baseline build is compared with zlib).
// Build: hipcc --offload-arch=gfx950 -O3 -I svim_amd/csrc [-DINFL_PROBE_NOLIT] [-DINFL_PROBE_NOCOPY] [-DINFL_HDR_BATCH=n] -o tools/micro/inflate_lanes_bench.bin tools/micro/inflate_lanes_bench.hip -lz
// Run:   inflate_lanes_bench.bin <file.bam> [blocks per launch = 32768] [repetitions = 3]
#include <hip/hip_runtime.h>
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "inflate_lanes.hpp"

struct Job { unsigned long long in_off, out_off; uint32_t in_bytes, out_bytes; };
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

#define HDR_T0 const unsigned long long h0 = __builtin_readcyclecounter();
#define HDR_T1 t_hdr += __builtin_readcyclecounter() - h0; hdr_events++;
#define INFL_TRIP_LIMIT 600000u
static_assert(6 % INFL_DEPTH == 0, "the loop body holds six trips");
__global__ __launch_bounds__(64) void k_lanes(const uint8_t* comp, const Job* jobs, long long n_jobs, uint8_t* out, uint8_t* redo, unsigned long long* prof, uint32_t* lens_scratch) {
    __shared__ uint32_t lds[64 * INFL_STRIDE / 4];
    const int lane = (int)threadIdx.x;
    const long long j = (long long)blockIdx.x * 64 + lane;
    InflLane L;
    Job job{0ull, 0ull, 0u, 0u};
    if (j < n_jobs) job = jobs[j];
    infl_init(L, comp + job.in_off, job.in_bytes, out + job.out_off, job.out_bytes, reinterpret_cast<uint8_t*>(lds) + lane * INFL_STRIDE,
              lens_scratch + (size_t)blockIdx.x * (64 * INFL_LENS_WORDS) + lane, 64u);
    if (j >= n_jobs) L.state = INFL_ST_DONE;
    uint32_t trips = 0, hdr_events = 0;
    unsigned long long t_hdr = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
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
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (j < n_jobs) redo[j] = L.state == INFL_ST_DONE ? (uint8_t)0 : (uint8_t)1;
    if (lane == 0) { prof[4 * blockIdx.x] = t1 - t0; prof[4 * blockIdx.x + 1] = trips; prof[4 * blockIdx.x + 2] = t_hdr; prof[4 * blockIdx.x + 3] = hdr_events; }
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s file.bam [blocks per launch] [reps]\n", argv[0]); return 2; }
    const size_t want = argc > 2 ? (size_t)atoll(argv[2]) : 32768;
    const int reps = argc > 3 ? atoi(argv[3]) : 3;
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    std::vector<uint8_t> file;
    { static uint8_t buf[1 << 20]; size_t k; while ((k = fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + k); }
    fclose(f);
    std::vector<Job> base; size_t at = 0;
    while (at + 18 <= file.size()) {
        const uint8_t* h = file.data() + at;
        if (h[0] != 0x1f || h[1] != 0x8b) break;
        const unsigned xlen = h[10] | (h[11] << 8);
        unsigned bsize = 0;
        for (unsigned p = 12; p + 4 <= 12 + xlen;) { const unsigned sl = h[p + 2] | (h[p + 3] << 8); if (h[p] == 'B' && h[p + 1] == 'C') bsize = (h[p + 4] | (h[p + 5] << 8)) + 1u; p += 4 + sl; }
        if (!bsize) break;
        const uint8_t* tr = h + bsize - 8;
        const uint32_t isize = tr[4] | (tr[5] << 8) | (tr[6] << 16) | ((uint32_t)tr[7] << 24);
        if (isize) base.push_back(Job{(unsigned long long)(at + 12 + xlen), 0ull, (uint32_t)(bsize - 12 - xlen - 8), isize});
        at += bsize;
    }
    const size_t nb = base.size();
    if (!nb) { fprintf(stderr, "no blocks\n"); return 2; }
    // replicate the file (its own copy of the compressed bytes per replica: the input of a real chunk is not L2-resident either)
    const size_t copies = (want + nb - 1) / nb, n = want;
    std::vector<Job> jobs(n);
    unsigned long long out_bytes = 0, in_bytes = 0;
    for (size_t k = 0; k < n; k++) { jobs[k] = base[k % nb]; jobs[k].in_off += (unsigned long long)(k / nb) * file.size(); jobs[k].out_off = out_bytes; out_bytes += jobs[k].out_bytes; in_bytes += jobs[k].in_bytes; }
    uint8_t *d_comp, *d_out, *d_redo; Job* d_jobs; unsigned long long* d_prof;
    const size_t nwaves = (n + 63) / 64;
    CHK(hipMalloc(&d_comp, copies * file.size() + 256)); CHK(hipMalloc(&d_out, out_bytes + 256)); CHK(hipMalloc(&d_redo, n + 64)); CHK(hipMalloc(&d_jobs, n * sizeof(Job)));
    CHK(hipMalloc(&d_prof, nwaves * 32));
    uint32_t* d_lens; CHK(hipMalloc(&d_lens, nwaves * 64 * INFL_LENS_WORDS * 4));
    for (size_t c = 0; c < copies; c++) CHK(hipMemcpy(d_comp + c * file.size(), file.data(), file.size(), hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_jobs, jobs.data(), n * sizeof(Job), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        CHK(hipMemset(d_out, 0, out_bytes));
        CHK(hipEventRecord(e0));
        k_lanes<<<(unsigned)nwaves, 64>>>(d_comp, d_jobs, (long long)n, d_out, d_redo, d_prof, d_lens);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    std::vector<unsigned long long> prof(nwaves * 4);
    CHK(hipMemcpy(prof.data(), d_prof, nwaves * 32, hipMemcpyDeviceToHost));
    std::vector<uint8_t> redo(n); CHK(hipMemcpy(redo.data(), d_redo, n, hipMemcpyDeviceToHost));
    size_t n_redo = 0; for (auto v : redo) n_redo += v;
    double cyc = 0, trips = 0, hdr = 0, hev = 0, cmax = 0;
    for (size_t w = 0; w < nwaves; w++) { cyc += (double)prof[4 * w]; trips += (double)prof[4 * w + 1]; hdr += (double)prof[4 * w + 2]; hev += (double)prof[4 * w + 3]; if ((double)prof[4 * w] > cmax) cmax = (double)prof[4 * w]; }
    printf("%zu blocks (%zu waves), %.1f MB -> %.1f MB: %.2f ms = %.1f GB/s inflated; given up %zu\n", n, nwaves, in_bytes / 1e6, out_bytes / 1e6, best, out_bytes / best / 1e6, n_redo);
    printf("per wave: %.0f cycles (max %.0f), %.0f trips, %.0f cycles per trip incl. headers, headers %.1f %% of the cycles in %.1f batches of %.0f cycles\n",
           cyc / nwaves, cmax, trips / nwaves, cyc / trips, 100.0 * hdr / cyc, hev / nwaves, hev ? hdr / hev : 0.0);
#if !defined(INFL_PROBE_NOLIT) && !defined(INFL_PROBE_NOCOPY)
    {   // the first blocks against zlib
        const size_t chk = n < 300 ? n : 300;
        std::vector<uint8_t> got(jobs[chk - 1].out_off + jobs[chk - 1].out_bytes);
        CHK(hipMemcpy(got.data(), d_out, got.size(), hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t k = 0; k < chk; k++) {
            if (redo[k]) continue;
            std::vector<uint8_t> ex(jobs[k].out_bytes);
            z_stream zs; memset(&zs, 0, sizeof zs); inflateInit2(&zs, -15);
            zs.next_in = file.data() + base[k % nb].in_off; zs.avail_in = jobs[k].in_bytes; zs.next_out = ex.data(); zs.avail_out = jobs[k].out_bytes;
            inflate(&zs, Z_FINISH); inflateEnd(&zs);
            if (memcmp(ex.data(), got.data() + jobs[k].out_off, ex.size()) != 0) bad++;
        }
        printf("first %zu blocks against zlib: %zu differ\n", chk, bad);
        if (bad) return 1;
    }
#endif
    return 0;
}
