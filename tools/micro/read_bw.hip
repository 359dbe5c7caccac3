// read_bw.hip - what a pure streaming READ reaches on this box: persistent waves, 16 B per lane and load (1 KiB per wave-instruction),
// U loads in flight per lane, B resident 256-thread blocks per CU.  The ceiling k_cigar_scan (collect.hip) is measured against.
//   hipcc --offload-arch=gfx950 -O3 -o read_bw.bin read_bw.hip && ./read_bw.bin [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int U>
__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ p, size_t n_vec, unsigned* out) {
    const size_t stride = (size_t)gridDim.x * 256 * U;
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n_vec; i += stride) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const size_t k = i + (size_t)u * 256; v[u] = k < n_vec ? p[k] : make_uint4(0, 0, 0, 0); }
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;            // keeps the loads alive
}

template <int U> static void run(const uint4* p, size_t n_vec, unsigned* out, int n_cu, int per_cu) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const unsigned grid = (unsigned)(n_cu * per_cu);
    k_read<U><<<grid, 256>>>(p, n_vec, out);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int it = 0; it < 5; it++) {
        CHECK(hipEventRecord(a));
        k_read<U><<<grid, 256>>>(p, n_vec, out);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    printf("loads in flight per lane %d, blocks/CU %d: %.3f ms  %.2f TB/s\n", U, per_cu, best, (double)n_vec * 16 / (best * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 5.7;
    const size_t n_vec = (size_t)(gib * (1ull << 30)) / 16;
    uint4* p; unsigned* out;
    CHECK(hipMalloc(&p, n_vec * 16)); CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(p, 1, n_vec * 16));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("%s, %d CUs, %.2f GiB\n", prop.name, n_cu, gib);
    for (int per_cu : {4, 6, 8}) { run<1>(p, n_vec, out, n_cu, per_cu); run<2>(p, n_vec, out, n_cu, per_cu); run<4>(p, n_vec, out, n_cu, per_cu); run<8>(p, n_vec, out, n_cu, per_cu); }
    return 0;
}
