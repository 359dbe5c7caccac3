// Rate of the edit-distance column update (svim_amd/csrc/myers_column.hpp) in REAL shader cycles: every wave reads s_memtime (shader clock) and
// s_memrealtime (constant 100 MHz) around its loop, so the clock the chip actually sustains under this load and the cycles per word-column
// are both measured, at 1..8 resident waves per SIMD (as far as the kernel's registers allow).
// Build: hipcc --offload-arch=gfx950 -O3 -I svim_amd/csrc -o tools/micro/column_clock.bin tools/micro/column_clock.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "myers_column.hpp"

template <int Q, int P>
__global__ __launch_bounds__(256) void k_columns(uint32_t* out, unsigned long long* clk, int n_cols, uint32_t seed) {
    uint32_t pv[Q], mv[Q], pl[P][Q];
#pragma unroll
    for (int q = 0; q < Q; q++) {
        pv[q] = ~0u; mv[q] = 0u;
#pragma unroll
        for (int b = 0; b < P; b++) pl[b][q] = seed * (q * 7 + b * 3 + 1) + threadIdx.x * 2654435761u;
    }
    uint32_t x = seed + threadIdx.x;
    int score = 0;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int j = 0; j < n_cols; j += 8) {
        x = x * 1664525u + 1013904223u;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint32_t nk[P];
#pragma unroll
            for (int b = 0; b < P; b++) nk[b] = (uint32_t)__builtin_amdgcn_sbfe((int)x, 2 * k + b, 1);
            unsigned carry = 0;
            uint32_t ph_prev = 0x80000000u, mh_prev = 0u;
            MYERS_COLUMN(Q, P, pl, pv, mv, nk, carry, ph_prev, mh_prev)
            score += (int)(ph_prev >> 31) - (int)(mh_prev >> 31);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t s = (uint32_t)score;
#pragma unroll
    for (int q = 0; q < Q; q++) s ^= pv[q] ^ mv[q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); clk[2 * w] = t1 - t0; clk[2 * w + 1] = r1 - r0; }
}

template <int Q>
static void run(uint32_t* out, unsigned long long* clk, unsigned long long* h) {
    const int n_cols = 16000;
    int maxb = 0; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&maxb, k_columns<Q, 2>, 256, 0);
    hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_columns<Q, 2>));
    printf("Q=%d: %d VGPRs, at most %d waves/SIMD\n", Q, fa.numRegs, maxb);
    for (int wps = 1; wps <= maxb && wps <= 8; wps++) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const int blocks = 256 * wps;
        k_columns<Q, 2><<<blocks, 256>>>(out, clk, n_cols, 1); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); k_columns<Q, 2><<<blocks, 256>>>(out, clk, n_cols, 1); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(h, clk, (size_t)blocks * 4 * 16, hipMemcpyDeviceToHost);
        double sc = 0, rc = 0;
        for (int w = 0; w < blocks * 4; w++) { sc += (double)h[2 * w]; rc += (double)h[2 * w + 1]; }
        sc /= blocks * 4; rc /= blocks * 4;
        const double ghz = sc / (rc / 0.1);                                   // shader cycles per ns (s_memrealtime: 100 MHz)
        const double cyc_wall = ms * 1e-3 * ghz * 1e9 / ((double)wps * n_cols);      // SIMD cycles per column of one wave at the measured clock
        printf("Q=%2d waves/SIMD %d: %.3f ms, shader clock %.3f GHz (memtime/memrealtime), wave loop %.0f shader cycles = %.1f per column;  per SIMD: %.1f cycles/column = %.2f per word-column\n",
               Q, wps, ms, ghz, sc, sc / n_cols, cyc_wall, cyc_wall / Q);
    }
}

int main() {
    uint32_t* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    unsigned long long* clk; (void)hipMalloc(&clk, 256 * 8 * 4 * 16);
    unsigned long long* h = (unsigned long long*)malloc(256 * 8 * 4 * 16);
    run<4>(out, clk, h); run<8>(out, clk, h); run<12>(out, clk, h); run<16>(out, clk, h);
    return 0;
}
