// Stand-alone rate of the edit-distance column update (svim_amd/csrc/myers_column.hpp): Q state words per lane, no memory traffic,
// 1..4 waves per SIMD.  Prints shader cycles per column and per word-column; compare with the issue bound 9.4*2 + 3*4 = 30.8 cycles.
// Build: hipcc --offload-arch=gfx950 -O3 -I svim_amd/csrc -o column_rate.bin tools/micro/column_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "myers_column.hpp"

template <int Q, int P>
__global__ __launch_bounds__(256) void k_columns(uint32_t* out, int n_cols, uint32_t seed) {
    uint32_t pv[Q], mv[Q], pl[P][Q];
#pragma unroll
    for (int q = 0; q < Q; q++) {
        pv[q] = ~0u; mv[q] = 0u;
#pragma unroll
        for (int b = 0; b < P; b++) pl[b][q] = seed * (q * 7 + b * 3 + 1) + threadIdx.x * 2654435761u;
    }
    uint32_t x = seed + threadIdx.x;
    int score = 0;
    for (int j = 0; j < n_cols; j++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t c = x >> 30;
        uint32_t nk[P];
#pragma unroll
        for (int b = 0; b < P; b++) nk[b] = ((c >> b) & 1u) - 1u;
        unsigned carry = 0;
        uint32_t ph_prev = 0x80000000u, mh_prev = 0u;
        MYERS_COLUMN(Q, P, pl, pv, mv, nk, carry, ph_prev, mh_prev)
        score += (int)(ph_prev >> 31) - (int)(mh_prev >> 31);
    }
    uint32_t s = (uint32_t)score;
#pragma unroll
    for (int q = 0; q < Q; q++) s ^= pv[q] ^ mv[q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int Q>
static void run(uint32_t* out) {
    const int n_cols = 20000;
    for (int wps : {1, 2, 3, 4}) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const int blocks = 256 * wps;
        k_columns<Q, 2><<<blocks, 256>>>(out, n_cols, 1); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); k_columns<Q, 2><<<blocks, 256>>>(out, n_cols, 1); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms * 1e-3 * 2.4e9 / ((double)wps * n_cols);          // SIMD cycles per column of one wave
        printf("Q=%2d waves/SIMD %d: %.3f ms  %.0f cycles/column  %.1f cycles/word-column (at 2.4 GHz)\n", Q, wps, ms, cyc, cyc / Q);
    }
}

int main() {
    uint32_t* out; (void)hipMalloc(&out, 256 * 4 * 256 * 4);
    run<16>(out); run<8>(out); run<4>(out);
    return 0;
}
