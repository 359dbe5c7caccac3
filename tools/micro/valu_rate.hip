// VALU issue-rate probe for gfx950: dependent vs independent 32-bit integer chains at 1..8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int CHAINS>
__global__ void k_chain(uint32_t* out, int iters, uint32_t seed) {
    uint32_t x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) x[c] = seed + threadIdx.x * 7 + c;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) x[c] = (x[c] ^ (x[c] >> 3)) + 0x9e3779b9u;      // 2 dependent VALU ops (v_lshrrev+v_xor fused? -> check ISA) per chain per r
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s ^= x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// carry chain as in the bit-vector kernels: 16 words, v_addc_co serial through VCC, plus independent filler per word
__global__ void k_carry(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a[16], b[16];
#pragma unroll
    for (int q = 0; q < 16; q++) { a[q] = seed * (q + 1) + threadIdx.x; b[q] = seed ^ (q * 77 + threadIdx.x); }
    for (int i = 0; i < iters; i++) {
        unsigned carry = 0;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            unsigned co;
            const uint32_t s = __builtin_addc(a[q] & b[q], b[q], carry, &co);
            carry = co;
            a[q] = s ^ b[q];
            b[q] = (b[q] >> 1) | (a[q] << 31);
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) s ^= a[q] ^ b[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static float time_ms(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
    uint32_t* out; hipMalloc(&out, 256 * 8 * 256 * 4 * 4);
    const int iters = 20000;
    const int ncu = 256;
    for (int wps = 1; wps <= 8; wps *= 2) {                 // waves per SIMD: blocks of 256 threads = 1 wave on each of the 4 SIMDs
        const int blocks = ncu * wps;
        float m1 = time_ms([&] { k_chain<1><<<blocks, 256>>>(out, iters, 1); });
        float m4 = time_ms([&] { k_chain<4><<<blocks, 256>>>(out, iters, 1); });
        float m8 = time_ms([&] { k_chain<8><<<blocks, 256>>>(out, iters / 2, 1); });
        float mc = time_ms([&] { k_carry<<<blocks, 256>>>(out, iters, 1); });
        // lane-ops per second: blocks*256 threads * iters*16*CHAINS*OPS
        printf("waves/SIMD %d: chain1 %.2f ms, chain4 %.2f ms, chain8(half iters) %.2f ms, carry16 %.2f ms\n", wps, m1, m4, m8, mc);
    }
    return 0;
}
