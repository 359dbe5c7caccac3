// Dependent-issue probe: v_xor_b32 (full rate) and v_alignbit_b32 (half rate) with 1, 2, 4 independent chains per wave at 1..8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define K(NAME, BODY)                                                                             \
    __global__ void NAME(uint32_t* out, int iters, uint32_t seed) {                                \
        uint32_t x0 = seed + threadIdx.x, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7, y = seed * 31 + threadIdx.x; \
        for (int i = 0; i < iters; i++) {                                                          \
            for (int r = 0; r < 16; r++) { asm volatile(BODY : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y)); } \
        }                                                                                          \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3;                           \
    }
// 4 instructions per asm block in every variant
K(k_xor_c1, "v_xor_b32 %0, %0, %4\nv_xor_b32 %0, %0, %4\nv_xor_b32 %0, %0, %4\nv_xor_b32 %0, %0, %4\n")
K(k_xor_c2, "v_xor_b32 %0, %0, %4\nv_xor_b32 %1, %1, %4\nv_xor_b32 %0, %0, %4\nv_xor_b32 %1, %1, %4\n")
K(k_xor_c4, "v_xor_b32 %0, %0, %4\nv_xor_b32 %1, %1, %4\nv_xor_b32 %2, %2, %4\nv_xor_b32 %3, %3, %4\n")
K(k_ab_c1, "v_alignbit_b32 %0, %0, %4, 1\nv_alignbit_b32 %0, %0, %4, 1\nv_alignbit_b32 %0, %0, %4, 1\nv_alignbit_b32 %0, %0, %4, 1\n")
K(k_ab_c4, "v_alignbit_b32 %0, %0, %4, 1\nv_alignbit_b32 %1, %1, %4, 1\nv_alignbit_b32 %2, %2, %4, 1\nv_alignbit_b32 %3, %3, %4, 1\n")
K(k_mix_c1, "v_xor_b32 %0, %0, %4\nv_alignbit_b32 %0, %0, %4, 1\nv_xor_b32 %0, %0, %4\nv_bitop3_b32 %0, %0, %4, %4 bitop3:0x96\n")
K(k_mix_c4, "v_xor_b32 %0, %0, %4\nv_alignbit_b32 %1, %1, %4, 1\nv_xor_b32 %2, %2, %4\nv_bitop3_b32 %3, %3, %4, %4 bitop3:0x96\n")

typedef void (*kern_t)(uint32_t*, int, uint32_t);
int main() {
    uint32_t* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    const int iters = 4000;
    struct { const char* name; kern_t k; } ks[] = {{"xor  1 chain ", k_xor_c1}, {"xor  2 chains", k_xor_c2}, {"xor  4 chains", k_xor_c4}, {"abit 1 chain ", k_ab_c1},
        {"abit 4 chains", k_ab_c4}, {"mix  1 chain ", k_mix_c1}, {"mix  4 chains", k_mix_c4}};
    for (auto& e : ks) {
        printf("%s:", e.name);
        for (int wps : {1, 2, 3, 4, 5, 8}) {
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            const int blocks = 256 * wps;
            e.k<<<blocks, 256>>>(out, iters, 1); (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0); e.k<<<blocks, 256>>>(out, iters, 1); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("  w%d %.2f", wps, ms * 1e-3 * 2.4e9 / ((double)wps * iters * 64.0));
        }
        printf("   (cycles per instruction per SIMD at 2.4 GHz)\n");
    }
    return 0;
}
