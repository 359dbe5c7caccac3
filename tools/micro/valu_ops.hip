// Per-instruction VALU throughput on gfx950 (cycles per wave64 instruction per SIMD at 8 waves/SIMD, 8 independent register chains).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define DEF_KERNEL(NAME, ASM)                                                                       \
    __global__ void NAME(uint32_t* out, int iters, uint32_t seed) {                                  \
        uint32_t x0 = seed + threadIdx.x, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7, x4 = x0 * 11, x5 = x0 * 13, x6 = x0 * 17, x7 = x0 * 19; \
        uint32_t y = seed * 31 + threadIdx.x, z = seed ^ 0x55aa55aau;                                   \
        for (int i = 0; i < iters; i++) {                                                            \
            for (int r = 0; r < 8; r++) {                                                            \
                asm volatile(ASM("%0") ASM("%1") ASM("%2") ASM("%3") ASM("%4") ASM("%5") ASM("%6") ASM("%7") \
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y), "v"(z) : "vcc"); \
            }                                                                                        \
        }                                                                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;         \
    }

#define A_XOR(R) "v_xor_b32 " R ", " R ", %8\n"
#define A_ANDOR(R) "v_and_or_b32 " R ", " R ", %8, %9\n"
#define A_BITOP3(R) "v_bitop3_b32 " R ", " R ", %8, %9 bitop3:0x96\n"
#define A_ALIGN(R) "v_alignbit_b32 " R ", " R ", %8, 1\n"
#define A_ADDU(R) "v_add_u32 " R ", " R ", %8\n"
#define A_ADDCO(R) "v_add_co_u32 " R ", vcc, " R ", %8\n"
#define A_ADDC(R) "v_addc_co_u32 " R ", vcc, " R ", %8, vcc\n"
#define A_LSHR(R) "v_lshrrev_b32 " R ", 1, " R "\n"
#define A_MOV(R) "v_mov_b32 " R ", %8\n"
#define A_XAD(R) "v_xad_u32 " R ", " R ", %8, %9\n"
#define A_OR3(R) "v_or3_b32 " R ", " R ", %8, %9\n"
#define A_LSHLOR(R) "v_lshl_or_b32 " R ", " R ", 1, %9\n"
#define A_CNDMASK(R) "v_cndmask_b32 " R ", " R ", %8, vcc\n"
#define A_BFI(R) "v_bfi_b32 " R ", " R ", %8, %9\n"
#define A_FMA(R) "v_fma_f32 " R ", " R ", %8, %9\n"
#define A_PKADD(R) "v_pk_add_u16 " R ", " R ", %8\n"

DEF_KERNEL(k_xor, A_XOR)
DEF_KERNEL(k_andor, A_ANDOR)
DEF_KERNEL(k_bitop3, A_BITOP3)
DEF_KERNEL(k_align, A_ALIGN)
DEF_KERNEL(k_addu, A_ADDU)
DEF_KERNEL(k_addco, A_ADDCO)
DEF_KERNEL(k_addc, A_ADDC)
DEF_KERNEL(k_lshr, A_LSHR)
DEF_KERNEL(k_mov, A_MOV)
DEF_KERNEL(k_xad, A_XAD)
DEF_KERNEL(k_or3, A_OR3)
DEF_KERNEL(k_lshlor, A_LSHLOR)
DEF_KERNEL(k_cndmask, A_CNDMASK)
DEF_KERNEL(k_bfi, A_BFI)
DEF_KERNEL(k_fma, A_FMA)
DEF_KERNEL(k_pkadd, A_PKADD)

typedef void (*kern_t)(uint32_t*, int, uint32_t);

int main() {
    uint32_t* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    const int iters = 4000;
    struct { const char* name; kern_t k; } ks[] = {{"v_xor_b32", k_xor}, {"v_and_or_b32", k_andor}, {"v_bitop3_b32", k_bitop3}, {"v_alignbit_b32", k_align},
        {"v_add_u32", k_addu}, {"v_add_co_u32", k_addco}, {"v_addc_co_u32", k_addc}, {"v_lshrrev_b32", k_lshr}, {"v_mov_b32", k_mov}, {"v_xad_u32", k_xad},
        {"v_or3_b32", k_or3}, {"v_lshl_or_b32", k_lshlor}, {"v_cndmask_b32", k_cndmask}, {"v_bfi_b32", k_bfi}, {"v_fma_f32", k_fma}, {"v_pk_add_u16", k_pkadd}};
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const double ghz = prop.clockRate / 1e6;
    printf("clockRate %.3f GHz, CUs %d\n", ghz, prop.multiProcessorCount);
    for (int wps : {1, 2, 8}) {
        for (auto& e : ks) {
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            const int blocks = prop.multiProcessorCount * wps;
            e.k<<<blocks, 256>>>(out, iters, 1); (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0); e.k<<<blocks, 256>>>(out, iters, 1); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_simd = (double)wps * iters * 64.0;
            printf("waves/SIMD %d  %-16s %7.3f ms  %.2f cycles/instr/SIMD (at %.2f GHz)\n", wps, e.name, ms, ms * 1e-3 * ghz * 1e9 / instr_per_simd, ghz);
        }
    }
    return 0;
}
