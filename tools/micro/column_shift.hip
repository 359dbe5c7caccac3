// Candidate forms of the column update's SHIFT step (phs = (ph << 1) | bit 31 of the word below, same for mh) and of what follows it, each CORRECT
// (the checksums of all forms must agree) and timed like column_clock.hip.  v_alignbit measures 4.4 cycles alone (valu_banks) but ~8.5 inside the update.
//   S=0 shipped: 2 v_alignbit + v_bitop3 + v_and
//   S=1 v_lshrrev (carry bit) + v_add_u32 (x + x) + v_or, twice; then v_bitop3 + v_and                      (6 full-rate + 2)
//   S=2 carries folded into the two final three-operand results: 2 v_lshrrev, 2 v_add_u32, 2 v_or, 2 v_bitop3   (8 full-rate)
//   S=3 shifts on two more carry chains: v_addc_co (x + x + carry in, carry out = bit 31), then v_bitop3 + v_and
//   S=4 v_lshrrev + v_lshl_or, twice; then v_bitop3 + v_and
// Build: hipcc --offload-arch=gfx950 -O3 -I svim_amd/csrc -o tools/micro/column_shift.bin tools/micro/column_shift.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define BITOP3(a_, b_, c_, tt_) ((uint32_t)__builtin_amdgcn_bitop3_b32((int)(a_), (int)(b_), (int)(c_), (tt_)))
#define SB() __builtin_amdgcn_sched_barrier(0)
#define OPAQUE(x_) asm("" : "+v"(x_))

template <int Q, int P, int S>
__device__ __forceinline__ void column(uint32_t (&pl)[P][Q], uint32_t (&pv)[Q], uint32_t (&mv)[Q], const uint32_t (&nk)[P], unsigned& carry, uint32_t& ph_prev, uint32_t& mh_prev) {
    constexpr int GQ = Q >= 4 ? 4 : Q;
    unsigned cyp = ph_prev >> 31, cym = mh_prev >> 31;           // S=3: the shift chains' carries
#pragma unroll
    for (int q0 = 0; q0 < Q; q0 += GQ) {
        const int gn = Q - q0 < GQ ? Q - q0 : GQ;
        uint32_t eq_[GQ], xv_[GQ], sum_[GQ], ph_[GQ], mh_[GQ], phs_[GQ], mhs_[GQ];
#pragma unroll
        for (int g = 0; g < GQ; g++) if (g < gn) {
            uint32_t e = pl[0][q0 + g] ^ nk[0];
            e = BITOP3(e, pl[1][q0 + g], nk[1], 0x60);
            eq_[g] = e;
        }
#pragma unroll
        for (int g = 0; g < GQ; g++) if (g < gn) {
            xv_[g] = eq_[g] | mv[q0 + g]; sum_[g] = eq_[g] & pv[q0 + g];
            OPAQUE(xv_[g]);
        }
        SB();
#pragma unroll
        for (int g = 0; g < GQ; g++) if (g < gn) { unsigned carry_out; sum_[g] = __builtin_addc(sum_[g], pv[q0 + g], carry, &carry_out); carry = carry_out; }
#pragma unroll
        for (int g = 0; g < GQ; g++) if (g < gn) sum_[g] = BITOP3(sum_[g], pv[q0 + g], eq_[g], 0xBE);
#pragma unroll
        for (int g = 0; g < GQ; g++) if (g < gn) {
            ph_[g] = BITOP3(mv[q0 + g], sum_[g], pv[q0 + g], 0xF1);
            mh_[g] = pv[q0 + g] & sum_[g];
        }
        SB();
        if (S == 0) {
#pragma unroll
            for (int g = 0; g < GQ; g++) if (g < gn) {
                phs_[g] = __builtin_amdgcn_alignbit(ph_[g], g ? ph_[g - 1] : ph_prev, 31);
                mhs_[g] = __builtin_amdgcn_alignbit(mh_[g], g ? mh_[g - 1] : mh_prev, 31);
            }
            SB();
#pragma unroll
            for (int g = 0; g < GQ; g++) if (g < gn) { pv[q0 + g] = BITOP3(mhs_[g], xv_[g], phs_[g], 0xF1); mv[q0 + g] = phs_[g] & xv_[g]; }
        } else if (S == 1 || S == 4) {
#pragma unroll
            for (int g = 0; g < GQ; g++) if (g < gn) {
                uint32_t cp = (g ? ph_[g - 1] : ph_prev) >> 31, cm = (g ? mh_[g - 1] : mh_prev) >> 31;
                if (S == 1) {
                    uint32_t a, b;                                      // x + x: the compiler would make it v_lshlrev_b32, which is half rate (valu_banks)
                    asm("v_add_u32 %0, %1, %1" : "=v"(a) : "v"(ph_[g])); asm("v_add_u32 %0, %1, %1" : "=v"(b) : "v"(mh_[g]));
                    OPAQUE(cp); OPAQUE(cm);
                    phs_[g] = a | cp; mhs_[g] = b | cm;
                    OPAQUE(phs_[g]); OPAQUE(mhs_[g]);
                } else {
                    OPAQUE(cp); OPAQUE(cm);
                    asm("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(phs_[g]) : "v"(ph_[g]), "v"(cp));
                    asm("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(mhs_[g]) : "v"(mh_[g]), "v"(cm));
                }
            }
            SB();
#pragma unroll
            for (int g = 0; g < GQ; g++) if (g < gn) { pv[q0 + g] = BITOP3(mhs_[g], xv_[g], phs_[g], 0xF1); mv[q0 + g] = phs_[g] & xv_[g]; }
        } else if (S == 2) {
            uint32_t cp_[GQ], cm_[GQ], a_[GQ], b_[GQ];
#pragma unroll
            for (int g = 0; g < GQ; g++) if (g < gn) {
                cp_[g] = (g ? ph_[g - 1] : ph_prev) >> 31; cm_[g] = (g ? mh_[g - 1] : mh_prev) >> 31;
                asm("v_add_u32 %0, %1, %1" : "=v"(a_[g]) : "v"(ph_[g])); asm("v_add_u32 %0, %1, %1" : "=v"(b_[g]) : "v"(mh_[g]));
                OPAQUE(cp_[g]); OPAQUE(cm_[g]);
            }
#pragma unroll
            for (int g = 0; g < GQ; g++) if (g < gn) { uint32_t xvc = xv_[g] | cp_[g], t = b_[g] | cm_[g]; OPAQUE(xvc); OPAQUE(t); phs_[g] = xvc; mhs_[g] = t; }
            SB();
#pragma unroll
            for (int g = 0; g < GQ; g++) if (g < gn) {
                pv[q0 + g] = BITOP3(mhs_[g], a_[g], phs_[g], 0xF1);              // t | ~(a | xvc)
                mv[q0 + g] = BITOP3(a_[g], cp_[g], xv_[g], 0xA8);                 // (a | cp) & xv
            }
        } else if (S == 3) {
#pragma unroll
            for (int g = 0; g < GQ; g++) if (g < gn) {
                unsigned co;
                phs_[g] = __builtin_addc(ph_[g], ph_[g], cyp, &co); cyp = co;
                mhs_[g] = __builtin_addc(mh_[g], mh_[g], cym, &co); cym = co;
            }
            SB();
#pragma unroll
            for (int g = 0; g < GQ; g++) if (g < gn) { pv[q0 + g] = BITOP3(mhs_[g], xv_[g], phs_[g], 0xF1); mv[q0 + g] = phs_[g] & xv_[g]; }
        }
        ph_prev = ph_[gn - 1]; mh_prev = mh_[gn - 1];
        SB();
    }
}

template <int Q, int P, int S>
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
            column<Q, P, S>(pl, pv, mv, nk, carry, ph_prev, mh_prev);
            score += (int)(ph_prev >> 31) - (int)(mh_prev >> 31);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t s = (uint32_t)score;
#pragma unroll
    for (int q = 0; q < Q; q++) s = s * 31u + (pv[q] ^ (mv[q] * 7u));
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int Q, int S>
static unsigned run(uint32_t* out, unsigned long long* clk, uint32_t* hout, const char* what) {
    const int n_cols = 16000;
    int maxb = 0; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&maxb, k_columns<Q, 2, S>, 256, 0);
    hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_columns<Q, 2, S>));
    unsigned sum = 0;
    for (int wps : {4, 6}) {
        if (wps > maxb) continue;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const int blocks = 256 * wps;
        k_columns<Q, 2, S><<<blocks, 256>>>(out, clk, n_cols, 1); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); k_columns<Q, 2, S><<<blocks, 256>>>(out, clk, n_cols, 1); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hout, out, 256 * 4, hipMemcpyDeviceToHost);
        sum = 0; for (int i = 0; i < 256; i++) sum = sum * 1000003u + hout[i];
        const double ghz = (double)h[0] / ((double)h[1] / 0.1);
        printf("Q=%2d S=%d %-52s %3d VGPRs, waves/SIMD %d: %.2f cycles per word-column at the measured %.2f GHz, checksum %08x\n", Q, S, what, fa.numRegs, wps, ms * 1e-3 * ghz * 1e9 / ((double)wps * n_cols) / Q, ghz, sum);
    }
    return sum;
}

template <int Q> static void all(uint32_t* out, unsigned long long* clk, uint32_t* hout) {
    const unsigned c0 = run<Q, 0>(out, clk, hout, "shipped (2 v_alignbit)");
    const unsigned c1 = run<Q, 1>(out, clk, hout, "v_lshrrev + v_add_u32 + v_or");
    const unsigned c2 = run<Q, 2>(out, clk, hout, "carries folded into the final v_bitop3 pair");
    const unsigned c3 = run<Q, 3>(out, clk, hout, "shift on carry chains (v_addc_co x + x + c)");
    const unsigned c4 = run<Q, 4>(out, clk, hout, "v_lshrrev + v_lshl_or");
    printf("Q=%d: checksums %s\n", Q, (c0 == c1 && c0 == c2 && c0 == c3 && c0 == c4) ? "agree" : "DIFFER");
}

int main() {
    uint32_t* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    unsigned long long* clk; (void)hipMalloc(&clk, 16);
    uint32_t hout[256];
    all<4>(out, clk, hout); all<8>(out, clk, hout); all<12>(out, clk, hout); all<16>(out, clk, hout);
    return 0;
}
