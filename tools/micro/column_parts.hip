// What each instruction class of the column update (svim_amd/csrc/myers_column.hpp) costs IN CONTEXT: the same loop as column_clock.hip with one class at a time
// replaced by a full-rate stand-in (results are garbage, the instruction count stays): the time that disappears is that class's real marginal cost.
//   V=0 the update as shipped;  1 no carry chain (v_add_u32 instead of v_addc_co);  2 shifts by v_lshrrev (full rate) instead of v_alignbit;  3 both;
//   4 every v_bitop3 replaced by a two-operand v_xor;  5 = 3 + 4 (only full-rate two-operand instructions left);  6 as shipped without sched_barriers;
//   7 as shipped, words in groups of 8;  8 as shipped in groups of 2
// Build: hipcc --offload-arch=gfx950 -O3 -I svim_amd/csrc -o tools/micro/column_parts.bin tools/micro/column_parts.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define BITOP3(a_, b_, c_, tt_) ((uint32_t)__builtin_amdgcn_bitop3_b32((int)(a_), (int)(b_), (int)(c_), (tt_)))

template <int V> __device__ __forceinline__ uint32_t b3(uint32_t a, uint32_t b, uint32_t c, uint32_t xorstandin) {
    return 0;
}
#define B3(V_, a_, b_, c_, tt_) ((V_ == 4 || V_ == 5) ? ((a_) ^ (c_)) : BITOP3(a_, b_, c_, tt_))
#define SB(V_) do { if (V_ != 6) __builtin_amdgcn_sched_barrier(0); } while (0)

template <int Q, int P, int V>
__device__ __forceinline__ void column(uint32_t (&pl)[P][Q], uint32_t (&pv)[Q], uint32_t (&mv)[Q], const uint32_t (&nk)[P], unsigned& carry, uint32_t& ph_prev, uint32_t& mh_prev) {
    constexpr int GQ0 = V == 7 ? 8 : (V == 8 ? 2 : 4);
    constexpr int GQ = Q >= GQ0 ? GQ0 : Q;
#pragma unroll
    for (int q0 = 0; q0 < Q; q0 += GQ) {
        const int gn = Q - q0 < GQ ? Q - q0 : GQ;
        uint32_t eq_[GQ], xv_[GQ], sum_[GQ], ph_[GQ], mh_[GQ], phs_[GQ], mhs_[GQ];
#pragma unroll
        for (int g = 0; g < GQ; g++) if (g < gn) {
            uint32_t e = pl[0][q0 + g] ^ nk[0];
            e = B3(V, e, pl[1][q0 + g], nk[1], 0x60);
            eq_[g] = e;
        }
#pragma unroll
        for (int g = 0; g < GQ; g++) if (g < gn) {
            xv_[g] = eq_[g] | mv[q0 + g]; sum_[g] = eq_[g] & pv[q0 + g];
            asm("" : "+v"(xv_[g]));
        }
        SB(V);
#pragma unroll
        for (int g = 0; g < GQ; g++) if (g < gn) {
            if (V == 1 || V == 3 || V == 5) sum_[g] = sum_[g] + pv[q0 + g];
            else { unsigned carry_out; sum_[g] = __builtin_addc(sum_[g], pv[q0 + g], carry, &carry_out); carry = carry_out; }
        }
#pragma unroll
        for (int g = 0; g < GQ; g++) if (g < gn) sum_[g] = B3(V, sum_[g], pv[q0 + g], eq_[g], 0xBE);
#pragma unroll
        for (int g = 0; g < GQ; g++) if (g < gn) {
            ph_[g] = B3(V, mv[q0 + g], sum_[g], pv[q0 + g], 0xF1);
            mh_[g] = pv[q0 + g] & sum_[g];
        }
        SB(V);
#pragma unroll
        for (int g = 0; g < GQ; g++) if (g < gn) {
            if (V == 2 || V == 3 || V == 5) { phs_[g] = (ph_[g] >> 1) ^ (g ? ph_[g - 1] : ph_prev); mhs_[g] = (mh_[g] >> 1) ^ (g ? mh_[g - 1] : mh_prev); asm("" : "+v"(phs_[g])); asm("" : "+v"(mhs_[g])); }
            else { phs_[g] = __builtin_amdgcn_alignbit(ph_[g], g ? ph_[g - 1] : ph_prev, 31); mhs_[g] = __builtin_amdgcn_alignbit(mh_[g], g ? mh_[g - 1] : mh_prev, 31); }
        }
        ph_prev = ph_[gn - 1]; mh_prev = mh_[gn - 1];
        SB(V);
#pragma unroll
        for (int g = 0; g < GQ; g++) if (g < gn) {
            pv[q0 + g] = B3(V, mhs_[g], xv_[g], phs_[g], 0xF1);
            mv[q0 + g] = phs_[g] & xv_[g];
        }
        SB(V);
    }
}

template <int Q, int P, int V>
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
            column<Q, P, V>(pl, pv, mv, nk, carry, ph_prev, mh_prev);
            score += (int)(ph_prev >> 31) - (int)(mh_prev >> 31);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t s = (uint32_t)score;
#pragma unroll
    for (int q = 0; q < Q; q++) s ^= pv[q] ^ mv[q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int Q, int V>
static void run(uint32_t* out, unsigned long long* clk, const char* what) {
    const int n_cols = 16000;
    int maxb = 0; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&maxb, k_columns<Q, 2, V>, 256, 0);
    hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_columns<Q, 2, V>));
    for (int wps : {4, 8}) {
        if (wps > maxb) continue;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const int blocks = 256 * wps;
        k_columns<Q, 2, V><<<blocks, 256>>>(out, clk, n_cols, 1); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); k_columns<Q, 2, V><<<blocks, 256>>>(out, clk, n_cols, 1); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double ghz = (double)h[0] / ((double)h[1] / 0.1);
        printf("Q=%2d V=%d %-58s %3d VGPRs, waves/SIMD %d: %.2f cycles per word-column at the measured %.2f GHz\n", Q, V, what, fa.numRegs, wps, ms * 1e-3 * ghz * 1e9 / ((double)wps * n_cols) / Q, ghz);
    }
}

template <int Q> static void all(uint32_t* out, unsigned long long* clk) {
    run<Q, 0>(out, clk, "as shipped");
    run<Q, 1>(out, clk, "no carry chain (v_add_u32)");
    run<Q, 2>(out, clk, "v_lshrrev + v_xor instead of v_alignbit (+2 instr/word)");
    run<Q, 3>(out, clk, "no carry chain, no v_alignbit");
    run<Q, 4>(out, clk, "v_xor instead of every v_bitop3");
    run<Q, 5>(out, clk, "only full-rate two-operand instructions");
    run<Q, 6>(out, clk, "as shipped without sched_barriers");
    run<Q, 7>(out, clk, "as shipped, groups of 8 words");
    run<Q, 8>(out, clk, "as shipped, groups of 2 words");
}

int main() {
    uint32_t* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    unsigned long long* clk; (void)hipMalloc(&clk, 16);
    all<8>(out, clk); all<12>(out, clk); all<16>(out, clk);
    return 0;
}
