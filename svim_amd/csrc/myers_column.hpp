// myers_column.hpp - the column update shared by every edit-distance kernel (edit.hip) and by tools/micro/column_rate.hip.
#pragma once
#include <stdint.h>

// One column of the multi-word recurrence; leaves the (plus, minus) bits pushed out of the last word in bit 31 of ph_prev_ / mh_prev_.
// Issue cost on gfx950, measured in real shader cycles per wave64 instruction and SIMD (tools/micro/gen_valu_banks.py -> profiles/r06_valu_banks.txt):
//   2.3  v_xor / v_and / v_or / v_add_u32 / v_lshrrev / v_not / v_mov, and v_bitop3 / v_fma reading three VGPRs unless all three sit in one register bank
//   4.4  v_addc_co / v_add_co / v_sub_co / v_cmp (everything that writes a carry or a lane mask), v_alignbit, v_lshlrev, v_bfe, v_bfi, v_perm, v_and_or,
//        v_lshl_or, v_lshl_add, v_add3, v_xad, v_or3, v_mov_dpp, v_readlane, every instruction with an SGPR operand, v_bitop3 / v_fma with three same-bank VGPRs
// but INSIDE the update a v_alignbit costs ~8.8 (tools/micro/column_parts.hip, column_shift.hip: profiles/r06_column_shift.txt), a v_addc_co its 4.4.  The shifts
// phs = (ph << 1) | (bit 31 of the word below) therefore ride on two more carry chains - x + x + carry in, carry out = bit 31 of x - instead of two v_alignbit:
// round 6, 42.2 -> 33.2 cycles per word-column at 16 words per lane, 43.1 -> 34.9 at 12, 44.6 -> 38.2 at 8 (the same twelve instructions per word:
// 9 full rate + 3 v_addc_co = 33.9 by the table above; A/B of the whole step at commit 993386d: profiles/r06_edit_carry_chain_shift_ab.txt).
// Left to itself the compiler emits each word's instructions almost back to back (a dependent chain mixing both rates stalls, see
// valu_dep.hip), so the words are processed in groups of 4 with the recurrence cut into phases, every phase running over the 4
// words before the next starts (sched_barrier keeps the phases apart).
#define MYERS_GROUP(Q_) ((Q_) >= 4 ? 4 : (Q_))
// v_bitop3_b32 (any function of three words): truth table = the function applied to 0xF0, 0xCC, 0xAA
#define BITOP3(a_, b_, c_, tt_) ((uint32_t)__builtin_amdgcn_bitop3_b32((int)(a_), (int)(b_), (int)(c_), (tt_)))
// The core: one column over Q_ words.
//   TOP_ = 1: the row above the column is the constant boundary (horizontal delta +1: the plus chain starts with carry 1, the minus chain with 0; cyp_ / cym_
//             are not read); TOP_ = 0: cyp_ / cym_ hold the bits (0 / 1) the words above pushed out.
//   carry_:   carry of the adder chain, in and out.
//   Out: cyp_ / cym_ = the (plus, minus) bits pushed out of the last word (the horizontal delta of its last row), ph_last_ / mh_last_ = that word's ph / mh
//   (bit 31 = the same bits) - a caller uses whichever it can hand on cheaply, the other is dead code.
#define MYERS_SHIFT(w_, TOP_, ph_, mh_, phs_, mhs_, cyp_, cym_) {                                                                   \
            unsigned co_;                                                                                                                      \
            if (TOP_ && (w_) == 0) { phs_ = __builtin_addc(ph_, ph_, 0u, &co_) | 1u; cyp_ = co_; mhs_ = __builtin_addc(mh_, mh_, 0u, &co_); cym_ = co_; } \
            else { phs_ = __builtin_addc(ph_, ph_, cyp_, &co_); cyp_ = co_;                      /* v_addc_co_u32: ph + ph + carry in; carry out = bit 31 of ph */ \
                   mhs_ = __builtin_addc(mh_, mh_, cym_, &co_); cym_ = co_; } }
#define MYERS_COLUMN_C(Q_, P_, TOP_, pl_, pv_, mv_, nk_, carry_, cyp_, cym_, ph_last_, mh_last_) {              \
    _Pragma("unroll") for (int q0 = 0; q0 < Q_; q0 += MYERS_GROUP(Q_)) {                                        \
        constexpr int GQ = MYERS_GROUP(Q_);                                                                     \
        const int gn = Q_ - q0 < GQ ? Q_ - q0 : GQ;                              /* words in this group (the last one may be short) */ \
        uint32_t eq_[GQ], xv_[GQ], sum_[GQ], ph_[GQ], mh_[GQ], phs_[GQ], mhs_[GQ];                              \
        _Pragma("unroll") for (int g = 0; g < GQ; g++) if (g < gn) {                                            \
            uint32_t e = pl_[0][q0 + g] ^ nk_[0];                                                               \
            if (P_ == 2) e = BITOP3(e, pl_[1][q0 + g], nk_[1], 0x60);             /* e & (p1 ^ n1) */              \
            else { _Pragma("unroll") for (int b = 1; b < P_; b++) e &= pl_[b][q0 + g] ^ nk_[b]; }               \
            eq_[g] = e;                                                                                         \
        }                                                                                                       \
        _Pragma("unroll") for (int g = 0; g < GQ; g++) if (g < gn) {                                            \
            xv_[g] = eq_[g] | mv_[q0 + g]; sum_[g] = eq_[g] & pv_[q0 + g];                                      \
            asm("" : "+v"(xv_[g]));     /* opaque: keeps `phs & xv` a two-operand v_and */                       \
        }                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        _Pragma("unroll") for (int g = 0; g < GQ; g++) if (g < gn) {                                            \
            unsigned carry_out;                                                                                 \
            sum_[g] = __builtin_addc(sum_[g], pv_[q0 + g], carry_, &carry_out);    /* v_addc_co_u32: the carry stays in an SGPR pair */ \
            carry_ = carry_out;                                                                                 \
        }                                                                                                       \
        _Pragma("unroll") for (int g = 0; g < GQ; g++) if (g < gn) sum_[g] = BITOP3(sum_[g], pv_[q0 + g], eq_[g], 0xBE);     /* xh = (sum ^ pv) | eq */ \
        _Pragma("unroll") for (int g = 0; g < GQ; g++) if (g < gn) {                                            \
            ph_[g] = BITOP3(mv_[q0 + g], sum_[g], pv_[q0 + g], 0xF1);             /* mv | ~(xh | pv) */           \
            mh_[g] = pv_[q0 + g] & sum_[g];                                                                     \
        }                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        _Pragma("unroll") for (int g = 0; g < GQ; g++) if (g < gn)                                              \
            MYERS_SHIFT(q0 + g, TOP_, ph_[g], mh_[g], phs_[g], mhs_[g], cyp_, cym_)                              \
        ph_last_ = ph_[gn - 1]; mh_last_ = mh_[gn - 1];                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        _Pragma("unroll") for (int g = 0; g < GQ; g++) if (g < gn) {                                            \
            pv_[q0 + g] = BITOP3(mhs_[g], xv_[g], phs_[g], 0xF1);                 /* mhs | ~(xv | phs) */         \
            mv_[q0 + g] = phs_[g] & xv_[g];                                                                     \
        }                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
    } }

// The same column for a ROW BLOCK that receives its top boundary from memory (d_edit_blocked, edit.hip): cyp_ / cym_ hold the (plus, minus) horizontal delta of the
// row above the block, hin_m_ = the minus bit again (0 / 1) - ORed into the match vector's first bit AFTER xv was taken, as in Myers' block formulation (the
// adder's carry does not cross a block boundary: carry_ comes in as 0).
#define MYERS_COLUMN_B(Q_, P_, pl_, pv_, mv_, nk_, hin_m_, carry_, cyp_, cym_, ph_last_, mh_last_) {              \
    _Pragma("unroll") for (int q0 = 0; q0 < Q_; q0 += MYERS_GROUP(Q_)) {                                        \
        constexpr int GQ = MYERS_GROUP(Q_);                                                                     \
        const int gn = Q_ - q0 < GQ ? Q_ - q0 : GQ;                              /* words in this group (the last one may be short) */ \
        uint32_t eq_[GQ], xv_[GQ], sum_[GQ], ph_[GQ], mh_[GQ], phs_[GQ], mhs_[GQ];                              \
        _Pragma("unroll") for (int g = 0; g < GQ; g++) if (g < gn) {                                            \
            uint32_t e = pl_[0][q0 + g] ^ nk_[0];                                                               \
            if (P_ == 2) e = BITOP3(e, pl_[1][q0 + g], nk_[1], 0x60);             /* e & (p1 ^ n1) */              \
            else { _Pragma("unroll") for (int b = 1; b < P_; b++) e &= pl_[b][q0 + g] ^ nk_[b]; }               \
            eq_[g] = e;                                                                                         \
        }                                                                                                       \
        _Pragma("unroll") for (int g = 0; g < GQ; g++) if (g < gn) {                                            \
            xv_[g] = eq_[g] | mv_[q0 + g];                                                                      \
            if (q0 == 0 && g == 0) eq_[g] |= (hin_m_);                 /* Myers' block rule: a -1 coming in from above is a match in row 1 */ \
            sum_[g] = eq_[g] & pv_[q0 + g];                                                                     \
            asm("" : "+v"(xv_[g]));     /* opaque: keeps `phs & xv` a two-operand v_and */                       \
        }                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        _Pragma("unroll") for (int g = 0; g < GQ; g++) if (g < gn) {                                            \
            unsigned carry_out;                                                                                 \
            sum_[g] = __builtin_addc(sum_[g], pv_[q0 + g], carry_, &carry_out);    /* v_addc_co_u32: the carry stays in an SGPR pair */ \
            carry_ = carry_out;                                                                                 \
        }                                                                                                       \
        _Pragma("unroll") for (int g = 0; g < GQ; g++) if (g < gn) sum_[g] = BITOP3(sum_[g], pv_[q0 + g], eq_[g], 0xBE);     /* xh = (sum ^ pv) | eq */ \
        _Pragma("unroll") for (int g = 0; g < GQ; g++) if (g < gn) {                                            \
            ph_[g] = BITOP3(mv_[q0 + g], sum_[g], pv_[q0 + g], 0xF1);             /* mv | ~(xh | pv) */           \
            mh_[g] = pv_[q0 + g] & sum_[g];                                                                     \
        }                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        _Pragma("unroll") for (int g = 0; g < GQ; g++) if (g < gn)                                              \
            MYERS_SHIFT(q0 + g, 0, ph_[g], mh_[g], phs_[g], mhs_[g], cyp_, cym_)                              \
        ph_last_ = ph_[gn - 1]; mh_last_ = mh_[gn - 1];                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        _Pragma("unroll") for (int g = 0; g < GQ; g++) if (g < gn) {                                            \
            pv_[q0 + g] = BITOP3(mhs_[g], xv_[g], phs_[g], 0xF1);                 /* mhs | ~(xv | phs) */         \
            mv_[q0 + g] = phs_[g] & xv_[g];                                                                     \
        }                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
    } }


// Word interface (tools/micro): the bits coming in from above are bit 31 of ph_prev_ / mh_prev_, the bits pushed out are left there.
#define MYERS_COLUMN(Q_, P_, pl_, pv_, mv_, nk_, carry_, ph_prev_, mh_prev_) {                                  \
    unsigned cyp_w_ = (ph_prev_) >> 31, cym_w_ = (mh_prev_) >> 31;                                               \
    MYERS_COLUMN_C(Q_, P_, 0, pl_, pv_, mv_, nk_, carry_, cyp_w_, cym_w_, ph_prev_, mh_prev_) }

