// gpk_diag_sched.hpp -- the static schedule of the 128x128 diagonal-block kernel (potrf_diag2_kernel, gpk_potrf.hip).
//
// The block is 8 x 8 tiles of 16 x 16, held in LDS.  One wave ("chain", wave 0) walks the serial dependency chain of the
// factorisation -- CHOL(s): 16 x 16 Cholesky + inverse of diagonal tile s; then the solve of tile (s+1, s) and the update of tile
// (s+1, s+1) -- while the other three ("bulk") run everything that is not on it: the remaining triangular solves, the rank-16
// updates, the write-back of finished column panels of L and the recursive-doubling merges of the inverse, all as 16 x 16 tile
// operations on MFMA.  The work is cut into PHASES separated by workgroup barriers; what every wave does in every phase is DATA:
// the table built by make_diag_sched() below at compile time (constant memory on the device).  The host-side checker
// (diag_sched_check.cpp, run by the CPU test suite) executes the same table with plain loops -- tasks of a phase in adversarial
// orders -- and compares with a reference factorisation, so the dependency structure is verified without a GPU.
//
// Phases:  2 s      (s = 0..7)  "P12(s)": chain: CHOL(s);                     bulk: UPD(s) = column s-1 applied to every tile behind it
//          2 s + 1  (s = 0..6)  "P3(s)" : chain: solve (s+1, s), update (s+1, s+1);   bulk: solves (i, s), i >= s + 2
//          15, 16, 17           tail of the inversion (what needs the last diagonal tile)
// plus, wherever the bulk waves have slack: the write-back of column panel s of L (any time after P3(s)) and the merge passes
//   M0(level, pair):  T  = C Ai     into the (otherwise unused) mirror tiles above the diagonal
//   M1(level, pair):  C' = -Di T    over the tiles of L they replace (after those were written back and last used as operands)
// of  [A 0; C D]^-1 = [Ai 0; -Di C Ai, Di],  levels 16 -> 32 -> 64 -> 128.
#pragma once
#include <stdint.h>

namespace gpk_diag {

constexpr int NT = 8;           // tiles per block edge
constexpr int NPH = 18;         // phases
constexpr int MAXT = 12;        // task slots per (phase, wave)

// ---- task word ----
// bits  0- 2 out_r   3- 5 out_c   6- 8 a_r   9-11 a_c   12-14 b_r   15-17 b_c   18-20 nkb (1..4)
//       21 init (accumulator starts from the output tile)   22 neg (subtract the product)   23 btrans (B tile read transposed)
//       24 dep (reads what the previous task of this wave wrote: no operand prefetch across it)
//       25-27 kind: 1 = tile product, 2 = write back a part of column panel out_r of L (a_r = part, a_c = parts)
//       31 valid
constexpr uint32_t K_MM = 1, K_STORE = 2;

constexpr uint32_t task_mm(int orow, int ocol, int ar, int ac, int br, int bc, int nkb, bool init, bool neg, bool btrans, bool dep) {
    return (uint32_t)orow | ((uint32_t)ocol << 3) | ((uint32_t)ar << 6) | ((uint32_t)ac << 9) | ((uint32_t)br << 12) | ((uint32_t)bc << 15) |
           ((uint32_t)nkb << 18) | ((uint32_t)init << 21) | ((uint32_t)neg << 22) | ((uint32_t)btrans << 23) | ((uint32_t)dep << 24) |
           (K_MM << 25) | (1u << 31);
}
constexpr uint32_t task_store(int s, int part, int parts) {
    return (uint32_t)s | ((uint32_t)part << 6) | ((uint32_t)parts << 9) | (1u << 18) | (K_STORE << 25) | (1u << 31);
}
constexpr int t_or(uint32_t t) { return t & 7; }
constexpr int t_oc(uint32_t t) { return (t >> 3) & 7; }
constexpr int t_ar(uint32_t t) { return (t >> 6) & 7; }
constexpr int t_ac(uint32_t t) { return (t >> 9) & 7; }
constexpr int t_br(uint32_t t) { return (t >> 12) & 7; }
constexpr int t_bc(uint32_t t) { return (t >> 15) & 7; }
constexpr int t_nkb(uint32_t t) { return (t >> 18) & 7; }
constexpr bool t_init(uint32_t t) { return (t >> 21) & 1; }
constexpr bool t_neg(uint32_t t) { return (t >> 22) & 1; }
constexpr bool t_btrans(uint32_t t) { return (t >> 23) & 1; }
constexpr bool t_dep(uint32_t t) { return (t >> 24) & 1; }
constexpr int t_kind(uint32_t t) { return (t >> 25) & 7; }

struct Sched {
    uint32_t t[NPH][4][MAXT];
    int load[NPH][4];       // work units per (phase, wave) (one unit = one 16x16x16 product); build-time bookkeeping
    int count[NPH][4];
    int overflow;           // != 0: a list did not fit (checked by a static_assert)
};

constexpr int ph_p12(int s) { return 2 * s; }
constexpr int ph_p3(int s) { return 2 * s + 1; }
constexpr int PH_T1 = 15, PH_T2 = 16, PH_T3 = 17;

constexpr void put(Sched& S, int ph, int wave, uint32_t task, int units) {
    if (S.count[ph][wave] >= MAXT) {
        S.overflow = 1;
        return;
    }
    S.t[ph][wave][S.count[ph][wave]++] = task;
    S.load[ph][wave] += units;
}
// to the least loaded of waves lo..hi
constexpr void put_balanced(Sched& S, int ph, int lo, int hi, uint32_t task, int units) {
    int best = lo;
    for (int w = lo + 1; w <= hi; ++w)
        if (S.load[ph][w] < S.load[ph][best]) best = w;
    put(S, ph, best, task, units);
}

// solve of tile (i, s):  L(i, s) = X(i, s) inv(L_ss)^T,  inv(L_ss) sits in tile (s, s)
constexpr uint32_t task_trsm(int i, int s, bool dep) { return task_mm(i, s, i, s, s, s, 1, false, false, true, dep); }
// update of tile (i, j) by column k:  S(i, j) -= L(i, k) L(j, k)^T
constexpr uint32_t task_upd(int i, int j, int k, bool dep) { return task_mm(i, j, i, k, j, k, 1, true, true, true, dep); }
// merge of the pair at tile offset o, half width h tiles:  pass 0, T(ti, tj) = sum_{kb >= tj} C(ti, kb) Ai(kb, tj) -> mirror tile (o + ti, o + h + tj)
constexpr uint32_t task_m0(int o, int h, int ti, int tj) {
    return task_mm(o + ti, o + h + tj, o + h + ti, o + tj, o + tj, o + tj, h - tj, false, false, false, false);
}
// pass 1, C'(ti, tj) = -sum_{kb <= ti} Di(ti, kb) T(kb, tj) -> tile (o + h + ti, o + tj)
constexpr uint32_t task_m1(int o, int h, int ti, int tj) {
    return task_mm(o + h + ti, o + tj, o + h + ti, o + h, o, o + h + tj, ti + 1, false, true, false, false);
}

constexpr void put_m0(Sched& S, int ph, int lo, int hi, int o, int h, int first, int last) {   // tasks first..last-1 of the h*h list, longest first
    int idx = 0;
    for (int tj = 0; tj < h; ++tj)
        for (int ti = 0; ti < h; ++ti, ++idx)
            if (idx >= first && idx < last) put_balanced(S, ph, lo, hi, task_m0(o, h, ti, tj), h - tj);
}
constexpr void put_m1(Sched& S, int ph, int lo, int hi, int o, int h) {
    for (int ti = h - 1; ti >= 0; --ti)
        for (int tj = 0; tj < h; ++tj) put_balanced(S, ph, lo, hi, task_m1(o, h, ti, tj), ti + 1);
}
constexpr void put_store(Sched& S, int ph, int s) {           // column panel s of L, rows split over the three bulk waves
    for (int part = 0; part < 3; ++part) put(S, ph, 1 + part, task_store(s, part, 3), 1);
}

constexpr Sched make_diag_sched() {
    Sched S{};
    for (int s = 0; s < NT; ++s) {
        // P12(s): the bulk waves apply column s - 1 to every tile behind it, except (s, s) (the chain did that one in P3(s - 1))
        if (s >= 1)
            for (int j = s; j < NT; ++j)
                for (int i = j; i < NT; ++i)
                    if (!(i == s && j == s)) put_balanced(S, ph_p12(s), 1, 3, task_upd(i, j, s - 1, false), 1);
        if (s + 1 < NT) {
            // P3(s): the chain solves (s+1, s) and updates (s+1, s+1) with it; the bulk waves solve the rest of column s
            put(S, ph_p3(s), 0, task_trsm(s + 1, s, false), 1);
            put(S, ph_p3(s), 0, task_upd(s + 1, s + 1, s, true), 1);
            for (int i = s + 2; i < NT; ++i) put_balanced(S, ph_p3(s), 1, 3, task_trsm(i, s, false), 1);
        }
    }
    // write-back of column panel s: after P3(s); before any merge overwrites its tiles
    for (int s = 0; s <= 5; ++s) put_store(S, ph_p3(s + 1), s);
    put_store(S, ph_p12(7), 6);
    put_store(S, PH_T1, 7);
    // level 16 -> 32 (pairs of tiles 2p, 2p + 1)
    for (int p = 0; p < 4; ++p) {
        const int ph0 = (p < 3) ? ph_p3(2 * p + 1) : ph_p12(7);
        const int ph1 = (p < 3) ? ph_p12(2 * p + 2) : PH_T1;
        put_m0(S, ph0, 1, 3, 2 * p, 1, 0, 1);
        if (p < 3) put_m1(S, ph1, 1, 3, 2 * p, 1);
        else put_m1(S, ph1, 0, 0, 2 * p, 1);          // the tail's first step: one task, on the chain wave (bulk waves write back panel 7)
    }
    // level 32 -> 64
    put_m0(S, ph_p12(3), 1, 3, 0, 2, 0, 4);
    put_m1(S, ph_p12(5), 1, 3, 0, 2);
    put_m0(S, ph_p12(7), 1, 3, 4, 2, 0, 4);
    put_m1(S, PH_T2, 0, 3, 4, 2);
    // level 64 -> 128: pass 0 under the last two CHOLs, pass 1 is the end of the tail
    put_m0(S, ph_p12(6), 1, 3, 0, 4, 0, 7);
    put_m0(S, ph_p12(7), 1, 3, 0, 4, 7, 16);
    put_m1(S, PH_T3, 0, 3, 0, 4);
    return S;
}

// ---- the table as the kernel reads it: two words per task, element offsets into the LDS image resolved (row pitch `ldp`) ----
//   w0: bits 0-15 offset of the output tile, bits 16-31 offset of the A tile            (store task: s | part << 8 | parts << 12)
//   w1: bits 0-15 offset of the B tile, 16-18 nkb, 19 init, 20 neg, 21 btrans, 22 dep, 23-25 kind, 31 valid
// Per (phase, wave): the tile products first (in schedule order), then the write-back tasks; at least one empty slot ends the list.
struct DevSched {
    uint32_t w[NPH][4][2 * MAXT];     // [.. 0 .. MAXT-1] = w0 of task q, [MAXT .. 2 MAXT-1] = w1 of task q
};
constexpr int dw1_nkb(uint32_t w1) { return (w1 >> 16) & 7; }
constexpr bool dw1_init(uint32_t w1) { return (w1 >> 19) & 1; }
constexpr bool dw1_neg(uint32_t w1) { return (w1 >> 20) & 1; }
constexpr bool dw1_btrans(uint32_t w1) { return (w1 >> 21) & 1; }
constexpr bool dw1_dep(uint32_t w1) { return (w1 >> 22) & 1; }
constexpr int dw1_kind(uint32_t w1) { return (w1 >> 23) & 7; }

constexpr DevSched make_dev_sched(int ldp) {
    const Sched S = make_diag_sched();
    DevSched D{};
    for (int ph = 0; ph < NPH; ++ph)
        for (int w = 0; w < 4; ++w) {
            int n = 0;
            for (int pass = 0; pass < 2; ++pass)
                for (int q = 0; q < S.count[ph][w]; ++q) {
                    const uint32_t t = S.t[ph][w][q];
                    const bool mm = t_kind(t) == (int)K_MM;
                    if (mm != (pass == 0)) continue;
                    uint32_t w0 = 0, w1 = 0;
                    if (mm) {
                        w0 = (uint32_t)(16 * t_or(t) * ldp + 16 * t_oc(t)) | ((uint32_t)(16 * t_ar(t) * ldp + 16 * t_ac(t)) << 16);
                        w1 = (uint32_t)(16 * t_br(t) * ldp + 16 * t_bc(t)) | ((uint32_t)t_nkb(t) << 16) | ((uint32_t)t_init(t) << 19) |
                             ((uint32_t)t_neg(t) << 20) | ((uint32_t)t_btrans(t) << 21) | ((uint32_t)t_dep(t) << 22) | (K_MM << 23) | (1u << 31);
                    } else {
                        w0 = (uint32_t)t_or(t) | ((uint32_t)t_ar(t) << 8) | ((uint32_t)t_ac(t) << 12);
                        w1 = (K_STORE << 23) | (1u << 31);
                    }
                    D.w[ph][w][n] = w0;
                    D.w[ph][w][MAXT + n] = w1;
                    ++n;
                }
        }
    return D;
}
constexpr bool sched_has_end_slot() {      // the kernel recognises the end of a list by an empty slot
    const Sched S = make_diag_sched();
    for (int ph = 0; ph < NPH; ++ph)
        for (int w = 0; w < 4; ++w)
            if (S.count[ph][w] >= MAXT) return false;
    return S.overflow == 0;
}

}  // namespace gpk_diag
