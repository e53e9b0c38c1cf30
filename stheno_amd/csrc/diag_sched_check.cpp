// diag_sched_check.cpp -- host-side verification of the static schedule of potrf_diag2_kernel (gpk_diag_sched.hpp).  No GPU.
//
//   g++ -O2 -std=c++17 -o diag_sched_check diag_sched_check.cpp && ./diag_sched_check
//
// 1. hazards: within every phase, no task of one wave writes a tile that a task of ANOTHER wave reads or writes (tasks of one
//    wave run in order, phases are separated by workgroup barriers) -- checked on explicit read / write sets;
// 2. numerics: the table is executed with plain loops on a random SPD block (tasks of a phase wave by wave, in both wave orders)
//    and the written-back factor L and the inverse W = inv(L) are compared with a textbook Cholesky / forward substitution;
// 3. bookkeeping: every tile of L is written back exactly once, after its last modification; prints the load per phase and wave.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

#include "gpk_diag_sched.hpp"

using namespace gpk_diag;

static const Sched SCHED = make_diag_sched();
constexpr int N = 128;

struct State {
    double S[N][N];          // the LDS image (all of it: tiles above the diagonal are scratch)
    double Ld[NT][16][16];   // diagonal tiles of L, kept for the write-back
    double outL[N][N];       // "global memory": the factor
    int stored[N][N];
};

static void chol_tile(State& st, int s) {       // what the chain wave does in P12(s): tile (s, s) -> Ld[s], inv -> tile (s, s)
    double a[16][16], l[16][16] = {}, w[16][16] = {};
    for (int r = 0; r < 16; ++r)
        for (int c = 0; c < 16; ++c) a[r][c] = (c <= r) ? st.S[16 * s + r][16 * s + c] : st.S[16 * s + c][16 * s + r];   // lower triangle only
    for (int j = 0; j < 16; ++j) {
        double d = a[j][j];
        for (int k = 0; k < j; ++k) d -= l[j][k] * l[j][k];
        l[j][j] = std::sqrt(d);
        for (int r = j + 1; r < 16; ++r) {
            double v = a[r][j];
            for (int k = 0; k < j; ++k) v -= l[r][k] * l[j][k];
            l[r][j] = v / l[j][j];
        }
    }
    for (int c = 0; c < 16; ++c)
        for (int r = 0; r < 16; ++r) {
            double v = (r == c) ? 1.0 : 0.0;
            for (int k = 0; k < r; ++k) v -= l[r][k] * w[k][c];
            w[r][c] = v / l[r][r];
        }
    for (int r = 0; r < 16; ++r)
        for (int c = 0; c < 16; ++c) {
            st.Ld[s][r][c] = l[r][c];
            st.S[16 * s + r][16 * s + c] = w[r][c];
        }
}

static void exec_task(State& st, uint32_t t) {
    if (t_kind(t) == (int)K_STORE) {
        const int s = t_or(t), part = t_ar(t), parts = t_ac(t);
        const int r0 = 16 * s, rows = N - r0;
        const int len = ((rows + parts - 1) / parts + 7) / 8 * 8;
        for (int r = r0 + part * len; r < r0 + (part + 1) * len && r < N; ++r)
            for (int c = 0; c < 16; ++c) {
                if (r < r0 + 16) {
                    if (c <= r - r0) { st.outL[r][r0 + c] = st.Ld[s][r - r0][c]; st.stored[r][r0 + c]++; }
                } else {
                    st.outL[r][r0 + c] = st.S[r][r0 + c];
                    st.stored[r][r0 + c]++;
                }
            }
        return;
    }
    double acc[16][16];
    const int orow = t_or(t), ocol = t_oc(t);
    for (int r = 0; r < 16; ++r)
        for (int c = 0; c < 16; ++c) acc[r][c] = t_init(t) ? st.S[16 * orow + r][16 * ocol + c] : 0.0;
    for (int kb = 0; kb < t_nkb(t); ++kb)
        for (int r = 0; r < 16; ++r)
            for (int c = 0; c < 16; ++c) {
                double sum = 0;
                for (int k = 0; k < 16; ++k) {
                    const double a = st.S[16 * t_ar(t) + r][16 * (t_ac(t) + kb) + k];
                    const double b = t_btrans(t) ? st.S[16 * t_br(t) + c][16 * (t_bc(t) + kb) + k]
                                                 : st.S[16 * (t_br(t) + kb) + k][16 * t_bc(t) + c];
                    sum += a * b;
                }
                acc[r][c] += t_neg(t) ? -sum : sum;
            }
    for (int r = 0; r < 16; ++r)
        for (int c = 0; c < 16; ++c) st.S[16 * orow + r][16 * ocol + c] = acc[r][c];
}

// read / write sets at tile granularity: tile id = 8 * row + col; Ld[s] = 64 + s
static void rw_sets(uint32_t t, std::set<int>& rd, std::set<int>& wr) {
    if (t_kind(t) == (int)K_STORE) {
        const int s = t_or(t);
        rd.insert(64 + s);
        for (int i = s + 1; i < NT; ++i) rd.insert(8 * i + s);
        return;
    }
    if (t_init(t)) rd.insert(8 * t_or(t) + t_oc(t));
    wr.insert(8 * t_or(t) + t_oc(t));
    for (int kb = 0; kb < t_nkb(t); ++kb) {
        rd.insert(8 * t_ar(t) + t_ac(t) + kb);
        rd.insert(t_btrans(t) ? 8 * t_br(t) + t_bc(t) + kb : 8 * (t_br(t) + kb) + t_bc(t));
    }
}

static int check_hazards() {
    int bad = 0;
    for (int ph = 0; ph < NPH; ++ph) {
        std::set<int> rd[4], wr[4];
        for (int w = 0; w < 4; ++w)
            for (int q = 0; q < SCHED.count[ph][w]; ++q) rw_sets(SCHED.t[ph][w][q], rd[w], wr[w]);
        if (ph % 2 == 0 && ph <= 14) {   // CHOL(s) on wave 0
            const int s = ph / 2;
            rd[0].insert(9 * s);
            wr[0].insert(9 * s);
            wr[0].insert(64 + s);
        }
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) {
                if (a == b) continue;
                for (int x : wr[a])
                    if (rd[b].count(x) || (a < b && wr[b].count(x))) {
                        printf("HAZARD phase %d: wave %d writes tile %d that wave %d %s\n", ph, a, x, b, rd[b].count(x) ? "reads" : "writes");
                        ++bad;
                    }
            }
        // inside a wave: a task that reads what the previous one wrote must carry the dep flag
        for (int w = 0; w < 4; ++w)
            for (int q = 1; q < SCHED.count[ph][w]; ++q) {
                std::set<int> r1, w1, r0, w0;
                rw_sets(SCHED.t[ph][w][q], r1, w1);
                for (int q0 = 0; q0 < q; ++q0) {
                    r0.clear(); w0.clear();
                    rw_sets(SCHED.t[ph][w][q0], r0, w0);
                    for (int x : w0)
                        if (r1.count(x) && !(q0 == q - 1 && t_dep(SCHED.t[ph][w][q]))) {
                            // allowed only if the producer is at least two tasks back (its store has retired before the prefetch of
                            // task q is issued: prefetch happens one task ahead) -- flag anything else
                            if (q0 >= q - 1) { printf("DEP phase %d wave %d task %d reads tile %d of task %d without the dep flag\n", ph, w, q, x, q0); ++bad; }
                        }
                }
            }
    }
    return bad;
}

static int run_numerics(bool reverse_waves, unsigned seed) {
    static State st;
    memset(&st, 0, sizeof(st));
    static double A[N][N], G[N][N];
    srand(seed);
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) G[i][j] = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            double s = (i == j) ? 0.5 : 0.0;
            for (int k = 0; k < N; ++k) s += G[i][k] * G[j][k] / N;
            A[i][j] = s;
        }
    // phase 0 of the kernel: the lower triangle in; everything above the diagonal is garbage (scratch tiles, never-read halves)
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) st.S[i][j] = (j <= i) ? A[i][j] : 1e300;
    for (int ph = 0; ph < NPH; ++ph) {
        for (int wi = 0; wi < 4; ++wi) {
            const int w = reverse_waves ? 3 - wi : wi;
            if (w == 0 && ph % 2 == 0 && ph <= 14) chol_tile(st, ph / 2);
            // the kernel runs a wave's tile products first, then its write-back tasks (make_dev_sched lowers the lists that way)
            for (int pass = 0; pass < 2; ++pass)
                for (int q = 0; q < SCHED.count[ph][w]; ++q)
                    if ((t_kind(SCHED.t[ph][w][q]) == (int)K_MM) == (pass == 0)) exec_task(st, SCHED.t[ph][w][q]);
        }
    }
    // reference
    static double L[N][N], W[N][N];
    memset(L, 0, sizeof(L));
    memset(W, 0, sizeof(W));
    for (int j = 0; j < N; ++j) {
        double d = A[j][j];
        for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
        L[j][j] = std::sqrt(d);
        for (int r = j + 1; r < N; ++r) {
            double v = A[r][j];
            for (int k = 0; k < j; ++k) v -= L[r][k] * L[j][k];
            L[r][j] = v / L[j][j];
        }
    }
    for (int c = 0; c < N; ++c)
        for (int r = c; r < N; ++r) {
            double v = (r == c) ? 1.0 : 0.0;
            for (int k = c; k < r; ++k) v -= L[r][k] * W[k][c];
            W[r][c] = v / L[r][r];
        }
    int bad = 0;
    double eL = 0, eW = 0;
    for (int i = 0; i < N; ++i)
        for (int j = 0; j <= i; ++j) {
            if (st.stored[i][j] != 1) { if (bad < 5) printf("tile entry (%d, %d) of L written back %d times\n", i, j, st.stored[i][j]); ++bad; }
            eL = std::fmax(eL, std::fabs(st.outL[i][j] - L[i][j]));
            eW = std::fmax(eW, std::fabs(st.S[i][j] - W[i][j]));     // the inverse ends up in place (lower triangle of S)
        }
    for (int i = 0; i < N; ++i)
        for (int j = i + 1; j < N; ++j)
            if (st.stored[i][j] != 0) { ++bad; printf("entry (%d, %d) above the diagonal was written back\n", i, j); }
    printf("numerics (waves %s, seed %u): max |L - ref| = %.3g, max |W - ref| = %.3g\n", reverse_waves ? "3..0" : "0..3", seed, eL, eW);
    if (!(eL < 1e-12) || !(eW < 1e-9)) ++bad;
    return bad;
}

int main() {
    static_assert(make_diag_sched().overflow == 0, "a task list of the diagonal-block schedule overflows MAXT");
    static_assert(sched_has_end_slot(), "every task list needs an empty slot at its end");
    {   // the lowered table: same number of tasks per list, products first
        constexpr DevSched D = make_dev_sched(130);
        for (int ph = 0; ph < NPH; ++ph)
            for (int w = 0; w < 4; ++w) {
                int n = 0, seen_store = 0;
                for (; n < MAXT && (D.w[ph][w][MAXT + n] >> 31); ++n) {
                    const int kind = dw1_kind(D.w[ph][w][MAXT + n]);
                    if (kind == (int)K_STORE) seen_store = 1;
                    if (kind == (int)K_MM && seen_store) { printf("lowered list %d/%d: product after a write-back\n", ph, w); return 1; }
                }
                if (n != SCHED.count[ph][w]) { printf("lowered list %d/%d has %d tasks, schedule %d\n", ph, w, n, SCHED.count[ph][w]); return 1; }
            }
    }
    int bad = check_hazards();
    for (unsigned seed = 1; seed <= 3; ++seed) {
        bad += run_numerics(false, seed);
        bad += run_numerics(true, seed);
    }
    printf("load (work units) per phase, waves 0..3 (wave 0 additionally runs CHOL in even phases <= 14):\n");
    for (int ph = 0; ph < NPH; ++ph)
        printf("  phase %2d: %2d %2d %2d %2d   tasks %d %d %d %d\n", ph, SCHED.load[ph][0], SCHED.load[ph][1], SCHED.load[ph][2], SCHED.load[ph][3],
               SCHED.count[ph][0], SCHED.count[ph][1], SCHED.count[ph][2], SCHED.count[ph][3]);
    printf("DIAG_SCHED %s (%d problems)\n", bad ? "FAIL" : "OK", bad);
    return bad ? 1 : 0;
}
