// pipe_check.cpp -- host-side check of the task list of the pipelined panel factorisation (gpk_potrf_pipe.hpp): plays the list
// against a model of the chain workgroup and of the progress words, for many panel shapes, and verifies that
//   * every task finds its dependencies satisfied by EARLIER tasks and by chain steps that can have run (so workers that take
//     tasks in list order cannot deadlock), and every progress word moves exactly j -> j + 1;
//   * at the end every piece of the panel has received exactly the updates of the right-looking sweep and every piece below a
//     factorised diagonal block has been solved.
// Plain C++ (g++ pipe_check.cpp && ./a.out); run by tests/test_pipe_tasks.py.  Test infrastructure, not part of libgpk.so.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gpk_potrf_pipe.hpp"

static int fail(const char* what, const PipeShape& sh, int j, int s, int cb) {
    std::printf("FAIL %s: R=%d npb=%d nd=%d  task j=%d s=%d cb=%d\n", what, sh.R, sh.npb, sh.nd, j, s, cb);
    return 1;
}

struct Model {
    PipeShape sh;
    std::vector<int> prog;      // [s][cb]
    std::vector<int> dflag;     // inverse of block j published
    int chain_j = 0;            // next chain step
    int chain_phase = 0;        // 0: factorise, 1: solve the rows of block j+1, 2: update diagonal block j+1
    bool chain_done = false;
    int& P(int s, int cb) { return prog[(size_t)s * sh.npb + cb]; }

    // let the chain run as far as its dependencies allow
    void run_chain() {
        while (!chain_done) {
            const int j = chain_j;
            if (chain_phase == 0) {
                dflag[j] = 1;
                if (j + 1 >= sh.npb) { chain_done = true; break; }
                chain_phase = 1;
            }
            if (chain_phase == 1) {
                bool ok = true;
                for (int h = 0; h < 2; ++h) {
                    const int s = 2 * (j + 1) + h;
                    if (s < sh.R && P(s, j) != j) ok = false;
                }
                if (!ok) break;
                for (int h = 0; h < 2; ++h) {
                    const int s = 2 * (j + 1) + h;
                    if (s < sh.R) P(s, j) = j + 1;
                }
                chain_phase = 2;
            }
            if (chain_phase == 2) {
                bool ok = true;
                for (int h = 0; h < 2; ++h) {
                    const int s = 2 * (j + 1) + h;
                    if (s < sh.R && P(s, j + 1) != j) ok = false;
                }
                if (!ok) break;
                for (int h = 0; h < 2; ++h) {
                    const int s = 2 * (j + 1) + h;
                    if (s < sh.R) P(s, j + 1) = j + 1;
                }
                if (j + 1 >= sh.nd) { chain_done = true; break; }
                chain_j = j + 1;
                chain_phase = 0;
            }
        }
    }
};

static int check(const PipeShape& sh) {
    Model m;
    m.sh = sh;
    m.prog.assign((size_t)sh.R * sh.npb, 0);
    m.dflag.assign(sh.npb, 0);
    long total = 0;
    for (int j = 0; j < sh.nd; ++j) {
        const int nt = pipe_step_tasks(sh, j);
        int prev_s = -1, prev_cb = -2;
        for (int q = 0; q < nt; ++q) {
            const PipeTask t = pipe_decode(sh, j, q);
            ++total;
            if (t.j != j || t.s < pipe_first_strip(sh, j) || t.s >= sh.R) return fail("strip range", sh, j, t.s, t.cb);
            m.run_chain();
            if (t.cb < 0) {
                if (!m.dflag[j]) return fail("solve before the inverse", sh, j, t.s, t.cb);
                if (m.P(t.s, j) != j) return fail("solve: updates missing", sh, j, t.s, t.cb);
                m.P(t.s, j) = j + 1;
            } else {
                if (t.cb <= j || t.cb >= sh.npb || t.cb > (t.s >> 1)) return fail("block range", sh, j, t.s, t.cb);
                if (m.P(t.s, j) != j + 1) return fail("update: own strip not solved", sh, j, t.s, t.cb);
                for (int h = 0; h < 2; ++h) {
                    const int sb = 2 * t.cb + h;
                    if (sb < sh.R && m.P(sb, j) != j + 1) return fail("update: rows of the column block not solved", sh, j, t.s, t.cb);
                }
                if (m.P(t.s, t.cb) != j) return fail("update: previous update missing", sh, j, t.s, t.cb);
                m.P(t.s, t.cb) = j + 1;
            }
            // list order: solves by strip, then updates by (strip, block)
            if (t.cb < 0) {
                if (prev_cb >= 0 || t.s != prev_s + 1 && prev_s >= 0) return fail("solve order", sh, j, t.s, t.cb);
            } else if (prev_cb >= 0) {
                if (!(t.s > prev_s || (t.s == prev_s && t.cb == prev_cb + 1))) return fail("update order", sh, j, t.s, t.cb);
            }
            prev_s = t.s;
            prev_cb = t.cb;
        }
    }
    m.run_chain();
    if (!m.chain_done) return fail("chain stuck", sh, m.chain_j, m.chain_phase, 0);
    // final state
    for (int s = 0; s < sh.R; ++s)
        for (int cb = 0; cb < sh.npb && cb <= (s >> 1); ++cb) {
            const bool diag = (s >> 1) == cb;
            int want;
            if (cb < sh.nd) want = diag ? cb : cb + 1;                 // below a factorised block: solved; the block itself: all updates
            else want = sh.nd;                                        // not factorised here: the updates of the nd steps
            if (diag && cb >= 1 && cb <= sh.nd) want = cb;            // diagonal blocks: updated by the chain itself (modelled as progress = cb)
            if (m.P(s, cb) != want) {
                std::printf("FAIL final state: R=%d npb=%d nd=%d  piece s=%d cb=%d has %d, expected %d\n", sh.R, sh.npb, sh.nd, s, cb,
                            m.P(s, cb), want);
                return 1;
            }
        }
    return 0;
}

int main() {
    int bad = 0;
    long shapes = 0;
    for (int npb = 1; npb <= 20 && !bad; ++npb)
        for (int extra = 0; extra <= 70 && !bad; ++extra) {       // strips below the panel's last block (0: the panel reaches the last row)
            for (int ragged = 0; ragged < 2 && !bad; ++ragged) {  // last block of the panel holds one strip only
                PipeShape sh;
                sh.npb = npb;
                sh.R = 2 * npb - (ragged && extra == 0 ? 1 : 0) + extra;
                if (sh.R < 1) continue;
                for (int nd = (npb > 1 ? npb - 1 : npb); nd <= npb && !bad; ++nd) {
                    if (nd < 1) continue;
                    sh.nd = nd;
                    bad |= check(sh);
                    ++shapes;
                }
            }
        }
    PipeShape big{256, 64, 64};
    bad |= check(big);
    PipeShape tall{512, 8, 8};
    bad |= check(tall);
    if (!bad) std::printf("pipe_check: %ld shapes OK\n", shapes + 2);
    return bad;
}
