// pipe_check.cpp -- host-side check of the task list of the pipelined panel factorisation (gpk_potrf_pipe.hpp): plays the list
// against a model of the chain workgroup, of the progress words and of the counters, for many panel shapes, and verifies that
//   * every task finds its dependencies satisfied by EARLIER tasks and by chain steps that can have run by then (so workers that
//     take tasks in list order cannot deadlock), and every progress word moves exactly j -> j + 1;
//   * at the end every piece of the panel has received exactly the updates of the right-looking sweep, every piece below a
//     factorised diagonal block has been solved, every diagonal block has received all its tiles and the chain has finished.
// Plain C++ (g++ pipe_check.cpp && ./a.out); run by tests/test_pipe_tasks.py.  Test infrastructure, not part of libgpk.so.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gpk_potrf_pipe.hpp"

static int fail(const char* what, const PipeShape& sh, const PipeTask& t) {
    std::printf("FAIL %s: R=%d R32=%d npb=%d nd=%d  task kind=%d j=%d s=%d cb=%d\n", what, sh.R, sh.R32, sh.npb, sh.nd, t.kind, t.j, t.s, t.cb);
    return 1;
}

struct Model {
    PipeShape sh;
    std::vector<int> prog, cnt;  // [s][cb]
    std::vector<int> dflag;      // inverse of block j published
    int chain_j = 0;             // next block of the chain
    int& P(int s, int cb) { return prog[(size_t)s * sh.npb + cb]; }
    int& C(int s, int cb) { return cnt[(size_t)s * sh.npb + cb]; }
    void run_chain() {           // let the chain run as far as its dependencies allow
        while (chain_j < sh.nd) {
            const int j = chain_j;
            if (j > 0 && C(2 * j, j) != pipe_xupdates(pipe_fine_strips(sh, j - 1))) break;
            dflag[j] = 1;
            ++chain_j;
        }
    }
};

static int check(const PipeShape& sh) {
    Model m;
    m.sh = sh;
    m.prog.assign((size_t)sh.R * sh.npb, 0);
    m.cnt.assign((size_t)sh.R * sh.npb, 0);
    m.dflag.assign(sh.npb, 0);
    const int nseg = pipe_num_segments(sh);
    long total = 0;
    for (int k = 0; k < nseg; ++k) {
        const int nt = pipe_segment_tasks(sh, k);
        for (int q = 0; q < nt; ++q) {
            const PipeTask t = pipe_decode(sh, k, q);
            ++total;
            const int j = t.j;
            if (j < 0 || j >= sh.nd) return fail("step range", sh, t);
            m.run_chain();
            const int s1 = 2 * (j + 1);
            if (t.kind == PIPE_SOLVE) {
                if (t.s < pipe_first_strip(sh, j) || t.s >= sh.R) return fail("strip range", sh, t);
                if (!m.dflag[j]) return fail("solve before the inverse", sh, t);
                if (m.P(t.s, j) != j) return fail("solve: updates missing", sh, t);
                m.P(t.s, j) = j + 1;
            } else if (t.kind == PIPE_UPDATE) {
                if (t.s < pipe_first_strip(sh, j) || t.s >= sh.R) return fail("strip range", sh, t);
                if (t.cb <= j || t.cb >= sh.npb || t.cb > (t.s >> 1)) return fail("block range", sh, t);
                if (m.P(t.s, j) != j + 1) return fail("update: own strip not solved", sh, t);
                for (int h = 0; h < 2; ++h) {
                    const int sb = 2 * t.cb + h;
                    if (sb < sh.R && m.P(sb, j) != j + 1) return fail("update: rows of the column block not solved", sh, t);
                }
                if (m.P(t.s, t.cb) != j) return fail("update: previous update missing", sh, t);
                m.P(t.s, t.cb) = j + 1;
            } else if (t.kind == PIPE_XSOLVE) {
                const int nq = pipe_fine_strips(sh, j);
                if (t.s < 0 || t.s >= nq) return fail("fine strip range", sh, t);
                const int s = s1 + (t.s >> 1);
                if (s >= sh.R) return fail("fine strip beyond the matrix", sh, t);
                if (!m.dflag[j]) return fail("xsolve before the inverse", sh, t);
                if (m.P(s, j) != j) return fail("xsolve: updates missing", sh, t);
                if (++m.C(s, j) == pipe_xsolves_in_strip(nq, t.s >> 1)) m.P(s, j) = j + 1;
            } else {
                const int nq = pipe_fine_strips(sh, j);
                if (t.s < 0 || t.s >= nq || t.cb < 0 || t.cb > (t.s >> 1)) return fail("tile range", sh, t);
                for (int h = 0; h < 2; ++h)
                    if (s1 + h < sh.R && m.P(s1 + h, j) != j + 1) return fail("xupdate: rows of the block not solved", sh, t);
                const int s = s1 + (t.s >> 1);
                if (m.P(s, j + 1) != j) return fail("xupdate: previous update missing", sh, t);
                ++m.C(s1, j + 1);
            }
        }
    }
    m.run_chain();
    if (m.chain_j != sh.nd) {
        std::printf("FAIL chain stuck at block %d: R=%d npb=%d nd=%d\n", m.chain_j, sh.R, sh.npb, sh.nd);
        return 1;
    }
    // final state
    for (int j = 0; j + 1 < sh.npb && j < sh.nd; ++j)
        if (m.C(2 * (j + 1), j + 1) != pipe_xupdates(pipe_fine_strips(sh, j))) {
            std::printf("FAIL diagonal block %d incomplete: R=%d npb=%d nd=%d\n", j + 1, sh.R, sh.npb, sh.nd);
            return 1;
        }
    for (int s = 0; s < sh.R; ++s)
        for (int cb = 0; cb < sh.npb && cb <= (s >> 1); ++cb) {
            const bool diag = (s >> 1) == cb;
            int want;
            if (diag) want = cb >= 1 ? (cb - 1 < sh.nd ? cb - 1 : sh.nd) : 0;      // the updates of the steps < cb - 1 are tasks of the list, the last one arrives in tiles
            else if (cb < sh.nd) want = cb + 1;                                      // below a factorised block: solved
            else want = sh.nd;                                                       // not factorised here: the updates of the nd steps
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
        for (int extra = 0; extra <= 70 && !bad; ++extra)          // fine strips below the panel's last block (0: the panel reaches the last row)
            for (int cut = 0; cut < 4 && !bad; ++cut) {            // fine strips missing from the last block of the matrix
                PipeShape sh;
                sh.npb = npb;
                sh.R32 = 4 * npb + extra - cut;
                if (extra > 0 && cut > 0 && extra < 4) continue;   // (a ragged end inside the first block below the panel is covered by extra alone)
                if (sh.R32 < 4 * (npb - 1) + 1) continue;          // the last block of the panel holds at least one row
                sh.R = (sh.R32 + 1) / 2;
                for (int nd = (npb > 1 ? npb - 1 : npb); nd <= npb && !bad; ++nd) {
                    if (nd < 1) continue;
                    if (nd == npb - 1 && extra > 0) continue;      // the last block is only left out when the panel reaches the last row
                    sh.nd = nd;
                    bad |= check(sh);
                    ++shapes;
                }
            }
    PipeShape big{256, 512, 64, 64};
    bad |= check(big);
    PipeShape tall{512, 1024, 8, 8};
    bad |= check(tall);
    if (!bad) std::printf("pipe_check: %ld shapes OK\n", shapes + 2);
    return bad;
}
