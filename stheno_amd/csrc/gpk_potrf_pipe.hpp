// gpk_potrf_pipe.hpp -- task list of the pipelined panel factorisation (potrf_pipe_kernel in gpk_potrf.hip).
//
// The panel: the columns [c0, c0 + w) of an n x n matrix, rows c0 .. n-1, all updates from the columns left of c0 applied.
// Coordinates below are relative to (c0, c0): 128-column blocks b = 0 .. npb-1, strips s = 0 .. R-1 of 64 rows (block b = strips
// 2b, 2b+1), fine strips of 32 rows inside a diagonal block (q = 0..3).  ONE workgroup (the chain) factorises and inverts the
// diagonal blocks 0 .. nd-1, one after the other; every other workgroup is a worker and takes the rest of the right-looking sweep
// as TASKS from an atomic counter, in the order of this file.  Step j (block column j) consists of
//
//   critical tasks -- what diagonal block j+1 waits for, cut fine (32 rows) because their latency is the chain's:
//     xsolve(j, q)      L[q, j] = A[q, j] inv(L_jj)^T             the fine strips q of block j+1
//     xupdate(j, q, h)  A[q, h] -= L[q, j] L[rows of h, j]^T       the lower 32 x 64 tiles of diagonal block j+1 (h = column half)
//   the rest, 64 rows x 128 columns per task, strips s >= s0(j) = 2 (j+1) + (j+1 < npb ? 2 : 0):
//     solve(j, s)       L[s, j] = A[s, j] inv(L_jj)^T
//     update(j, s, cb)  A[s, cb] -= L[s, j] L[rows of cb, j]^T     cb = j+1 .. min(s >> 1, npb-1)
//   split in two parts: A(j) = the strips of block j+2 (what the critical tasks of step j+1 wait for), B(j) = everything below.
//
// List order:  crit(0) | A(0) crit(1) | B(0) A(1) crit(2) | B(1) A(2) crit(3) | ...  | B(nd-1)
// so the critical tasks of step j+1 are taken BEFORE the bulk of step j: the chain runs one step ahead of the sweep (look-ahead of
// depth 1), the workers that hold them simply wait for the chain.  Dependencies travel through one progress word per (strip,
// block) -- the number of rank-128 updates applied to the piece, block + 1 once it is solved -- plus one counter word per
// (strip, block) for what several tasks complete together (the fine strips of a strip; the tiles of a diagonal block).  Every
// dependency of a task is a task EARLIER in the list or a chain step that only needs earlier tasks, so workers that take their
// tasks in list order can never all be waiting (the earliest unfinished task is always runnable), whatever part of the grid is
// resident.  pipe_check.cpp plays the list against a model of these rules.
#pragma once

#if defined(__HIPCC__)
#define GPK_HD __host__ __device__ __forceinline__
#else
#define GPK_HD inline
#endif

#define GPK_PIPE_MAX_BLOCKS 64        // npb <= this
#define GPK_PIPE_MAX_SEGS (3 * GPK_PIPE_MAX_BLOCKS + 2)
#define GPK_PIPE_CTRL_HEAD 96         // control words before the progress arrays: [0] task counter, [1] abort, [2] fill-tile counter, [3] role ticket (first workgroup to arrive = the chain), [16 + j] inverse of block j published
#define GPK_PIPE_STRIP 64
#define GPK_PIPE_FINE 32

struct PipeShape {
    int R;      // strips of 64 rows (the last may be ragged)
    int R32;    // fine strips of 32 rows
    int npb;    // 128-column blocks of the panel
    int nd;     // diagonal blocks the chain factorises (npb, or npb - 1 when the last one is left to a separate launch)
};

enum { PIPE_SOLVE = 0, PIPE_UPDATE = 1, PIPE_XSOLVE = 2, PIPE_XUPDATE = 3 };
struct PipeTask {
    int kind, j, s, cb;   // solve: strip s; update: strip s, block cb; xsolve: fine strip s = q; xupdate: fine strip s = q, half cb = h
};

// ---- critical tasks of step j ----
GPK_HD int pipe_fine_strips(const PipeShape& sh, int j) {      // fine strips of block j+1 (0 if the block is not in the panel)
    if (j + 1 >= sh.npb) return 0;
    const int left = sh.R32 - 4 * (j + 1);
    return left < 0 ? 0 : (left > 4 ? 4 : left);
}
GPK_HD int pipe_xupdates(int nq) { return nq <= 0 ? 0 : (nq == 1 ? 1 : (nq == 2 ? 2 : (nq == 3 ? 4 : 6))); }    // sum over q < nq of 1 + (q >> 1)
GPK_HD int pipe_crit_tasks(const PipeShape& sh, int j) {
    const int nq = pipe_fine_strips(sh, j);
    return nq + pipe_xupdates(nq);
}
GPK_HD int pipe_xsolves_in_strip(int nq, int h) {              // fine strips of block j+1 inside its 64-row strip h
    const int left = nq - 2 * h;
    return left < 0 ? 0 : (left > 2 ? 2 : left);
}

// ---- the rest of step j ----
GPK_HD int pipe_first_strip(const PipeShape& sh, int j) { return 2 * (j + 1) + ((j + 1 < sh.npb) ? 2 : 0); }

// updates of strip s in step j: column blocks j+1 .. min(s >> 1, npb - 1)
GPK_HD int pipe_updates_of_strip(const PipeShape& sh, int j, int s) {
    if (j + 1 >= sh.npb) return 0;
    const int top = (s >> 1) < sh.npb - 1 ? (s >> 1) : sh.npb - 1;
    return top > j ? top - j : 0;
}

// strips [lo, hi) of step j (lo >= pipe_first_strip; clipped to R)
GPK_HD int pipe_range_solves(const PipeShape& sh, int lo, int hi) {
    if (hi > sh.R) hi = sh.R;
    return hi > lo ? hi - lo : 0;
}
GPK_HD int pipe_range_updates(const PipeShape& sh, int j, int lo, int hi) {
    if (j + 1 >= sh.npb) return 0;
    if (hi > sh.R) hi = sh.R;
    int cnt = 0;
    // strips of the blocks below the panel's last block all take npb - 1 - j updates; the ones inside the panel are counted one by one
    const int flat = 2 * (sh.npb - 1);
    int s = lo;
    for (; s < hi && s < flat; ++s) cnt += pipe_updates_of_strip(sh, j, s);
    if (s < hi) cnt += (hi - s) * (sh.npb - 1 - j);
    return cnt;
}
GPK_HD int pipe_range_tasks(const PipeShape& sh, int j, int lo, int hi) {
    return pipe_range_solves(sh, lo, hi) + pipe_range_updates(sh, j, lo, hi);
}
// task q of the strips [lo, hi) of step j: the solves by strip, then the updates by (strip, block)
GPK_HD PipeTask pipe_decode_range(const PipeShape& sh, int j, int lo, int hi, int q) {
    PipeTask t;
    t.j = j;
    if (hi > sh.R) hi = sh.R;
    const int nsolve = pipe_range_solves(sh, lo, hi);
    if (q < nsolve) {
        t.kind = PIPE_SOLVE;
        t.s = lo + q;
        t.cb = -1;
        return t;
    }
    t.kind = PIPE_UPDATE;
    int u = q - nsolve;
    const int flat = 2 * (sh.npb - 1);
    int s = lo;
    for (; s < hi && s < flat; ++s) {
        const int c = pipe_updates_of_strip(sh, j, s);
        if (u < c) {
            t.s = s;
            t.cb = j + 1 + u;
            return t;
        }
        u -= c;
    }
    const int per = sh.npb - 1 - j;
    t.s = s + u / per;
    t.cb = j + 1 + u % per;
    return t;
}

// ---- the list: segments 0 .. 3 nd + 1 ----
//   segment 0 = crit(0);  segment 1 + 3 g + r:  r = 0: B(g - 1) (empty for g = 0),  r = 1: A(g),  r = 2: crit(g + 1)   (g = 0 .. nd - 1)
//   segment 3 nd + 1 = B(nd - 1)
GPK_HD int pipe_num_segments(const PipeShape& sh) { return 3 * sh.nd + 2; }
GPK_HD void pipe_segment(int k, int& part, int& j) {          // part 0 = B, 1 = A, 2 = crit
    if (k == 0) { part = 2; j = 0; return; }
    const int g = (k - 1) / 3, r = (k - 1) % 3;
    part = r;
    j = r == 0 ? g - 1 : (r == 1 ? g : g + 1);
}
GPK_HD int pipe_segment_tasks(const PipeShape& sh, int k) {
    int part, j;
    pipe_segment(k, part, j);
    if (j < 0 || j >= sh.nd) return 0;
    const int s0 = pipe_first_strip(sh, j);
    if (part == 2) return pipe_crit_tasks(sh, j);
    if (part == 1) return pipe_range_tasks(sh, j, s0, s0 + 2);
    return pipe_range_tasks(sh, j, s0 + 2, sh.R);
}
GPK_HD PipeTask pipe_decode(const PipeShape& sh, int k, int q) {
    int part, j;
    pipe_segment(k, part, j);
    const int s0 = pipe_first_strip(sh, j);
    if (part == 1) return pipe_decode_range(sh, j, s0, s0 + 2, q);
    if (part == 0) return pipe_decode_range(sh, j, s0 + 2, sh.R, q);
    PipeTask t;
    t.j = j;
    const int nq = pipe_fine_strips(sh, j);
    if (q < nq) {
        t.kind = PIPE_XSOLVE;
        t.s = q;
        t.cb = -1;
        return t;
    }
    const int u = q - nq;                 // (q, h): (0,0) (1,0) (2,0) (2,1) (3,0) (3,1)
    t.kind = PIPE_XUPDATE;
    t.s = u < 2 ? u : 2 + ((u - 2) >> 1);
    t.cb = u < 2 ? 0 : ((u - 2) & 1);
    return t;
}

GPK_HD int pipe_ctrl_words(const PipeShape& sh) { return GPK_PIPE_CTRL_HEAD + 2 * sh.R * sh.npb; }
