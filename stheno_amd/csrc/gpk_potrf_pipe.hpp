// gpk_potrf_pipe.hpp -- task list of the pipelined panel factorisation (potrf_pipe_kernel in gpk_potrf.hip).
//
// The panel: the columns [c0, c0 + w) of an n x n matrix, rows c0 .. n-1, all updates from the columns left of c0 applied.
// Coordinates below are relative to (c0, c0): 128-column blocks b = 0 .. npb-1, strips s = 0 .. R-1 of 64 rows (block b = strips
// 2b, 2b+1).  ONE workgroup (the chain) walks down the diagonal: factorise block j, solve the 128 rows of block j+1 against it,
// apply their rank-128 update to diagonal block j+1, factorise that, ...  Every other workgroup is a worker and takes the rest of
// the right-looking sweep as TASKS from an atomic counter, in the order of this file:
//
//   step j = 0 .. nd-1 (nd = the diagonal blocks the chain factorises):
//     solve(j, s)        L[s, j] = A[s, j] inv(L_jj)^T                    s = s0(j) .. R-1
//     update(j, s, cb)   A[s, cb] -= L[s, j] L[cb rows, j]^T               s = s0(j) .. R-1,  cb = j+1 .. min(s >> 1, npb-1)
//   with s0(j) = 2 (j+1) + (j+1 < npb ? 2 : 0): the two strips of block j+1 belong to the chain while block j+1 is in the panel.
//
// Dependencies travel through one progress word per (strip, block): the number of rank-128 updates applied to that piece, and
// (block + 1) once it is final (solved).  solve(j, s) waits for the chain's inverse of block j and for progress[s][j] == j;
// update(j, s, cb) for progress == j + 1 of [s][j] and of the strips of block cb in column j, and for progress[s][cb] == j.
// Every dependency of a task is a task EARLIER in the list or a chain step <= its own, and chain step j only needs tasks of the
// steps < j: workers that take their tasks in list order can never all be waiting (the earliest unfinished task is always
// runnable), whatever part of the grid is resident.
#pragma once

#if defined(__HIPCC__)
#define GPK_HD __host__ __device__ __forceinline__
#else
#define GPK_HD inline
#endif

#define GPK_PIPE_MAX_BLOCKS 64        // npb <= this (step offsets live in LDS)
#define GPK_PIPE_CTRL_HEAD 96         // control words before the progress array: [0] task counter, [1] abort, [16 + j] chain flags
#define GPK_PIPE_STRIP 64

struct PipeShape {
    int R;      // strips of 64 rows (the last may be ragged)
    int npb;    // 128-column blocks of the panel
    int nd;     // diagonal blocks the chain factorises (npb, or npb - 1 when the last one is left to a separate launch)
};

struct PipeTask {
    int j, s, cb;   // cb < 0: solve(j, s)
};

GPK_HD int pipe_first_strip(const PipeShape& sh, int j) { return 2 * (j + 1) + ((j + 1 < sh.npb) ? 2 : 0); }

GPK_HD int pipe_solves(const PipeShape& sh, int j) {
    const int s0 = pipe_first_strip(sh, j);
    return sh.R > s0 ? sh.R - s0 : 0;
}

// updates of strip s in step j: column blocks j+1 .. min(s >> 1, npb - 1)
GPK_HD int pipe_updates_of_strip(const PipeShape& sh, int j, int s) {
    if (j + 1 >= sh.npb) return 0;
    const int top = (s >> 1) < sh.npb - 1 ? (s >> 1) : sh.npb - 1;
    return top > j ? top - j : 0;
}

GPK_HD int pipe_updates(const PipeShape& sh, int j) {
    if (j + 1 >= sh.npb) return 0;
    const int s0 = pipe_first_strip(sh, j);
    int cnt = 0;
    // strips of the blocks below the panel's last block all take npb - 1 - j updates; the ones inside the panel are counted one by one
    const int flat = 2 * (sh.npb - 1);
    int s = s0;
    for (; s < sh.R && s < flat; ++s) cnt += pipe_updates_of_strip(sh, j, s);
    if (s < sh.R) cnt += (sh.R - s) * (sh.npb - 1 - j);
    return cnt;
}

GPK_HD int pipe_step_tasks(const PipeShape& sh, int j) { return pipe_solves(sh, j) + pipe_updates(sh, j); }

// task q (0 <= q < pipe_step_tasks) of step j
GPK_HD PipeTask pipe_decode(const PipeShape& sh, int j, int q) {
    PipeTask t;
    t.j = j;
    const int s0 = pipe_first_strip(sh, j);
    const int nsolve = pipe_solves(sh, j);
    if (q < nsolve) {
        t.s = s0 + q;
        t.cb = -1;
        return t;
    }
    int u = q - nsolve;
    const int flat = 2 * (sh.npb - 1);
    int s = s0;
    for (; s < sh.R && s < flat; ++s) {
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

GPK_HD int pipe_ctrl_words(const PipeShape& sh) { return GPK_PIPE_CTRL_HEAD + sh.R * sh.npb; }
