// gpk_gemm_stream.hpp -- a workgroup's SEQUENCE of 128 x 128 output tiles as one uninterrupted operand pipeline (round 4).
//
// gemm_tile (gpk_gemm_tile.hpp) computes one tile: request C, fill the operand pipeline, k loop, drain, store.  With two workgroups
// per CU the fill / drain / store phases of one tile are covered by the other workgroup's k loop -- but a lone fp64 wave issues an
// MFMA only every ~140 cycles (profiles/r03_experiments.md section 1), so the pipe runs at 46 % while one of the two is outside its k
// loop, and at K = 1024 a tile is outside for ~15 of its ~270 us; at the K <= 512 of the batched factorisations for a quarter of
// its time.  Here the workgroup knows its NEXT tile while it works on the current one:
//   * the k loop is the software-pipelined one of gemm_tile (global loads two chunks ahead, LDS writes one chunk ahead, fragment
//     reads one phase ahead, one barrier per chunk inside the MFMA stream), and its load side simply runs on INTO the next tile:
//     the last two chunk bodies of a tile request chunks 0 and 1 of the next one, write chunk 0 to LDS and read its first
//     fragments -- the next tile starts with a full pipeline, there is no fill and no drain between tiles;
//   * the stores of a finished tile are interleaved with the loads of the next tile's C values into the registers they free.
// What is left between the last MFMA of a tile and the first of the next is one pass of stores + loads at the memory system's pace.
//
// A tile can ride in the stream ("streamable") when it needs no bounds checks, 16-byte loads are legal and it has an even number
// (>= 2) of k-chunks, so that every tile starts in LDS stage 0.  Everything else goes through gemm_tile.
#pragma once
#include "gpk_gemm_tile.hpp"

namespace {

// A task of a workgroup's scheduler.  Plain integers, all WAVE-UNIFORM (the schedulers pass them through readfirstlane: tile decoding
// goes through a float square root, and values the compiler cannot prove uniform would put the whole load side -- pointers, chunk
// counters, the jump to the next tile -- on the vector ALU behind divergent branches).
struct TileRef {
    int seg;                  // which GemmArgs of the launch (Sched::args(seg))
    int ti, tj;
    int b, b2;                // batch indices
    int task;                 // the scheduler's own handle (e.g. the persistent kernel's task index)
    int kind;                 // 0: none, 1: streamable, 2: a task for the legacy path
    __device__ __forceinline__ void uniform() {
        seg = __builtin_amdgcn_readfirstlane(seg); ti = __builtin_amdgcn_readfirstlane(ti); tj = __builtin_amdgcn_readfirstlane(tj);
        b = __builtin_amdgcn_readfirstlane(b); b2 = __builtin_amdgcn_readfirstlane(b2);
        task = __builtin_amdgcn_readfirstlane(task); kind = __builtin_amdgcn_readfirstlane(kind);
    }
};

constexpr int STREAM_SMEM = 2 * 2 * op_bytes(128);     // two stages of two operand tiles

template <typename T>
__device__ __forceinline__ void tile_k_range(const GemmArgs<T>& p, int m0, int n0, int& kc0, int& nk) {
    constexpr int BK = Traits<T>::BK;
    nk = (p.K + BK - 1) / BK;
    if (p.tri_k_lo) nk = min(nk, (m0 + 128 + BK - 1) / BK);
    if (p.tri_k_lo_b) nk = min(nk, (n0 + 128 + BK - 1) / BK);
    kc0 = p.tri_k ? m0 / BK : 0;
    if (kc0 > nk - 1) kc0 = nk > 0 ? nk - 1 : 0;
}

template <typename T, bool EDGE>
__device__ __forceinline__ bool tile_streamable(const GemmArgs<T>& p, int ti, int tj) {
    constexpr int BK = Traits<T>::BK;
    const int m0 = ti * 128, n0 = tj * 128;
    int kc0, nk;
    tile_k_range(p, m0, n0, kc0, nk);
    const int n = nk - kc0;
    bool ok = n >= 2 && (n & 1) == 0;
    if (EDGE) ok = ok && p.vec_ok && (p.K % BK == 0) && m0 + 128 <= p.M && n0 + 128 <= p.N;
    return ok;
}

// Sched: args(seg) -> const GemmArgs<T>& (kernel arguments: scalar loads); prefetch() -- called at the start of a tile (may start an asynchronous claim); resolve(smem) -> TileRef -- called once per
// tile at its end by ALL threads, returns the task after the one already known as `nxt` (kind 0 if there is none).
// On entry `cur` is streamable; on return `cur` / `nxt` are the first two tasks NOT done (kinds 0 / 2 possible).
template <typename T, bool A_KMAJ, bool B_KMAJ, bool EDGE, typename Sched>
__device__ __forceinline__ void gemm_stream(TileRef& cur, TileRef& nxt, Sched& sched, char* smem) {
    typedef typename Traits<T>::acc_t acc_t;
    typedef typename Traits<T>::vec_t vec_t;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef typename std::conditional<sizeof(T) == 8, double, f32x2>::type frag_t;
    constexpr int TS = 128, NT = 256;
    constexpr int BK = Traits<T>::BK;
    constexpr int VEC_ = Traits<T>::VEC;
    constexpr int NV = TS * 8 / NT;      // 16-byte vectors a thread moves per operand tile and k-chunk (4)
    constexpr int FR = 4, FRM = 4;       // 16 x 16 fragments per wave: 64 x 64 of the tile
    constexpr int WT = 64, WTM = 64;
    constexpr int OPB = op_bytes(TS), STAGE = 2 * OPB;
    constexpr int NSL = FRM * FR / 2;    // slices of a phase's MFMAs
    static_assert(NSL == 8 && FRM + FR == 8 && 2 * NV == 8, "eight slices, eight fragments, eight vectors per phase");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, kq = lane >> 4;
    const int swz = (lr >> 1) & 7;

    acc_t acc[FRM][FR];
    frag_t Fa[2][FRM], Fb[2][FR];
    vec_t qa[NV], qb[NV];
    const T* pa[NV];
    const T* pb[NV];

    // ---- the load side: a pointer set that walks the k-chunks of one tile and then jumps to the next tile ----
    int irem = 0;                        // chunks of the tile being loaded that are still to be requested (incl. the one pointed at)
    int64_t stepA = 0, stepB = 0, curA = 0, curB = 0;
    TileRef inext;                       // where the load side goes when irem reaches 0 (kind 1) -- or nowhere
    inext.kind = 0;
    auto repoint = [&](const TileRef& t) {
        const GemmArgs<T>& p = sched.args(t.seg);
        const int m0 = t.ti * TS, n0 = t.tj * TS;
        int kc0, nk;
        tile_k_range(p, m0, n0, kc0, nk);
        const T* A = p.A + t.b * p.sA + t.b2 * p.sA2;
        const T* B = p.B + t.b * p.sB + t.b2 * p.sB2;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (A_KMAJ) {
                pa[i] = A + (int64_t)(m0 + (tid >> 3) + (NT / 8) * i) * p.lda + (int64_t)kc0 * BK + (tid & 7) * VEC_;
            } else {
                constexpr int CPR = TS / VEC_;
                const int id = tid + NT * i;
                pa[i] = A + ((int64_t)kc0 * BK + id / CPR) * p.lda + m0 + (id % CPR) * VEC_;
            }
            if (B_KMAJ) {
                pb[i] = B + (int64_t)(n0 + (tid >> 3) + (NT / 8) * i) * p.ldb + (int64_t)kc0 * BK + (tid & 7) * VEC_;
            } else {
                constexpr int CPR = TS / VEC_;
                const int id = tid + NT * i;
                pb[i] = B + ((int64_t)kc0 * BK + id / CPR) * p.ldb + n0 + (id % CPR) * VEC_;
            }
        }
        stepA = A_KMAJ ? (int64_t)BK : (int64_t)BK * p.lda;
        stepB = B_KMAJ ? (int64_t)BK : (int64_t)BK * p.ldb;
        irem = nk - kc0;
    };
    // before the eight vector loads of a chunk: jump to the next tile if this one is exhausted; the pointers move on behind the loads
    // unless they sit on the last chunk of the last tile (which is then simply read again: nobody uses it, nothing is read out of range)
    auto arm = [&]() {
        if (irem == 0 && inext.kind == 1) {
            repoint(inext);
            inext.kind = 0;
        }
        curA = irem > 1 ? stepA : 0;
        curB = irem > 1 ? stepB : 0;
        if (irem > 0) --irem;
    };
    auto g_issue1 = [&](int j) {
        if (j < NV) { qa[j] = *reinterpret_cast<const vec_t*>(pa[j]); pa[j] += curA; }
        else { qb[j - NV] = *reinterpret_cast<const vec_t*>(pb[j - NV]); pb[j - NV] += curB; }
    };
    auto g_commit1 = [&](int stage, int j) {
        char* dA = smem + stage * STAGE;
        if (j < NV) sstore1<T, TS, A_KMAJ, NT>(dA, qa[j], tid, j);
        else sstore1<T, TS, B_KMAJ, NT>(dA + OPB, qb[j - NV], tid, j - NV);
    };
    auto frag = [&](auto kmaj_c, const char* lds, int rowbase, int ph) -> frag_t {
        constexpr bool KMAJ = decltype(kmaj_c)::value;
        if constexpr (sizeof(T) == 8) {
            return fragread<T, TS, KMAJ>(lds, rowbase, lr, ph * 4 + kq, swz);
        } else {
            // fp32: two consecutive k values per lane (step 2 ph' + e of the chunk <-> k = 8 ph + 2 kq + e, the same permutation of the
            // contraction order for both operands)
            if constexpr (KMAJ) {
                const int u = ph * 4 + kq;
                return *reinterpret_cast<const f32x2*>(lds + (rowbase + lr) * 128 + ((((u >> 1) ^ swz) << 4) + (u & 1) * 8));
            } else {
                const int k = ph * 8 + 2 * kq;
                f32x2 v;
                v[0] = fragread<T, TS, false>(lds, rowbase, lr, k, swz);
                v[1] = fragread<T, TS, false>(lds, rowbase, lr, k + 1, swz);
                return v;
            }
        }
    };
    auto f_read1 = [&](auto set_c, int stage, int ph, int j) {
        constexpr int set = decltype(set_c)::value;
        const char* sA = smem + stage * STAGE;
        if (j < FRM) Fa[set][j] = frag(std::integral_constant<bool, A_KMAJ>{}, sA, wm * WTM + j * 16, ph);
        else Fb[set][j - FRM] = frag(std::integral_constant<bool, B_KMAJ>{}, sA + OPB, wn * WT + (j - FRM) * 16, ph);
    };
    auto f_mma1 = [&](auto set_c, int q) {
        constexpr int set = decltype(set_c)::value;
        const int fj = q >> 1, f0 = (q & 1) * 2;
        if constexpr (sizeof(T) == 8) {
            acc[f0][fj] = Traits<T>::mfma(Fa[set][f0], Fb[set][fj], acc[f0][fj]);
            acc[f0 + 1][fj] = Traits<T>::mfma(Fa[set][f0 + 1], Fb[set][fj], acc[f0 + 1][fj]);
        } else {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                acc[f0][fj] = Traits<T>::mfma((T)Fa[set][f0][e], (T)Fb[set][fj][e], acc[f0][fj]);
                acc[f0 + 1][fj] = Traits<T>::mfma((T)Fa[set][f0 + 1][e], (T)Fb[set][fj][e], acc[f0 + 1][fj]);
            }
        }
    };
    typedef std::integral_constant<int, 0> F0;
    typedef std::integral_constant<int, 1> F1;
    // One chunk (see gemm_tile's pipelined loop): LDS stage `stage` holds it, fragment set 0 its phase 0, qa / qb the chunk after it.
    auto chunk = [&](int stage) {
        arm();
#pragma unroll
        for (int q = 0; q < NSL; ++q) {          // phase 0: multiply set 0, read phase 1 into set 1, write the next chunk
            f_mma1(F0{}, q);
            f_read1(F1{}, stage, 1, q);
            g_commit1(stage ^ 1, q);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < NSL; ++q) {          // phase 1: multiply set 1, read phase 2 into set 0, request the chunk after next
            f_mma1(F1{}, q);
            f_read1(F0{}, stage, 2, q);
            g_issue1(q);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < NSL; ++q) {          // phase 2: multiply set 0, read phase 3 into set 1
            f_mma1(F0{}, q);
            f_read1(F1{}, stage, 3, q);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < NSL / 2; ++q) f_mma1(F1{}, q);     // phase 3, first half
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave has read this stage and written the other
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = NSL / 2; q < NSL; ++q) {    // phase 3, second half: read phase 0 of the next chunk into set 0
            f_mma1(F1{}, q);
            f_read1(F0{}, stage ^ 1, 0, 2 * (q - NSL / 2));
            f_read1(F0{}, stage ^ 1, 0, 2 * (q - NSL / 2) + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- C side ----
    auto c_index = [&](const GemmArgs<T>& p, int m0, int n0, int fi, int fj, int i, int64_t ld) -> int64_t {
        const int row = m0 + wm * WTM + fi * 16 + Traits<T>::crow(lane, i);
        const int col = n0 + wn * WT + fj * 16 + lr;
        return (int64_t)row * ld + col;
    };
    auto load_c = [&](const TileRef& t, int fi, int fj) {          // raw C values of fragment (fi, fj) of tile t (if it reads C)
        const GemmArgs<T>& p = sched.args(t.seg);
        if (!p.has_beta) return;
        const T* Cin = p.Cin + t.b * p.sC + t.b2 * p.sC2;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[fi][fj][i] = Cin[c_index(p, t.ti * TS, t.tj * TS, fi, fj, i, p.ldcin)];
    };

    // ---- cold start: C values and the first two chunks of `cur` ----
    repoint(cur);
#pragma unroll
    for (int fi = 0; fi < FRM; ++fi)
#pragma unroll
        for (int fj = 0; fj < FR; ++fj) load_c(cur, fi, fj);
    arm();
#pragma unroll
    for (int j = 0; j < 2 * NV; ++j) g_issue1(j);
#pragma unroll
    for (int j = 0; j < 2 * NV; ++j) g_commit1(0, j);
    __syncthreads();
    arm();
#pragma unroll
    for (int j = 0; j < 2 * NV; ++j) g_issue1(j);
#pragma unroll
    for (int j = 0; j < FRM + FR; ++j) f_read1(F0{}, 0, 0, j);

    while (true) {
        const GemmArgs<T>& p = sched.args(cur.seg);
        const int m0 = cur.ti * TS, n0 = cur.tj * TS;
        int kc0, nk;
        tile_k_range(p, m0, n0, kc0, nk);
        sched.prefetch();
        // accumulators: C * (beta / alpha), so that the epilogue is a pure store of alpha * acc
#pragma unroll
        for (int fi = 0; fi < FRM; ++fi)
#pragma unroll
            for (int fj = 0; fj < FR; ++fj)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[fi][fj][i] = p.has_beta ? acc[fi][fj][i] * p.beta_over_alpha : T(0);
        inext = nxt;          // where the load side goes after `cur` (it is inside `cur` now: cold start, or it jumped here two chunks ago)
        for (int c = nk - kc0; c > 0; c -= 2) {
            chunk(0);
            chunk(1);
        }
        // ---- tile done: its successor's pipeline is full if it is streamable (kind 1) ----
        const bool go_on = nxt.kind == 1;
        TileRef nn = sched.resolve(smem);
        T* C = p.C + cur.b * p.sC + cur.b2 * p.sC2;
#pragma unroll
        for (int fi = 0; fi < FRM; ++fi)
#pragma unroll
            for (int fj = 0; fj < FR; ++fj) {
#pragma unroll
                for (int i = 0; i < 4; ++i) C[c_index(p, m0, n0, fi, fj, i, p.ldc)] = p.alpha * acc[fi][fj][i];
                if (go_on) load_c(nxt, fi, fj);                  // the next tile's C values into the registers just stored
            }
        cur = nxt;
        nxt = nn;
        if (!go_on) break;
    }
    // (every wave has passed a barrier after its last LDS read of operands it needed; the fragments read for a successor that
    // does not exist are not used)
    __syncthreads();
}

}  // namespace
