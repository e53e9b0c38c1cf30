// gpk_gemm.hip -- the one dense contraction of the GP hot path, on MFMA.
//
//   C[m][n] = alpha * sum_k a(m,k) * b(n,k) + beta * C[m][n]
//
// Used for: the Cholesky trailing SYRK update, the panel TRSM (multiplication by
// an inverted diagonal block), the strip update, the merge of diagonal-block
// inverses, the blocked TRSM behind posterior conditioning, and the SYRK of
// the inducing-point ELBO.  (Call sites in the reference that this replaces:
// LAPACK dpotrf/dtrsm/dgemm underneath B.cholesky / B.iqf / B.solve,
// stheno/random.py:274-276, stheno/model/observations.py:300-335.)
//
// Design (gfx950):
//  * 128x128 block tile, 256 threads = 4 waves in a 2x2 grid, each wave owns a
//    64x64 sub-tile = 4x4 MFMA 16x16x4 fragments (f64: 128 accumulator VGPRs).
//    __launch_bounds__(256, 2): two workgroups per CU, so one workgroup's
//    global->LDS staging overlaps the other's MFMA phase.
//  * K is consumed in 128-byte chunks (16 f64 / 32 f32), register-staged
//    (global_load_dwordx4 -> ds_write_b128) into a double-buffered LDS tile,
//    one barrier per chunk; the next chunk's global loads are issued before the
//    current chunk's MFMAs.
//  * LDS image of a k-contiguous operand: [128 rows][128 B], 16-byte chunks
//    XOR-swizzled with (row >> 1) & 7 so that the MFMA fragment reads
//    (16 rows x 2 k per 32-lane group) are bank-conflict-free for f64 and the
//    8-lane ds_write_b128 groups are conflict-free.  An operand stored with the
//    row index contiguous is staged as [BK][128 + 16] (pad keeps the two
//    k-rows of a 32-lane group on different bank halves).
//  * blockIdx -> tile mapping: plain row-major (triangular for the lower-only mode);
//    workgroup b runs on XCD b % 8, so consecutive tiles of a tile row spread over the 8
//    L2s.  Three other orders were built and measured in rounds 1-3 and are gone: an 8x8 super-tile per XCD (4 % SLOWER on the
//    Cholesky trailing update: ragged diagonal super-tiles unbalance the XCDs, the 256 MB MALL already serves the panel re-reads),
//    a row-pair order for triangular grids (no change) and a column-major walk for wide rectangular problems (cfg5: 87 vs 64 ms
//    per step although it cut the fabric-side re-reads of K_zx).  profiles/r01_experiments.md, r02_experiments.md.
//  * accumulators are initialised with C * (beta / alpha) so the epilogue is a
//    pure store of alpha * acc (exact for alpha = -1, beta = 1).
#include "gpk_common.hpp"
#include "gpk_gemm_tile.hpp"
#include <type_traits>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

__device__ __forceinline__ void tri_decode(int s, int& I, int& J) {
    int i = (int)((sqrtf(8.f * (float)s + 1.f) - 1.f) * 0.5f);
    while ((i + 1) * (i + 2) / 2 <= s) ++i;
    while (i * (i + 1) / 2 > s) --i;
    I = i;
    J = s - i * (i + 1) / 2;
}

template <typename T>
__device__ __forceinline__ bool decode_tile(const GemmArgs<T>& p, int bid, int& ti, int& tj) {
    const bool tri = p.lower_only && p.tiles_m == p.tiles_n;   // square: triangular enumeration
    if (tri) {
        tri_decode(bid, ti, tj);
        return ti < p.tiles_m;
    }
    if (p.tri_k_lo_b) {   // B lower triangular in k: column tile tj runs tj + 1 blocks of k -- longest columns first
        tj = p.tiles_n - 1 - bid / p.tiles_m;
        ti = bid - (bid / p.tiles_m) * p.tiles_m;
        return tj >= 0;
    }
    ti = bid / p.tiles_n;
    tj = bid - ti * p.tiles_n;
    if (p.tri_k_lo && ti < p.tiles_m) ti = p.tiles_m - 1 - ti;
    return ti < p.tiles_m && (!p.lower_only || tj <= ti);
}

// A square lower-only problem restricted to column groups (GemmArgs::colmask; persistent kernel): group g = tile columns
// [g W, (g + 1) W) with the tile rows from g W down -- a triangle of W (W + 1) / 2 tiles on the diagonal, full rows of W below.
// Groups in ascending order, inside a group the triangle first, then row-major.  Everything here is wave-uniform scalar work
// (at most 64 groups), twice per tile.
template <typename T>
__device__ __forceinline__ bool decode_striped(const GemmArgs<T>& p, int bid, int& ti, int& tj) {
    unsigned long long m = p.colmask;
    const int W = p.grp_tiles;
    int rem = bid;
    while (m != 0ull) {
        const int g = __builtin_ctzll(m);
        m &= m - 1ull;
        const int c0 = g * W;
        if (c0 >= p.tiles_n) break;
        const int w = min(W, p.tiles_n - c0);
        const int head = w * (w + 1) / 2;
        const int cnt = head + (p.tiles_m - c0 - w) * w;
        if (rem < cnt) {
            if (rem < head) {
                int i, j;
                tri_decode(rem, i, j);
                ti = c0 + i; tj = c0 + j;
            } else {
                rem -= head;
                const int r = rem / w;
                ti = c0 + w + r; tj = c0 + (rem - r * w);
            }
            return true;
        }
        rem -= cnt;
    }
    ti = tj = 0;
    return false;
}

// NCT: column tiles per workgroup (1, or 2 = a TS x 2TS output: the in-place panel TRSM of the
// Cholesky needs ONE workgroup to own all 128 columns of its rows -- see gpk_gemm_launch2).
// PIPE: which k loops the tile body holds (gemm_tile).  0 = the bounds-checked round-1 loop ONLY: what UNALIGNED fp64 problems are
// sent to -- in the kernel that holds both loops the fp64 tile body needs 256 registers and 360-480 bytes of scratch, most of it in
// that loop (3 x slower per tile); on its own it needs none.
template <typename T, int TS, bool A_KMAJ, bool B_KMAJ, bool EDGE, int NCT = 1, int PIPE = GPK_GEMM_PIPE>
__global__ __launch_bounds__(256, (TS == 128 || NCT == 2 ? 2 : 4)) void gemm_kernel(GemmArgs<T> p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * (1 + NCT) * op_bytes(TS)];
    int ti, tj;
    if constexpr (TS == 128 && NCT == 1) {
        const int bid = (int)blockIdx.x;
        if (bid >= p.split_from) {
            const int qd = bid - p.split_from;
            if (!decode_tile(p, p.split_from + (qd >> 2), ti, tj)) return;
            const int ti2 = 2 * ti + ((qd >> 1) & 1), tj2 = 2 * tj + (qd & 1);
            if ((p.lower_only && tj2 > ti2) || ti2 * 64 >= p.M || tj2 * 64 >= p.N) return;
            gemm_tile<T, 64, A_KMAJ, B_KMAJ, EDGE, 1, 4, false, PIPE>(p, ti2, tj2, blockIdx.y, blockIdx.z, smem);
            return;
        }
    }
    if (p.xcd_batch > 0) {       // workgroup L runs on XCD L % 8: the tiles of matrix m = (L / 8 / tiles) * 8 + L % 8 all land there
        const int L = (int)blockIdx.x, s = L >> 3;
        const int m = (s / p.xcd_tiles) * 8 + (L & 7);
        if (m >= p.xcd_batch) return;
        if (!decode_tile(p, s % p.xcd_tiles, ti, tj)) return;
        gemm_tile<T, TS, A_KMAJ, B_KMAJ, EDGE, NCT, 4, false, PIPE>(p, ti, tj, m, 0, smem);
        return;
    }
    if (!decode_tile(p, (int)blockIdx.x, ti, tj)) return;
    gemm_tile<T, TS, A_KMAJ, B_KMAJ, EDGE, NCT, 4, false, PIPE>(p, ti, tj, blockIdx.y, blockIdx.z, smem);
}

// The panel solve of the blocked Cholesky,  P <- P inv(L_cc)^T  (both operands k-contiguous, one workgroup owns all 128 columns of its
// rows: in place), with the zero half of the triangular operand skipped fragment by fragment (gemm_tile, TRIB).
template <typename T, int TS, bool EDGE, int NCT>
__global__ __launch_bounds__(256, 2) void gemm_trib_kernel(GemmArgs<T> p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * (1 + NCT) * op_bytes(TS)];
    int ti, tj;
    if (!decode_tile(p, (int)blockIdx.x, ti, tj)) return;
    gemm_tile<T, TS, true, true, EDGE, NCT, 4, true>(p, ti, tj, blockIdx.y, blockIdx.z, smem);
}

// A (M x K) LOWER TRIANGULAR, few tiles (the leaves `inv(L_qq) B_q` of the recursive solve: 1024 x 1024 against 2048 columns):
// a row tile at row m0 runs m0 + TS of k, so with one tile per workgroup the long tiles finish alone -- one wave per SIMD, which
// issues an MFMA every ~140 cycles (profiles/r03_experiments.md, sections 1 and 11: 62 us = 35 TFLOP/s for 2.1 GFLOP).  Here a
// workgroup takes the PAIR of 32-row tiles i and tiles_m - 1 - i of one 64-column tile: every task has the same K (M + 32), there
// are 4x as many workgroups as 64 x 64 tiles gave (two per CU at the leaf's size: two waves per SIMD from start to end).
template <typename T, bool B_KMAJ, bool EDGE>
__global__ __launch_bounds__(256, 4) void gemm_trilo_pair_kernel(GemmArgs<T> p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * (1 + 2) * op_bytes(32)];
    const int half = p.tiles_m >> 1;                      // tiles_m is even (launcher)
    const int tj = (int)blockIdx.x / half, pi = (int)blockIdx.x - tj * half;
    gemm_tile<T, 32, true, B_KMAJ, EDGE, 2>(p, p.tiles_m - 1 - pi, tj, blockIdx.y, blockIdx.z, smem);
    gemm_tile<T, 32, true, B_KMAJ, EDGE, 2>(p, pi, tj, blockIdx.y, blockIdx.z, smem);
}

// ---- persistent variant: a resident set of workgroups pulls tiles of up to two problems ("segments")
// from one device-side counter.  It exists for the look-ahead Cholesky (gpk_potrf.hip):
//  * the trailing update of one outer step is two problems -- the next panel's strip (written OUT OF
//    PLACE into the panel workspace) and the lower triangle behind it -- that should share one
//    launch, one ramp and one tail;
//  * `reserve`: the workgroups that find themselves on one of the CUs named by `rkeys` (one CU per XCD,
//    identified by the CU / SH / SE fields of HW_REG_HW_ID) exit at once.  The grid has exactly one
//    workgroup per residency slot and block b goes to XCD b % 8, so every XCD receives as many workgroups as
//    it has slots, the pair on the reserved CU leaves, and that CU stays EMPTY for as long as the update
//    runs.  The library's helper stream is created with a CU mask of exactly those CUs
//    (gpk_helper_stream): the serial panel chain of the next outer step of the Cholesky runs there
//    meanwhile.  Without this nothing co-resides with the update -- its two workgroups per CU hold 144 of
//    160 KiB LDS and 444 of 512 VGPRs, and a second stream starves until the grid drains (measured,
//    profiles/r01_experiments.md; an UNMASKED second stream starves even beside emptied CUs, r02 notes).
//    Placement only: results do not depend on which workgroups leave.
// ctrl[0] = tile counter, ctrl[1] = leavers; zeroed by the launcher (memset node) per launch.
template <typename T>
struct PersistArgs {
    GemmArgs<T> seg[GPK_PERSIST_MAX_SEG];
    int first[GPK_PERSIST_MAX_SEG + 1];   // first tile of segment i (first[nseg] = all tiles; unused segments are empty)
    int sig[GPK_PERSIST_MAX_SEG];         // GpkSeg::signal
    int ntiles;               // all tiles
    int ntasks, split_from;   // tasks = tiles, except that the tiles from split_from on are handed out as four quarter tiles each
    unsigned* ctrl;
    int reserve;
    int max_leave;
    int chunks;               // tasks are claimed in chunks of 64 per XCD (gpk_claim_task): ctrl[8 + x] = XCD x's counter, ctrl[16 + 4 x ..] its slots
    unsigned rkeys[8];
    long long* prof;          // development aid: 8 slots (6 stamps) for each of the first 8 tiles of every workgroup (nullable)
};

template <typename T, int TS, bool EDGE, int NW = 4>
__device__ __forceinline__ void persist_body(const PersistArgs<T>& p, char* smem) {
    // both operands are k-contiguous: their LDS images use TS * 128 of each op_bytes(TS) slot; the broadcast
    // word lives in the unused tail of the first slot (one more byte of LDS would cost the 64-tile kernel
    // its fourth workgroup per CU)
    // (an explicit LDS pointer: through a generic `volatile int&` the compiler emitted FLAT loads / stores for this word, and a flat
    // load's s_waitcnt vmcnt(0) at the hand-over waited for the finished tile's stores to drain)
    typedef __attribute__((address_space(3))) volatile int lds_word_t;
    lds_word_t& s_tile = *(lds_word_t*)(uint32_t)(uintptr_t)(smem + TS * 128);
    const int tid = threadIdx.x;
    unsigned xcc0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc0));
    xcc0 &= 7u;
    unsigned* const cl_ctr = p.chunks ? &p.ctrl[8 + xcc0] : &p.ctrl[0];         // what a claim increments (gpk_claim_task)
    unsigned* const ch_ctr = p.chunks ? &p.ctrl[0] : nullptr;
    unsigned* const ch_slots = p.chunks ? &p.ctrl[16 + 4 * xcc0] : nullptr;
    if (p.reserve) {
        if (tid == 0) {
            unsigned xcc, hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            const unsigned key = ((hw >> 8) & 0xffu) + 1u;                 // CU_ID[11:8] SH_ID[12] SE_ID[15:13]
            bool leave = (p.rkeys[xcc & 7u] == key);
            // never more than `max_leave` leavers (the pair of workgroups of each reserved CU): were the rest of
            // the chip blocked by somebody else's kernel, the whole grid could otherwise drain through the
            // reserved CUs and leave the update undone
            if (leave && atomicAdd(&p.ctrl[1], 1u) >= (unsigned)p.max_leave) leave = false;
            s_tile = leave ? -1 : 0;
        }
        __syncthreads();
        if (s_tile < 0) return;
        __syncthreads();
    }
    // Two tasks are held at any time: the one being worked on and its successor, whose C tile the current tile's k loop pulls into
    // the L2 (gemm_tile, pf_c).  The claim for the task after that is started at the beginning of a tile -- its latency hides under
    // the k loop -- and handed round at its end.  The barriers of the hand-over wait for the LDS only: the stores of the finished
    // tile drain while the next one starts (a __syncthreads() here waited 20-35 us for them, profiles/r04_gemm_checks_tileprof_1.log).
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto claim_now = [&]() -> int {
        if (tid == 0) s_tile = gpk_claim_task(cl_ctr, ch_ctr, ch_slots);
        lds_barrier();
        const int v = __builtin_amdgcn_readfirstlane(s_tile);     // uniform by construction; tell the compiler (scalar loads of the segment)
        lds_barrier();
        return v;
    };
    struct Task { int sgi, ti, tj, quad, reps; bool ok, quarter; };
    auto decode = [&](int t) -> Task {
        Task k;
        k.ok = t < p.ntasks;
        k.quarter = (TS == 128) && t >= p.split_from;
        k.quad = k.quarter ? ((t - p.split_from) & 3) : 0;
        const int tt = k.quarter ? p.split_from + ((t - p.split_from) >> 2) : t;
        static_assert(GPK_PERSIST_MAX_SEG == 4, "segment selection below");
        k.sgi = (tt >= p.first[2]) ? ((tt >= p.first[3]) ? 3 : 2) : ((tt >= p.first[1]) ? 1 : 0);
        k.ti = k.tj = 0;
        k.reps = 1;
        if (!k.ok) return k;
        const GemmArgs<T>& g = p.seg[k.sgi];
        const int tl = tt - p.first[k.sgi];
        if (g.tri_k_lo_b && g.pair_cols) {
            // B lower triangular in k: column tile c runs c + 1 blocks of k.  One task = the tiles c and
            // tiles_n - 1 - c of one tile row, tiles_n + 1 blocks together whatever c is: equal tasks.
            const int half = g.tiles_n >> 1;
            k.ti = tl / half;
            k.tj = tl - k.ti * half;
            k.reps = 2;
        } else if (g.colmask != 0ull) {
            k.ok = decode_striped(g, tl, k.ti, k.tj);
        } else {
            k.ok = decode_tile(g, tl, k.ti, k.tj);
        }
        k.ti = __builtin_amdgcn_readfirstlane(k.ti);
        k.tj = __builtin_amdgcn_readfirstlane(k.tj);
        return k;
    };
    int nlocal = 0;
    int t = claim_now(), tn = claim_now();
    while (t < p.ntasks) {
        // The claim for the task after `tn` is made INSIDE the tile body, between its k loop and its stores (gemm_tile, claim_ctr): there
        // nothing else of the wave is outstanding, so the wait for the atomic's result costs its own round trip and no more.  At the top
        // of the iteration -- where it used to be -- the compiler's s_waitcnt vmcnt(0) behind it also waited for the previous tile's
        // stores to drain, BEFORE this tile's C tile was even requested.
        int nxt = -1;
        const Task k = decode(t);
        const GemmArgs<T>& g = p.seg[k.sgi];
        bool ok = k.ok;
        if constexpr (TS == 128) {
            if (ok && k.quarter) {              // the last, partial round of the launch: 64 x 64 quarters of a 128-tile
                const int ti2 = 2 * k.ti + (k.quad >> 1), tj2 = 2 * k.tj + (k.quad & 1);
                if (!((g.lower_only && tj2 > ti2) || ti2 * 64 >= g.M || tj2 * 64 >= g.N))
                    gemm_tile<T, 64, true, true, EDGE, 1, NW, false, 1>(g, ti2, tj2, 0, 0, smem, nullptr);
                ok = false;
            }
        }
        if (ok) {
            long long* pr = (p.prof != nullptr && nlocal < 8) ? p.prof + ((int64_t)blockIdx.x * 8 + nlocal) * 8 : nullptr;
            ++nlocal;
            // where the successor will read its C tile from (a plain 128-tile that reads C; anything else: no prefetch)
            const T* pf_c = nullptr;
            int64_t pf_ld = 0;
            if constexpr (TS == 128) {
                const Task kn = decode(tn);
                if (kn.ok && !kn.quarter && kn.reps == 1) {
                    const GemmArgs<T>& gn = p.seg[kn.sgi];
                    if (gn.has_beta && kn.ti * TS + TS <= gn.M && kn.tj * TS + TS <= gn.N) {
                        pf_c = gn.Cin + (int64_t)kn.ti * TS * gn.ldcin + (int64_t)kn.tj * TS;
                        pf_ld = gn.ldcin;
                    }
                }
            }
#pragma unroll 1
            for (int r = 0; r < k.reps; ++r)     // ONE call site: a second inlined copy of the tile body costs registers
                gemm_tile<T, TS, true, true, EDGE, 1, NW, false, 1>(g, k.ti, (k.reps == 2 && r == 0) ? g.tiles_n - 1 - k.tj : k.tj, 0, 0, smem, pr,
                                                                    (r == k.reps - 1) ? pf_c : nullptr, pf_ld, (r == k.reps - 1) ? cl_ctr : nullptr, &nxt, ch_ctr, ch_slots);
            if (p.sig[k.sgi]) {                  // somebody outside this launch waits for the tiles of this segment (the look-ahead's next chain)
                gpk_barrier_stores_done();     // every wave's stores of the tile are out
                if (tid == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_fetch_add(&p.ctrl[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        if (!ok && tid == 0) nxt = gpk_claim_task(cl_ctr, ch_ctr, ch_slots);      // (no full tile body ran: a quarter tile of the last round, an empty task)
        if (tid == 0) s_tile = nxt;
        lds_barrier();
        t = tn;
        tn = __builtin_amdgcn_readfirstlane(s_tile);
        lds_barrier();
    }
}

template <typename T, int TS, bool EDGE>
__global__ __launch_bounds__(256, (TS == 128 ? 2 : 4)) void gemm_persist_kernel(PersistArgs<T> p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * op_bytes(TS)];
    persist_body<T, TS, EDGE>(p, smem);
}

// The same body under its own name: the second launch of an update, on the helper stream once the panel chain is
// done (gpk_gemm_persist_rejoin) -- a few workgroups that pull tiles from the counter of the launch above.  A
// separate symbol so that kernel traces / rocprofv3 --stats keep the two apart.
template <typename T, int TS, bool EDGE>
__global__ __launch_bounds__(256, (TS == 128 ? 2 : 4)) void gemm_persist_helper_kernel(PersistArgs<T> p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * op_bytes(TS)];
    persist_body<T, TS, EDGE>(p, smem);
}

template <typename T, int TS, bool EDGE, int PIPE = GPK_GEMM_PIPE>
void launch_layout(bool a_kmaj, bool b_kmaj, dim3 grid, hipStream_t stream, const GemmArgs<T>& args) {
    if (a_kmaj && b_kmaj)
        hipLaunchKernelGGL((gemm_kernel<T, TS, true, true, EDGE, 1, PIPE>), grid, dim3(256), 0, stream, args);
    else if (a_kmaj && !b_kmaj)
        hipLaunchKernelGGL((gemm_kernel<T, TS, true, false, EDGE, 1, PIPE>), grid, dim3(256), 0, stream, args);
    else if (!a_kmaj && b_kmaj)
        hipLaunchKernelGGL((gemm_kernel<T, TS, false, true, EDGE, 1, PIPE>), grid, dim3(256), 0, stream, args);
    else
        hipLaunchKernelGGL((gemm_kernel<T, TS, false, false, EDGE, 1, PIPE>), grid, dim3(256), 0, stream, args);
}

// ---- measurement hook: HIP events around every GEMM launch (opt-in, see gpk.h) ----
struct ProfSlot {
    hipEvent_t a, b;
    int variant;
    double flops;
    bool ended;
};
struct Prof {
    bool on = false;
    std::vector<ProfSlot> used, pool;
    ProfSlot* begin(int variant, double flops, hipStream_t stream) {
        ProfSlot s;
        if (!pool.empty()) {
            s = pool.back();
            pool.pop_back();
        } else {
            if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess) return nullptr;
        }
        s.variant = variant;
        s.flops = flops;
        s.ended = false;
        (void)hipEventRecord(s.a, stream);
        used.push_back(s);
        return &used.back();
    }
    void end(ProfSlot* s, hipStream_t stream) { (void)hipEventRecord(s->b, stream); s->ended = true; }
};
Prof g_prof;

GPK_KNOB(int64_t, g_small_tile_below, 512);   // tuning knob (gpk_tune(1, v)); r01 sweep (old k loop): 256 -> 1024 = -1 % POTRF time; r04 sweep with the pipelined
                                              // 128-tile loop: 1024 -> 512 = TRSM 9.85 -> 9.24 ms (fp64 cfg2), 18.0 -> 17.5 (fp32 cfg3), POTRF / batched unchanged
                                              // (profiles/r04_sweep_small_tile_threshold.log): a launch of 512 128-tiles is exactly one round
GPK_KNOB(int, g_trib, 1);                     // tuning knob (gpk_tune(36, v)): panel solves skip the zero half of the inverted diagonal block per fragment
GPK_KNOB(int, g_xcd_batch, 1);              // tuning knob (gpk_tune(45, v)): batched 128-tile launches keep every matrix on one XCD (r04: 15.57 -> 15.37 ms per 512 x 2048^2 fp32 POTRF)
GPK_KNOB(int, g_trilo_pairs, 1);            // tuning knob (gpk_tune(42, v)): small products with a lower-triangular A take gemm_trilo_pair_kernel
GPK_KNOB(int, g_split_tail, 1);               // tuning knob (gpk_tune(31, v)): cut the last, partial round of a 128-tile launch into quarter tiles

}  // namespace

namespace {
long long* g_tile_prof = nullptr;       // development aid (gpk_tune_tile_prof)
int64_t g_tile_prof_only = -1;          // tuning knob (gpk_tune(20, v)): stamp only the v-th persistent launch since the knob was set (-1: every one)
int64_t g_tile_prof_count = 0;
GPK_KNOB(int, g_persist_chunks, 0);              // tuning knob (gpk_tune(58, v)): the persistent update hands its tiles out in chunks of 64 per XCD (gpk_claim_task)
GPK_KNOB(int64_t, g_persist_chunks_min, 2048);   // tuning knob (gpk_tune(59, v)): ... from this many tiles on
GPK_KNOB(int64_t, g_persist_small_below, 512);   // tuning knob (gpk_tune(8, v)): the persistent update takes 64x64 tiles below this many 128-tiles
int g_cu_count[64] = {0};
int device_cus() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (g_cu_count[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        g_cu_count[dev] = n;
    }
    return g_cu_count[dev];
}
}  // namespace

void gpk_set_tile_prof(long long* dev_buf) { g_tile_prof = dev_buf; }

// Tuning knobs of this file (gpk_tune in include/gpk.h): A/B runs on the GPU box.
void gpk_tune_gemm(int key, int64_t value) {
    if (key == 1) GPK_KNOB_SET(g_small_tile_below = value;);
    if (key == 8) GPK_KNOB_SET(g_persist_small_below = value;);
    if (key == 58) GPK_KNOB_SET(g_persist_chunks = (int)value;);
    if (key == 59) GPK_KNOB_SET(g_persist_chunks_min = value;);
    if (key == 31) GPK_KNOB_SET(g_split_tail = (int)value;);
    if (key == 36) GPK_KNOB_SET(g_trib = (int)value;);
    if (key == 42) GPK_KNOB_SET(g_trilo_pairs = (int)value;);
    if (key == 45) GPK_KNOB_SET(g_xcd_batch = (int)value;);
    if (key == 20) { g_tile_prof_only = value; g_tile_prof_count = 0; }
}

// the hooks for launches made outside this file (the mixed-phase batched steps of gpk_potrf.hip); nullptr when the hooks are off
void* gpk_prof_begin(int variant, double flops, hipStream_t stream) { return g_prof.on ? (void*)g_prof.begin(variant, flops, stream) : nullptr; }
void gpk_prof_end(void* slot, hipStream_t stream) {
    if (slot != nullptr) g_prof.end(static_cast<ProfSlot*>(slot), stream);
}

extern "C" int gpk_prof_start(void) {
    for (auto& s : g_prof.used) g_prof.pool.push_back(s);
    g_prof.used.clear();
    g_prof.on = true;
    return GPK_OK;
}

// variant: 16*(64x64 tiles) + 8*(f64) + 4*(A k-major) + 2*(B k-major) + 1*(edge-checked kernel); + 32: the persistent update, 64 +: panel_step_kernel,
// 96 +: gemm_trilo_pair_kernel, 128 +: gemm_trib_kernel; or -1 for all
extern "C" int gpk_prof_stop(int variant, double* total_ms, int64_t* launches, double* useful_flops) {
    g_prof.on = false;
    double ms = 0, fl = 0;
    int64_t n = 0;
    for (auto& s : g_prof.used) {
        if (!s.ended) continue;          // (a launch that bailed out between its two events)
        if (hipEventSynchronize(s.b) != hipSuccess) return GPK_ERR_LAUNCH;
        if (variant >= 0 && s.variant != variant) continue;
        float t = 0;
        if (hipEventElapsedTime(&t, s.a, s.b) != hipSuccess) return GPK_ERR_LAUNCH;
        ms += t;
        fl += s.flops;
        ++n;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = n;
    if (useful_flops) *useful_flops = fl;
    return GPK_OK;
}

template <typename T>
int gpk_gemm_launch2(bool a_kmaj, bool b_kmaj, int64_t M, int64_t N, int64_t K, T alpha,
                     const T* A, int64_t lda, int64_t sA, int64_t sA2, const T* B, int64_t ldb,
                     int64_t sB, int64_t sB2, T beta, T* C, int64_t ldc, int64_t sC, int64_t sC2,
                     int64_t batch, int64_t batch2, int flags, hipStream_t stream, const T* colscale, T* colss, int64_t ldss) {
    const bool lower_only = (flags & 1) != 0;
    const bool fused_cols = colscale != nullptr || colss != nullptr;      // (gpk_gemm_colscale: the 128-tile kernels' epilogue)
    if (fused_cols && (batch != 1 || batch2 != 1 || lower_only || (const void*)A == (const void*)C || (const void*)B == (const void*)C))
        return GPK_ERR_ARG(17);
    if (M <= 0 || N <= 0 || batch <= 0 || batch2 <= 0) return GPK_OK;
    if (M > INT32_MAX || N > INT32_MAX || K > INT32_MAX || batch > 65535 || batch2 > 65535)
        return GPK_ERR_ARG(3);
    if (alpha == T(0)) return GPK_ERR_ARG(6);   // scale-only is not a use of this path
    if (K < 0) return GPK_ERR_ARG(3);
    // the C tile's per-lane byte offsets are 32 bits (gemm_tile: up to 12 rows * ldc * sizeof(T))
    if (ldc >= GPK_C_LD_MAX) return GPK_ERR_ARG(15);
    constexpr int VEC = Traits<T>::VEC;
    constexpr int BK = Traits<T>::BK;

    GemmArgs<T> g;
    g.A = A; g.B = B; g.C = C; g.Cin = C;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldcin = ldc;
    g.sA = sA; g.sB = sB; g.sC = sC;
    g.sA2 = sA2; g.sB2 = sB2; g.sC2 = sC2;
    g.M = (int)M; g.N = (int)N; g.K = (int)K;
    g.alpha = alpha;
    g.has_beta = (beta != T(0)) ? 1 : 0;
    g.beta_over_alpha = g.has_beta ? beta / alpha : T(0);
    // Tile choice: 128x128 unless that grid cannot even give every CU one workgroup; then
    // 64x64 tiles (4x the workgroups) -- the narrow panel/merge/solve GEMMs of the path.
    int ts = 128, nct = 1;
    {
        const int64_t tm = gpk_cdiv(M, 128), tn = gpk_cdiv(N, 128);
        const int64_t t128 = (lower_only && tm == tn ? tm * (tm + 1) / 2 : tm * tn) * batch * batch2;
        if (t128 < g_small_tile_below && !fused_cols) ts = 64;
        // In-place use (C aliases the A operand: the panel TRSM  P <- P inv(L_cc)^T  of the Cholesky):
        // every workgroup reads the full K range of its rows of A and then overwrites a column slice
        // of them, so ONE workgroup must own all N columns of a row tile -- with two column tiles a
        // late workgroup would read what its neighbour already overwrote.
        if ((const void*)A == (const void*)C || (const void*)B == (const void*)C) {
            if (N > 128 || ((const void*)B == (const void*)C)) return GPK_ERR_ARG(16);   // cannot be made race-free
            if (ts == 64 && a_kmaj && b_kmaj)
                nct = 2;          // 64 x 128 per workgroup: keeps the 2x finer row split of the small-tile path
            else
                ts = 128;
        }
    }
    g.tiles_m = (int)gpk_cdiv(M, ts);
    g.tiles_n = (int)gpk_cdiv(N, ts * nct);
    g.lower_only = lower_only ? 1 : 0;
    g.tri_k = (flags & 2) ? 1 : 0;
    g.tri_k_lo = (flags & 4) ? 1 : 0;
    g.tri_k_lo_b = (flags & 8) ? 1 : 0;
    g.pair_cols = 0; g.colmask = 0; g.grp_tiles = 0;

    const bool tri = lower_only && g.tiles_m == g.tiles_n;
    const int64_t total = tri ? (int64_t)g.tiles_m * (g.tiles_m + 1) / 2
                                     : (int64_t)g.tiles_m * g.tiles_n;
    int64_t gridx = total;

    const bool aligned = ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && (lda % VEC == 0) &&
                         (ldb % VEC == 0) && (sA % VEC == 0) && (sB % VEC == 0) &&
                         (sA2 % VEC == 0) && (sB2 % VEC == 0);
    // (K < BK -- K = 0 above all, an empty contraction: C = beta C -- goes to the bounds-checked kernels: the pipelined loop of the
    // others runs at least one chunk, `while (true) { chunk; if (--left == 0) break; }` with left = 0 would spin for 2^32 of them)
    const bool edge = !aligned || (M % ts) || (N % (ts * nct)) || (K % BK) || K < BK || lda >= GPK_PIPE_LD_MAX || ldb >= GPK_PIPE_LD_MAX;
    g.vec_ok = aligned ? 1 : 0;

    // The last, partial round.  The hardware hands out workgroups in index order as slots free up, so with equal tiles the last
    // `total mod slots` of them run alone at the end while the other slots idle -- up to a whole tile time per launch (measured:
    // 1664 tiles = 3.25 rounds of 512 slots cost the time of 4).  When that remainder is at most half a round, its tiles are
    // launched as four 64 x 64 quarter tiles each: they fit one or two rounds of ~0.35 tile times.  Same results (a tile's entries are
    // computed by the same k order whichever kernel body does it).
    g.split_from = INT32_MAX;
    g.colscale = colscale; g.colss = colss; g.ldss = ldss;
    g.xcd_batch = 0; g.xcd_tiles = 0;
    if (g_split_tail && !fused_cols && ts == 128 && nct == 1 && batch == 1 && batch2 == 1 && (flags & (2 | 4 | 8)) == 0 &&
        (const void*)A != (const void*)C && (const void*)B != (const void*)C) {      // (in-place: one workgroup must own all columns of its rows)
        const int64_t slots = (int64_t)device_cus() * 2;
        const int64_t rem = total % slots;
        if (total > slots && rem > 0 && 2 * rem <= slots) {
            g.split_from = (int)(total - rem);
            gridx = total + 3 * rem;
        }
    }
    dim3 grid((unsigned)gridx, (unsigned)batch, (unsigned)batch2);
    // (many small problems only: the 25 strided K-slices of cfg5's split-K SYRK, 528 tiles each, deal badly over 8 XCDs -- 64.4 vs 56.4 ms per step)
    if (g_xcd_batch && (flags & 16) == 0 && ts == 128 && nct == 1 && batch >= 64 && batch2 == 1 && g.split_from == INT32_MAX && gridx * gpk_cdiv(batch, 8) * 8 < INT32_MAX) {
        g.xcd_batch = (int)batch; g.xcd_tiles = (int)gridx;
        grid = dim3((unsigned)(gridx * gpk_cdiv(batch, 8) * 8), 1, 1);
    }
    // triangular A, a grid too small to keep two waves per SIMD busy to the end: pairs of 32-row tiles (gemm_trilo_pair_kernel)
    if (g_trilo_pairs && flags == 4 && a_kmaj && ts == 64 && nct == 1 && batch == 1 && batch2 == 1 && M % 64 == 0 && M >= 256 &&
        (const void*)A != (const void*)C && (const void*)B != (const void*)C && gpk_cdiv(M, 64) * gpk_cdiv(N, 64) <= 2 * (int64_t)device_cus()) {
        g.tiles_m = (int)(M / 32);
        g.tiles_n = (int)gpk_cdiv(N, 64);
        const bool e2 = !aligned || (N % 64) || (K % BK);
        const dim3 pgrid((unsigned)((g.tiles_m / 2) * g.tiles_n), 1, 1);
        ProfSlot* ps = nullptr;
        if (g_prof.on) ps = g_prof.begin(96 + (sizeof(T) == 8 ? 8 : 0) + (b_kmaj ? 2 : 0) + (e2 ? 1 : 0), (double)M * (double)N * (double)K, stream);
        if (b_kmaj) {
            if (e2) hipLaunchKernelGGL((gemm_trilo_pair_kernel<T, true, true>), pgrid, dim3(256), 0, stream, g);
            else hipLaunchKernelGGL((gemm_trilo_pair_kernel<T, true, false>), pgrid, dim3(256), 0, stream, g);
        } else {
            if (e2) hipLaunchKernelGGL((gemm_trilo_pair_kernel<T, false, true>), pgrid, dim3(256), 0, stream, g);
            else hipLaunchKernelGGL((gemm_trilo_pair_kernel<T, false, false>), pgrid, dim3(256), 0, stream, g);
        }
        if (ps) g_prof.end(ps, stream);
        GPK_CHECK_LAUNCH();
        return GPK_OK;
    }
    const bool trib = (flags & 16) && g_trib && a_kmaj && b_kmaj && g.tiles_n == 1 && N <= 128 && g.split_from == INT32_MAX;
    ProfSlot* slot = nullptr;
    if (g_prof.on) {
        // useful (algorithmic) flops: a lower-only update counts the symmetric half
        // (a triangular operand halves the multiply-adds actually needed: the TRSM / TRMM count)
        const double fl = (lower_only ? 1.0 : 2.0) * ((flags & (2 | 4 | 8)) ? 0.5 : 1.0) * (double)M * (double)N * (double)K * (double)batch * (double)batch2;
        // (gemm_trib_kernel is a kernel of its own: codes 128 + ...; round 3 filed it under gemm_kernel's code)
        const int code = (trib && (nct == 2 || ts == 128)) ? 128 + (sizeof(T) == 8 ? 8 : 0) + (edge ? 1 : 0) + (ts == 64 ? 16 : 0)
                                                           : (sizeof(T) == 8 ? 8 : 0) + (a_kmaj ? 4 : 0) + (b_kmaj ? 2 : 0) + (edge ? 1 : 0) + (ts == 64 ? 16 : 0);
        slot = g_prof.begin(code, fl, stream);
    }
    if (trib && nct == 2) {
        if (edge) hipLaunchKernelGGL((gemm_trib_kernel<T, 64, true, 2>), grid, dim3(256), 0, stream, g);
        else hipLaunchKernelGGL((gemm_trib_kernel<T, 64, false, 2>), grid, dim3(256), 0, stream, g);
    } else if (trib && ts == 128 && nct == 1) {
        if (edge) hipLaunchKernelGGL((gemm_trib_kernel<T, 128, true, 1>), grid, dim3(256), 0, stream, g);
        else hipLaunchKernelGGL((gemm_trib_kernel<T, 128, false, 1>), grid, dim3(256), 0, stream, g);
    } else if (nct == 2) {
        if (edge)
            hipLaunchKernelGGL((gemm_kernel<T, 64, true, true, true, 2>), grid, dim3(256), 0, stream, g);
        else
            hipLaunchKernelGGL((gemm_kernel<T, 64, true, true, false, 2>), grid, dim3(256), 0, stream, g);
    } else if (sizeof(T) == 8 && !aligned) {        // (unaligned fp64: the kernels that hold the bounds-checked loop only, see gemm_kernel)
        if constexpr (sizeof(T) == 8) {
            if (ts == 128) launch_layout<T, 128, true, 0>(a_kmaj, b_kmaj, grid, stream, g);
            else launch_layout<T, 64, true, 0>(a_kmaj, b_kmaj, grid, stream, g);
        }
    } else if (ts == 128) {
        if (edge)
            launch_layout<T, 128, true>(a_kmaj, b_kmaj, grid, stream, g);
        else
            launch_layout<T, 128, false>(a_kmaj, b_kmaj, grid, stream, g);
    } else {
        if (edge)
            launch_layout<T, 64, true>(a_kmaj, b_kmaj, grid, stream, g);
        else
            launch_layout<T, 64, false>(a_kmaj, b_kmaj, grid, stream, g);
    }
    if (slot) g_prof.end(slot, stream);
    GPK_CHECK_LAUNCH();
    return GPK_OK;
}

template <typename T>
int gpk_gemm_launch(bool a_kmaj, bool b_kmaj, int64_t M, int64_t N, int64_t K, T alpha,
                    const T* A, int64_t lda, int64_t sA, const T* B, int64_t ldb, int64_t sB,
                    T beta, T* C, int64_t ldc, int64_t sC, int64_t batch, int flags,
                    hipStream_t stream) {
    return gpk_gemm_launch2<T>(a_kmaj, b_kmaj, M, N, K, alpha, A, lda, sA, 0, B, ldb, sB, 0, beta, C,
                               ldc, sC, 0, batch, 1, flags, stream, nullptr, nullptr, 0);
}

// ---- the library's helper stream: confined (CU mask) to one CU per XCD, whose identities the persistent
// update is told to keep clear of ----
namespace {
struct HelperDev {
    int state = 0;              // 0 = not tried, 1 = ready, -1 = unavailable
    hipStream_t aux = nullptr;
    hipStream_t side = nullptr;  // a second stream on the same CUs (gpk_helper_side_stream)
    int side_state = 0;
    unsigned keys[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
HelperDev g_helper[64];
std::mutex g_helper_mutex;

__global__ void helper_census_kernel(unsigned* out) {
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        const unsigned key = ((hw >> 8) & 0xffu) + 1u;
        const unsigned old = atomicCAS(&out[xcc & 7u], 0u, key);
        if (old != 0u && old != key) atomicAdd(&out[8], 1u);     // a second CU on this XCD: the mask is not what we think
    }
}
}  // namespace

// One-time per device (first look-ahead factorisation, or gpk_init): creates the stream, runs a 64-workgroup
// census on it and reads 36 bytes back -- the only host synchronisation and the only allocation of the library.
int gpk_helper_stream(hipStream_t* aux, unsigned keys[8]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return GPK_ERR_LAUNCH;
    std::lock_guard<std::mutex> lock(g_helper_mutex);
    HelperDev& h = g_helper[dev];
    if (h.state == 0) {
        h.state = -1;
        // CU-mask bit b of this API: XCD b % 8, then SE (b / 8) % 4, then the (b / 32)-th ACTIVE CU of that SE
        // (measured on MI355X, profiles/r02_experiments.md; an XCD with no bit set is left unrestricted, so every
        // XCD gets its bit).  Bits 0..7 = the first active CU of SE 0 on each of the 8 XCDs.
        uint32_t mask[8] = {0xffu, 0, 0, 0, 0, 0, 0, 0};
        hipStream_t st = nullptr;
        unsigned* d = nullptr;
        if (hipExtStreamCreateWithCUMask(&st, 8, mask) == hipSuccess && hipMalloc(&d, 9 * sizeof(unsigned)) == hipSuccess) {
            unsigned host[9] = {0};
            bool ok = hipMemsetAsync(d, 0, sizeof(host), st) == hipSuccess;
            if (ok) {
                hipLaunchKernelGGL(helper_census_kernel, dim3(64), dim3(64), 0, st, d);
                ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess &&
                     hipMemcpy(host, d, sizeof(host), hipMemcpyDeviceToHost) == hipSuccess;
            }
            int nk = 0;
            for (int x = 0; x < 8; ++x) nk += host[x] != 0;
            if (ok && host[8] == 0 && nk == 8) {
                for (int x = 0; x < 8; ++x) h.keys[x] = host[x];
                h.aux = st;
                h.state = 1;
            }
        }
        if (d) (void)hipFree(d);
        if (h.state != 1) {
            // no CU mask (or not the layout above): a plain helper stream, nothing reserved -- the look-ahead then
            // does not overlap (results are the same)
            if (st) (void)hipStreamDestroy(st);
            if (hipStreamCreateWithFlags(&h.aux, hipStreamNonBlocking) == hipSuccess) h.state = 2;
        }
    }
    if (h.state <= 0) return GPK_ERR_LAUNCH;
    *aux = h.aux;
    for (int x = 0; x < 8; ++x) keys[x] = (h.state == 1) ? h.keys[x] : 0u;
    return GPK_OK;
}

// A SECOND stream with the helper stream's CU mask (round 6): memory-bound side work -- the matrix-vector products of a right-hand
// side that rides through the look-ahead factorisation -- runs there BESIDE the serial panel chain, on the CUs the trailing update
// keeps empty.  (An unmasked stream would take residency slots the persistent update counts on: measured, +1.7 ms per cfg2 eval.)
// *side = nullptr when the device has no masked helper stream: the caller runs that work in line.
int gpk_helper_side_stream(hipStream_t* side) {
    hipStream_t aux;
    unsigned keys[8];
    *side = nullptr;
    const int st = gpk_helper_stream(&aux, keys);
    if (st != GPK_OK) return st;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return GPK_ERR_LAUNCH;
    std::lock_guard<std::mutex> lock(g_helper_mutex);
    HelperDev& h = g_helper[dev];
    if (h.state == 1 && h.side_state == 0) {
        h.side_state = -1;
        uint32_t mask[8] = {0xffu, 0, 0, 0, 0, 0, 0, 0};
        hipStream_t s2 = nullptr;
        if (hipExtStreamCreateWithCUMask(&s2, 8, mask) == hipSuccess) {
            h.side = s2;
            h.side_state = 1;
        }
    }
    if (h.side_state == 1) *side = h.side;
    return GPK_OK;
}

// Destroy the helper streams (gpk_shutdown): a process that exits with a CU-masked stream still alive can crash in
// the runtime's / a profiler's own teardown (seen with rocprofv3 around a torch process).
void gpk_helper_shutdown() {
    std::lock_guard<std::mutex> lock(g_helper_mutex);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (int dev = 0; dev < 64; ++dev) {
        HelperDev& h = g_helper[dev];
        if (h.aux != nullptr || h.side != nullptr) (void)hipSetDevice(dev);
        if (h.side != nullptr) {
            (void)hipStreamSynchronize(h.side);
            (void)hipStreamDestroy(h.side);
        }
        if (h.aux != nullptr) {
            (void)hipStreamSynchronize(h.aux);
            (void)hipStreamDestroy(h.aux);
        }
        h = HelperDev();
    }
    (void)hipSetDevice(cur);
}

// ---- persistent launch: up to GPK_PERSIST_MAX_SEG k-major problems  C = Cin + alpha * A B^T  in one resident grid ----

// tiles of a lower-only problem (M >= N: the square part's lower triangle + every row below it) restricted to column groups (decode_striped); area: the elements of C they cover,
// counted as the library counts a symmetric update (the diagonal tiles' upper halves are not useful work)
static int64_t striped_tiles(int64_t M, int64_t N, int ts, int64_t grp, uint64_t mask, double* area) {
    const int64_t tiles_m = gpk_cdiv(M, ts), tiles_n = gpk_cdiv(N, ts), W = grp / ts;
    int64_t n = 0;
    for (int g = 0; g < 64; ++g) {
        if (!((mask >> g) & 1u)) continue;
        const int64_t c0 = g * W;
        if (c0 >= tiles_n) break;
        const int64_t w = (W < tiles_n - c0) ? W : tiles_n - c0;
        n += w * (w + 1) / 2 + (tiles_m - c0 - w) * w;
        if (area != nullptr) {
            const double rows = (double)(M - g * grp), left = (double)(N - g * grp), cols = left < (double)grp ? left : (double)grp;
            *area += cols * rows - 0.5 * cols * cols;
        }
    }
    return n;
}

template <typename T>
int gpk_gemm_persist_launch(const GpkSeg<T>* segs, int nseg, T alpha, unsigned* ctrl, int reserve,
                            hipStream_t stream, GpkPersistSaved* saved, bool ctrl_zeroed) {
    if (saved != nullptr) { saved->valid = 0; saved->signal_tiles = 0; }
    if (nseg < 1 || nseg > GPK_PERSIST_MAX_SEG) return GPK_ERR_ARG(2);
    if (ctrl == nullptr) return GPK_ERR_ARG(4);
    if (alpha == T(0)) return GPK_ERR_ARG(3);
    constexpr int VEC = Traits<T>::VEC;
    constexpr int BK = Traits<T>::BK;
    int64_t t128 = 0;
    for (int i = 0; i < nseg; ++i) {
        const GpkSeg<T>& q = segs[i];
        if (q.M > INT32_MAX || q.N > INT32_MAX || q.K > INT32_MAX || q.K <= 0) return GPK_ERR_ARG(1);
        if (q.M <= 0 || q.N <= 0) continue;
        if (q.ldc >= GPK_C_LD_MAX || q.ldcin >= GPK_C_LD_MAX) return GPK_ERR_ARG(1);
        if (q.colmask != 0) {        // column groups of a square lower-only problem
            // (M > N: rows below the square part -- the look-ahead Cholesky carries the rows of K(x*, x) under the matrix it factorises)
            if (!q.lower_only || q.M < q.N || q.tri_b || q.grp < 128 || q.grp % 128 != 0) return GPK_ERR_ARG(1);
            t128 += striped_tiles(q.M, q.N, 128, q.grp, q.colmask, nullptr);
            continue;
        }
        const int64_t tm = gpk_cdiv(q.M, 128), tn = gpk_cdiv(q.N, 128);
        t128 += (q.lower_only && tm == tn) ? tm * (tm + 1) / 2 : tm * tn;
    }
    if (t128 == 0) return GPK_OK;
    const int ts = (t128 < g_persist_small_below) ? 64 : 128;

    PersistArgs<T> pa;
    bool edge = false;
    int64_t total = 0;
    double flops = 0;
    int live = 0;
    for (int i = 0; i < nseg; ++i) {
        const GpkSeg<T>& q = segs[i];
        if (q.M <= 0 || q.N <= 0) continue;
        GemmArgs<T>& g = pa.seg[live];
        g.A = q.A; g.B = q.B; g.C = q.C; g.Cin = q.Cin;
        g.lda = q.lda; g.ldb = q.ldb; g.ldc = q.ldc; g.ldcin = q.ldcin;
        g.sA = g.sB = g.sC = g.sA2 = g.sB2 = g.sC2 = 0;
        g.M = (int)q.M; g.N = (int)q.N; g.K = (int)q.K;
        g.alpha = alpha; g.has_beta = q.Cin != nullptr ? 1 : 0; g.beta_over_alpha = g.has_beta ? T(1) / alpha : T(0);
        if (!g.has_beta) g.Cin = q.C;
        g.tiles_m = (int)gpk_cdiv(q.M, ts); g.tiles_n = (int)gpk_cdiv(q.N, ts);
        g.lower_only = q.lower_only ? 1 : 0;
        g.tri_k = g.tri_k_lo = 0;
        g.colscale = nullptr; g.colss = nullptr; g.ldss = 0; g.xcd_batch = 0; g.xcd_tiles = 0;
        g.tri_k_lo_b = q.tri_b ? 1 : 0;
        const bool aligned = ((uintptr_t)q.A % 16 == 0) && ((uintptr_t)q.B % 16 == 0) && (q.lda % VEC == 0) &&
                             (q.ldb % VEC == 0);
        g.vec_ok = aligned ? 1 : 0;
        edge = edge || !aligned || (q.M % ts) || (q.N % ts) || (q.K % BK) || q.lda >= GPK_PIPE_LD_MAX || q.ldb >= GPK_PIPE_LD_MAX;
        const bool tri = g.lower_only && g.tiles_m == g.tiles_n;
        int64_t nt = tri ? (int64_t)g.tiles_m * (g.tiles_m + 1) / 2 : (int64_t)g.tiles_m * g.tiles_n;
        g.pair_cols = (q.tri_b == 2 && !g.lower_only && g.tiles_n >= 2 && g.tiles_n % 2 == 0) ? 1 : 0;
        if (g.pair_cols) nt = (int64_t)g.tiles_m * (g.tiles_n / 2);
        g.colmask = q.colmask; g.grp_tiles = q.colmask != 0 ? (int)(q.grp / ts) : 0;
        double area = 0;            // elements of C the segment updates (striped segments)
        if (q.colmask != 0) nt = striped_tiles(q.M, q.N, ts, q.grp, q.colmask, &area);
        if (nt == 0) continue;
        pa.first[live] = (int)total;
        pa.sig[live] = q.signal ? 1 : 0;
        if (q.signal && saved != nullptr) saved->signal_tiles += (int)nt;
        total += nt;
        flops += q.colmask != 0 ? 2.0 * area * (double)q.K : (q.lower_only || q.tri_b ? 1.0 : 2.0) * (double)q.M * (double)q.N * (double)q.K;
        ++live;
    }
    if (total > INT32_MAX / 2) return GPK_ERR_ARG(1);
    if (live == 0) return GPK_OK;
    for (int i = live; i < GPK_PERSIST_MAX_SEG; ++i) { pa.seg[i] = pa.seg[0]; pa.sig[i] = 0; }
    for (int i = live; i <= GPK_PERSIST_MAX_SEG; ++i) pa.first[i] = (int)total;      // (empty segments: never selected)
    pa.ntiles = (int)total;
    pa.ntasks = (int)total;
    pa.split_from = INT32_MAX;
    pa.ctrl = ctrl;
    pa.chunks = (g_persist_chunks && total >= g_persist_chunks_min) ? 1 : 0;
    pa.prof = nullptr;
    if (g_tile_prof != nullptr) {
        if (g_tile_prof_only < 0 || g_tile_prof_count == g_tile_prof_only) pa.prof = g_tile_prof;
        ++g_tile_prof_count;
    }
    if (!ctrl_zeroed && hipMemsetAsync(ctrl, 0, GPK_PERSIST_CTRL_WORDS * sizeof(unsigned), stream) != hipSuccess) return GPK_ERR_LAUNCH;

    const int per_cu = (ts == 128) ? 2 : 4;
    int64_t slots = (int64_t)device_cus() * per_cu;
    int64_t gridx = total < slots ? total : slots;
    if (total < slots) reserve = 0;      // a grid that does not fill the chip leaves CUs free by itself
    pa.reserve = 0;
    pa.max_leave = 0;
    for (int x = 0; x < 8; ++x) pa.rkeys[x] = 0;
    if (reserve) {
        hipStream_t aux;
        unsigned keys[8];
        if (gpk_helper_stream(&aux, keys) == GPK_OK) {
            int nk = 0;
            for (int x = 0; x < 8; ++x) { pa.rkeys[x] = keys[x]; nk += keys[x] != 0; }
            pa.reserve = nk > 0 ? 1 : 0;
            pa.max_leave = per_cu * nk;
        }
    }
    if (g_split_tail && ts == 128 && total > gridx) {        // the last, partial round as quarter tiles (see gpk_gemm_launch2)
        bool plain = true;
        for (int i = 0; i < live; ++i) plain = plain && !pa.seg[i].pair_cols && !pa.seg[i].tri_k_lo_b;
        const int64_t workers = gridx - pa.max_leave;
        const int64_t rem = total % workers;
        if (plain && workers > 0 && rem > 0 && 2 * rem <= workers) {
            pa.split_from = (int)(total - rem);
            pa.ntasks = (int)(total + 3 * rem);
        }
    }
    ProfSlot* slot = nullptr;
    if (g_prof.on) slot = g_prof.begin(32 + (sizeof(T) == 8 ? 8 : 0) + (edge ? 1 : 0) + (ts == 64 ? 16 : 0), flops, stream);
    dim3 grid((unsigned)gridx);
    if (ts == 128) {
        if (edge) hipLaunchKernelGGL((gemm_persist_kernel<T, 128, true>), grid, dim3(256), 0, stream, pa);
        else hipLaunchKernelGGL((gemm_persist_kernel<T, 128, false>), grid, dim3(256), 0, stream, pa);
    } else {
        if (edge) hipLaunchKernelGGL((gemm_persist_kernel<T, 64, true>), grid, dim3(256), 0, stream, pa);
        else hipLaunchKernelGGL((gemm_persist_kernel<T, 64, false>), grid, dim3(256), 0, stream, pa);
    }
    if (slot) g_prof.end(slot, stream);
    GPK_CHECK_LAUNCH();
    if (saved != nullptr && pa.reserve) {
        static_assert(sizeof(PersistArgs<T>) <= sizeof(saved->bytes), "GpkPersistSaved too small");
        memcpy(saved->bytes, &pa, sizeof(pa));
        saved->ts = ts;
        saved->edge = edge ? 1 : 0;
        saved->per_cu = per_cu;
        saved->valid = 1;
    }
    return GPK_OK;
}

// The CUs a reserving update keeps clear REJOIN it once the helper stream has nothing else to do: the same kernel, the
// same tile counter, a grid of just the reserved slots, enqueued on the (CU-masked) helper stream behind the chain.  Tiles
// are claimed atomically, so whoever runs takes what is left; results do not depend on it.
template <typename T>
int gpk_gemm_persist_rejoin(const GpkPersistSaved* saved, hipStream_t helper_stream) {
    if (saved == nullptr || !saved->valid) return GPK_OK;
    PersistArgs<T> pa;
    memcpy(&pa, saved->bytes, sizeof(pa));
    pa.reserve = 0;
    pa.prof = nullptr;
    dim3 grid((unsigned)(8 * saved->per_cu));
    if (saved->ts == 128) {
        if (saved->edge) hipLaunchKernelGGL((gemm_persist_helper_kernel<T, 128, true>), grid, dim3(256), 0, helper_stream, pa);
        else hipLaunchKernelGGL((gemm_persist_helper_kernel<T, 128, false>), grid, dim3(256), 0, helper_stream, pa);
    } else {
        if (saved->edge) hipLaunchKernelGGL((gemm_persist_helper_kernel<T, 64, true>), grid, dim3(256), 0, helper_stream, pa);
        else hipLaunchKernelGGL((gemm_persist_helper_kernel<T, 64, false>), grid, dim3(256), 0, helper_stream, pa);
    }
    GPK_CHECK_LAUNCH();
    return GPK_OK;
}

// ---------------------------------------------------------------------------
// One step of the blocked Cholesky below a factorised 128x128 diagonal block, in ONE launch (gpk_potrf.hip, single matrices):
//   phase A   L[r, c:c+128] = P[r, :] inv(L_cc)^T          the panel solve, strips of TS rows (one workgroup per strip)
//   phase B   A[r, cb] -= L[r, c:c+128] L[cb rows, c:c+128]^T   the rank-128 update of the columns of the outer panel still to come,
//                                                          one workgroup per (strip, 128-column block) on or below the diagonal
// Phase B of a workgroup needs phase A of its own strip and of the (up to 128 / TS) strips that hold the rows of its column block:
// one flag word per strip, published by the strip's phase-A workgroup (stores -> barrier -> agent-scope release -> flag) and
// polled by one lane of the consumers (relaxed agent loads -> agent-scope acquire -> barrier).  Phase-A workgroups have the lowest
// block indices and never wait before they publish, so the launch makes progress whatever part of the grid is resident.  Against
// two launches (panel solve, then update) the chain of a 128-column step loses a kernel boundary and a ramp, and with TS = 32 the
// strips are twice as fine as the 64-row tiles of the plain kernel (a lone wave per SIMD issues its MFMAs at a fraction of the
// pipe rate: the step is latency-bound, not throughput-bound).
// ---------------------------------------------------------------------------
namespace {
template <typename T>
struct PanelStepArgs {
    GemmArgs<T> trsm;     // A = P (rows below the diagonal block; M x 128), B = inv(L_cc) (128 x 128), C = P (in place)
    GemmArgs<T> upd;      // A = B = the panel after phase A (M x 128 / N x 128), C = the trailing columns (M x N)
    unsigned* flags;      // one word per strip, zero before the launch
    int nstrips, ncb;     // strips of TS rows; 128-column blocks to update (0: panel solve only)
};

template <typename T, int TS, bool EDGE>
__global__ __launch_bounds__(256, 2) void panel_step_kernel(PanelStepArgs<T> p) {
    constexpr int NCT = 128 / TS;
    __shared__ __attribute__((aligned(16))) char smem[2 * (1 + NCT) * op_bytes(TS)];
    const int tid = threadIdx.x;
    int strip = (int)blockIdx.x, cb = 0;
    if (strip >= p.nstrips) {                  // an update-only workgroup: (strip, cb >= 1), strips from the first one that reaches block cb
        int r = strip - p.nstrips;
        for (cb = 1; cb < p.ncb; ++cb) {
            const int cnt = p.nstrips - cb * NCT;
            if (r < cnt) break;
            r -= cnt;
        }
        if (cb >= p.ncb) return;
        strip = cb * NCT + r;
    }
    if (cb == 0) {
        gemm_tile<T, TS, true, true, EDGE, NCT, 4, true>(p.trsm, strip, 0, 0, 0, smem);      // (TRIB: inv(L_cc) is lower triangular)
        gpk_barrier_stores_done();              // every wave's stores of the strip are out
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&p.flags[strip], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (p.ncb == 0) return;
    }
    if (tid == 0) {
        // the rows of column block cb are strips cb NCT .. cb NCT + NCT - 1; plus this workgroup's own strip when another one solved it
        int lo = cb * NCT, hi = lo + NCT;
        if (hi > p.nstrips) hi = p.nstrips;
        for (int s = lo; s < hi + 1; ++s) {
            const int f = (s < hi) ? s : strip;
            while (__hip_atomic_load(&p.flags[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    gemm_tile<T, TS, true, true, EDGE, NCT>(p.upd, strip, cb, 0, 0, smem);
}
}  // namespace

// A: n x n matrix (ld), c: first column of the factorised diagonal block, W = inv(L_cc) (128 x 128, row-major), ke: end of the outer
// panel (columns [c + 128, ke) are updated; ke <= c + 128: panel solve only).  flags: >= ceil((n - c - 128) / 32) zeroed words.
template <typename T>
int gpk_panel_step_launch(T* A, int64_t n, int64_t ld, int64_t c, const T* W, int64_t ke, unsigned* flags, hipStream_t stream) {
    const int64_t r1 = c + GPK_DB;
    const int64_t m = n - r1;
    if (m <= 0) return GPK_OK;
    if (ke > n) ke = n;
    const int64_t ncols = ke > r1 ? ke - r1 : 0;
    constexpr int VEC = Traits<T>::VEC;
    constexpr int BK = Traits<T>::BK;
    const int ts = (m <= 4096) ? 32 : 64;       // latency-bound steps take the finer strips
    PanelStepArgs<T> pa;
    GemmArgs<T>& g = pa.trsm;
    T* P = A + r1 * ld + c;
    g.A = P; g.B = W; g.C = P; g.Cin = P;
    g.lda = ld; g.ldb = GPK_DB; g.ldc = ld; g.ldcin = ld;
    g.sA = g.sB = g.sC = g.sA2 = g.sB2 = g.sC2 = 0;
    g.M = (int)m; g.N = GPK_DB; g.K = GPK_DB;
    g.alpha = T(1); g.beta_over_alpha = T(0); g.has_beta = 0;
    g.tiles_m = (int)gpk_cdiv(m, ts); g.tiles_n = 1;
    g.lower_only = 0; g.tri_k = 0; g.tri_k_lo = 0; g.tri_k_lo_b = 0; g.pair_cols = 0; g.colmask = 0; g.grp_tiles = 0;
    g.split_from = INT32_MAX;
    g.colscale = nullptr; g.colss = nullptr; g.ldss = 0; g.xcd_batch = 0; g.xcd_tiles = 0;
    const bool aligned = ((uintptr_t)P % 16 == 0) && ((uintptr_t)W % 16 == 0) && (ld % VEC == 0);
    g.vec_ok = aligned ? 1 : 0;
    GemmArgs<T>& u = pa.upd;
    u = g;
    u.A = P; u.B = P; u.C = A + r1 * ld + r1; u.Cin = u.C;
    u.ldb = ld;
    u.N = (int)ncols;
    u.alpha = T(-1); u.beta_over_alpha = T(-1); u.has_beta = 1;
    u.tiles_n = (int)gpk_cdiv(ncols > 0 ? ncols : 1, GPK_DB);
    const bool edge = !aligned || (m % ts) || (ncols % GPK_DB) || (GPK_DB % BK) || ld >= GPK_PIPE_LD_MAX;
    pa.flags = flags;
    pa.nstrips = (int)gpk_cdiv(m, ts);
    pa.ncb = (int)gpk_cdiv(ncols, GPK_DB);
    const int nct = GPK_DB / ts;
    int64_t grid = pa.nstrips;
    for (int cb = 1; cb < pa.ncb; ++cb) {
        const int64_t cnt = pa.nstrips - (int64_t)cb * nct;
        if (cnt > 0) grid += cnt;
    }
    ProfSlot* slot = nullptr;
    if (g_prof.on) slot = g_prof.begin(64 + (sizeof(T) == 8 ? 8 : 0) + (edge ? 1 : 0), (double)m * GPK_DB * GPK_DB + 2.0 * (double)m * (double)ncols * GPK_DB, stream);
    if (ts == 32) {
        if (edge) hipLaunchKernelGGL((panel_step_kernel<T, 32, true>), dim3((unsigned)grid), dim3(256), 0, stream, pa);
        else hipLaunchKernelGGL((panel_step_kernel<T, 32, false>), dim3((unsigned)grid), dim3(256), 0, stream, pa);
    } else {
        if (edge) hipLaunchKernelGGL((panel_step_kernel<T, 64, true>), dim3((unsigned)grid), dim3(256), 0, stream, pa);
        else hipLaunchKernelGGL((panel_step_kernel<T, 64, false>), dim3((unsigned)grid), dim3(256), 0, stream, pa);
    }
    if (slot) g_prof.end(slot, stream);
    GPK_CHECK_LAUNCH();
    return GPK_OK;
}
template int gpk_panel_step_launch<double>(double*, int64_t, int64_t, int64_t, const double*, int64_t, unsigned*, hipStream_t);
template int gpk_panel_step_launch<float>(float*, int64_t, int64_t, int64_t, const float*, int64_t, unsigned*, hipStream_t);

#define GPK_INST(T)                                                                                  \
    template int gpk_gemm_launch2<T>(bool, bool, int64_t, int64_t, int64_t, T, const T*, int64_t,    \
                                     int64_t, int64_t, const T*, int64_t, int64_t, int64_t, T, T*,   \
                                     int64_t, int64_t, int64_t, int64_t, int64_t, int, hipStream_t, const T*, T*, int64_t); \
    template int gpk_gemm_launch<T>(bool, bool, int64_t, int64_t, int64_t, T, const T*, int64_t,     \
                                    int64_t, const T*, int64_t, int64_t, T, T*, int64_t, int64_t,    \
                                    int64_t, int, hipStream_t);                                      \
    template int gpk_gemm_persist_launch<T>(const GpkSeg<T>*, int, T, unsigned*, int, hipStream_t, GpkPersistSaved*, bool); \
    template int gpk_gemm_persist_rejoin<T>(const GpkPersistSaved*, hipStream_t);
GPK_INST(double)
GPK_INST(float)
