// gpk_gemm_tile.hpp -- the workgroup-level MFMA tile of the library: one TS x (NCT * TS) output tile of
//   C[m][n] = alpha * sum_k a(m,k) b(n,k) + beta * C[m][n]
// computed by the calling workgroup from register-staged, double-buffered LDS operand tiles (design notes at the top of
// gpk_gemm.hip).  Device code only; included by gpk_gemm.hip (every GEMM kernel) and gpk_potrf.hip (the pipelined panel
// factorisation, whose worker workgroups run panel solves and rank-128 updates as tasks).
#pragma once
#include "gpk_common.hpp"
#include <type_traits>

// The k loop of the 128-tile: 0 = one chunk of global loads in flight, MFMA phase, write phase, barrier (rounds 1-3);
// 1 = the software-pipelined loop (see gemm_tile), 2 = ... and a ragged K's partial last chunk behind it (bounds-checked kernels; the
// persistent kernels, whose K is a panel width, pass 1).  A build-time switch: both loops in one kernel cost registers.
#ifndef GPK_GEMM_PIPE
#define GPK_GEMM_PIPE 2
#endif
#ifndef GPK_GEMM_PIPE64
#define GPK_GEMM_PIPE64 1      // ... for the 64-tile too
#endif
namespace {

// One operand tile of TS rows in LDS, either image: k-contiguous [TS][128 B] or
// row-contiguous [BK][TS + 16] (BK * sizeof(T) = 128 B), so (TS + 16) * 128 bytes cover both.
constexpr int op_bytes(int ts) { return (ts + 16) * 128; }
// leading dimensions (elements) from which the pipelined k loop's 32-bit operand offsets could overflow (128 rows * ld * 8 bytes < 2^32)
constexpr int64_t GPK_PIPE_LD_MAX = (int64_t)1 << 21;
// leading dimensions of C (elements) from which the per-lane 32-bit byte offset of a C fragment (<= 12 rows * ldc * 8 bytes) could overflow
constexpr int64_t GPK_C_LD_MAX = (int64_t)1 << 25;

template <typename T>
struct GemmArgs {
    const T* A;
    const T* B;
    T* C;
    const T* Cin;               // where beta * C is read from (== C unless the update goes out of place)
    int64_t lda, ldb, ldc, ldcin, sA, sB, sC;
    int64_t sA2, sB2, sC2;      // second (blockIdx.z) batch level
    int M, N, K;
    T alpha, beta_over_alpha;
    int has_beta;
    int tiles_m, tiles_n;
    int lower_only;
    int vec_ok;       // pointers / leading dimensions allow 16-byte loads
    int tri_k;        // operands are lower-triangular in k (zero for k < row index): start at k = m0
    int tri_k_lo;     // A (M x K) is lower triangular (zero for k > row): stop at k = m0 + TS; tile rows run
                      // longest-first (bottom rows first)
    int tri_k_lo_b;   // B (N x K) is lower triangular (zero for k > column index n): stop at k = n0 + tile width
    int pair_cols;    // persistent kernel, tri_k_lo_b: one task = column tiles c and tiles_n - 1 - c of a tile row
    const T* colscale;   // (128-tile kernels, optional) C[m][n] = alpha * sum * colscale[n]: a column scaling folded into the store
    T* colss;            // (optional) column sums of squares of the UNSCALED alpha * sum over each wave's 64 rows: row 2 ti + wm of a
    int64_t ldss;        //   [2 tiles_m][ldss] buffer (the caller adds the rows up) -- the pseudo-point path's Q_x_diag without a pass over V
    int xcd_batch;    // (knob 45) batched launch as a 1-D grid with all tiles of a matrix on ONE XCD: xcd_batch = batch size, else 0
    int xcd_tiles;    //   tiles per matrix of such a launch
    unsigned long long colmask;   // persistent kernel, square lower-only segments: only the column groups named here (0: all) --
    int grp_tiles;                //   group g = tile columns [g grp_tiles, (g + 1) grp_tiles), rows from its first column's tile down
    int split_from;   // plain launches of 128-tiles: block indices from here on are QUARTER tiles (64 x 64) of the tiles split_from,
                      // split_from + 1, ... -- the last, partial round of a launch cut four times finer (see gpk_gemm_launch2)
};

// ---- task claims of the persistent kernels ------------------------------------
// Plain: the next index of ONE device-wide counter.  Chained (chain_ctr != nullptr; round 6): `ctr` is the counter of the calling
// workgroup's XCD, and tasks are handed out in CHUNKS of 64 consecutive indices per XCD -- the workgroup whose claim opens a chunk
// (local index a multiple of 64) takes the next chunk number from the device-wide counter `chain_ctr` and publishes it in one of four
// tagged slots of its XCD, the 63 others read it there.  64 consecutive tiles of a trailing update are 8 tile rows x the 8 tile columns
// of a column group: the 64 workgroups of an XCD then share 16 operand panels instead of holding up to 128 different ones, start
// together and walk k together, so a k-slice of a panel is fetched into the XCD's L2 once instead of once per tile.
__device__ __forceinline__ int gpk_claim_task(unsigned* ctr, unsigned* chain_ctr, unsigned* chain_slots) {
    const unsigned i = atomicAdd(ctr, 1u);
    if (chain_ctr == nullptr) return (int)i;
    const unsigned k = i >> 6, want = (k + 1u) & 0xfffu;
    unsigned* slot = chain_slots + (k & 3u);
    unsigned v;
    if ((i & 63u) == 0u) {
        v = (want << 20) | (atomicAdd(chain_ctr, 1u) & 0xfffffu);
        __hip_atomic_store(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        do {
            v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while ((v >> 20) != want);        // (published right behind the opening claim, which precedes this one)
    }
    const unsigned c = v & 0xfffffu;
    return c >= 0x7ffffu ? 0x7fffffff : (int)(c * 64u + (i & 63u));
}

// ---- global -> registers ---------------------------------------------------
// `fast`: (EDGE kernels only) this tile's rows and this k-chunk lie fully inside the operand and
// 16-byte loads are legal -- interior tiles of a ragged problem take the vector path too.
template <typename T, int TS, bool KMAJ, bool EDGE, int NT = 256>
__device__ __forceinline__ void gload(typename Traits<T>::vec_t (&r)[TS * 8 / NT], const T* __restrict__ base,
                                      int64_t ld, int r0, int k0, int R, int K, int tid, bool fast) {
    typedef typename Traits<T>::vec_t vec_t;
    constexpr int VEC = Traits<T>::VEC;
    if (KMAJ) {
        const int c = tid & 7, rr0 = tid >> 3;
#pragma unroll
        for (int i = 0; i < TS * 8 / NT; ++i) {
            const int row = r0 + rr0 + (NT / 8) * i;
            const int k = k0 + c * VEC;
            const T* p = base + (int64_t)row * ld + k;
            if (!EDGE || fast) {
                r[i] = *reinterpret_cast<const vec_t*>(p);
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) r[i][v] = (row < R && k + v < K) ? p[v] : T(0);
            }
        }
    } else {
        constexpr int CPR = TS / VEC;   // 16-byte chunks per k-row
#pragma unroll
        for (int i = 0; i < TS * 8 / NT; ++i) {
            const int id = tid + NT * i;
            const int krow = id / CPR, cc = id % CPR;
            const int k = k0 + krow;
            const int row = r0 + cc * VEC;
            const T* p = base + (int64_t)k * ld + row;
            if (!EDGE || fast) {
                r[i] = *reinterpret_cast<const vec_t*>(p);
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) r[i][v] = (k < K && row + v < R) ? p[v] : T(0);
            }
        }
    }
}

// ---- registers -> LDS --------------------------------------------------------
template <typename T, int TS, bool KMAJ, int NT = 256>
__device__ __forceinline__ void sstore(char* lds, const typename Traits<T>::vec_t (&r)[TS * 8 / NT], int tid) {
    typedef typename Traits<T>::vec_t vec_t;
    constexpr int VEC = Traits<T>::VEC;
    if (KMAJ) {
        const int c = tid & 7, rr0 = tid >> 3;
#pragma unroll
        for (int i = 0; i < TS * 8 / NT; ++i) {
            const int rr = rr0 + (NT / 8) * i;
            const int off = rr * 128 + ((c ^ ((rr >> 1) & 7)) << 4);
            *reinterpret_cast<vec_t*>(lds + off) = r[i];
        }
    } else {
        constexpr int CPR = TS / VEC;
#pragma unroll
        for (int i = 0; i < TS * 8 / NT; ++i) {
            const int id = tid + NT * i;
            const int krow = id / CPR, cc = id % CPR;
            const int off = krow * ((TS + 16) * (int)sizeof(T)) + cc * 16;
            *reinterpret_cast<vec_t*>(lds + off) = r[i];
        }
    }
}


// vector i alone (the pipelined loop writes one vector per slice of MFMAs)
template <typename T, int TS, bool KMAJ, int NT = 256>
__device__ __forceinline__ void sstore1(char* lds, const typename Traits<T>::vec_t& r, int tid, int i) {
    typedef typename Traits<T>::vec_t vec_t;
    constexpr int VEC = Traits<T>::VEC;
    int off;
    if (KMAJ) {
        const int c = tid & 7, rr = (tid >> 3) + (NT / 8) * i;
        off = rr * 128 + ((c ^ ((rr >> 1) & 7)) << 4);
    } else {
        constexpr int CPR = TS / VEC;
        const int id = tid + NT * i;
        off = (id / CPR) * ((TS + 16) * (int)sizeof(T)) + (id % CPR) * 16;
    }
    *reinterpret_cast<vec_t*>(lds + off) = r;
}

// ---- LDS -> MFMA operand -----------------------------------------------------
// rowbase: first tile row of this 16-row fragment; lr = lane & 15; k = element index in chunk.
template <typename T, int TS, bool KMAJ>
__device__ __forceinline__ T fragread(const char* lds, int rowbase, int lr, int k, int swz) {
    constexpr int VEC = Traits<T>::VEC;
    int off;
    if (KMAJ) {
        const int chunk = k / VEC, within = k % VEC;
        off = (rowbase + lr) * 128 + ((chunk ^ swz) << 4) + within * (int)sizeof(T);
    } else {
        off = k * ((TS + 16) * (int)sizeof(T)) + (rowbase + lr) * (int)sizeof(T);
    }
    return *reinterpret_cast<const T*>(lds + off);
}

// claim_ctr / claimed (/ chain_ctr / chain_slots) (persistent kernel): see the claim below the k loop and gpk_claim_task.
// pf_c / pf_ld (pipelined 128-tile only): first element and leading dimension of the C tile this workgroup will READ NEXT (the
// persistent kernel knows its next task).  Round 4, from the per-tile time stamps (profiles/r04_gemm_checks_tileprof_1.log): a
// workgroup waits 20-45 us for the 1024 cache lines of its C tile (HBM misses, a few dozen in flight per CU) before its first
// MFMA, and the other workgroup of the CU covers that at the lone-wave rate only.  Each thread therefore touches one line of the
// next tile per chunk during the last few chunks of this one -- just in time: the XCD's 4 MiB L2 turns over in ~25 us -- so that the
// next tile's C loads are L2 hits.
// NCT: column tiles per workgroup (1, or 2 = a TS x 2TS output: the in-place panel TRSM of the
// Cholesky needs ONE workgroup to own all 128 columns of its rows -- see gpk_gemm_launch2).
// One output tile (ti, tj) of one problem, computed by the calling workgroup (256 threads).  `smem`:
// 2 * (1 + NCT) * op_bytes(TS) bytes of LDS, free on entry; every wave has passed a barrier after its
// last LDS read when the function returns.
// NW: waves of the workgroup, as an (NW / 2) x 2 grid over the tile.  NW = 4 (256 threads): 64 x 64 of a 128-tile per wave -- what
// every kernel of this file instantiates.  NW = 8 (512 threads, 32 x 64 per wave, two workgroups = FOUR waves per SIMD) was built
// and measured twice on the hypothesis that two waves per SIMD starve the matrix pipe whenever one of them waits: round 3 with the
// old k loop (same results, 2-4 % SLOWER: fp64 8192^3 68.6 vs 71.4 TFLOP/s, POTRF N = 16384 28.6 vs 28.1 ms), round 4 with the
// pipelined loop in a persistent kernel at 128 registers (1-1.5 % slower: the K = 1024 fp64 update 3.89 vs 3.84 ms, the fp32 look-ahead
// 90.6 vs 89.3 ms; profiles/r04_ab_nw8.log).  The kernels were removed, the parameter stays.
// TRIB: the B operand (N x K) is LOWER TRIANGULAR and the tile starts at column 0 of it (the panel solve P inv(L_cc)^T of the
// Cholesky): a 16-column fragment at columns j0.. only needs k < j0 + 16, the MFMAs beyond are skipped per fragment and per group of
// k values -- 7/16 of the multiply-adds of a 128-column solve (the k loop itself still streams all of the operands).
template <typename T, int TS, bool A_KMAJ, bool B_KMAJ, bool EDGE, int NCT, int NW = 4, bool TRIB = false, int PIPE = GPK_GEMM_PIPE>
__device__ __forceinline__ void gemm_tile(const GemmArgs<T>& p, int ti, int tj, int64_t b, int64_t b2, char* smem,
                                          long long* prof = nullptr, const T* pf_c = nullptr, int64_t pf_ld = 0,
                                          unsigned* claim_ctr = nullptr, int* claimed = nullptr,
                                          unsigned* chain_ctr = nullptr, unsigned* chain_slots = nullptr) {
    if (prof != nullptr && threadIdx.x == 0) prof[0] = wall_clock64();
    typedef typename Traits<T>::acc_t acc_t;
    typedef typename Traits<T>::vec_t vec_t;
    constexpr int BK = Traits<T>::BK;
    constexpr int VEC_ = Traits<T>::VEC;
    constexpr int NT = 64 * NW;          // threads
    // HALFW: eight waves on a 32-row tile (the latency-critical strips of the pipelined panel).  An operand tile is 256 vectors: the
    // two halves of the workgroup (waves 0-3 / 4-7) each stage and multiply HALF of the NCT column tiles, the lower half also
    // stages A; inside a half the four waves form the usual 2 x 2 grid of 16 x 16 fragments.
    constexpr bool HALFW = (NW == 8 && TS == 32);
    static_assert(!HALFW || NCT % 2 == 0, "two halves share the column tiles");
    constexpr int NTL = HALFW ? 256 : NT;   // threads that move ONE operand tile
    constexpr int NCW = HALFW ? NCT / 2 : NCT;   // column tiles a wave works on
    constexpr int NV = TS * 8 / NTL;     // 16-byte vectors a thread moves per operand tile and k-chunk
    constexpr int FR = TS / 32;          // 16x16 fragments per wave along the columns
    constexpr int FRM = HALFW ? 1 : TS / (8 * NW);   // ... along the rows
    constexpr int WT = TS / 2;           // wave sub-tile width
    constexpr int WTM = 16 * FRM;        // ... height
    constexpr int OPB = op_bytes(TS), STAGE = (1 + NCT) * OPB;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = HALFW ? (wave >> 2) : 0;            // which half of the column tiles (HALFW)
    const int ltid = HALFW ? (tid & 255) : tid;          // thread index inside the group that moves one operand tile
    const int wgrid = HALFW ? (wave & 3) : wave;
    const int wm = wgrid >> 1, wn = wgrid & 1;
    const int cfirst = half * NCW;                       // first column tile of this wave
    const int lr = lane & 15, kq = lane >> 4;
    const int swz = (lr >> 1) & 7;

    const T* __restrict__ A = p.A + b * p.sA + b2 * p.sA2;
    const T* __restrict__ B = p.B + b * p.sB + b2 * p.sB2;
    T* __restrict__ C = p.C + b * p.sC + b2 * p.sC2;
    const T* __restrict__ Cin = p.Cin + b * p.sC + b2 * p.sC2;

    // tile indices come out of a float square root (decode_tile) or an LDS broadcast: tell the compiler they are wave-uniform, so
    // that everything derived from them -- the tile's base addresses above all -- is scalar arithmetic
    ti = __builtin_amdgcn_readfirstlane(ti);
    tj = __builtin_amdgcn_readfirstlane(tj);
    const int m0 = ti * TS, n0 = tj * TS * NCT;

    // C addressing: a wave-uniform base per (fragment, accumulator register) + ONE per-lane byte offset (round 4: the 64 loads and 64
    // stores of a tile used to compute a 64-bit row * ld + col per element on the vector ALU -- 6 + 5 us of a K = 1024 tile's ~270
    // went into ISSUING them).  crow(lane, i) = crow(lane, 0) + rstep * i.
    constexpr int RSTEP = (sizeof(T) == 8) ? 4 : 1;
    const unsigned lane_off_in = ((unsigned)Traits<T>::crow(lane, 0) * (unsigned)p.ldcin + (unsigned)lr) * (unsigned)sizeof(T);
    const unsigned lane_off_out = ((unsigned)Traits<T>::crow(lane, 0) * (unsigned)p.ldc + (unsigned)lr) * (unsigned)sizeof(T);
    auto c_base = [&](int64_t ld, int c, int fi, int fj, int i) -> int64_t {      // (uniform) element offset of the fragment register's first row / column
        return (int64_t)(m0 + wm * WTM + fi * 16 + RSTEP * i) * ld + (n0 + (cfirst + c) * TS + wn * WT + fj * 16);
    };

    acc_t acc[NCW][FRM][FR];
#pragma unroll
    for (int c = 0; c < NCW; ++c) {
        if (p.has_beta) {
#pragma unroll
            for (int fi = 0; fi < FRM; ++fi)
#pragma unroll
                for (int fj = 0; fj < FR; ++fj)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = m0 + wm * WTM + fi * 16 + Traits<T>::crow(lane, i);
                        const int col = n0 + (cfirst + c) * TS + wn * WT + fj * 16 + lr;
                        T v = T(0);
                        if (!EDGE || (row < p.M && col < p.N))
                            v = *reinterpret_cast<const T*>(reinterpret_cast<const char*>(Cin + c_base(p.ldcin, c, fi, fj, i)) + lane_off_in);
                        acc[c][fi][fj][i] = v * p.beta_over_alpha;
                    }
        } else {
#pragma unroll
            for (int fi = 0; fi < FRM; ++fi)
#pragma unroll
                for (int fj = 0; fj < FR; ++fj)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[c][fi][fj][i] = T(0);
        }
    }

    int nk = (p.K + BK - 1) / BK;
    if (p.tri_k_lo) nk = min(nk, (m0 + TS + BK - 1) / BK);   // A vanishes right of its diagonal
    if (p.tri_k_lo_b) nk = min(nk, (n0 + TS * NCT + BK - 1) / BK);   // B vanishes right of its diagonal
    int kc0 = p.tri_k ? m0 / BK : 0;            // all-zero k-chunks of triangular operands are skipped
    if (kc0 > nk - 1) kc0 = nk > 0 ? nk - 1 : 0;
    // Register staging.  Small tiles (TS = 64) serve the narrow, latency-bound GEMMs of the path
    // (panel / merge / solve), where a workgroup is often alone on its CU: they keep TWO k-chunks of
    // global loads in flight (PF2); the 128-tile kernel hides the latency with its second workgroup.
    constexpr bool PF2 = (TS <= 64);
    vec_t ra[PF2 ? 2 : 1][NV], rb[PF2 ? 2 : 1][NCW][NV];

    const bool a_in = EDGE && p.vec_ok && (m0 + TS <= p.M), b_in = EDGE && p.vec_ok && (n0 + TS * NCT <= p.N);
    auto issue = [&](auto set_c, int kc) {           // global -> register set `set`
        constexpr int set = decltype(set_c)::value;
        const bool k_in = (kc + 1) * BK <= p.K;
        if (!HALFW || half == 0) gload<T, TS, A_KMAJ, EDGE, NTL>(ra[set], A, p.lda, m0, kc * BK, p.M, p.K, ltid, a_in && k_in);
#pragma unroll
        for (int c = 0; c < NCW; ++c)
            gload<T, TS, B_KMAJ, EDGE, NTL>(rb[set][c], B, p.ldb, n0 + (cfirst + c) * TS, kc * BK, p.N, p.K, ltid, b_in && k_in);
    };
    auto commit = [&](auto set_c, int stage) {       // register set -> LDS stage
        constexpr int set = decltype(set_c)::value;
        char* dA = smem + stage * STAGE;
        if (!HALFW || half == 0) sstore<T, TS, A_KMAJ, NTL>(dA, ra[set], ltid);
#pragma unroll
        for (int c = 0; c < NCW; ++c) sstore<T, TS, B_KMAJ, NTL>(dA + (1 + cfirst + c) * OPB, rb[set][c], ltid);
    };
    auto mma = [&](int stage, int kc) {
        // scheduler hint: interleave the LDS reads with the MFMAs of this k-chunk (measured +2 % for fp64,
        // -7 % for fp32, whose paired-k reads already leave fewer LDS instructions)
        if constexpr (sizeof(T) == 8) __builtin_amdgcn_iglp_opt(0);
        const char* sA = smem + stage * STAGE;
        const char* sB = sA + OPB;
        if constexpr (sizeof(T) == 4 && A_KMAJ && B_KMAJ) {
            // fp32, both operands k-contiguous: a lane fetches TWO consecutive k values with one
            // ds_read_b64 and feeds them to two successive MFMAs.  The contraction order inside the
            // chunk is permuted identically for A and B (step 2p+e, lane group kq <-> k = 8p + 2kq + e),
            // which is harmless for a sum.  Halves the LDS read instructions and removes the 2-way
            // bank conflict of ds_read_b32 on a 128-byte row pitch (bank = dword mod 32: the two
            // rows 2a, 2a+1 of a 32-lane group alias); the b64 pattern equals the fp64 one.
            typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int pp = 0; pp < BK / 8; ++pp) {
                f2 a2[FRM], b2[NCW][FR];
                const int u = pp * 4 + kq;           // 8-byte unit of the 128-byte row
                const int uo = (((u >> 1) ^ swz) << 4) + (u & 1) * 8;
#pragma unroll
                for (int f = 0; f < FRM; ++f) a2[f] = *reinterpret_cast<const f2*>(sA + (wm * WTM + f * 16 + lr) * 128 + uo);
#pragma unroll
                for (int f = 0; f < FR; ++f) {
#pragma unroll
                    for (int c = 0; c < NCW; ++c)
                        b2[c][f] = *reinterpret_cast<const f2*>(sB + (cfirst + c) * OPB + (wn * WT + f * 16 + lr) * 128 + uo);
                }
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int c = 0; c < NCW; ++c)
#pragma unroll
                        for (int fj = 0; fj < FR; ++fj) {
                            if constexpr (TRIB) {        // (wave-uniform) the 8 k values of this group all lie right of the fragment's columns
                                if (kc * BK + pp * 8 >= (cfirst + c) * TS + wn * WT + fj * 16 + 16) continue;
                            }
#pragma unroll
                            for (int fi = 0; fi < FRM; ++fi)
                                acc[c][fi][fj] = Traits<T>::mfma((T)a2[fi][e], (T)b2[c][fj][e], acc[c][fi][fj]);
                        }
            }
            return;
        }
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            T a[FRM], bb[NCW][FR];
            const int k = kk * 4 + kq;
#pragma unroll
            for (int f = 0; f < FRM; ++f) a[f] = fragread<T, TS, A_KMAJ>(sA, wm * WTM + f * 16, lr, k, swz);
#pragma unroll
            for (int f = 0; f < FR; ++f) {
#pragma unroll
                for (int c = 0; c < NCW; ++c)
                    bb[c][f] = fragread<T, TS, B_KMAJ>(sB + (cfirst + c) * OPB, wn * WT + f * 16, lr, k, swz);
            }
#pragma unroll
            for (int c = 0; c < NCW; ++c)
#pragma unroll
                for (int fj = 0; fj < FR; ++fj) {
                    if constexpr (TRIB) {
                        if (kc * BK + kk * 4 >= (cfirst + c) * TS + wn * WT + fj * 16 + 16) continue;
                    }
#pragma unroll
                    for (int fi = 0; fi < FRM; ++fi) acc[c][fi][fj] = Traits<T>::mfma(a[fi], bb[c][fj], acc[c][fi][fj]);
                }
        }
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, PF2 ? 1 : 0> S1;

    if (prof != nullptr && threadIdx.x == 0) {      // C tile requested, accumulators being initialised
        prof[1] = wall_clock64();
        prof[4] = (long long)__builtin_readcyclecounter();   // shader-clock ticks: with the 100 MHz stamps they give the clock the k loop ran at
    }

    // ---- PIPE: the software-pipelined k loop of the 128-tile (round 4) ----------------------------------------------------
    // A lone fp64 wave issues an MFMA every ~140 cycles, two waves of a SIMD together every 64 (profiles/r03_experiments.md section 1):
    // the matrix pipe runs at full rate only while BOTH waves of a SIMD have an MFMA to issue.  The loop below therefore never
    // leaves the MFMA stream: a chunk is four PHASES (4 k values in fp64, 8 in fp32: one register set of fragments each);
    // while phase p multiplies, the fragments of phase p + 1 are read from LDS into the other register set, the NEXT chunk (global
    // loads issued one chunk earlier) is written to the other LDS stage (phase 0) and the chunk after that is requested from
    // global memory (phase 1); the one barrier of the chunk sits in the middle of phase 3's MFMAs, and the first fragments of the
    // next chunk are read behind it, under the rest of phase 3.  sched_group_barrier pins that interleaving.
    // (bounds-checked kernels: the interior tiles of a ragged problem take it too -- cfg5's N = 200000, the look-ahead at orders
    // that are not multiples of 128)
    // (round 4, later: the 64-tile -- 32 x 32 per wave, four workgroups per CU: the narrow panel / solve GEMMs, 23 % of a cfg2 step -- takes
    // the same loop with four slices per phase instead of eight)
    constexpr bool PIPE_KERNEL = PIPE != 0 && NCT == 1 && !TRIB && NW == 4 && (TS == 128 || (TS == 64 && GPK_GEMM_PIPE64 != 0));
    // A ragged K (bounds-checked kernels): the loop takes the whole chunks, the last, partial chunk goes through the bounds-checked
    // loads of the loop below, once (round 4: with `p.K % BK == 0` as a condition EVERY tile of such a problem took the loop below,
    // which in the fp64 kernels -- both loops in one function, 256 registers -- runs with spills: fp64 15000^2 K = 1000 21 TFLOP/s
    // against 55 at K = 1024)
    const bool k_tail = EDGE && PIPE_KERNEL && PIPE >= 2 && (nk * BK > p.K);
    const int nk_pipe = k_tail ? nk - 1 : nk;
    // A ragged M / N (bounds-checked kernels): the tiles on the matrix edge take the loop too -- rows past the end of an operand are
    // CLAMPED to its last row when the per-thread pointers are set up (nothing changes inside the loop): those rows of the product
    // are computed from duplicates and never stored.  (An operand stored K x M, rows of it contiguous, needs M to be a whole number of
    // 16-byte vectors for that.)  Round 4: fp64 8000 x 2000 x 15008 ran at 21 TFLOP/s because its 78 edge tiles of 1008 took the loop
    // below -- in the fp64 kernels with spills, 3 x slower per tile -- and sat at the end of the launch.
    // (PIPE >= 2 only: the persistent fp64 kernel answered one more `min` per pointer with spills inside the loop)
    constexpr bool CLAMP = PIPE >= 2;
    const bool a_ok = a_in || (CLAMP && EDGE && p.vec_ok && (A_KMAJ || (p.M % VEC_ == 0 && p.M >= VEC_)));
    const bool b_ok = b_in || (CLAMP && EDGE && p.vec_ok && (B_KMAJ || (p.N % VEC_ == 0 && p.N >= VEC_)));
    const bool pipe_tile = PIPE_KERNEL && (!EDGE || (a_ok && b_ok && nk_pipe > kc0 && (PIPE >= 2 || p.K % BK == 0)));
    if constexpr (PIPE_KERNEL) if (pipe_tile) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef typename std::conditional<sizeof(T) == 8, double, f32x2>::type frag_t;
        constexpr int KPP = (sizeof(T) == 8) ? 4 : 8;      // k values per phase
        constexpr int NPH = BK / KPP;                       // phases per chunk
        static_assert(NPH == 4, "four phases per chunk");
        frag_t Fa[2][FRM], Fb[2][FR];
        vec_t qa[NV], qb[NV];
        // Operand addresses.  BUF (kernels without bounds checks): buffer loads -- ONE wave-uniform 64-bit origin per operand (the tile's
        // first row at the current chunk, moved on by the scalar ALU) in a buffer descriptor, one per-lane 32-bit byte offset per
        // operand (voffset) and the uniform offset of the thread's i-th vector (soffset).  Per-thread 64-bit pointers take 4 NV vector
        // registers and two vector adds per load: 250 -> 232 registers for the fp64 persistent kernel, the fp32 look-ahead 90.7 -> 89.3 ms
        // (profiles/r04_experiments.md section 12).  Offsets are 32 bits: 128 rows * ld * sizeof(T) < 2^32 -- the launchers send leading
        // dimensions >= GPK_PIPE_LD_MAX to the bounds-checked kernels.
        // !BUF (bounds-checked kernels): per-thread pointers.  Those kernels hold this loop AND the one below for their edge tiles and sit
        // at the scalar-register limit; the descriptors' 20 scalar registers pushed the fp64 ones into spilling accumulators inside the loop.
        constexpr bool BUF = !EDGE;
        auto uni = [](const void* q) -> const char* {            // (only ever fed to a descriptor: the address space does not matter)
            const uint64_t v = reinterpret_cast<uint64_t>(q);
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
            return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
        };
        auto rsrc = [](const char* base) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 0xffffffffu, 0x00020000); };
        constexpr int RPP_A = A_KMAJ ? NT / 8 : NT / (TS / VEC_), RPP_B = B_KMAJ ? NT / 8 : NT / (TS / VEC_);     // rows a pass of the workgroup covers
        static_assert(NT % (TS / VEC_) == 0, "a thread's vectors sit in the same columns of successive k rows");
        auto lane_off = [&](auto kmaj_c, int64_t ld) -> unsigned {
            constexpr bool KMAJ = decltype(kmaj_c)::value;
            constexpr int CPR = TS / VEC_;
            return KMAJ ? ((unsigned)(tid >> 3) * (unsigned)ld + (unsigned)(tid & 7) * VEC_) * (unsigned)sizeof(T)
                        : ((unsigned)(tid / CPR) * (unsigned)ld + (unsigned)(tid % CPR) * VEC_) * (unsigned)sizeof(T);
        };
        const char* ua = nullptr;               // BUF: the operands' origins, lane offsets and the offset of one pass of the workgroup
        const char* ub = nullptr;
        unsigned la = 0, lb = 0, passA = 0, passB = 0;
        const T* pa[NV];                        // !BUF: per-thread source pointers of the NV vectors of an operand tile
        const T* pb[NV];
        if constexpr (BUF) {
            la = lane_off(std::integral_constant<bool, A_KMAJ>{}, p.lda);
            lb = lane_off(std::integral_constant<bool, B_KMAJ>{}, p.ldb);
            ua = uni(A_KMAJ ? A + (int64_t)m0 * p.lda + (int64_t)kc0 * BK : A + (int64_t)kc0 * BK * p.lda + m0);
            ub = uni(B_KMAJ ? B + (int64_t)n0 * p.ldb + (int64_t)kc0 * BK : B + (int64_t)kc0 * BK * p.ldb + n0);
            passA = __builtin_amdgcn_readfirstlane((unsigned)RPP_A * (unsigned)p.lda * (unsigned)sizeof(T));
            passB = __builtin_amdgcn_readfirstlane((unsigned)RPP_B * (unsigned)p.ldb * (unsigned)sizeof(T));
        } else {
            auto init_ptrs = [&](const T* (&ptr)[NV], auto kmaj_c, const T* base, int64_t ld, int r0, int R) {      // R: rows of the operand
                constexpr bool KMAJ = decltype(kmaj_c)::value;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    if (KMAJ) {
                        const int row = CLAMP ? min(r0 + (tid >> 3) + (NT / 8) * i, R - 1) : r0 + (tid >> 3) + (NT / 8) * i;
                        ptr[i] = base + (int64_t)row * ld + (int64_t)kc0 * BK + (tid & 7) * VEC_;
                    } else {
                        constexpr int CPR = TS / VEC_;
                        const int id = tid + NT * i;
                        const int col = CLAMP ? min(r0 + (id % CPR) * VEC_, R - VEC_) : r0 + (id % CPR) * VEC_;
                        ptr[i] = base + ((int64_t)kc0 * BK + id / CPR) * ld + col;
                    }
                }
            };
            init_ptrs(pa, std::integral_constant<bool, A_KMAJ>{}, A, p.lda, m0, p.M);
            init_ptrs(pb, std::integral_constant<bool, B_KMAJ>{}, B, p.ldb, n0, p.N);
        }
        // a chunk along k, in bytes (BUF) / elements (!BUF)
        const int64_t stepA = (A_KMAJ ? (int64_t)BK : (int64_t)BK * p.lda) * (BUF ? (int64_t)sizeof(T) : 1);
        const int64_t stepB = (B_KMAJ ? (int64_t)BK : (int64_t)BK * p.ldb) * (BUF ? (int64_t)sizeof(T) : 1);
        // one vector of the next-but-one chunk: global -> registers (and the origin / pointer moves on by a chunk)
        // (they stop at the tile's last chunk: past the end of the k range the loop re-reads that chunk and nobody uses it --
        // one loop body for every chunk instead of three tail variants, which cost 800 bytes of scratch)
        int64_t curA = 0, curB = 0;
        auto g_issue1 = [&](int j) {
            if constexpr (BUF) {
                if (j < NV) {
                    qa[j] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc(ua), la, (unsigned)j * passA, 0));
                    if (j == NV - 1) ua += curA;
                } else {
                    qb[j - NV] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc(ub), lb, (unsigned)(j - NV) * passB, 0));
                    if (j == 2 * NV - 1) ub += curB;
                }
            } else {
                if (j < NV) { qa[j] = *reinterpret_cast<const vec_t*>(pa[j]); pa[j] += curA; }
                else { qb[j - NV] = *reinterpret_cast<const vec_t*>(pb[j - NV]); pb[j - NV] += curB; }
            }
        };
        int adv = nk_pipe - kc0 - 1;            // how many more times they may move on
        auto g_arm = [&]() { curA = adv > 0 ? stepA : 0; curB = adv > 0 ? stepB : 0; --adv; };
        // one vector of the next chunk: registers -> LDS stage
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
                // two consecutive k values per lane: step 2 * ph' + e of the chunk <-> k = 8 ph + 2 kq + e, the same permutation
                // of the contraction order for both operands
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
        // fragment j of a phase (0 .. FRM - 1: A, FRM .. FRM + FR - 1: B): LDS -> register set
        auto f_read1 = [&](auto set_c, int stage, int ph, int j) {
            constexpr int set = decltype(set_c)::value;
            const char* sA = smem + stage * STAGE;
            if (j < FRM) Fa[set][j] = frag(std::integral_constant<bool, A_KMAJ>{}, sA, wm * WTM + j * 16, ph);
            else Fb[set][j - FRM] = frag(std::integral_constant<bool, B_KMAJ>{}, sA + OPB, wn * WT + (j - FRM) * 16, ph);
        };
        // A phase is NSL slices: one fragment read, one vector moved, CPS fragment products each.  128-tile: 8 slices of 2 products
        // (fragment column q / 2, rows 2 (q % 2), 2 (q % 2) + 1); 64-tile: 4 slices of 1.
        // (the formulas also cover 8 waves on the 128-tile -- 4 slices of 2 products, the 6 fragments read 2 + 2 + 2 + 0: measured in round 4,
        // see NW above)
        constexpr int NSL = 2 * NV;                                  // one vector moved per slice
        constexpr int CPS = FRM * FR / NSL;                          // fragment products per slice
        constexpr int NFG = FRM + FR;                                // fragments of a phase
        constexpr int RPS = (NFG + NSL - 1) / NSL;                   // fragments read per slice
        constexpr int RPS2 = (NFG + NSL / 2 - 1) / (NSL / 2);        // ... in the half phase behind the barrier
        static_assert(CPS * NSL == FRM * FR && NSL % 2 == 0 && CPS >= 1, "slices cover the fragment products and the vectors of a chunk");
        auto f_reads = [&](auto set_c, int stage, int ph, int q, int per) {      // the fragments [q per, (q + 1) per) of a phase
#pragma unroll
            for (int u = 0; u < per; ++u)
                if (q * per + u < NFG) f_read1(set_c, stage, ph, q * per + u);
        };
        auto f_mma1 = [&](auto set_c, int q) {
            constexpr int set = decltype(set_c)::value;
#pragma unroll
            for (int e = 0; e < (sizeof(T) == 8 ? 1 : 2); ++e)
#pragma unroll
                for (int u = 0; u < CPS; ++u) {
                    const int idx = q * CPS + u, fj = idx / FRM, fi = idx % FRM;
                    if constexpr (sizeof(T) == 8) acc[0][fi][fj] = Traits<T>::mfma(Fa[set][fi], Fb[set][fj], acc[0][fi][fj]);
                    else acc[0][fi][fj] = Traits<T>::mfma((T)Fa[set][fi][e], (T)Fb[set][fj][e], acc[0][fi][fj]);
                }
        };
        typedef std::integral_constant<int, 0> F0;
        typedef std::integral_constant<int, 1> F1;
        // the next tile's C lines, one per thread and chunk over the last PFN chunks (see pf_c above); every other chunk the same
        // instruction re-reads an operand address that is in the L1 anyway -- the loop body stays free of branches
        constexpr int LPR = TS * (int)sizeof(T) / 128;          // cache lines per tile row
        constexpr int PFN = TS == 128 ? TS * LPR / NT : 0;      // lines per thread: 4 (fp64) / 2 (fp32)
        const bool pf_on = pf_c != nullptr && TS == 128 && (!BUF || pf_ld < GPK_PIPE_LD_MAX);
        const int64_t pf_step = (int64_t)(NT / (LPR > 0 ? LPR : 1)) * pf_ld * (int64_t)sizeof(T);
        const unsigned pf_off = ((unsigned)(tid / (LPR > 0 ? LPR : 1)) * (unsigned)pf_ld) * (unsigned)sizeof(T) + (unsigned)(tid % (LPR > 0 ? LPR : 1)) * 128u;
        const char* pf_base = BUF ? uni(pf_c) : nullptr;        // BUF: uniform origin + one 32-bit lane offset, like the operands
        const char* pf_lane = (BUF || !pf_on) ? nullptr          // !BUF: a per-thread pointer
                                              : reinterpret_cast<const char*>(pf_c + (int64_t)(tid / (LPR > 0 ? LPR : 1)) * pf_ld) + (tid % (LPR > 0 ? LPR : 1)) * 128;
        int left = nk_pipe - kc0;               // chunks of this tile still to be multiplied (>= 1)
        int pfv = 0;
        auto pf_tick = [&]() {
            asm volatile("" ::"v"(pfv));        // (the previous one is back: it is a chunk old)
            const int j = left - 3;             // chunks left - 3 = PFN - 1 ... 0: the tile's last lines go last
            const bool on = pf_on && j >= 0 && j < PFN;         // (uniform)
            if constexpr (BUF) {
                pfv = (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc(on ? pf_base + j * pf_step : ua), on ? pf_off : la, 0, 0);
            } else {
                const char* a = on ? pf_lane + j * pf_step : reinterpret_cast<const char*>(pa[0]);
                pfv = *reinterpret_cast<const int*>(a);
            }
        };
        // One chunk.  On entry: LDS stage `stage` holds the chunk, fragment set 0 its phase 0, qa / qb the next chunk; it writes that
        // one to the other stage, requests the one after it and reads the next chunk's phase 0 at the end.
        // The source order IS the schedule: sched_barrier(0) after every slice keeps the compiler from regrouping it.
        auto chunk = [&](int stage) {
            constexpr bool W = true, G = true;
            g_arm();
#pragma unroll
            for (int q = 0; q < NSL; ++q) {          // phase 0: multiply set 0, read phase 1 into set 1, write the next chunk
                f_mma1(F0{}, q);
                f_reads(F1{}, stage, 1, q, RPS);
                if constexpr (W) g_commit1(stage ^ 1, q);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < NSL; ++q) {          // phase 1: multiply set 1, read phase 2 into set 0, request the chunk after next
                f_mma1(F1{}, q);
                f_reads(F0{}, stage, 2, q, RPS);
                if constexpr (G) g_issue1(q);
                if (q == NSL - 1) pf_tick();         // (behind the chunk's own loads: nothing younger than it is waited for within the next chunk)
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < NSL; ++q) {          // phase 2: multiply set 0, read phase 3 into set 1
                f_mma1(F0{}, q);
                f_reads(F1{}, stage, 3, q, RPS);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < NSL / 2; ++q) f_mma1(F1{}, q);     // phase 3, first half
            __builtin_amdgcn_sched_barrier(0);
            // every wave has read this stage and written the other one
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = NSL / 2; q < NSL; ++q) {    // phase 3, second half: read phase 0 of the next chunk into set 0
                f_mma1(F1{}, q);
                if constexpr (W) f_reads(F0{}, stage ^ 1, 0, q - NSL / 2, RPS2);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto g_issue = [&]() {
#pragma unroll
            for (int j = 0; j < 2 * NV; ++j) g_issue1(j);
        };
        g_arm();
        g_issue();
#pragma unroll
        for (int j = 0; j < 2 * NV; ++j) g_commit1(0, j);
        __syncthreads();
        if (prof != nullptr && threadIdx.x == 0) prof[6] = wall_clock64();      // C values and the first chunk have arrived
        g_arm();
        g_issue();
#pragma unroll
        for (int j = 0; j < FRM + FR; ++j) f_read1(F0{}, 0, 0, j);
        while (true) {
            chunk(0);
            if (--left == 0) break;
            chunk(1);
            if (--left == 0) break;
        }
        asm volatile("" ::"v"(pfv));
    }
    if (EDGE && PIPE >= 2 && pipe_tile && k_tail) {          // the partial last chunk
        issue(S0{}, nk - 1);
        __syncthreads();                        // (the waves' last fragment reads of the loop above)
        commit(S0{}, 0);
        __syncthreads();
        mma(0, nk - 1);
        __syncthreads();
    }
    if (!pipe_tile) {
    if (PF2) {
        issue(S0{}, kc0);
        if (kc0 + 1 < nk) issue(S1{}, kc0 + 1);
        commit(S0{}, 0);
        __syncthreads();
        for (int kc = kc0; kc < nk; kc += 2) {
            // LDS stage 0 holds chunk kc, register set 1 holds chunk kc + 1
            if (kc + 2 < nk) issue(S0{}, kc + 2);
            mma(0, kc);
            if (kc + 1 >= nk) break;
            commit(S1{}, 1);
            __syncthreads();
            // LDS stage 1 holds chunk kc + 1, register set 0 holds chunk kc + 2
            if (kc + 3 < nk) issue(S1{}, kc + 3);
            mma(1, kc + 1);
            if (kc + 2 < nk) commit(S0{}, 0);
            __syncthreads();
        }
    } else {
        issue(S0{}, kc0);
        commit(S0{}, 0);
        __syncthreads();
        for (int kc = kc0; kc < nk; ++kc) {
            const bool more = (kc + 1 < nk);
            if (more) issue(S0{}, kc + 1);
            mma((kc - kc0) & 1, kc);
            if (more) commit(S0{}, (kc + 1 - kc0) & 1);
            __syncthreads();
        }
    }
    }

    if (prof != nullptr && threadIdx.x == 0) {      // k loop done
        prof[2] = wall_clock64();
        prof[5] = (long long)__builtin_readcyclecounter();
    }
    // (persistent kernel) thread 0 claims the workgroup's next-but-one task HERE: no load and no store of this wave is outstanding, so
    // waiting for the atomic's result costs one round trip of wave 0 and nothing else
    if (claim_ctr != nullptr && threadIdx.x == 0) *claimed = gpk_claim_task(claim_ctr, chain_ctr, chain_slots);
    // (in-place use: every global read of this workgroup's rows of A happened above)
    // Fused column statistics / scaling (round 4; `gpk_gemm_colscale`, what SURVEY 8(b) called gpk_syrk_scaled, split where the path
    // needs it): V = L_z^{-1} K_zx leaves this kernel already multiplied by K_n^{-1/2} per column, and the column sums of squares of the
    // unscaled V (`B.matmul_diag`, observations.py:305) are written per 64-row slab on the way -- the stand-alone scaling pass and the
    // reduction pass over the 3.3 GB of V are gone.
    if constexpr (TS == 128 && NCT == 1 && NW == 4 && !TRIB) {
        if (p.colscale != nullptr || p.colss != nullptr) {
#pragma unroll
            for (int fj = 0; fj < FR; ++fj) {
                const int col = n0 + wn * WT + fj * 16 + lr;
                const bool cvalid = !EDGE || col < p.N;
                T ss = T(0);
#pragma unroll
                for (int fi = 0; fi < FRM; ++fi)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const bool rvalid = !EDGE || m0 + wm * WTM + fi * 16 + Traits<T>::crow(lane, i) < p.M;      // (rows past M: computed from clamped rows)
                        const T v = rvalid ? p.alpha * acc[0][fi][fj][i] : T(0);
                        ss += v * v;
                    }
                ss += __shfl_xor(ss, 16);
                ss += __shfl_xor(ss, 32);
                if (p.colss != nullptr && kq == 0 && cvalid) p.colss[(int64_t)(2 * ti + wm) * p.ldss + col] = ss;
                if (p.colscale != nullptr) {
                    const T sc = cvalid ? p.colscale[col] : T(1);
#pragma unroll
                    for (int fi = 0; fi < FRM; ++fi)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[0][fi][fj][i] *= sc;
                }
            }
        }
    }
    // (Round 4 also built the store of a finished 128-tile THROUGH LDS -- the tile laid out row-major in the free operand stages and
    // written as 16-byte stores of whole rows, 16 wave-instructions per thread instead of 64, full 128-byte lines -- and measured it
    // against this loop on one box: fp64 K = 1024 update 62.7 vs 64.0 TFLOP/s, fp32 137 vs 138, POTRF fp64 26.3-26.7 vs 26.1 ms, batched
    // fp32 POTRF 15.3 vs 15.55 ms: the three extra barriers cost what the wider stores save.  profiles/r04_ab_c_tile_through_lds.log.)
    {
#pragma unroll
    for (int c = 0; c < NCW; ++c)
#pragma unroll
        for (int fi = 0; fi < FRM; ++fi)
#pragma unroll
            for (int fj = 0; fj < FR; ++fj)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = m0 + wm * WTM + fi * 16 + Traits<T>::crow(lane, i);
                    const int col = n0 + (cfirst + c) * TS + wn * WT + fj * 16 + lr;
                    if (!EDGE || (row < p.M && col < p.N))
                        *reinterpret_cast<T*>(reinterpret_cast<char*>(C + c_base(p.ldc, c, fi, fj, i)) + lane_off_out) = p.alpha * acc[c][fi][fj][i];
                }
    }
    if (prof != nullptr) {
        __builtin_amdgcn_s_waitcnt(0);          // (profiling only) stores retired
        if (threadIdx.x == 0) prof[3] = wall_clock64();
    }
}

}  // namespace
