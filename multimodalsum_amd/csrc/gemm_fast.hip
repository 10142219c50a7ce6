// bf16 GEMMs with direct global->LDS DMA (global_load_lds_dwordx4) for gfx950.
//
//   NT:  C[m][n] = epi(alpha * sum_k A[m][k] * B[n][k] + bias[n]) (+C),   A [M,K], B [N,K] both K-contiguous
//        (forward x W^T; input gradients dy (W^T)^T through the transposed weight shadow)
//   TN:  C[m][n] = sum_k A[k][m] * B[k][n],   A [K,M], B [K,N] both reduction-major
//        (weight gradients dy^T x straight from the activations)
//
// 256x256 tiles (the step's products) run on FOUR waves of 128x128 each (gemm_nt_w4_kernel, gemm_tn_w4_kernel: one wave per
// SIMD, accumulators in all 256 AGPRs, source-level software pipeline, every epilogue form); 256x128 / 128x128 tiles (small
// outputs) on the eight- / four-wave ring kernels (gemm_nt_ring_kernel, gemm_tn_ring_kernel).  Operands live in LDS in a
// swizzled image that the DMA writes linearly (wave-uniform base + lane*16 B): the XOR is applied to the per-lane SOURCE
// address and again on the fragment read (guide rule 21).  Rows past M/N are clamped (their results are never stored).
#include "gemm_common.h"
#include <stdlib.h>
#include <algorithm>

namespace {

// ---------------------------------------------------------------------------------------------
// Epilogue through LDS.  In the accumulator layout a lane owns ONE column and 16 scattered rows, so a
// direct store is 16 two-byte stores per 32x32 tile (128 per lane for a 256x256 tile) and the store
// issue, not the MFMAs, bounds every K<=4096 product.  Instead each 32-row band of the block tile is
// staged in LDS as f32 [32][BN] (the operand stages are dead by now) and written back row-wise:
// 8 consecutive columns per thread = one 16-byte store (two for f32 outputs), with the bias / GELU /
// ReLU / accumulate work vectorised on the same 8 columns.
// ---------------------------------------------------------------------------------------------
// The bias of the 8 columns a thread stores in every pass (0 where there is none).  Loaded ONCE per tile and retired here:
// a load inside the store loop makes the compiler wait vmcnt(0) there, and vmcnt retires in order, so that wait also sits
// out the acknowledgement of every earlier store -- the tile would leave one 16-byte store group at a time (measured: a
// biased M=20480 N=K=1024 product took 124 us against 75 us for the same product without bias; 73 us with this).
template <int BN, int THREADS>
__device__ __forceinline__ void epilogue_bias(const GemmArgs& p, int n0, int ks, int tid, float (&bv)[8]) {
    constexpr int CPR = BN / 8;
    const bool has_bias = (p.flags & MMSUM_GEMM_BIAS) && (ks == 0) && !(p.flags & MMSUM_GEMM_COLSUM);
    const int col = n0 + (tid % CPR) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = 0.f;
    if (has_bias) {
        if (col + 8 <= p.N && (((uintptr_t)p.bias) & 15) == 0) {
            const f32x4_t a = *reinterpret_cast<const f32x4_t*>(p.bias + col);
            const f32x4_t b = *reinterpret_cast<const f32x4_t*>(p.bias + col + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { bv[e] = a[e]; bv[4 + e] = b[e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (col + e < p.N) bv[e] = p.bias[col + e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(bv[e]));
}

// Workgroup barrier that orders LDS traffic only: the wave's LDS operations are retired (lgkmcnt) and then the raw s_barrier
// is taken.  __syncthreads() must not be used in the epilogue: its fence also waits vmcnt(0), i.e. for the acknowledgement of
// every global store issued so far -- with one barrier per staged band the tile then left in four bursts, each paying a full
// store round trip (measured at M=64,512: the epilogue took 40-52 % of every K=1024 launch; the main loop alone runs at
// 1.3-1.5 PFLOP/s).  With this barrier the stores of one band stay in flight under the staging of the next, and the last
// band's stores under the next tile's ring fill.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// Where a pass of the staged epilogue takes the accumulators of row band i from: the kernel's register array, or (four-wave TN
// kernel) accumulators that live in named AGPRs and are read out one band at a time.
struct AccArray {
    template <int TN> __device__ __forceinline__ void operator()(int i, const f32x16_t (*acc)[TN], f32x16_t (&band)[TN]) const {
#pragma unroll
        for (int j = 0; j < TN; ++j) band[j] = acc[i][j];
    }
};

// Edge tiles (rows past M / columns past N inside the tile, unaligned C): every access guarded, scalar fallbacks.
template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, int OUT, int LDS_BYTES = 4 * (BM + BN) * SLAB_BYTES, int LAY = LAY_32, typename AccSrc = AccArray>
__device__ __forceinline__ void epilogue_edge(const GemmArgs& p, const f32x16_t (&acc)[BM / WAVES_M / 32][BN / WAVES_N / 32],
                                           char* smem, int m0, int n0, int ks, int wm, int wn, int tid, int lane, const float (&bv)[8], AccSrc src = AccSrc{}) {
    constexpr int TM = BM / WAVES_M / 32, TN = BN / WAVES_N / 32;
    constexpr int THREADS = WAVES_M * WAVES_N * 64;
    constexpr int CPR = BN / 8;                       // 8-column chunks per row
    constexpr int BAND = 32 * BN;                     // floats per staged 32-row band
    constexpr int CHUNKS = WAVES_M * 32 * CPR;        // per pass: one band of every wave row
    // pass i stages tile row i of EVERY wave row (WAVES_M bands side by side); with room for two sets of bands the
    // passes alternate between them and one barrier per pass orders everything (the readers of a set are two
    // barriers behind its next writers), otherwise a second barrier guards the reuse
    constexpr int NSETS = (LDS_BYTES >= 2 * WAVES_M * BAND * 4) ? 2 : 1;
    constexpr int NIT = (CHUNKS + THREADS - 1) / THREADS;
    constexpr bool kAuxIn = (EPI == MMSUM_EPI_GELU_BWD || EPI == MMSUM_EPI_RELU_BWD);
    constexpr bool kPre = kAuxIn || OUT == OUT_T_ACC || OUT == OUT_F32_ACC;       // the pass reads something from global memory
    static_assert(LDS_BYTES >= WAVES_M * BAND * 4, "epilogue staging does not fit the kernel's LDS");
    float* stage0 = reinterpret_cast<float*>(smem);
    const bool do_colsum = (p.flags & MMSUM_GEMM_COLSUM) != 0;     // bias slot = f32 output: += column sums of the stored tile
    const bool do_colsq = (p.flags & MMSUM_GEMM_COLSUM2) != 0;     // ... and, N floats further, += column sums of its squares
    static_assert(THREADS % CPR == 0, "a thread must own one 8-column chunk for the column sums");
    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, csq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bf16_t* Ct = static_cast<bf16_t*>(p.C);
    float* Cf = static_cast<float*>(p.C);
    bf16_t* aux = static_cast<bf16_t*>(p.aux);
    constexpr bool kF32Out = (OUT == OUT_F32_ACC || OUT == OUT_F32_ATOMIC || OUT == OUT_F32);
    const bool vec_ok = kF32Out ? ((((uintptr_t)p.C) & 15) == 0 && (p.ldc & 3) == 0)
                                : ((((uintptr_t)p.C) & 15) == 0 && (p.ldc & 7) == 0);
    const bool aux_vec = aux != nullptr && ((((uintptr_t)p.aux) & 15) == 0) && ((p.ldaux & 7) == 0);
    // What a pass reads from global memory (saved pre-activation of the *_BWD epilogues, C of an accumulating store) is
    // requested one pass AHEAD, before the previous pass's stores are issued: vmcnt retires in issue order, so a load issued
    // behind stores would make its consumer wait for their acknowledgement; issued in front of them it only waits for itself.
    // ONE register set: an iteration consumes its operand and at once requests the same iteration's operand of the NEXT pass,
    // before its own store is issued.
    u32x4_t aux_pre[NIT], c_pre[NIT];
    f32x4_t cf_pre[NIT][2];
    auto prefetch1 = [&](int i, int it) {
        if constexpr (kPre) {
            const int c = tid + it * THREADS;
            const int wr = c / (32 * CPR), lr = (c / CPR) % 32, cc = (c % CPR) * 8;
            const int row = m0 + wr * (TM * 32) + i * 32 + lr, col = n0 + cc;
            // unconditional loads from clamped (always valid) addresses: a load inside a divergent branch makes hipcc fall
            // back to vmcnt(0) at its use; rows / columns past the edge are simply not stored
            const long rc = row < p.M ? row : p.M - 1, cl = col < p.N ? col : p.N - 8;
            if constexpr (kAuxIn) aux_pre[it] = *reinterpret_cast<const u32x4_t*>(aux + rc * p.ldaux + cl);
            if constexpr (OUT == OUT_T_ACC) c_pre[it] = *reinterpret_cast<const u32x4_t*>(Ct + rc * p.ldc + cl);
            if constexpr (OUT == OUT_F32_ACC) {
                cf_pre[it][0] = *reinterpret_cast<const f32x4_t*>(Cf + rc * p.ldc + cl);
                cf_pre[it][1] = *reinterpret_cast<const f32x4_t*>(Cf + rc * p.ldc + cl + 4);
            }
        }
    };
#pragma unroll
    for (int it = 0; it < NIT; ++it) prefetch1(0, it);
    lds_barrier();                                    // every wave is done reading the operand stages
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        float* set = stage0 + (NSETS == 2 ? (i & 1) * (WAVES_M * BAND) : 0);
        if (NSETS == 1 && i > 0) lds_barrier();
        {
            float* stage = set + wm * BAND;
            f32x16_t band[TN];
            src(i, acc, band);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = wn * (TN * 32) + j * 32;
#pragma unroll
                for (int r = 0; r < 16; ++r) stage[blk_row<LAY>(r, lane) * BN + col + blk_col<LAY>(r, lane)] = band[j][r];
            }
        }
        lds_barrier();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = tid + it * THREADS;
            if (c >= CHUNKS) break;
            const int wr = c / (32 * CPR), lr = (c / CPR) % 32, cc = (c % CPR) * 8;
            const float* stage = set + wr * BAND;
            const int row = m0 + wr * (TM * 32) + i * 32 + lr, col = n0 + cc;
            // take this iteration's prefetched operands and request the next pass's at once (unconditionally: see prefetch1)
            u32x4_t aux_now, c_now;
            f32x4_t cf_now[2];
            if constexpr (kAuxIn) aux_now = aux_pre[it];
            if constexpr (OUT == OUT_T_ACC) c_now = c_pre[it];
            if constexpr (OUT == OUT_F32_ACC) { cf_now[0] = cf_pre[it][0]; cf_now[1] = cf_pre[it][1]; }
            if (i + 1 < TM) prefetch1(i + 1, it);
            if (row >= p.M || col >= p.N) continue;
            float v[8];
            {
                const f32x4_t a = *reinterpret_cast<const f32x4_t*>(stage + lr * BN + cc);
                const f32x4_t b = *reinterpret_cast<const f32x4_t*>(stage + lr * BN + cc + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
            }
            const int nvalid = min(8, p.N - col);
            const bool full = nvalid == 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * p.alpha + bv[e];
            const long o = (long)row * p.ldc + col;
            if constexpr (EPI == MMSUM_EPI_GELU) {
                if (aux) {
                    const long oa = (long)row * p.ldaux + col;
                    if (full && aux_vec) {
                        bf16_t t[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) t[e] = (bf16_t)v[e];
                        u32x4_t w;
                        __builtin_memcpy(&w, t, 16);
                        *reinterpret_cast<u32x4_t*>(aux + oa) = w;
                    } else {
                        for (int e = 0; e < nvalid; ++e) aux[oa + e] = (bf16_t)v[e];
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gelu_fast_f(v[e]);
            } else if constexpr (EPI == MMSUM_EPI_GELU_BWD || EPI == MMSUM_EPI_RELU_BWD) {
                bf16_t t[8];                          // always the prefetched vector: the host admits these epilogues for
                __builtin_memcpy(t, &aux_now, 16);                  // aligned operands and N % 8 == 0 only (no scalar path)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if constexpr (EPI == MMSUM_EPI_GELU_BWD) v[e] *= gelu_grad_fast_f((float)t[e]);
                    else v[e] = ((float)t[e] > 0.f) ? v[e] : 0.f;
                }
            } else if constexpr (EPI == MMSUM_EPI_RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (do_colsum) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float t = (e < nvalid) ? (float)(bf16_t)v[e] : 0.f;     // what a separate pass over the bf16 result would add
                    csum[e] += t;
                    csq[e] = fmaf(t, t, csq[e]);
                }
            }
            if constexpr (OUT == OUT_T_ACC) {
                bf16_t t[8];
                __builtin_memcpy(t, &c_now, 16);
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = (bf16_t)(v[e] + (float)t[e]);
                u32x4_t w;
                __builtin_memcpy(&w, t, 16);
                *reinterpret_cast<u32x4_t*>(Ct + o) = w;
            } else if constexpr (OUT == OUT_F32_ACC) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    *reinterpret_cast<f32x4_t*>(Cf + o + 4 * h) = f32x4_t{v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]} + cf_now[h];
            } else if constexpr (OUT == OUT_T) {
                if (full && vec_ok) {
                    bf16_t t[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) t[e] = (bf16_t)v[e];
                    u32x4_t w;
                    __builtin_memcpy(&w, t, 16);
                    *reinterpret_cast<u32x4_t*>(Ct + o) = w;
                } else {
                    for (int e = 0; e < nvalid; ++e) Ct[o + e] = (bf16_t)v[e];
                }
            } else if constexpr (OUT == OUT_F32_ATOMIC) {
                for (int e = 0; e < nvalid; ++e) atomicAdd(Cf + o + e, v[e]);
            } else {
                if (full && vec_ok) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) *reinterpret_cast<f32x4_t*>(Cf + o + 4 * h) = f32x4_t{v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]};
                } else {
                    for (int e = 0; e < nvalid; ++e) Cf[o + e] = v[e];
                }
            }
        }
    }
    if (do_colsum) {
        // every thread owns one 8-column chunk (tid % CPR) in all passes: fold the THREADS / CPR partials through LDS
        float* red = reinterpret_cast<float*>(smem);
        constexpr int NPART = THREADS / CPR;
        for (int w = 0; w < (do_colsq ? 2 : 1); ++w) {
            lds_barrier();
#pragma unroll
            for (int e = 0; e < 8; ++e) red[(tid / CPR) * BN + (tid % CPR) * 8 + e] = w ? csq[e] : csum[e];
            lds_barrier();
            for (int c = tid; c < BN; c += THREADS) {
                float t = 0.f;
#pragma unroll
                for (int k = 0; k < NPART; ++k) t += red[k * BN + c];
                if (n0 + c < p.N) atomicAdd(const_cast<float*>(p.bias) + (long)w * p.N + n0 + c, t);
            }
        }
    }
}

// Interior tiles (the whole BM x BN tile inside the matrix, 16-byte aligned C / aux): the same staging, with everything that
// does not depend on data taken out of the loop -- the staged epilogue above spends ~2,600 instructions per wave and tile on
// guards, index arithmetic and run-time flags (measured at M = 64,512, N = 4096, K = 1024: 232 us of a 696 us launch with the
// global stores REMOVED, against 386 us for the main loop alone).  Here a thread's rows are base + compile-time constants,
// the LDS addresses are immediates, and bias / alpha / activation / accumulate / column sums are compile-time forms.
template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, int OUT, int CS, int LDS_BYTES, int LAY = LAY_32, typename AccSrc = AccArray>
__device__ __forceinline__ void epilogue_interior(const GemmArgs& p, const f32x16_t (&acc)[BM / WAVES_M / 32][BN / WAVES_N / 32],
                                                  char* smem, int m0, int n0, int wm, int wn, int tid, int lane, const float (&bv)[8], AccSrc src = AccSrc{}) {
    constexpr int TM = BM / WAVES_M / 32, TN = BN / WAVES_N / 32;
    constexpr int THREADS = WAVES_M * WAVES_N * 64;
    // LAY_16T: a lane holds four consecutive columns of a row per quarter -> one 16-byte LDS write per quarter (64 per wave and
    // tile instead of 256 four-byte ones) into rows padded by four floats (the 16 rows of a write then start 4 banks apart)
    constexpr int PITCH = LAY == LAY_16T ? BN + 4 : BN;
    constexpr int CPR = BN / 8, BAND = 32 * PITCH;
    constexpr int RPI = THREADS / CPR;                 // rows one iteration of the write-back covers
    constexpr int NIT = WAVES_M * 32 / RPI;
    static_assert(THREADS % CPR == 0 && RPI <= 32 && 32 % RPI == 0 && NIT * RPI == WAVES_M * 32, "write-back geometry");
    constexpr int NSETS = (LDS_BYTES >= 2 * WAVES_M * BAND * 4) ? 2 : 1;
    constexpr bool kAuxIn = (EPI == MMSUM_EPI_GELU_BWD || EPI == MMSUM_EPI_RELU_BWD);
    constexpr bool kPre = kAuxIn || OUT == OUT_T_ACC || OUT == OUT_F32_ACC;
    constexpr bool kF32Out = (OUT == OUT_F32_ACC || OUT == OUT_F32);
    float* stage0 = reinterpret_cast<float*>(smem);
    const int trow = tid / CPR, cc = (tid % CPR) * 8;
    // row of iteration `it` of pass `i` = m0 + trow + ROW(i, it), ROW a compile-time constant
    auto row_of = [](int i, int it) { return ((it * RPI) / 32) * (TM * 32) + i * 32 + (it * RPI) % 32; };
    const long cbase = (long)(m0 + trow) * p.ldc + n0 + cc;
    const long abase = (long)(m0 + trow) * p.ldaux + n0 + cc;
    bf16_t* Ct = static_cast<bf16_t*>(p.C);
    float* Cf = static_cast<float*>(p.C);
    bf16_t* aux = static_cast<bf16_t*>(p.aux);
    const float alpha = p.alpha;
    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, csq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // CS: 1 = column sums, 2 = + sums of squares
    // ONE register set: an iteration consumes its operand and at once requests the same iteration's operand of the NEXT pass,
    // before its own store is issued (vmcnt retires in issue order: a load issued behind stores would wait for their
    // acknowledgement; issued in front of them it only waits for the stores of a pass ago).
    u32x4_t aux_pre[NIT], c_pre[NIT];
    f32x4_t cf_pre[NIT][2];
    auto prefetch1 = [&](int i, int it) {
        if constexpr (kPre) {
            const long r = row_of(i, it);
            if constexpr (kAuxIn) aux_pre[it] = *reinterpret_cast<const u32x4_t*>(aux + abase + r * p.ldaux);
            if constexpr (OUT == OUT_T_ACC) c_pre[it] = *reinterpret_cast<const u32x4_t*>(Ct + cbase + r * p.ldc);
            if constexpr (OUT == OUT_F32_ACC) {
                cf_pre[it][0] = *reinterpret_cast<const f32x4_t*>(Cf + cbase + r * p.ldc);
                cf_pre[it][1] = *reinterpret_cast<const f32x4_t*>(Cf + cbase + r * p.ldc + 4);
            }
        }
    };
#pragma unroll
    for (int it = 0; it < NIT; ++it) prefetch1(0, it);
    lds_barrier();                                    // every wave is done reading the operand stages
    const float* rd0 = stage0 + trow * PITCH + cc;    // this thread's read position inside a band set (+ compile-time offsets)
    float* wr0 = stage0 + wm * BAND + wn * (TN * 32) +
                 (LAY == LAY_16T ? (lane & 15) * PITCH + 4 * (lane >> 4) : LAY == LAY_16 ? (4 * (lane >> 4)) * BN + (lane & 15) : (4 * (lane >> 5)) * BN + (lane & 31));
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        constexpr int SETF = WAVES_M * BAND;
        const int seto = (NSETS == 2 ? (i & 1) * SETF : 0);
        if (NSETS == 1 && i > 0) lds_barrier();
        f32x16_t band[TN];
        src(i, acc, band);
        if constexpr (LAY == LAY_16T) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<f32x4_t*>(wr0 + seto + (16 * (q >> 1)) * PITCH + j * 32 + 16 * (q & 1)) =
                        f32x4_t{band[j][4 * q], band[j][4 * q + 1], band[j][4 * q + 2], band[j][4 * q + 3]};
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    wr0[seto + (LAY == LAY_16 ? (16 * (r >> 3) + (r & 3)) * BN + 16 * ((r >> 2) & 1) : ((r & 3) + 8 * (r >> 2)) * BN) + j * 32] = band[j][r];
        }
        lds_barrier();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            u32x4_t aux_now, c_now;
            f32x4_t cf_now[2];
            if constexpr (kAuxIn) aux_now = aux_pre[it];
            if constexpr (OUT == OUT_T_ACC) c_now = c_pre[it];
            if constexpr (OUT == OUT_F32_ACC) { cf_now[0] = cf_pre[it][0]; cf_now[1] = cf_pre[it][1]; }
            if (i + 1 < TM) prefetch1(i + 1, it);
            const float* src = rd0 + seto + ((it * RPI) / 32) * BAND + ((it * RPI) % 32) * PITCH;
            const f32x4_t a = *reinterpret_cast<const f32x4_t*>(src);
            const f32x4_t b = *reinterpret_cast<const f32x4_t*>(src + 4);
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = fmaf(a[e], alpha, bv[e]); v[4 + e] = fmaf(b[e], alpha, bv[4 + e]); }
            const long r = row_of(i, it);
            if constexpr (EPI == MMSUM_EPI_GELU) {
                if (aux != nullptr) {
                    bf16_t t[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) t[e] = (bf16_t)v[e];
                    u32x4_t w;
                    __builtin_memcpy(&w, t, 16);
                    *reinterpret_cast<u32x4_t*>(aux + abase + r * p.ldaux) = w;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gelu_fast_f(v[e]);
            } else if constexpr (kAuxIn) {
                bf16_t t[8];
                __builtin_memcpy(t, &aux_now, 16);
                if constexpr (EPI == MMSUM_EPI_GELU_BWD) {
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const gelu_f32x2_t g = gelu_grad_fast2(gelu_f32x2_t{(float)t[e], (float)t[e + 1]});
                        v[e] *= g.x;
                        v[e + 1] *= g.y;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = ((float)t[e] > 0.f) ? v[e] : 0.f;
                }
            } else if constexpr (EPI == MMSUM_EPI_RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if constexpr (kF32Out) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x4_t w = f32x4_t{v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]};
                    if constexpr (OUT == OUT_F32_ACC) w = w + cf_now[h];
                    *reinterpret_cast<f32x4_t*>(Cf + cbase + r * p.ldc + 4 * h) = w;
                }
            } else {
                bf16_t t[8];
                if constexpr (OUT == OUT_T_ACC) {
                    __builtin_memcpy(t, &c_now, 16);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)t[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = (bf16_t)v[e];
                if constexpr (CS != 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float tf = (float)t[e];                                       // what a separate pass over the bf16 result would add
                        csum[e] += tf;
                        if constexpr (CS == 2) csq[e] = fmaf(tf, tf, csq[e]);
                    }
                }
                u32x4_t w;
                __builtin_memcpy(&w, t, 16);
                *reinterpret_cast<u32x4_t*>(Ct + cbase + r * p.ldc) = w;
            }
        }
    }
    if constexpr (CS != 0) {
        // every thread owns one 8-column chunk (tid % CPR) in all passes: fold the THREADS / CPR partials through LDS
        float* red = reinterpret_cast<float*>(smem);
        constexpr int NPART = THREADS / CPR;
        lds_barrier();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[(tid / CPR) * BN + (tid % CPR) * 8 + e] = csum[e];
            if constexpr (CS == 2) red[NPART * BN + (tid / CPR) * BN + (tid % CPR) * 8 + e] = csq[e];
        }
        lds_barrier();
        for (int c = tid; c < CS * BN; c += THREADS) {
            const int w = c / BN, cc = c % BN;
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < NPART; ++k) t += red[w * NPART * BN + k * BN + cc];
            atomicAdd(const_cast<float*>(p.bias) + (long)w * p.N + n0 + cc, t);
        }
    }
}

// Interior tiles of a kernel whose accumulators are in the LAY_16T layout (a lane holds four consecutive columns of a row) and
// whose epilogue reads nothing from global memory (plain / GELU / ReLU, bf16 output): alpha, bias and the activation are applied
// on the registers, the result is converted to bf16 THERE and staged as bf16 -- one 8-byte LDS write per quarter (64 per wave and
// tile, against 256 four-byte writes of the f32 staging) into rows padded by 16 bytes (the 16 rows x 4 column groups of a write
// then fall in 64 different banks), half the LDS bytes both ways, and nothing but a 16-byte LDS read and a 16-byte global store
// per 8 outputs on the way out.  The saved pre-activation of the GELU form is staged and stored the same way.  Column sums (CS)
// are taken on the way out from the bf16 values, i.e. of exactly what a separate pass over the stored tile would read.
template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, int CS, int LDS_BYTES, typename AccSrc = AccArray>
__device__ __forceinline__ void epilogue_interior_packed(const GemmArgs& p, const f32x16_t (&acc)[BM / WAVES_M / 32][BN / WAVES_N / 32],
                                                         char* smem, int m0, int n0, int ks, int wm, int wn, int tid, int lane, AccSrc src = AccSrc{}) {
    static_assert(EPI == MMSUM_EPI_NONE || EPI == MMSUM_EPI_GELU || EPI == MMSUM_EPI_RELU, "epilogues that read global memory stage in f32");
    constexpr int TM = BM / WAVES_M / 32, TN = BN / WAVES_N / 32;
    constexpr int THREADS = WAVES_M * WAVES_N * 64;
    constexpr int PITCH = BN * 2 + 16;                  // bytes per staged row
    constexpr int BANDB = 32 * PITCH, SETB = WAVES_M * BANDB;
    constexpr int CPR = BN / 8, RPI = THREADS / CPR, NIT = WAVES_M * 32 / RPI;
    constexpr bool kAux = EPI == MMSUM_EPI_GELU;
    static_assert(THREADS % CPR == 0 && RPI <= 32 && 32 % RPI == 0 && NIT * RPI == WAVES_M * 32, "write-back geometry");
    static_assert(LDS_BYTES >= (kAux ? 4 : 2) * SETB, "two sets of bands (+ two of the saved pre-activation)");
    const bool has_aux = kAux && p.aux != nullptr;     // workgroup-uniform
    bf16_t* Ct = static_cast<bf16_t*>(p.C);
    bf16_t* aux = static_cast<bf16_t*>(p.aux);
    const float alpha = p.alpha;
    // bias of this lane's columns: 4 consecutive ones per 16-column quarter, 2 TN quarters per row.  Loaded once per tile and
    // retired here (see epilogue_bias)
    f32x4_t bq[TN][2];
    {
        const bool has_bias = (p.flags & MMSUM_GEMM_BIAS) && (ks == 0) && !(p.flags & MMSUM_GEMM_COLSUM);
        const float* bsrc = p.bias + n0 + wn * (TN * 32) + 4 * (lane >> 4);
        const bool vec = (((uintptr_t)p.bias) & 15) == 0 && (n0 & 3) == 0;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int sj = 0; sj < 2; ++sj) {
                bq[j][sj] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                if (has_bias) {
                    if (vec) bq[j][sj] = *reinterpret_cast<const f32x4_t*>(bsrc + j * 32 + 16 * sj);
                    else
#pragma unroll
                        for (int e = 0; e < 4; ++e) bq[j][sj][e] = bsrc[j * 32 + 16 * sj + e];
                }
            }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int sj = 0; sj < 2; ++sj) asm volatile("" : "+v"(bq[j][sj]));
    }
    const int trow = tid / CPR, cc = (tid % CPR) * 8;
    auto row_of = [](int i, int it) { return ((it * RPI) / 32) * (TM * 32) + i * 32 + (it * RPI) % 32; };
    const long cbase = (long)(m0 + trow) * p.ldc + n0 + cc;
    const long abase = (long)(m0 + trow) * p.ldaux + n0 + cc;
    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, csq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    char* wr0 = smem + wm * BANDB + (lane & 15) * PITCH + (wn * (TN * 32) + 4 * (lane >> 4)) * 2;
    const char* rd0 = smem + trow * PITCH + cc * 2;
    auto pack4 = [](const float (&v)[4]) {
        bf16_t t[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = (bf16_t)v[e];
        u32x2_t w;
        __builtin_memcpy(&w, t, 8);
        return w;
    };
    lds_barrier();                                    // every wave is done reading the operand stages
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int seto = (i & 1) * SETB;             // the readers of a set are two barriers behind its next writers
        f32x16_t band[TN];
        src(i, acc, band);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(band[j][4 * q + e], alpha, bq[j][q & 1][e]);
                const int off = seto + (16 * (q >> 1)) * PITCH + (j * 32 + 16 * (q & 1)) * 2;
                if constexpr (kAux) {
                    if (has_aux) *reinterpret_cast<u32x2_t*>(wr0 + 2 * SETB + off) = pack4(v);
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const gelu_f32x2_t g = gelu_fast2(gelu_f32x2_t{v[e], v[e + 1]});
                        v[e] = g.x;
                        v[e + 1] = g.y;
                    }
                } else if constexpr (EPI == MMSUM_EPI_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                *reinterpret_cast<u32x2_t*>(wr0 + off) = pack4(v);
            }
        lds_barrier();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int lo = seto + ((it * RPI) / 32) * BANDB + ((it * RPI) % 32) * PITCH;
            const long r = row_of(i, it);
            const u32x4_t w = *reinterpret_cast<const u32x4_t*>(rd0 + lo);
            if constexpr (kAux) {
                if (has_aux) *reinterpret_cast<u32x4_t*>(aux + abase + r * p.ldaux) = *reinterpret_cast<const u32x4_t*>(rd0 + 2 * SETB + lo);
            }
            if constexpr (CS != 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float tf = __builtin_bit_cast(float, (e & 1) ? (w[e >> 1] & 0xffff0000u) : (w[e >> 1] << 16));
                    csum[e] += tf;
                    if constexpr (CS == 2) csq[e] = fmaf(tf, tf, csq[e]);
                }
            }
            *reinterpret_cast<u32x4_t*>(Ct + cbase + r * p.ldc) = w;
        }
    }
    if constexpr (CS != 0) {
        // every thread owns one 8-column chunk (tid % CPR) in all passes: fold the THREADS / CPR partials through LDS
        float* red = reinterpret_cast<float*>(smem);
        constexpr int NPART = THREADS / CPR;
        lds_barrier();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[(tid / CPR) * BN + (tid % CPR) * 8 + e] = csum[e];
            if constexpr (CS == 2) red[NPART * BN + (tid / CPR) * BN + (tid % CPR) * 8 + e] = csq[e];
        }
        lds_barrier();
        for (int c = tid; c < CS * BN; c += THREADS) {
            const int w = c / BN, col = c % BN;
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < NPART; ++k) t += red[w * NPART * BN + k * BN + col];
            atomicAdd(const_cast<float*>(p.bias) + (long)w * p.N + n0 + col, t);
        }
    }
}

// WIDE = the workgroup has the whole 512-register file per wave (the four-wave kernels: accumulators in AGPRs, 256 VGPRs for the
// epilogue): GELU' / ReLU' and the column sums then take the lean form too, and CS (column sums of the stored tile) is a
// compile-time property of the kernel -- one lean body per instantiation keeps the lane-constant addresses the compiler hoists
// out of the tile loop inside the register file.  On the eight-wave kernels (128 registers beside the accumulators) the
// derivative arithmetic of a pass spilled 167 registers in the lean form: they keep the guarded one, with column sums by flag.
template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, int OUT, int LDS_BYTES = 4 * (BM + BN) * SLAB_BYTES, int LAY = LAY_32, typename AccSrc = AccArray,
          bool WIDE = false, int CS = 0>
__device__ __forceinline__ void epilogue_staged(const GemmArgs& p, const f32x16_t (&acc)[BM / WAVES_M / 32][BN / WAVES_N / 32],
                                                char* smem, int m0, int n0, int ks, int wm, int wn, int tid, int lane, AccSrc src = AccSrc{}) {
    constexpr bool kF32 = (OUT == OUT_F32_ACC || OUT == OUT_F32_ATOMIC || OUT == OUT_F32);
    constexpr bool kAuxIn = (EPI == MMSUM_EPI_GELU_BWD || EPI == MMSUM_EPI_RELU_BWD);
    const bool aligned = (((uintptr_t)p.C) & 15) == 0 && (p.ldc & (kF32 ? 3 : 7)) == 0 &&
                         (p.aux == nullptr || ((((uintptr_t)p.aux) & 15) == 0 && (p.ldaux & 7) == 0));
    const bool interior = m0 + BM <= p.M && n0 + BN <= p.N && aligned;        // workgroup-uniform
    if constexpr (LAY == LAY_16T && OUT == OUT_T && !kAuxIn) {
        static_assert(WIDE, "the packed epilogue is the four-wave kernels'");
        if (interior) {
            epilogue_interior_packed<BM, BN, WAVES_M, WAVES_N, EPI, CS, LDS_BYTES, AccSrc>(p, acc, smem, m0, n0, ks, wm, wn, tid, lane, src);
            return;
        }
    }
    float bv[8];
    epilogue_bias<BN, WAVES_M * WAVES_N * 64>(p, n0, ks, tid, bv);
    if constexpr (!(LAY == LAY_16T && OUT == OUT_T && !kAuxIn) && OUT != OUT_F32_ATOMIC && (WIDE || !kAuxIn)) {
        if (interior && (WIDE || !(p.flags & MMSUM_GEMM_COLSUM))) {
            epilogue_interior<BM, BN, WAVES_M, WAVES_N, EPI, OUT, WIDE ? CS : 0, LDS_BYTES, LAY, AccSrc>(p, acc, smem, m0, n0, wm, wn, tid, lane, bv, src);
            return;
        }
    }
    epilogue_edge<BM, BN, WAVES_M, WAVES_N, EPI, OUT, LDS_BYTES, LAY, AccSrc>(p, acc, smem, m0, n0, ks, wm, wn, tid, lane, bv, src);
}

__device__ __forceinline__ void dma16(const bf16_t* gsrc, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// NT ring kernel: 4 LDS stages of ONE 32-deep k-slab each; the DMA runs three slabs ahead and is
// retired with counted s_waitcnt vmcnt(N) (never 0 in steady state) + a raw s_barrier, so a slab's
// load latency is covered by three slabs of MFMA work instead of one.
//   iteration s:  wait(own DMA of slab s landed) ; barrier ; issue DMA of slab s+3 into slot (s+3)&3
//                 (that slot was last read in iteration s-1, which every wave finished before this
//                 barrier) ; MFMAs of slab s.
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WAVES_M, int WAVES_N, int NSTAGE_ = 4>
struct RingCfg {
    static constexpr int NW = WAVES_M * WAVES_N;
    static constexpr int THREADS = NW * 64;
    static constexpr int TM = BM / WAVES_M / 32, TN = BN / WAVES_N / 32;
    static constexpr int A_BYTES = BM * SLAB_BYTES, B_BYTES = BN * SLAB_BYTES;
    static constexpr int STAGE = A_BYTES + B_BYTES;
    static constexpr int NSTAGE = NSTAGE_;
    static constexpr int PA = BM / 16, PB = BN / 16;
    static constexpr int PPW = (PA + PB) / NW;                    // DMA instructions per wave per slab
    static_assert(PA % NW == 0 && PB % NW == 0, "pieces must split evenly over the waves");
};

inline int cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v & ~7;
        if (n <= 0) n = 256;
    }
    return n;
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NT ring kernels: v_mfma_f32_16x16x32_bf16 (the chip holds a higher clock on this shape than on 32x32x16 at equal cycles per FLOP)
__device__ __forceinline__ Frag nt_frag(const char* slab, int row0, int lane) { return lds_frag16(slab, row0, lane); }
__device__ __forceinline__ void nt_mma(f32x16_t& acc, const Frag& a, const Frag& b) { mma_slab16(acc, a, b); }

// ---------------------------------------------------------------------------------------------
// 64-deep stages with 128-byte LDS rows (the four-wave kernel below).  A 32-deep stage takes 64 bytes of every operand row,
// HALF a cache line per DMA row: twice the L2 requests the bytes need (TCC_REQ: 62 bytes per request against 120).  Here
// eight consecutive lanes of a DMA instruction fetch one 128-byte row (one request).
//   LDS image: row r at r * 128; 16-byte chunk c (k = 8c..8c+7) of row r at position c ^ ((r >> 1) & 7): a 16x16x32
//   fragment read (lane = row & 15, k group lane >> 4) then spreads each of ds_read_b128's 16-lane groups over all 64 banks.
//   The DMA writes LDS linearly (lane L -> byte 16 L of the piece), so the XOR sits on the SOURCE chunk of the lane.
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WAVES_M, int WAVES_N>
struct K64Cfg {
    static constexpr int NW = WAVES_M * WAVES_N;
    static constexpr int ROWB = 128;
    static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    static constexpr int PA = BM / 8, PB = BN / 8;                 // 1-KiB pieces: 8 rows x 128 B
    static constexpr int PPW = (PA + PB) / NW, PAW = PA / NW, PBW = PB / NW;
    static_assert(PA % NW == 0 && PB % NW == 0, "pieces must split evenly over the waves");
};

// One 1-KiB piece of an operand stage: 8 rows x 128 B through a buffer load whose resource starts at the tile's first row
// (uniform), whose per-lane offset (row, chunk) is fixed for the tile and whose k offset is a scalar: no address arithmetic
// per instruction.  I = index among this wave's pieces of the operand; `stage` = LDS base of the operand's stage.
template <typename C, int I>
__device__ __forceinline__ void k64_piece(char* stage, const bf16_t* base, int voff, int k0, int wave) {
#if defined(__HIP_DEVICE_COMPILE__)        // the host pass of hipcc rejects the buffer-resource builtins inside templates
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(base), 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(stage + (I * C::NW + wave) * 1024), 16, voff, k0 * 2, 0, 0);
#endif
}

// FS ("fused split"): the form for products whose tile list fills a fraction of the CUs (the small-batch step: 72 tiles of 1,152
// decoder rows).  A CU takes in ~35 GB/s whatever the ring depth (tools/gemm_ksweep.py: one 128x128 tile costs 14 us per 1,024 of K, four
// times its MFMA time), so the reduction of a tile is cut into p.fsplit slices on as many CUs; the slices meet like mmsum_dec_gemm's: every
// slice stores its f32 accumulators write-through into its slab of p.split_ws, drains them, takes a ticket; the LAST arriver reads the
// other slabs past the caches, adds the slices IN SLICE ORDER (its own accumulators in their place: the sum does not depend on who
// arrived last) and runs the ordinary epilogue.  The ticket word is left zero for the stream's next product.
template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, int OUT, bool FS = false>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_nt_ring_kernel(GemmArgs p) {
    constexpr int NSTAGE = 4, AHEAD = NSTAGE - 1;   // three slabs of DMA in flight beyond the one being consumed
    using Cfg = RingCfg<BM, BN, WAVES_M, WAVES_N, NSTAGE>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    int m_cap;
    apply_live_rows(p, m_cap);                    // the tile list shrinks with the live row count
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int tiles = tiles_m * tiles_n;
    const int nsp = FS ? p.fsplit : p.splitk;       // reduction slices per tile
    const int total = tiles * nsp;
    void* const C0 = p.C;
    // persistent: a workgroup walks virtual ids blockIdx.x, +gridDim.x, ... (the grid is a multiple of 8, so every id
    // of a workgroup lands on its own XCD's chunk of the tile order)
    for (int vid = blockIdx.x; vid < total; vid += gridDim.x) {
    const int wg = xcd_remap(vid, total);
    const int ks = wg / tiles;
    p.C = (!FS && (p.flags & MMSUM_GEMM_SLABS)) ? static_cast<void*>(static_cast<float*>(C0) + (long)ks * m_cap * p.ldc) : C0;
    const int t = wg % tiles;
    int tm, tn;
    tile_coords(t, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    const int nslab_total = p.K / 32;
    const int per = ((nslab_total + nsp - 1) / nsp + 1) & ~1;       // even number of slabs per split
    const int s_beg = ks * per, s_end = min(nslab_total, s_beg + per);
    const int ns = s_end - s_beg;

    const bf16_t* A = static_cast<const bf16_t*>(p.A);
    const bf16_t* A2 = static_cast<const bf16_t*>(p.A2);
    const bf16_t* B = static_cast<const bf16_t*>(p.B);

    f32x16_t acc[Cfg::TM][Cfg::TN];
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j) acc[i][j] = zero_acc();

    // one slab piece: 16 rows x 64 B; lane -> (row = lane>>2, physical chunk = lane&3).  The row part of every piece's
    // source address is fixed for the tile: computed once, the slab loop only adds the k offset.
    const int prow = lane >> 2;
    long offA[Cfg::PPW], offA2[Cfg::PPW];          // element offsets (B pieces use offA's slots past PA)
#pragma unroll
    for (int i = 0; i < Cfg::PPW; ++i) {
        const bool isA = i * Cfg::NW < Cfg::PA;
        const int rb = isA ? i * Cfg::NW + wave : i * Cfg::NW - Cfg::PA + wave;
        const int row = rb * 16 + prow;
        const int c = (lane & 3) ^ ((row >> 2) & 3);
        int grow = (isA ? m0 : n0) + row;
        const int lim = isA ? p.M : p.N;
        grow = grow < lim ? grow : lim - 1;
        offA[i] = (long)((isA && p.conv_wp) ? conv_row(p, grow) : grow) * (isA ? p.lda : p.ldb) + c * 8;      // implicit convolution: the pixel's padded row
        offA2[i] = isA ? (long)grow * p.lda2 + c * 8 : 0;
    }
    auto issue = [&](int si) {          // si: slab index relative to s_beg
        char* As = smem + (si % NSTAGE) * Cfg::STAGE;
        char* Bs = As + Cfg::A_BYTES;
        int k0 = (s_beg + si) * 32;
        const int kb = k0;
        const bool second = A2 != nullptr && k0 >= p.ksplit;
        const bf16_t* Ab = second ? A2 : A;
        if (second) k0 -= p.ksplit;
        if (p.conv_wp) k0 = conv_koff(p, k0);                 // implicit convolution: (tap row offset) * lda + channel
#pragma unroll
        for (int i = 0; i < Cfg::PPW; ++i) {
            if (i * Cfg::NW < Cfg::PA) dma16(Ab + (second ? offA2[i] : offA[i]) + k0, As + (i * Cfg::NW + wave) * 1024);
            else dma16(B + offA[i] + kb, Bs + (i * Cfg::NW - Cfg::PA + wave) * 1024);
        }
    };

    if (ns > 0) {
        issue(0);
        if (ns > 1) issue(1);
        if (AHEAD > 2 && ns > 2) issue(2);
        auto wait_slab = [&](int si) {
            const int ahead = ns - 1 - si;                       // slabs issued after slab si (capped at AHEAD - 1)
            if (AHEAD > 2 && ahead >= 2) wait_vmcnt<2 * Cfg::PPW>();
            else if (ahead >= 1) wait_vmcnt<Cfg::PPW>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
        };
        for (int si = 0; si < ns; ++si) {
            wait_slab(si);
            const char* As = smem + (si % NSTAGE) * Cfg::STAGE;
            const char* Bs = As + Cfg::A_BYTES;
            // this slab's first fragments are requested BEFORE the DMA of slab si+3 is issued (different ring slots): the
            // address arithmetic and the DMA instructions then run under the LDS latency instead of in front of it
            Frag b[Cfg::TN];
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j) b[j] = nt_frag(Bs, wn * (Cfg::TN * 32) + j * 32, lane);
            Frag a0 = nt_frag(As, wm * (Cfg::TM * 32), lane);
            if (si + AHEAD < ns) issue(si + AHEAD);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i) {
                const Frag a = i == 0 ? a0 : nt_frag(As, wm * (Cfg::TM * 32) + i * 32, lane);
#pragma unroll
                for (int j = 0; j < Cfg::TN; ++j) nt_mma(acc[i][j], a, b[j]);
            }
            __builtin_amdgcn_s_setprio(0);
        }
    }
    bool finish = true;
    if constexpr (FS) {
        constexpr int WORDS = Cfg::TM * Cfg::TN * 16;                 // accumulator words per lane
        float* const slabs = p.split_ws + (long)t * nsp * (BM * BN);
        float* const mine = slabs + (long)ks * (BM * BN) + wave * (WORDS * 64) + lane;
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    __hip_atomic_store(mine + ((i * Cfg::TN + j) * 16 + e) * 64, acc[i][j][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's slab words are in memory ...
        lds_barrier();                                                // ... and so are the other three waves' (and nobody reads the operand stages any more)
        if (tid == 0) *reinterpret_cast<volatile unsigned*>(smem) = __hip_atomic_fetch_add(p.split_tickets + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lds_barrier();
        const unsigned ticket = *reinterpret_cast<volatile unsigned*>(smem);
        finish = ticket == (unsigned)(nsp - 1);
        if (finish) {
            if (tid == 0) __hip_atomic_store(p.split_tickets + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the stream's next product
            const float* const theirs = slabs + wave * (WORDS * 64) + lane;
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i) {
                float o[MMSUM_NT_FSPLIT_MAX][Cfg::TN][16];
#pragma unroll
                for (int sl = 0; sl < MMSUM_NT_FSPLIT_MAX; ++sl)
                    if (sl < nsp && sl != ks) {
#pragma unroll
                        for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
                            for (int e = 0; e < 16; ++e)
                                o[sl][j][e] = __hip_atomic_load(theirs + (long)sl * (BM * BN) + ((i * Cfg::TN + j) * 16 + e) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float v = 0.f;
#pragma unroll
                        for (int sl = 0; sl < MMSUM_NT_FSPLIT_MAX; ++sl)
                            if (sl < nsp) v += sl == ks ? acc[i][j][e] : o[sl][j][e];
                        acc[i][j][e] = v;
                    }
            }
        }
    }
    if (finish) {
        const int eks = FS ? 0 : ks;
        if constexpr (OUT == OUT_F32_ATOMIC) {
            // f32 atomics want 128 contiguous bytes per half-wave instruction: that is the direct accumulator layout
            gemm_epilogue<bf16_t, BM / WAVES_M / 32, BN / WAVES_N / 32, EPI, OUT, LAY_16>(p, acc, m0 + wm * (BM / WAVES_M), n0 + wn * (BN / WAVES_N), eks, lane);
        } else {
            epilogue_staged<BM, BN, WAVES_M, WAVES_N, EPI, OUT, Cfg::NSTAGE * Cfg::STAGE, LAY_16>(p, acc, smem, m0, n0, eks, wm, wn, tid, lane);
        }
    }
    lds_barrier();            // the staging reads are done before the next tile's DMA lands in the same LDS (stores stay in flight)
    }
}


// ---------------------------------------------------------------------------------------------
// Four-wave form of the 256x256 NT tile: 2 x 2 waves, each a 128x128 block (256 accumulator registers = every AGPR, one
// wave per SIMD), 64-deep stages in the 128-byte-row image above.
// Why four waves: with eight, every 64 k of a tile move 64 KB into LDS and 192 KB out of it (each wave reads 128 + 64
// operand rows), and the LDS -- about 64 B/clk for the DMA's writes, 256 for reads -- is then busy ~1800 of the 2048 cycles
// the MFMAs need: the eight-wave loop runs at 1.6 PFLOP/s without its DMA and at 1.1 with a DMA that only hits L2, in
// whatever order the instructions come (32- or 64-deep stages, staggered wave rows, pieces spread between MFMAs).
// 128x128 wave tiles read 128 KB: 1.74 / 1.30 PFLOP/s for the same two measurements.
// One wave per SIMD has nobody to hide behind, so the loop is software-pipelined in the source and the MFMAs are inline asm
// (accumulators pinned in AGPRs; "memory" keeps the reads and DMA written between groups of four where they are):
//   phase 0 of stage s: 64 MFMAs on the first 32 k (set X) | 16 fragment reads of the second 32 k (set Y) | DMA: A of stage s+2
//   wait (own DMA of stage s+1) ; barrier                  -- one per stage, between its halves
//   phase 1:            64 MFMAs on Y | 16 reads of stage s+1's first 32 k (X) | DMA: B of stage s+2
// LDS = 160 KB: THREE stages of A (32 KB each) and TWO of B.  B is the weight matrix (a few MB, L2-resident, shared by every
// tile of a column): one stage ahead is enough.  A streams from HBM: its DMA runs two stages ahead, so at the barrier only
// the 8 youngest pieces (A of stage s+2) may still be in flight -- s_waitcnt vmcnt(8), not 0.
// ---------------------------------------------------------------------------------------------
template <int EPI, int OUT, int CS = 0>
__global__ __launch_bounds__(256, 1) void gemm_nt_w4_kernel(GemmArgs p) {
    constexpr int BM = 256, BN = 256, WAVES_M = 2, WAVES_N = 2, TM = 4, TN = 4;
    using C = K64Cfg<BM, BN, WAVES_M, WAVES_N>;
    static_assert(C::PAW == 8 && C::PBW == 8, "8 + 8 pieces per wave and stage: one behind each group of four MFMAs");
    constexpr int B_BASE = 3 * C::A_BYTES;                        // [A0 | A1 | A2 | B0 | B1]
    // The bf16-output forms issue their MFMAs with the operand roles swapped -- the weight fragment as the MFMA's A operand -- which
    // hands every lane four consecutive COLUMNS of an output row per quarter (LAY_16T) instead of four consecutive rows of a column:
    // the epilogues that read nothing from global memory (plain / GELU / ReLU) then convert on the registers and stage bf16 in
    // 8-byte pieces (epilogue_interior_packed), the accumulating store stages f32 in 16-byte pieces.  Same products, same order of
    // summation.  Columns stay on the lanes where the wider pieces cost registers the epilogue does not have (GELU' / ReLU' with
    // their saved operand: 20 spills; the f32 accumulate with two vectors of C per chunk: 174) and for the f32-atomic form, which
    // stores element-wise from the registers.
    constexpr bool SWAP = OUT == OUT_T_ACC || (OUT == OUT_T && EPI != MMSUM_EPI_GELU_BWD && EPI != MMSUM_EPI_RELU_BWD);
    constexpr int LAYW = SWAP ? LAY_16T : LAY_16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    int m_cap;
    apply_live_rows(p, m_cap);
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int tiles = tiles_m * tiles_n;
    const int total = tiles * p.splitk;
    void* const C0 = p.C;
    const bf16_t* A = static_cast<const bf16_t*>(p.A);
    const bf16_t* A2 = static_cast<const bf16_t*>(p.A2);
    const bf16_t* B = static_cast<const bf16_t*>(p.B);
    // fragment geometry: lane -> row lane & 15 of a 16-row block, k group lane >> 4; the second 32 k of a stage = offset ^ 64
    const int r16 = lane & 15, kg = lane >> 4;
    const int fo = r16 * C::ROWB + (((kg ^ (r16 >> 1)) & 7) << 4);
    const int a_frag0 = wm * (TM * 32) * C::ROWB + fo, b_frag0 = B_BASE + wn * (TN * 32) * C::ROWB + fo;
    const int ydelta = (fo ^ 64) - fo;

    for (int vid = blockIdx.x; vid < total; vid += gridDim.x) {
    const int wg = xcd_remap(vid, total);
    const int ks = wg / tiles;
    p.C = (p.flags & MMSUM_GEMM_SLABS) ? static_cast<void*>(static_cast<float*>(C0) + (long)ks * m_cap * p.ldc) : C0;
    const int t = wg % tiles;
    int tm, tn;
    tile_coords(t, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int nslab_total = p.K / 32;
    const int per = ((nslab_total + p.splitk - 1) / p.splitk + 1) & ~1;
    const int s_beg = ks * per, s_end = min(nslab_total, s_beg + per);
    const int nst = (s_end - s_beg) / 2, k_beg = s_beg * 32;
    if (nst <= 0) continue;      // never: gemm_glds_eligible admits this kernel only when every reduction slice holds a stage

    // 256 accumulator registers = all AGPRs of the wave: the MFMAs are issued from inline asm with "+a" operands so that the
    // accumulators never move (left to the compiler they were copied to VGPRs and back around every MFMA: 600 moves per 128)
    // They are not zeroed per tile (256 v_accvgpr_write, which the compiler moreover emitted once per path: 512 per tile): the
    // first 64 MFMAs of a tile take the constant 0 as their C operand (W4_MMA0).
    f32x4_t c[TM][TN][4];

    // DMA sources: lane -> (row = lane >> 3 of the piece, chunk position lane & 7); rows past the edge re-read the last one
    // (the lane index is re-read opaquely here, as it is for the epilogue below: from the kernel's `lane` the compiler hoists the
    // lane-constant parts of these offsets to kernel entry and, with no register to carry them across the main loop, spills them)
    int lane_t;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_t));
    int offA[C::PAW], offA2[C::PAW], offB[C::PBW];
#pragma unroll
    for (int i = 0; i < C::PAW; ++i) {
        const int row = (i * C::NW + wave) * 8 + (lane_t >> 3);
        const int ch = ((lane_t & 7) ^ ((row >> 1) & 7)) * 16;
        const int ra = m0 + row < p.M ? row : p.M - 1 - m0, rb = n0 + row < p.N ? row : p.N - 1 - n0;
        offA[i] = (p.conv_wp ? conv_row(p, m0 + ra) : ra) * (int)p.lda * 2 + ch;      // implicit convolution: the pixel's padded row (base = the buffer)
        offA2[i] = ra * (int)p.lda2 * 2 + ch;
        offB[i] = rb * (int)p.ldb * 2 + ch;
    }
    const bf16_t* baseA = p.conv_wp ? A : A + (long)m0 * p.lda;
    const bf16_t* baseA2 = A2 ? A2 + (long)m0 * p.lda2 : A;
    const bf16_t* baseB = B + (long)n0 * p.ldb;
    // piece I of A / B of stage `st` (stage index relative to k_beg)
#define W4_A(I, ST)                                                                                    \
    {                                                                                                  \
        const int k0_ = k_beg + (ST) * 64;                                                             \
        const bool second_ = A2 != nullptr && k0_ >= p.ksplit;                                         \
        k64_piece<C, I>(smem + ((ST) % 3) * C::A_BYTES, second_ ? baseA2 : baseA, second_ ? offA2[I] : offA[I],                                  \
                        second_ ? k0_ - p.ksplit : (p.conv_wp ? conv_koff(p, k0_) : k0_), wave);                                                  \
    }
#define W4_B(I, ST) k64_piece<C, I>(smem + B_BASE + ((ST) & 1) * C::B_BYTES, baseB, offB[I], k_beg + (ST) * 64, wave);
#define W4_ALL(M, ST) M(0, ST) M(1, ST) M(2, ST) M(3, ST) M(4, ST) M(5, ST) M(6, ST) M(7, ST)
    {
        Frag aX[TM], bX[TN], aY[TM], bY[TN];
        // prologue, in the order the steady state issues: A(0) B(0) A(1) B(1) A(2)
        W4_ALL(W4_A, 0) W4_ALL(W4_B, 0)
        if (nst > 1) { W4_ALL(W4_A, 1) W4_ALL(W4_B, 1) wait_vmcnt<16>(); } else { wait_vmcnt<0>(); }
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) bX[j].c[h] = *reinterpret_cast<const u32x4_t*>(smem + b_frag0 + (j * 32 + h * 16) * C::ROWB);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) aX[i].c[h] = *reinterpret_cast<const u32x4_t*>(smem + a_frag0 + (i * 32 + h * 16) * C::ROWB);
        // four MFMAs: A block I x B block J (quarters q = 2 si + sj)
#define W4_MMA(FA, FB, I, J)                                                                                          \
        _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                                  \
            if constexpr (SWAP) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %1, %0" : "+a"(c[I][J][q]) : "v"(FA[I].c[q >> 1]), "v"(FB[J].c[q & 1]) : "memory"); \
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c[I][J][q]) : "v"(FA[I].c[q >> 1]), "v"(FB[J].c[q & 1]) : "memory");
        // the same with C = 0: the first MFMA into each accumulator of a tile
#define W4_MMA0(FA, FB, I, J)                                                                                         \
        _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                                  \
            if constexpr (SWAP) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %1, 0" : "=a"(c[I][J][q]) : "v"(FA[I].c[q >> 1]), "v"(FB[J].c[q & 1]) : "memory"); \
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c[I][J][q]) : "v"(FA[I].c[q >> 1]), "v"(FB[J].c[q & 1]) : "memory");
        // LEFT = stages after this one: >= 2 steady state; 1: nothing left to request; 0: the last stage
        auto stage = [&](int st, int abuf, auto left_c, auto first_c) {
            constexpr int LEFT = decltype(left_c)::value;
            constexpr bool FIRST = decltype(first_c)::value;
            const char* As = smem + abuf * C::A_BYTES + a_frag0;
            const char* Bs = smem + (st & 1) * C::B_BYTES + b_frag0;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                if constexpr (FIRST) { W4_MMA0(aX, bX, g >> 2, g & 3) } else { W4_MMA(aX, bX, g >> 2, g & 3) }
                if (g < 8) bY[g >> 1].c[g & 1] = *reinterpret_cast<const u32x4_t*>(Bs + ((g >> 1) * 32 + (g & 1) * 16) * C::ROWB + ydelta);
                else aY[(g - 8) >> 1].c[g & 1] = *reinterpret_cast<const u32x4_t*>(As + (((g - 8) >> 1) * 32 + (g & 1) * 16) * C::ROWB + ydelta);
                // A of stage st+2 into the A stage that stage st-1 left at the previous barrier (one piece behind every other group)
                if constexpr (LEFT >= 2) {
                    switch (g) {
                        case 0: W4_A(0, st + 2) break; case 2: W4_A(1, st + 2) break; case 4: W4_A(2, st + 2) break; case 6: W4_A(3, st + 2) break;
                        case 8: W4_A(4, st + 2) break; case 10: W4_A(5, st + 2) break; case 12: W4_A(6, st + 2) break; case 14: W4_A(7, st + 2) break;
                        default: break;
                    }
                }
            }
            if constexpr (LEFT >= 2) wait_vmcnt<8>(); else if constexpr (LEFT == 1) wait_vmcnt<0>();     // stage st+1 has landed; A of st+2 may be in flight
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int anext = abuf == 2 ? 0 : abuf + 1;
            const char* An = smem + anext * C::A_BYTES + a_frag0;
            const char* Bn = smem + ((st + 1) & 1) * C::B_BYTES + b_frag0;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                W4_MMA(aY, bY, g >> 2, g & 3)
                if constexpr (LEFT >= 1) {
                    if (g < 8) bX[g >> 1].c[g & 1] = *reinterpret_cast<const u32x4_t*>(Bn + ((g >> 1) * 32 + (g & 1) * 16) * C::ROWB);
                    else aX[(g - 8) >> 1].c[g & 1] = *reinterpret_cast<const u32x4_t*>(An + (((g - 8) >> 1) * 32 + (g & 1) * 16) * C::ROWB);
                }
                // B of stage st+2 into the B stage this one leaves
                if constexpr (LEFT >= 2) {
                    switch (g) {
                        case 0: W4_B(0, st + 2) break; case 2: W4_B(1, st + 2) break; case 4: W4_B(2, st + 2) break; case 6: W4_B(3, st + 2) break;
                        case 8: W4_B(4, st + 2) break; case 10: W4_B(5, st + 2) break; case 12: W4_B(6, st + 2) break; case 14: W4_B(7, st + 2) break;
                        default: break;
                    }
                }
            }
        };
        // stage 0 defines the accumulators (C = 0 in its first 64 MFMAs), in whichever of the three forms the tile's length asks for
        if (nst >= 3) stage(0, 0, std::integral_constant<int, 2>{}, std::true_type{});
        else if (nst == 2) stage(0, 0, std::integral_constant<int, 1>{}, std::true_type{});
        else stage(0, 0, std::integral_constant<int, 0>{}, std::true_type{});
        int st = 1, abuf = 1;
        for (; st + 2 < nst; ++st) { stage(st, abuf, std::integral_constant<int, 2>{}, std::false_type{}); abuf = abuf == 2 ? 0 : abuf + 1; }
        if (st + 1 < nst) { stage(st, abuf, std::integral_constant<int, 1>{}, std::false_type{}); abuf = abuf == 2 ? 0 : abuf + 1; ++st; }
        if (st < nst) stage(st, abuf, std::integral_constant<int, 0>{}, std::false_type{});
        asm volatile("s_nop 15\n s_nop 15" ::: "memory");      // MFMAs issued from inline asm: the compiler does not know results are still in the pipeline
#undef W4_MMA0
#undef W4_MMA
    }
#undef W4_ALL
#undef W4_B
#undef W4_A
    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = c[i][j][r >> 2][r & 3];
    // The lane index the epilogue's addresses derive from is re-read here, opaquely: from the kernel's `lane` the compiler hoists
    // every lane-constant part of those addresses to kernel entry and carries them across the main loop, which has no register to
    // spare (it spilled them around -- and, for the variants with more epilogue state, inside -- the MFMA loop: 4 .. 133 VGPRs
    // per variant; none now).
    int lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    if constexpr (OUT == OUT_F32_ATOMIC) {
        gemm_epilogue<bf16_t, TM, TN, EPI, OUT, LAYW>(p, acc, m0 + wm * (BM / WAVES_M), n0 + wn * (BN / WAVES_N), ks, lane_e);
    } else {
        epilogue_staged<BM, BN, WAVES_M, WAVES_N, EPI, OUT, 3 * C::A_BYTES + 2 * C::B_BYTES, LAYW, AccArray, true, CS>(p, acc, smem, m0, n0, ks, wm, wn,
                                                                                                                     wave * 64 + lane_e, lane_e);
    }
    lds_barrier();
    }
}

// ---------------------------------------------------------------------------------------------
// "TN" ring kernel: C[m][n] = sum_k A[k][m] * B[k][n] with BOTH operands reduction-major (A [K,M],
// B [K,N], row-major).  This is the weight-gradient product dW = dy^T x taken straight from the
// activations as the forward/backward passes left them (dy [rows, N_out], x [rows, K_in]): no
// transposed copies.  Same 4-stage DMA ring as above; a stage is 32 reduction rows x BM (and x BN)
// columns kept ROW-major in LDS, and the MFMA operands (8 consecutive k per lane) are gathered by the
// gfx950 transposing LDS read ds_read_b64_tr_b16 (guide T10): a 16-lane group reads a 4(k) x 16(m)
// block and lane i receives column i.  A 32-lane half therefore touches 4 k-rows x 64 B per
// instruction; the 64-byte chunks of a row are XOR-ed with (k & 3) (on the DMA SOURCE address, the
// LDS destination of global_load_lds is linear) so the four rows fall in four different bank groups.
// Rows past K and columns past M/N are fetched from a 16-byte zero page instead of being clamped,
// so any K works (zeros add nothing to the reduction).
// ---------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) uint32_t g_zero_page[4];

#include "gemm_tn_slab.inc"

template <int BM, int BN, int TM, int TN>
__device__ __forceinline__ void tn_slab(f32x16_t (&acc)[TM][TN], const uint32_t (&addrA)[TM], const uint32_t (&addrB)[TN]) {
    if constexpr (BM == 256 && BN == 256) { TN_SLAB_4x2_256_256(acc, addrA, addrB); }
    else if constexpr (BM == 256 && BN == 128) { TN_SLAB_2x2_256_128(acc, addrA, addrB); }
    else { TN_SLAB_2x2_128_128(acc, addrA, addrB); }
}

// per-lane source of one 1-KiB DMA piece of a [32 k][BW] stage (1024/(2 BW) reduction rows per piece)
struct TnPiece { const bf16_t* src; int k; bool col_ok; };
template <int BW>
__device__ __forceinline__ TnPiece tn_piece(const bf16_t* __restrict__ g, long ld, int c0, int C, int k0, int piece, int lane) {
    constexpr int ROWB = BW * 2, CPR = ROWB / 16, RPP = 1024 / ROWB;
    const int r = piece * RPP + lane / CPR;
    const int pc = lane % CPR;
    const int c64 = (pc >> 2) ^ (r & 3);
    const int col = c0 + (c64 * 4 + (pc & 3)) * 8;
    TnPiece t;
    t.k = k0 + r;
    t.col_ok = col < C;
    t.src = g + (long)t.k * ld + (t.col_ok ? col : 0);
    return t;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, int OUT>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_tn_ring_kernel(GemmArgs p) {
    using Cfg = RingCfg<BM, BN, WAVES_M, WAVES_N>;
    constexpr int ROWB_A = BM * 2, ROWB_B = BN * 2;
    static_assert((Cfg::TM == 4 && Cfg::TN == 2 && BM == 256 && BN == 256) || (Cfg::TM == 2 && Cfg::TN == 2), "no slab schedule for this tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    int m_cap;
    apply_live_rows(p, m_cap);                    // reduction-major product: the live count limits K
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles = tiles_m * tiles_n;
    const int ks = wg / tiles;
    if (p.flags & MMSUM_GEMM_SLABS) p.C = static_cast<float*>(p.C) + (long)ks * p.M * p.ldc;
    const int t = wg % tiles;
    int tm, tn;
    tile_coords(t, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    const int nslab_total = (p.K + 31) / 32;
    const int per = (nslab_total + p.splitk - 1) / p.splitk;
    const int s_beg = ks * per, s_end = min(nslab_total, s_beg + per);
    const int ns = s_end - s_beg;

    const bf16_t* A = static_cast<const bf16_t*>(p.A);
    const bf16_t* B = static_cast<const bf16_t*>(p.B);

    f32x16_t acc[Cfg::TM][Cfg::TN];
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j) acc[i][j] = zero_acc();

    // transposing-read lane geometry: group g = lane>>4 -> (16-column half g&1, k half g>>1); lane 4q+p of the
    // group addresses row q, columns 4p..4p+3 of its 4 x 16 block
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    uint32_t offA[Cfg::TM], offB[Cfg::TN];
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i) offA[i] = lds0 + (8 * (g >> 1) + q) * ROWB_A + (((wm * Cfg::TM + i) ^ q) * 64) + 32 * (g & 1) + 8 * pp;
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
        offB[j] = lds0 + Cfg::A_BYTES + (8 * (g >> 1) + q) * ROWB_B + (((wn * Cfg::TN + j) ^ q) * 64) + 32 * (g & 1) + 8 * pp;

    // DMA sources: pointer + reduction row per piece, advanced by 32 rows per issued slab (issue order is sequential)
    TnPiece pc[Cfg::PPW];
#pragma unroll
    for (int i = 0; i < Cfg::PPW; ++i) {
        if (i * Cfg::NW < Cfg::PA) pc[i] = tn_piece<BM>(A, p.lda, m0, p.M, s_beg * 32, i * Cfg::NW + wave, lane);
        else pc[i] = tn_piece<BN>(B, p.ldb, n0, p.N, s_beg * 32, i * Cfg::NW - Cfg::PA + wave, lane);
    }
    const long stepA = 32 * p.lda, stepB = 32 * p.ldb;
    const uintptr_t zero_page = (uintptr_t)g_zero_page;
    auto issue = [&](int si) {
        char* As = smem + (si & 3) * Cfg::STAGE;
        char* Bs = As + Cfg::A_BYTES;
#pragma unroll
        for (int i = 0; i < Cfg::PPW; ++i) {
            const bool ok = pc[i].col_ok && pc[i].k < p.K;
            const uintptr_t src = ok ? (uintptr_t)pc[i].src : zero_page;
            if (i * Cfg::NW < Cfg::PA) {
                dma16(reinterpret_cast<const bf16_t*>(src), As + (i * Cfg::NW + wave) * 1024);
                pc[i].src += stepA;
            } else {
                dma16(reinterpret_cast<const bf16_t*>(src), Bs + (i * Cfg::NW - Cfg::PA + wave) * 1024);
                pc[i].src += stepB;
            }
            pc[i].k += 32;
        }
    };

    if (ns > 0) {
        issue(0);
        if (ns > 1) issue(1);
        if (ns > 2) issue(2);
        for (int si = 0; si < ns; ++si) {
            const int ahead = ns - 1 - si;
            if (ahead >= 2) wait_vmcnt<2 * Cfg::PPW>();
            else if (ahead == 1) wait_vmcnt<Cfg::PPW>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (si + 3 < ns) issue(si + 3);
            const uint32_t stage = (uint32_t)((si & 3) * Cfg::STAGE);
            uint32_t addrA[Cfg::TM], addrB[Cfg::TN];
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i) addrA[i] = offA[i] + stage;
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j) addrB[j] = offB[j] + stage;
            tn_slab<BM, BN, Cfg::TM, Cfg::TN>(acc, addrA, addrB);
        }
        // the MFMAs were issued from inline asm: the compiler does not know results are still in the pipeline
        asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    }
    if constexpr (OUT == OUT_F32_ATOMIC) {
        gemm_epilogue<bf16_t, BM / WAVES_M / 32, BN / WAVES_N / 32, EPI, OUT>(p, acc, m0 + wm * (BM / WAVES_M), n0 + wn * (BN / WAVES_N), ks, lane);
    } else {
        epilogue_staged<BM, BN, WAVES_M, WAVES_N, EPI, OUT>(p, acc, smem, m0, n0, ks, wm, wn, tid, lane);
    }
}

// ---------------------------------------------------------------------------------------------
// Four-wave form of the 256x256 TN tile (see gemm_nt_w4_kernel for why four waves): 2 x 2 waves of 128x128, the same 4-stage
// ring of 32-deep stages and transposing reads as gemm_tn_ring_kernel, 256 accumulator registers pinned in AGPRs, and the
// loop software-pipelined in the source: the 32 MFMAs of a stage run on fragments read during the previous stage, while the
// next stage's 32 transposing reads (one pair behind every other MFMA) and the 8 DMA pieces of the stage three ahead (one
// behind every fourth) are issued between them.  One barrier per stage; at it only the youngest 8 pieces may be in flight.
// ---------------------------------------------------------------------------------------------
#include "gemm_tn_w4_acc.inc"
typedef __attribute__((ext_vector_type(4))) short tn_s16x4_t;
typedef __attribute__((ext_vector_type(8))) short tn_s16x8_t;
typedef __attribute__((address_space(3))) tn_s16x4_t* tn_lds_s16x4_ptr;

// BS (column sums of A, i.e. the bias gradient of the Linear whose weight gradient this product is: MMSUM_GEMM_COLSUM): the waves
// of the first tile column (tn == 0) add up the A fragments they hold for the MFMAs anyway -- four v_dot2c_f32_bf16 per fragment
// against (1, 1); the two waves of a wave row hold the same A blocks and take two each: 16 per wave and stage, in the shadow of
// the stage's 32 MFMAs -- and leave bias[m] += sum_k A[k][m] with one f32 atomic per column and slice.  The separate column-sum passes over dq / dk / dv (983 MB per decoder layer) are gone.
template <int OUT, bool BS = false>
__global__ __launch_bounds__(256, 1) void gemm_tn_w4_kernel(GemmArgs p) {
    constexpr int BM = 256, BN = 256, WAVES_M = 2, WAVES_N = 2, TM = 4, TN = 4;
    using Cfg = RingCfg<BM, BN, WAVES_M, WAVES_N>;
    constexpr int ROWB = BM * 2;                                   // 512-byte LDS rows (one reduction row of 256 columns) for A and B
    static_assert(Cfg::PPW == 8 && Cfg::PA == 4 * Cfg::NW, "8 pieces per wave and stage: 4 of A, 4 of B");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    int m_cap;
    apply_live_rows(p, m_cap);                    // reduction-major product: the live count limits K
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles = tiles_m * tiles_n;
    const int ks = wg / tiles;
    if (p.flags & MMSUM_GEMM_SLABS) p.C = static_cast<float*>(p.C) + (long)ks * p.M * p.ldc;
    const int t = wg % tiles;
    int tm, tn;
    tile_coords(t, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int nslab_total = (p.K + 31) / 32;
    const int per = (nslab_total + p.splitk - 1) / p.splitk;
    const int s_beg = ks * per, s_end = min(nslab_total, s_beg + per);
    const int ns = s_end - s_beg;
    const bf16_t* A = static_cast<const bf16_t*>(p.A);
    const bf16_t* B = static_cast<const bf16_t*>(p.B);
    // weight gradient of the implicit 3x3 convolution (mmsum_conv3x3_wgrad): B is the padded activation image [rows, C]; the tile's 256
    // output columns are channels c0.. of ONE tap, i.e. the same rows shifted by the tap's offset in the padded image
    long b_off = n0;
    int b_lim = p.N - n0;
    if (p.conv_wp) {
        const int tap = n0 >> p.conv_cshift, c0 = n0 & ((1 << p.conv_cshift) - 1);
        const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
        b_off = (long)((ky - 1) * p.conv_wp + (kx - 1)) * p.ldb + c0;
        b_lim = (1 << p.conv_cshift) - c0;
    }

    const bool do_bsum = BS && tn == 0;                            // workgroup-uniform; wave (wm, wn) sums A blocks 2 wn, 2 wn + 1 of its row
    float bsum[2] = {0.f, 0.f};
    if constexpr (BS) p.flags &= ~MMSUM_GEMM_COLSUM;                 // here the flag means sums of A, not of the stored tile: the epilogue must not see it
    // accumulator (i, j) = a[16 (4 i + j) : +15]: all 256 AGPRs, addressed by name (gemm_tn_w4_acc.inc).  The compiler does not know
    // they are in use: nothing else in this kernel may need an AGPR (register pressure stays below 256 VGPRs; the epilogue takes
    // the accumulators one row band -- 64 registers -- at a time).
    {
        const bf16x8_t zero = __builtin_bit_cast(bf16x8_t, u32x4_t{0u, 0u, 0u, 0u});
        TNW4_ZERO(zero)
    }

    // transposing-read lane geometry (as in gemm_tn_ring_kernel): group g = lane >> 4 -> (16-column half g & 1, k half g >> 1);
    // lane 4q+p of the group addresses row q, columns 4p..4p+3 of its 4 x 16 block.  Byte offsets inside a stage.
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    int offA[TM], offB[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) offA[i] = (8 * (g >> 1) + q) * ROWB + (((wm * TM + i) ^ q) * 64) + 32 * (g & 1) + 8 * pp;
#pragma unroll
    for (int j = 0; j < TN; ++j) offB[j] = Cfg::A_BYTES + (8 * (g >> 1) + q) * ROWB + (((wn * TN + j) ^ q) * 64) + 32 * (g & 1) + 8 * pp;
    // operand of 16-deep step s16 of a 32-column block: rows 16 s16 + {0..3, 4..7} (+8 for the upper half-wave)
    auto tr_frag = [&](const char* stage, int off, int s16) {
        const char* ptr = stage + off + s16 * (16 * ROWB);
        const tn_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tn_lds_s16x4_ptr)(ptr));
        const tn_s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tn_lds_s16x4_ptr)(ptr + 4 * ROWB));
        return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    // DMA sources: buffer loads.  The resource of a stage starts at the stage's first reduction row and the tile's first column
    // and ends with the matrix (rows past K read as zeros in hardware); a lane's (row in the stage, column) is a 32-bit offset
    // fixed for the tile, out of range for columns past M / N.
    int voff[Cfg::PPW];
#pragma unroll
    for (int i = 0; i < Cfg::PPW; ++i) {
        const bool isA = i * Cfg::NW < Cfg::PA;
        const int piece = isA ? i * Cfg::NW + wave : i * Cfg::NW - Cfg::PA + wave;
        constexpr int CPR = ROWB / 16, RPP = 1024 / ROWB;
        const int r = piece * RPP + lane / CPR, pcn = lane % CPR;
        const int c64 = (pcn >> 2) ^ (r & 3);
        const int col = (c64 * 4 + (pcn & 3)) * 8;                  // inside the tile
        const int lim = isA ? p.M - m0 : b_lim;
        voff[i] = col < lim ? (r * (int)(isA ? p.lda : p.ldb) + col) * 2 : (int)0x80000000u;
    }
    // The DMA is issued from inline asm: to the compiler a buffer_load ... lds is a store to LDS that the transposing reads
    // behind it might alias, and it answers with s_waitcnt vmcnt(0) after every piece (measured: the loop at a quarter of its
    // speed).  Its ordering is by hand anyway: counted vmcnt waits and the stage barrier.
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    auto issue_piece = [&](auto ic, int si) {
        constexpr int i = decltype(ic)::value;
        constexpr bool isA = i * Cfg::NW < Cfg::PA;
        const uint32_t dst = lds0 + (si & 3) * Cfg::STAGE + (isA ? (i * Cfg::NW + wave) * 1024 : Cfg::A_BYTES + (i * Cfg::NW - Cfg::PA + wave) * 1024);
        const long k0 = (long)(s_beg + si) * 32;
        const long ld = isA ? p.lda : p.ldb;
        const uintptr_t base = (uintptr_t)((isA ? A + m0 : B + b_off) + k0 * ld);
        const long rows_left = (long)p.K - k0;                        // > 0: only existing stages are requested
        const long bytes = ((rows_left - 1) * ld + (((isA ? p.M - m0 : b_lim) + 7) & ~7)) * 2;      // whole 16-byte chunks of the last row (the pitch covers them)
        u32x4_t rsrc;                                                 // raw buffer resource: base, stride 0, num_records, dword 3 as make_buffer_rsrc's
        rsrc[0] = (uint32_t)base;
        rsrc[1] = (uint32_t)(base >> 32) & 0xffffu;
        rsrc[2] = bytes > 0x7fffffffL ? 0x7fffffffu : (uint32_t)bytes;
        rsrc[3] = 0x00020000u;
        unsigned keep;
        const int vo = voff[i];                                       // (an asm operand cannot name a captured array element)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(vo), "s"(dst), "s"(rsrc) : "memory");
    };
    // Reduction rows past K in the last stage: a lane of an LDS-DMA whose source lies outside the buffer may leave its LDS bytes as
    // they were, so those rows of the stage are zeroed by hand when the stage is requested (the slot is free by then, and the
    // DMA writes other rows).  Zeros add nothing to the reduction: any K works.
    const int k_tail = p.K & 31;                                   // rows of the last stage that exist (0 = all 32)
    auto zero_tail = [&](int si) {
        if (k_tail != 0 && s_beg + si == nslab_total - 1) {
            char* slot = smem + (si & 3) * Cfg::STAGE;
            for (int o = k_tail * ROWB + tid * 16; o < 32 * ROWB; o += 256 * 16) {
                *reinterpret_cast<u32x4_t*>(slot + o) = u32x4_t{0u, 0u, 0u, 0u};
                *reinterpret_cast<u32x4_t*>(slot + Cfg::A_BYTES + o) = u32x4_t{0u, 0u, 0u, 0u};
            }
        }
    };
#define TNW4_ISSUE_ALL(SI) { zero_tail(SI); issue_piece(std::integral_constant<int, 0>{}, SI); issue_piece(std::integral_constant<int, 1>{}, SI); \
                             issue_piece(std::integral_constant<int, 2>{}, SI); issue_piece(std::integral_constant<int, 3>{}, SI); \
                             issue_piece(std::integral_constant<int, 4>{}, SI); issue_piece(std::integral_constant<int, 5>{}, SI); \
                             issue_piece(std::integral_constant<int, 6>{}, SI); issue_piece(std::integral_constant<int, 7>{}, SI); }

    if (ns > 0) {
        bf16x8_t aX[TM][2], bX[TN][2], aY[TM][2], bY[TN][2];
        TNW4_ISSUE_ALL(0)
        if (ns > 1) TNW4_ISSUE_ALL(1)
        if (ns > 2) TNW4_ISSUE_ALL(2)
        if (ns > 2) wait_vmcnt<16>(); else if (ns > 1) wait_vmcnt<8>(); else wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // zero_tail's stores
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int s16 = 0; s16 < 2; ++s16) { aX[i][s16] = tr_frag(smem, offA[i], s16); bX[i][s16] = tr_frag(smem, offB[i], s16); }
        // LEFT = stages after this one (capped at 3): >= 3 request stage si+3; >= 2: stage si+2 is in flight at the barrier
        auto stage = [&](int si, bf16x8_t (&aC)[TM][2], bf16x8_t (&bC)[TN][2], bf16x8_t (&aN)[TM][2], bf16x8_t (&bN)[TN][2], auto left_c) {
            constexpr int LEFT = decltype(left_c)::value;
            if constexpr (LEFT >= 2) wait_vmcnt<8>(); else if constexpr (LEFT == 1) wait_vmcnt<0>();     // stage si+1 has landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const char* nxt = smem + ((si + 1) & 3) * Cfg::STAGE;
#pragma unroll
            for (int m = 0; m < 32; ++m) {
                const int s16 = m >> 4, i = (m >> 2) & 3, j = m & 3;
                TNW4_MFMA(4 * i + j, aC[i][s16], bC[j][s16])
                if constexpr (LEFT >= 1) {
                    if ((m & 1) == 0) {
                        const int f = m >> 1;                           // 0..7: A fragments (block f >> 1, step f & 1); 8..15: B
                        if (f < 8) aN[f >> 1][f & 1] = tr_frag(nxt, offA[f >> 1], f & 1);
                        else bN[(f - 8) >> 1][f & 1] = tr_frag(nxt, offB[(f - 8) >> 1], f & 1);
                    }
                }
                if constexpr (LEFT >= 3) {
                    switch (m) {
                        case 1: issue_piece(std::integral_constant<int, 0>{}, si + 3); break;
                        case 5: issue_piece(std::integral_constant<int, 1>{}, si + 3); break;
                        case 9: issue_piece(std::integral_constant<int, 2>{}, si + 3); break;
                        case 13: issue_piece(std::integral_constant<int, 3>{}, si + 3); break;
                        case 17: issue_piece(std::integral_constant<int, 4>{}, si + 3); break;
                        case 21: issue_piece(std::integral_constant<int, 5>{}, si + 3); break;
                        case 25: issue_piece(std::integral_constant<int, 6>{}, si + 3); break;
                        case 29: issue_piece(std::integral_constant<int, 7>{}, si + 3); break;
                        default: break;
                    }
                }
            }
            if constexpr (LEFT >= 3) zero_tail(si + 3);
            if constexpr (BS) {
                if (do_bsum) {
                    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
                    const bf16x2_t ones = {(bf16_t)1.f, (bf16_t)1.f};
                    auto add8 = [&](float t, const bf16x8_t f) {
                        t = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 0, 1), ones, t, false);
                        t = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 2, 3), ones, t, false);
                        t = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 4, 5), ones, t, false);
                        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 6, 7), ones, t, false);
                    };
#pragma unroll
                    for (int i = 0; i < TM; ++i)    // scalar branches (selecting the fragments per register would cost as much as the sums)
                        if ((i >> 1) == wn) bsum[i & 1] = add8(add8(bsum[i & 1], aC[i][0]), aC[i][1]);
                }
            }
        };
#define TNW4_XY(SI, L) stage(SI, aX, bX, aY, bY, std::integral_constant<int, L>{});
#define TNW4_YX(SI, L) stage(SI, aY, bY, aX, bX, std::integral_constant<int, L>{});
        int si = 0;
        for (; si + 4 < ns; si += 2) { TNW4_XY(si, 3) TNW4_YX(si + 1, 3) }
        switch (ns - si) {                                             // 1..4 stages left, set X current
            case 4: TNW4_XY(si, 3) TNW4_YX(si + 1, 2) TNW4_XY(si + 2, 1) TNW4_YX(si + 3, 0) break;
            case 3: TNW4_XY(si, 2) TNW4_YX(si + 1, 1) TNW4_XY(si + 2, 0) break;
            case 2: TNW4_XY(si, 1) TNW4_YX(si + 1, 0) break;
            default: TNW4_XY(si, 0) break;
        }
#undef TNW4_XY
#undef TNW4_YX
        asm volatile("s_nop 15\n s_nop 15" ::: "memory");      // MFMAs issued from inline asm: results may still be in the pipeline
    }
#undef TNW4_ISSUE_ALL
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");          // MFMAs issued from inline asm: results may still be in the pipeline
    if constexpr (BS) {
        if (do_bsum) {      // lanes l and l + 32 hold the two k halves of column l & 31 of each A block
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const float t = wave_half_sum(bsum[ii]);
                const int col = m0 + (wm * TM + 2 * wn + ii) * 32 + (lane & 31);
                if (lane < 32 && col < p.M) atomicAdd(const_cast<float*>(p.bias) + col, t * p.alpha);
            }
        }
    }
    struct NamedAcc {
        __device__ __forceinline__ void operator()(int i, const f32x16_t (*)[TN], f32x16_t (&band)[TN]) const {
#pragma unroll
            for (int j = 0; j < TN; ++j) { TNW4_READ(4 * i + j, band[j]) }
        }
    };
    static_assert(OUT != OUT_F32_ATOMIC, "the atomic epilogue wants the whole accumulator array");
    f32x16_t unused[TM][TN];
    epilogue_staged<BM, BN, WAVES_M, WAVES_N, MMSUM_EPI_NONE, OUT, 4 * (BM + BN) * SLAB_BYTES, false, NamedAcc>(p, unused, smem, m0, n0, ks, wm, wn, tid, lane, NamedAcc{});
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int OUT>
int launch_tn_one(const GemmArgs& a, hipStream_t stream) {
    using R = RingCfg<BM, BN, WAVES_M, WAVES_N>;
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    const size_t lds = R::NSTAGE * R::STAGE;
    // function-local static: initialised once, thread-safe (C++11), for the one device this process drives
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_ring_kernel<BM, BN, WAVES_M, WAVES_N, MMSUM_EPI_NONE, OUT>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return MMSUM_ERR_HIP;
    gemm_tn_ring_kernel<BM, BN, WAVES_M, WAVES_N, MMSUM_EPI_NONE, OUT><<<dim3(tiles * a.splitk), dim3(R::THREADS), lds, stream>>>(a);
    return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
int launch_tn_cfg(const GemmArgs& a, hipStream_t stream) {
    switch (out_mode_of(a)) {
        case OUT_T: return launch_tn_one<BM, BN, WAVES_M, WAVES_N, OUT_T>(a, stream);
        case OUT_F32_ACC: return launch_tn_one<BM, BN, WAVES_M, WAVES_N, OUT_F32_ACC>(a, stream);
        case OUT_F32_ATOMIC: return launch_tn_one<BM, BN, WAVES_M, WAVES_N, OUT_F32_ATOMIC>(a, stream);
        case OUT_F32: return launch_tn_one<BM, BN, WAVES_M, WAVES_N, OUT_F32>(a, stream);
        default: return MMSUM_ERR_BAD_SHAPE;
    }
}

// persistent workgroups: at most one per CU walks the tile list (measured +0..8 % on multi-round shapes: no relaunch
// between a CU's tiles)
inline int ring_grid(const GemmArgs& a, int bm, int bn) {
    const int tiles = ((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn) * a.splitk;
    const int cus = cu_count();
    return tiles > cus ? cus : tiles;
}

inline bool nt_one_round(const GemmArgs& a) { return (long)((a.M + 127) / 128) * ((a.N + 127) / 128) * a.splitk <= cu_count(); }

// Reduction slices per tile of the fused-split form (1: not taken).  Taken when the caller lent a workspace, the 128x128 tile list
// fills at most half the CUs and every slice keeps >= 768 of K (the slabs of a tile's slices go through memory: 64 KB each way per
// slice; at K = 1,024 that costs more than the shorter loop saves: 16.5 against 13.7 us); the slab words of a tile's slices + the
// ticket words must fit.
inline int nt_fused_split(const GemmArgs& a) {
    if (a.split_ws == nullptr || a.splitk != 1 || a.conv_wp != 0 || out_mode_of(a) == OUT_F32_ATOMIC) return 1;
    const long tiles = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    int s = (int)std::min<long>(MMSUM_NT_FSPLIT_MAX, cu_count() / std::max<long>(tiles, 1));
    const int ns = a.K / 32;
    for (; s >= 2; --s) {
        const int per = ((ns + s - 1) / s + 1) & ~1;
        if (per >= 24 && (s - 1) * per < ns && MMSUM_NT_FSPLIT_TICKET_BYTES + tiles * s * (128L * 128 * 4) <= a.split_ws_bytes) return s;
    }
    return 1;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, int OUT>
int launch_one(const GemmArgs& a, hipStream_t stream) {
    using R = RingCfg<BM, BN, WAVES_M, WAVES_N>;
    const size_t lds = R::NSTAGE * R::STAGE;
    if constexpr (BM == 128 && BN == 128 && WAVES_M == 4 && OUT != OUT_F32_ATOMIC) {       // the one-round form (eight waves): the only one a split tile list reaches
        const int fs = nt_fused_split(a);
        if (fs > 1) {
            static const hipError_t attr_fs = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_ring_kernel<BM, BN, WAVES_M, WAVES_N, EPI, OUT, true>),
                                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (attr_fs != hipSuccess) return MMSUM_ERR_HIP;
            GemmArgs b = a;
            b.fsplit = fs;
            b.split_tickets = reinterpret_cast<unsigned*>(a.split_ws);
            b.split_ws = reinterpret_cast<float*>(reinterpret_cast<char*>(a.split_ws) + MMSUM_NT_FSPLIT_TICKET_BYTES);
            const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
            gemm_nt_ring_kernel<BM, BN, WAVES_M, WAVES_N, EPI, OUT, true><<<dim3(tiles * fs), dim3(R::THREADS), lds, stream>>>(b);
            return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
        }
    }
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_ring_kernel<BM, BN, WAVES_M, WAVES_N, EPI, OUT>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return MMSUM_ERR_HIP;
    gemm_nt_ring_kernel<BM, BN, WAVES_M, WAVES_N, EPI, OUT><<<dim3(ring_grid(a, BM, BN)), dim3(R::THREADS), lds, stream>>>(a);
    return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
int launch_cfg(const GemmArgs& a, hipStream_t stream) {
    const int epi = (a.flags >> 3) & 7, out = out_mode_of(a);
#define FAST_CASE(E, O) if (epi == E && out == O) return launch_one<BM, BN, WAVES_M, WAVES_N, E, O>(a, stream);
    FAST_CASE(MMSUM_EPI_NONE, OUT_T) FAST_CASE(MMSUM_EPI_NONE, OUT_T_ACC) FAST_CASE(MMSUM_EPI_NONE, OUT_F32_ACC)
    FAST_CASE(MMSUM_EPI_NONE, OUT_F32_ATOMIC) FAST_CASE(MMSUM_EPI_NONE, OUT_F32)
    FAST_CASE(MMSUM_EPI_GELU, OUT_T) FAST_CASE(MMSUM_EPI_GELU_BWD, OUT_T) FAST_CASE(MMSUM_EPI_RELU, OUT_T) FAST_CASE(MMSUM_EPI_RELU_BWD, OUT_T)
#undef FAST_CASE
    return MMSUM_ERR_BAD_SHAPE;
}

template <int EPI, int OUT, int CS = 0>
int launch_w4_one(const GemmArgs& a, hipStream_t stream) {
    using C = K64Cfg<256, 256, 2, 2>;
    const size_t lds = 3 * C::A_BYTES + 2 * C::B_BYTES;             // 160 KB: the whole LDS of a CU
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_w4_kernel<EPI, OUT, CS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return MMSUM_ERR_HIP;
    gemm_nt_w4_kernel<EPI, OUT, CS><<<dim3(ring_grid(a, 256, 256)), dim3(256), lds, stream>>>(a);
    return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
}
int launch_w4(const GemmArgs& a, hipStream_t stream) {
    const int epi = (a.flags >> 3) & 7, out = out_mode_of(a);
    if (a.flags & MMSUM_GEMM_COLSUM) {        // column sums of the stored bf16 tile in the epilogue (bias gradients)
        const bool sq = (a.flags & MMSUM_GEMM_COLSUM2) != 0;       // + sums of squares: the BatchNorm statistics of a convolution's output
        if (epi == MMSUM_EPI_GELU_BWD && out == OUT_T && !sq) return launch_w4_one<MMSUM_EPI_GELU_BWD, OUT_T, 1>(a, stream);
        if (epi == MMSUM_EPI_NONE && out == OUT_T) return sq ? launch_w4_one<MMSUM_EPI_NONE, OUT_T, 2>(a, stream) : launch_w4_one<MMSUM_EPI_NONE, OUT_T, 1>(a, stream);
        return MMSUM_ERR_BAD_SHAPE;           // gemm_glds_eligible admits column sums for these two forms only
    }
#define W4_CASE(E, O) if (epi == E && out == O) return launch_w4_one<E, O>(a, stream);
    W4_CASE(MMSUM_EPI_NONE, OUT_T) W4_CASE(MMSUM_EPI_NONE, OUT_T_ACC) W4_CASE(MMSUM_EPI_NONE, OUT_F32_ACC)
    W4_CASE(MMSUM_EPI_NONE, OUT_F32_ATOMIC) W4_CASE(MMSUM_EPI_NONE, OUT_F32)
    W4_CASE(MMSUM_EPI_GELU, OUT_T) W4_CASE(MMSUM_EPI_RELU, OUT_T) W4_CASE(MMSUM_EPI_GELU_BWD, OUT_T) W4_CASE(MMSUM_EPI_RELU_BWD, OUT_T)
#undef W4_CASE
    return MMSUM_ERR_BAD_SHAPE;               // gemm_glds_eligible admits no other combination
}

// Tile shape that keeps the 256 CUs busiest for a problem: fraction of the last round of workgroups that is filled x
// fraction of the padded tile area that is real x relative main-loop efficiency of the shape (operand bytes per FLOP halve
// from 128^2 to 256^2 and the L2 -> LDS DMA rate, not the MFMA rate, bounds the small tiles: measured 0.50 and 0.72).
inline double tile_score(int M, int N, int splitk, int bm, int bn, double eff) {
    const long tiles = (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn) * splitk;
    const long rounds = (tiles + 255) / 256;
    const double fill = (double)tiles / (double)(rounds * 256);
    const double waste = ((double)M * N) / ((double)((M + bm - 1) / bm * bm) * ((N + bn - 1) / bn * bn));
    return fill * waste * eff;
}
enum { TILE_256x256 = 0, TILE_256x128 = 1, TILE_128x128 = 2 };
inline int choose_tile(const GemmArgs& a) {
    const double s256 = tile_score(a.M, a.N, a.splitk, 256, 256, 1.00);
    const double s128 = tile_score(a.M, a.N, a.splitk, 128, 128, 0.50);
    const double s2x1 = tile_score(a.M, a.N, a.splitk, 256, 128, 0.72);
    if (s256 >= s128 && s256 >= s2x1) return TILE_256x256;
    return s2x1 >= s128 ? TILE_256x128 : TILE_128x128;
}

}  // namespace

// The staged epilogue reads its global operands (saved pre-activation of GELU' / ReLU', C of an accumulating store) as
// 16-byte vectors only -- a scalar fallback inside the kernel makes hipcc wait vmcnt(0) at every join, i.e. for every store
// issued before it.  Products whose operands do not allow that (ragged N, unaligned pitch) take the generic kernel.
static bool epilogue_reads_vectorisable(const GemmArgs& a) {
    const int epi = (a.flags >> 3) & 7, out = out_mode_of(a);
    const bool aux_in = epi == MMSUM_EPI_GELU_BWD || epi == MMSUM_EPI_RELU_BWD;
    if (!aux_in && out != OUT_T_ACC && out != OUT_F32_ACC) return true;           // nothing is read
    if (a.N % 8) return false;
    if (aux_in && (a.aux == nullptr || (((uintptr_t)a.aux) & 15) || (a.ldaux & 7))) return false;
    if (out == OUT_T_ACC && ((((uintptr_t)a.C) & 15) || (a.ldc & 7))) return false;
    if (out == OUT_F32_ACC && ((((uintptr_t)a.C) & 15) || (a.ldc & 3))) return false;
    return true;
}

bool gemm_glds_eligible(int dtype, const GemmArgs& a) {
    if (dtype != MMSUM_BF16) return false;
    if (a.flags & (MMSUM_GEMM_A_T | MMSUM_GEMM_B_T)) return false;
    if (a.K < 64 || a.K % 64) return false;
    if (a.A2 && (a.ksplit % 64)) return false;
    if (a.splitk > 1) {           // every reduction slice holds at least one 64-deep stage (the four-wave kernel's first stage defines its accumulators)
        const int ns = a.K / 32, per = ((ns + a.splitk - 1) / a.splitk + 1) & ~1;
        if ((a.splitk - 1) * per >= ns) return false;
    }
    const int epi = (a.flags >> 3) & 7, out = out_mode_of(a);
    if (epi != MMSUM_EPI_NONE && out != OUT_T) return false;     // rare combinations stay on the generic kernel
    return epilogue_reads_vectorisable(a);
}

// Column sums of A in the weight-gradient product (MMSUM_GEMM_COLSUM with A_T | B_T): only the four-wave 256x256 kernel with
// split-K slabs carries them
bool gemm_tn_colsum_ok(int dtype, const GemmArgs& a) {
    return gemm_tn_eligible(dtype, a) && choose_tile(a) == TILE_256x256 && out_mode_of(a) == OUT_F32 && a.bias != nullptr;
}

// A [K,M] and B [K,N] reduction-major bf16 (flags A_T | B_T): the weight-gradient layout
bool gemm_tn_eligible(int dtype, const GemmArgs& a) {
    if (dtype != MMSUM_BF16) return false;
    if ((a.flags & (MMSUM_GEMM_A_T | MMSUM_GEMM_B_T)) != (MMSUM_GEMM_A_T | MMSUM_GEMM_B_T)) return false;
    if (a.A2 != nullptr || (a.flags & MMSUM_GEMM_BIAS) || ((a.flags >> 3) & 7) != MMSUM_EPI_NONE || a.aux != nullptr) return false;
    // 16-byte column chunks: a ragged last chunk must still lie inside the row (leading dimension padded)
    if ((a.lda & 7) || (a.ldb & 7) || a.lda < ((a.M + 7) & ~7) || a.ldb < ((a.N + 7) & ~7)) return false;
    if ((((uintptr_t)a.A) | ((uintptr_t)a.B)) & 15) return false;
    return epilogue_reads_vectorisable(a);
}

template <int OUT, bool BS = false>
int launch_tn_w4_one(const GemmArgs& a, hipStream_t stream) {
    using R = RingCfg<256, 256, 2, 2>;
    const int tiles = ((a.M + 255) / 256) * ((a.N + 255) / 256);
    const size_t lds = R::NSTAGE * R::STAGE;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_w4_kernel<OUT, BS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return MMSUM_ERR_HIP;
    gemm_tn_w4_kernel<OUT, BS><<<dim3(tiles * a.splitk), dim3(256), lds, stream>>>(a);
    return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
}
int launch_tn_w4(const GemmArgs& a, hipStream_t stream) {
    if (a.flags & MMSUM_GEMM_COLSUM) return out_mode_of(a) == OUT_F32 ? launch_tn_w4_one<OUT_F32, true>(a, stream) : MMSUM_ERR_BAD_SHAPE;
    switch (out_mode_of(a)) {
        case OUT_T: return launch_tn_w4_one<OUT_T>(a, stream);
        case OUT_F32_ACC: return launch_tn_w4_one<OUT_F32_ACC>(a, stream);
        case OUT_F32: return launch_tn_w4_one<OUT_F32>(a, stream);
        default: return launch_tn_cfg<256, 256, 2, 4>(a, stream);       // f32 atomics: the eight-wave kernel
    }
}

int launch_gemm_tn(const GemmArgs& a, hipStream_t stream) {
    switch (choose_tile(a)) {
        case TILE_256x256: return launch_tn_w4(a, stream);
        case TILE_256x128: return launch_tn_cfg<256, 128, 4, 2>(a, stream);
        default: return launch_tn_cfg<128, 128, 2, 2>(a, stream);
    }
}

int launch_gemm_tn_w4(const GemmArgs& a, hipStream_t stream) { return launch_tn_w4(a, stream); }      // mmsum_conv3x3_wgrad: the four-wave kernel whatever the tile rule says

int launch_gemm_glds(const GemmArgs& a, hipStream_t stream) {
    switch (choose_tile(a)) {
        case TILE_256x256: return launch_w4(a, stream);
        case TILE_256x128: return launch_cfg<256, 128, 4, 2>(a, stream);
        default:
            // a tile list of one round of the CUs or less: nobody shares the CU, so EIGHT waves (64x32 each) take the tile -- two per SIMD
            // cover each other's DMA issue and LDS waits the way two co-resident four-wave workgroups do on the large products
            // (tools/gemm_small_bench.py, 1,152 x 1,024 x 1,024: 17.8 -> 13.7 us)
            if (nt_one_round(a)) return launch_cfg<128, 128, 4, 2>(a, stream);
            return launch_cfg<128, 128, 2, 2>(a, stream);
    }
}

static const int kTileDims[3][2] = {{256, 256}, {256, 128}, {128, 128}};
GemmPlan plan_gemm_glds(const GemmArgs& a) {
    const int tile = choose_tile(a);
    const int* d = kTileDims[tile];
    const int fs = tile == TILE_128x128 ? nt_fused_split(a) : 1;
    if (fs > 1) return GemmPlan{MMSUM_PLAN_NT_RING, d[0], d[1], ((a.M + 127) / 128) * ((a.N + 127) / 128) * fs};
    return GemmPlan{MMSUM_PLAN_NT_RING, d[0], d[1], ring_grid(a, d[0], d[1])};
}
GemmPlan plan_gemm_tn(const GemmArgs& a) {
    const int* d = kTileDims[choose_tile(a)];
    return GemmPlan{MMSUM_PLAN_TN_RING, d[0], d[1], ((a.M + d[0] - 1) / d[0]) * ((a.N + d[1] - 1) / d[1]) * a.splitk};
}
