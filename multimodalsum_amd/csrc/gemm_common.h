// Shared by the generic and the LDS-DMA GEMM kernels: argument block and the fused epilogue.
#pragma once
#include "mmsum_device.h"
#include "mmsum_kernels.h"

struct GemmArgs {
    const void* A; const void* A2; const void* B; void* C; const float* bias; void* aux;
    int M, N, K; long lda, lda2, ldb, ldc, ldaux; int ksplit; float alpha; int flags; int splitk;
    const int* live;     // device int32 or NULL: live rows of the row-streamed operand (M for natural A, K for the A_T|B_T product)
    const float* alpha_dev;   // device f32 or NULL: alpha is multiplied by it when the kernel runs (an upstream gradient scale)
    // Implicit 3x3 convolution (stride 1, padding 1) as the NT product y[(n,y,x), co] = sum_{tap, c} A[pixel + tap][c] W[co][tap * C + c]
    // (mmsum_conv3x3_gemm; conv_wp == 0: an ordinary product).  A = the activations in the PADDED NHWC layout [(n, H + 2, W + 2), C]
    // with zero borders: the operand row of output pixel m for tap (ky, kx) is row conv_row(m) + ky * Wp + kx -- the im2col matrix is
    // never materialised: the LDS-DMA pieces of a 64-deep (32-deep) stage read C-contiguous runs of one tap.  C is a power of two >= 64.
    int conv_wp, conv_w, conv_hw, conv_hpwp, conv_cshift;
    // Fused split of small NT products (gemm_nt_ring_kernel<..., FS>): the caller's workspace (split_ws_bytes; the launcher cuts it into the
    // ticket words and the slabs), slices per tile.  NULL / 0: not lent.
    float* split_ws; long split_ws_bytes; unsigned* split_tickets; int fsplit;
};
#define MMSUM_NT_FSPLIT_MAX 4
#define MMSUM_NT_FSPLIT_TICKET_BYTES 4096

// Padded-layout row of output pixel m = (n, y, x) for tap (0, 0): n * Hp * Wp + y * Wp + x.
__device__ __forceinline__ int conv_row(const GemmArgs& p, int m) {
    const int n = m / p.conv_hw, rem = m - n * p.conv_hw;
    const int y = rem / p.conv_w, x = rem - y * p.conv_w;
    return n * p.conv_hpwp + y * p.conv_wp + x;
}
// Element offset (to add to the row's address) of reduction index k0 = tap * C + c0, k0 a multiple of 32 (scalar arithmetic).
__device__ __forceinline__ int conv_koff(const GemmArgs& p, int k0) {
    const int tap = k0 >> p.conv_cshift, c0 = k0 & ((1 << p.conv_cshift) - 1);
    const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;            // tap / 3, tap % 3 for tap <= 8
    return (ky * p.conv_wp + kx) * (int)p.lda + c0;
}

// Live row count (device-resident, so one captured HIP graph serves every batch): rows at and past it are neither read
// nor written.  Natural A: limits M.  Reduction-major product (A_T | B_T, the weight gradient): limits K.
__device__ __forceinline__ void apply_live_rows(GemmArgs& p, int& m_cap) {
    m_cap = p.M;
    if (p.alpha_dev != nullptr) p.alpha *= *p.alpha_dev;
    if (p.live != nullptr) {
        const int lv = max(0, __builtin_amdgcn_readfirstlane(*p.live));
        if ((p.flags & (MMSUM_GEMM_A_T | MMSUM_GEMM_B_T)) == (MMSUM_GEMM_A_T | MMSUM_GEMM_B_T)) p.K = min(p.K, lv);
        else p.M = min(p.M, lv);
    }
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    return 0.5f * (1.f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

// bf16 epilogues: Phi(x) by Abramowitz-Stegun 7.1.26 (|error| < 8e-8 absolute, far below bf16 rounding): one
// v_exp + one v_rcp + 6 FMAs instead of erff's ~40 instructions with branches; the exp is shared with the pdf term.
__device__ __forceinline__ void gelu_terms_fast(float x, float& cdf, float& e) {
    const float t = __builtin_amdgcn_rcpf(1.f + 0.23164189f * fabsf(x));          // 0.3275911 / sqrt(2)
    e = __expf(-0.5f * x * x);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float q = 0.5f * poly * e;                                               // Phi(-|x|), no cancellation
    cdf = x >= 0.f ? 1.f - q : q;
}
__device__ __forceinline__ float gelu_fast_f(float x) {
    float cdf, e;
    gelu_terms_fast(x, cdf, e);
    return x * cdf;
}
__device__ __forceinline__ float gelu_grad_fast_f(float x) {
    float cdf, e;
    gelu_terms_fast(x, cdf, e);
    return cdf + x * 0.39894228040143267794f * e;
}

// The same two functions on PAIRS of values, written on 2-vectors so that the polynomial, the squares and the products issue
// as packed f32 instructions (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth of work per issue slot); |x| rides on the source
// modifiers of the non-packed instructions.  Per element 7 + 2 transcendental instructions instead of 15 + 2: in the four-wave
// GEMM a 256x256 tile's GELU cost 9.8 us of a 39 us tile (in-kernel stamps, tools/w4_stamps.py).
//   gelu(x)  = max(x, 0) - |x| q(|x|),   gelu'(x) = 1/2 + copysign(1/2 - q, x) + x pdf(x),   q = Phi(-|x|) as above.
typedef float gelu_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_q_e_pair(gelu_f32x2_t x, gelu_f32x2_t& q, gelu_f32x2_t& e) {
    gelu_f32x2_t t;
    t.x = __builtin_amdgcn_rcpf(fmaf(fabsf(x.x), 0.23164189f, 1.f));
    t.y = __builtin_amdgcn_rcpf(fmaf(fabsf(x.y), 0.23164189f, 1.f));
    const gelu_f32x2_t a = (x * x) * gelu_f32x2_t{-0.72134752044f, -0.72134752044f};            // -x^2/2 log2(e)
    e.x = __builtin_amdgcn_exp2f(a.x);
    e.y = __builtin_amdgcn_exp2f(a.y);
    const gelu_f32x2_t c5 = {0.5307027145f, 0.5307027145f}, c4 = {-0.7265760135f, -0.7265760135f}, c3 = {0.7107068705f, 0.7107068705f},
                       c2 = {-0.142248368f, -0.142248368f}, c1 = {0.127414796f, 0.127414796f};         // the A-S coefficients, halved
    const gelu_f32x2_t poly = t * (c1 + t * (c2 + t * (c3 + t * (c4 + t * c5))));
    q = poly * e;
}
__device__ __forceinline__ gelu_f32x2_t gelu_fast2(gelu_f32x2_t x) {
    gelu_f32x2_t q, e, r;
    gelu_q_e_pair(x, q, e);
    r.x = fmaf(-fabsf(x.x), q.x, fmaxf(x.x, 0.f));
    r.y = fmaf(-fabsf(x.y), q.y, fmaxf(x.y, 0.f));
    return r;
}
__device__ __forceinline__ gelu_f32x2_t gelu_grad_fast2(gelu_f32x2_t x) {
    gelu_f32x2_t q, e, h;
    gelu_q_e_pair(x, q, e);
    const gelu_f32x2_t d = gelu_f32x2_t{0.5f, 0.5f} - q;                                            // >= 0
    h.x = __builtin_copysignf(d.x, x.x);
    h.y = __builtin_copysignf(d.y, x.y);
    return (h + gelu_f32x2_t{0.5f, 0.5f}) + (x * e) * gelu_f32x2_t{0.39894228040143267794f, 0.39894228040143267794f};
}

// Epilogue of one wave: acc[i][j] is the 32x32 tile at rows row0 + i*32, columns col0 + j*32.
// OUT selects the store form at compile time (the runtime-flag version costs ~250 instructions per
// element): 0 = store T, 1 = T += , 2 = f32 += , 3 = f32 atomic += , 4 = store f32.
enum { OUT_T = 0, OUT_T_ACC = 1, OUT_F32_ACC = 2, OUT_F32_ATOMIC = 3, OUT_F32 = 4 };

// Position of accumulator register r of a 32x32 block for this lane, by accumulator layout LAY:
//   LAY_32   the block is one 32x32x16 MFMA result (column on the lane);
//   LAY_16   four 16x16x32 results, registers 4q..4q+3 = quarter q = 2 (row half) + (col half), each with rows 4 (lane >> 4) + e
//            and column lane & 15;
//   LAY_16T  the same four quarters from MFMAs issued with the operand roles swapped (weights as the MFMA's A operand): row
//            lane & 15 and the four CONSECUTIVE columns 4 (lane >> 4) + e -- what a lane holds of a quarter is 8 contiguous bytes of a
//            bf16 output row.
enum { LAY_32 = 0, LAY_16 = 1, LAY_16T = 2 };
template <int LAY> __device__ __forceinline__ int blk_row(int r, int lane) {
    return LAY == LAY_16 ? 16 * (r >> 3) + 4 * (lane >> 4) + (r & 3) : LAY == LAY_16T ? 16 * (r >> 3) + (lane & 15) : acc_row(r, lane);
}
template <int LAY> __device__ __forceinline__ int blk_col(int r, int lane) {
    return LAY == LAY_16 ? 16 * ((r >> 2) & 1) + (lane & 15) : LAY == LAY_16T ? 16 * ((r >> 2) & 1) + 4 * (lane >> 4) + (r & 3) : (lane & 31);
}

template <typename T, int TM, int TN, int EPI, int OUT, int LAY = LAY_32>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, const f32x16_t (&acc)[TM][TN], int row0, int col0, int ks, int lane) {
    const bool has_bias = (p.flags & MMSUM_GEMM_BIAS) && (ks == 0);
    float* Cf = static_cast<float*>(p.C);
    T* Ct = static_cast<T*>(p.C);
    T* aux = static_cast<T*>(p.aux);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = col0 + j * 32 + blk_col<LAY>(r, lane);
                const bool col_ok = col < p.N;
                const float bv = (has_bias && col_ok) ? p.bias[col] : 0.f;
                const int row = row0 + i * 32 + blk_row<LAY>(r, lane);
                if (!col_ok || row >= p.M) continue;
                float v = acc[i][j][r] * p.alpha + bv;
                if constexpr (EPI == MMSUM_EPI_GELU) {
                    if (aux) aux[(long)row * p.ldaux + col] = from_f32<T>(v);
                    v = gelu_f(v);
                } else if constexpr (EPI == MMSUM_EPI_GELU_BWD) {
                    v *= gelu_grad_f(to_f32(aux[(long)row * p.ldaux + col]));
                } else if constexpr (EPI == MMSUM_EPI_RELU) {
                    v = fmaxf(v, 0.f);
                } else if constexpr (EPI == MMSUM_EPI_RELU_BWD) {
                    v = (to_f32(aux[(long)row * p.ldaux + col]) > 0.f) ? v : 0.f;
                }
                const long o = (long)row * p.ldc + col;
                if constexpr (OUT == OUT_T) Ct[o] = from_f32<T>(v);
                else if constexpr (OUT == OUT_T_ACC) Ct[o] = from_f32<T>(to_f32(Ct[o]) + v);
                else if constexpr (OUT == OUT_F32_ACC) Cf[o] += v;
                else if constexpr (OUT == OUT_F32_ATOMIC) atomicAdd(Cf + o, v);
                else Cf[o] = v;
            }
        }
    }
}

inline int out_mode_of(const GemmArgs& a) {
    const bool f32 = a.flags & MMSUM_GEMM_OUT_F32, acc = a.flags & MMSUM_GEMM_ACCUM;
    if (a.splitk > 1 && !(a.flags & MMSUM_GEMM_SLABS)) return OUT_F32_ATOMIC;
    if (a.flags & MMSUM_GEMM_SLABS) return OUT_F32;
    if (f32) return acc ? OUT_F32_ACC : OUT_F32;
    return acc ? OUT_T_ACC : OUT_T;
}

// Grouped rasterisation: logical tile ids walk GROUP_M tile-rows at a time, so that the 32 workgroups of an XCD (32 consecutive
// logical tiles: xcd_remap below) run an 8 x 4 block of tiles and share its 8 A and 4 B panels in the XCD's 4 MiB L2.
// (Measured and not kept, round 3: walking groups of four tile COLUMNS down the rows instead, so that the four B panels stay
// L2-resident -- within +-1 % on every shape of the step at M = 64,512)
__device__ __forceinline__ void tile_coords(int t, int tiles_m, int tiles_n, int& tm, int& tn) {
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * tiles_n;
    const int g = t / per_group;
    const int first_m = g * GROUP_M;
    const int gm = min(GROUP_M, tiles_m - first_m);
    const int in = t - g * per_group;
    tm = first_m + in % gm;
    tn = in / gm;
}

// bijective XCD remap (blocks b and b+8 share an XCD): consecutive logical ids land on one XCD
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
    return (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
}

struct GemmPlan { int kernel, bm, bn, grid; };                  // kernel: MMSUM_PLAN_* of include/mmsum_hip.h
GemmPlan plan_gemm_glds(const GemmArgs& a);
GemmPlan plan_gemm_tn(const GemmArgs& a);
int launch_gemm_glds(const GemmArgs& a, hipStream_t stream);   // gemm_fast.hip
bool gemm_glds_eligible(int dtype, const GemmArgs& a);
int launch_gemm_skinny(const GemmArgs& a, hipStream_t stream);   // gemm_skinny.hip: M <= 64 (decode step)
bool gemm_skinny_eligible(int dtype, const GemmArgs& a);
int launch_gemm_skinny_f32(const GemmArgs& a, hipStream_t stream);   // gemm_skinny.hip: f32 x, W, out with M <= 96 (decode step of the f32 mode)
bool gemm_skinny_f32_eligible(int dtype, const GemmArgs& a);
int launch_gemm_tn(const GemmArgs& a, hipStream_t stream);     // gemm_fast.hip: A [K,M], B [K,N] (weight gradients)
int launch_gemm_tn_w4(const GemmArgs& a, hipStream_t stream);  // gemm_fast.hip: the 256x256 four-wave TN kernel (implicit-convolution weight gradient)
bool gemm_tn_eligible(int dtype, const GemmArgs& a);
bool gemm_tn_colsum_ok(int dtype, const GemmArgs& a);          // MMSUM_GEMM_COLSUM on the weight-gradient product: column sums of A
