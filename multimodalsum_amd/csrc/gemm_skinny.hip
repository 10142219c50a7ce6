// Skinny bf16 GEMM for the single-token decode step of generation (multimodalsum_amd/generation.py):
//   out[M, N] = epi(alpha * x[M, K] . W[N, K]^T + bias),  M <= 128 hypothesis rows, N, K = model dimensions.
// The product is a weight stream: W (2 N K bytes) is read once, x (<= 64 rows) stays in L2.  The tiled kernels put
// one workgroup on a 128- or 256-column tile, i.e. 8 workgroups for N = 1024 -- 3 % of the chip pulling the
// weights (measured 26 us per product, half of a decode step).  Here a workgroup owns 32 output columns, its four
// waves split K four ways and read both operands straight from global memory in the MFMA operand layout (two 16-byte
// loads per lane per 32-deep slab and operand, no LDS in the loop), the partial accumulators meet in LDS and the waves
// share the bias / GELU / store work.  N = 1024 gives 32 workgroups of 4 (or 8, see below) waves streaming.
#include "gemm_common.h"

namespace {

// NW waves split K NW ways.  Four waves keep every load of a K = 1024 product in flight at once (8 slabs per wave); products with a
// longer reduction or more than 32 rows take eight waves (K = 4096: two batches of 8 slabs per wave instead of four; 96 rows: a third of
// the fragment loads and MFMAs per wave).  The partial accumulators of ALL waves meet in LDS and every wave finishes its share of the
// accumulator registers (sum, bias / GELU, store), instead of one wave adding up seven others.
//
// AF32 (MMSUM_GEMM_A_F32): x is f32 while W is bf16 -- the LM head of the decode step takes the final LayerNorm's output un-rounded.
// A lane splits its eight f32 values of a chunk into hi = bf16(x) and lo = bf16(x - hi) and issues the slab's MFMAs twice: the
// product carries 16 significant bits of x (the weights are the model's bf16 weights either way) and accumulates in f32.
// CF32 (MMSUM_GEMM_OUT_F32): the result is stored as f32 (decode logits: the quantity that is ranked keeps its f32 accumulator).
__device__ __forceinline__ void split_hi_lo(const float* src, bool valid, u32x4_t& hi, u32x4_t& lo) {
    bf16_t h[8], l[8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        f32x4_t v = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (valid) v = *reinterpret_cast<const f32x4_t*>(src + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            h[4 * q + e] = (bf16_t)v[e];
            l[4 * q + e] = (bf16_t)(v[e] - (float)h[4 * q + e]);
        }
    }
    __builtin_memcpy(&hi, h, 16);
    __builtin_memcpy(&lo, l, 16);
}

template <int MT, int EPI, int NW, bool AF32 = false, bool CF32 = false>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float (*red)[MT][16][64] = reinterpret_cast<float (*)[MT][16][64]>(smem_raw);      // [NW][MT][16][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * 32;
    const bf16_t* A = static_cast<const bf16_t*>(p.A);
    const bf16_t* A2 = static_cast<const bf16_t*>(p.A2);
    const bf16_t* B = static_cast<const bf16_t*>(p.B);
    const int nslab = p.K / 32, per = nslab / NW;
    const int s0 = wave * per, s1 = s0 + per;
    const int ln = lane & 31;
    const int nrow = n0 + ln;
    const bf16_t* brow = B + (long)(nrow < p.N ? nrow : p.N - 1) * p.ldb;
    f32x16_t acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = zero_acc();
#pragma unroll 8
    for (int s = s0; s < s1; ++s) {
        int k0 = s * 32;
        const Frag b = global_frag<bf16_t>(brow + k0, lane, true);
        const bf16_t* Ab = A;
        long lda = p.lda;
        if (A2 != nullptr && k0 >= p.ksplit) { Ab = A2; lda = p.lda2; k0 -= p.ksplit; }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = mt * 32 + ln;
            if constexpr (AF32) {
                const float* arow = static_cast<const float*>(p.A) + (long)(m < p.M ? m : 0) * p.lda + k0;
                Frag hi, lo;
#pragma unroll
                for (int i = 0; i < 2; ++i) split_hi_lo(arow + lane_chunk<bf16_t>(lane >> 5, i) * 8, m < p.M, hi.c[i], lo.c[i]);
                mma_slab<bf16_t>(acc[mt], hi, b);
                mma_slab<bf16_t>(acc[mt], lo, b);
            } else {
                const Frag a = global_frag<bf16_t>(Ab + (long)(m < p.M ? m : 0) * lda + k0, lane, m < p.M);
                mma_slab<bf16_t>(acc[mt], a, b);
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][mt][r][lane] = acc[mt][r];
    __syncthreads();
    // wave w finishes accumulator registers w, w + NW, ... of the MT * 16
    const bool col_ok = nrow < p.N;
    const float bv = ((p.flags & MMSUM_GEMM_BIAS) && col_ok) ? p.bias[nrow] : 0.f;
    bf16_t* C = static_cast<bf16_t*>(p.C);
    for (int i = wave; i < MT * 16; i += NW) {
        const int mt = i / 16, r = i % 16;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w][mt][r][lane];
        v = v * p.alpha + bv;
        if constexpr (EPI == MMSUM_EPI_GELU) v = gelu_fast_f(v);
        const int m = mt * 32 + acc_row(r, lane);
        if (col_ok && m < p.M) {
            if constexpr (CF32) static_cast<float*>(p.C)[(long)m * p.ldc + nrow] = v;
            else C[(long)m * p.ldc + nrow] = (bf16_t)v;
        }
    }
}

// The same product on 16-column workgroups and v_mfma_f32_16x16x32_bf16 (lane (r = l & 15, g = l >> 4) holds 8 consecutive k of row r:
// one 16-byte load per lane, operand and slab): twice the workgroups for the same N.  A workgroup streams its weights at only ~16 GB/s
// (measured: the K = 4096, N = 1024 product took 16.7 us on 32 workgroups whatever the number of loads in flight), so the products with
// N <= 4096 -- everything in the decode step but the LM head -- are bounded by how many CUs pull weights.  MB = 16-row blocks of x.
template <int MB, int EPI, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_skinny16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float (*red)[MB][4][64] = reinterpret_cast<float (*)[MB][4][64]>(smem_raw);       // [NW][MB][4][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * 16;
    const bf16_t* A = static_cast<const bf16_t*>(p.A);
    const bf16_t* A2 = static_cast<const bf16_t*>(p.A2);
    const bf16_t* B = static_cast<const bf16_t*>(p.B);
    const int nslab = p.K / 32, per = nslab / NW;
    const int s0 = wave * per, s1 = s0 + per;
    const int lr = lane & 15, kg = (lane >> 4) * 8;
    const int nrow = n0 + lr;
    const bf16_t* brow = B + (long)(nrow < p.N ? nrow : p.N - 1) * p.ldb + kg;
    f32x4_t acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int s = s0; s < s1; ++s) {
        int k0 = s * 32;
        const u32x4_t b = *reinterpret_cast<const u32x4_t*>(brow + k0);
        const bf16_t* Ab = A;
        long lda = p.lda;
        if (A2 != nullptr && k0 >= p.ksplit) { Ab = A2; lda = p.lda2; k0 -= p.ksplit; }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int m = mb * 16 + lr;
            u32x4_t a = u32x4_t{0, 0, 0, 0};
            if (m < p.M) a = *reinterpret_cast<const u32x4_t*>(Ab + (long)m * lda + k0 + kg);
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc[mb], 0, 0, 0);
        }
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave][mb][e][lane] = acc[mb][e];
    __syncthreads();
    const bool col_ok = nrow < p.N;
    const float bv = ((p.flags & MMSUM_GEMM_BIAS) && col_ok) ? p.bias[nrow] : 0.f;
    bf16_t* C = static_cast<bf16_t*>(p.C);
    for (int i = wave; i < MB * 4; i += NW) {
        const int mb = i / 4, e = i % 4;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w][mb][e][lane];
        v = v * p.alpha + bv;
        if constexpr (EPI == MMSUM_EPI_GELU) v = gelu_fast_f(v);
        const int m = mb * 16 + 4 * (lane >> 4) + e;          // 16x16 result: row 4 (l >> 4) + e, column l & 15
        if (col_ok && m < p.M) C[(long)m * p.ldc + nrow] = (bf16_t)v;
    }
}

}  // namespace

bool gemm_skinny_eligible(int dtype, const GemmArgs& a) {
    if (dtype != MMSUM_BF16 || a.M > 128 || a.splitk != 1 || a.live != nullptr || a.alpha_dev != nullptr) return false;
    if (a.flags & (MMSUM_GEMM_A_T | MMSUM_GEMM_B_T | MMSUM_GEMM_ACCUM | MMSUM_GEMM_SLABS | MMSUM_GEMM_COLSUM)) return false;
    const int epi = (a.flags >> 3) & 7;
    if (!(epi == MMSUM_EPI_NONE || (epi == MMSUM_EPI_GELU && a.aux == nullptr))) return false;
    // f32 result / f32 x: the decode LM head only (32-column workgroups, plain epilogue, one operand tensor)
    if ((a.flags & (MMSUM_GEMM_OUT_F32 | MMSUM_GEMM_A_F32)) && (epi != MMSUM_EPI_NONE || a.A2 != nullptr || a.M > 64)) return false;
    if ((a.flags & MMSUM_GEMM_A_F32) && !(a.flags & MMSUM_GEMM_OUT_F32)) return false;
    if (a.K % 128 || (a.A2 && a.ksplit % 32)) return false;
    if (a.N < 256) return false;                 // tiny outputs: nothing to gain
    return true;
}

template <int MT, int EPI, int NW, bool AF32 = false, bool CF32 = false>
int launch_skinny_one(const GemmArgs& a, hipStream_t stream) {
    const size_t lds = (size_t)NW * MT * 16 * 64 * sizeof(float);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_skinny_kernel<MT, EPI, NW, AF32, CF32>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return MMSUM_ERR_HIP;
    gemm_skinny_kernel<MT, EPI, NW, AF32, CF32><<<dim3((a.N + 31) / 32), dim3(NW * 64), lds, stream>>>(a);
    return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
}

template <int MB, int EPI, int NW>
int launch_skinny16_one(const GemmArgs& a, hipStream_t stream) {
    const size_t lds = (size_t)NW * MB * 4 * 64 * sizeof(float);
    gemm_skinny16_kernel<MB, EPI, NW><<<dim3((a.N + 15) / 16), dim3(NW * 64), lds, stream>>>(a);
    return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
}

int launch_gemm_skinny(const GemmArgs& a, hipStream_t stream) {
    const int epi = (a.flags >> 3) & 7;
    if (a.flags & MMSUM_GEMM_OUT_F32) {                        // decode logits (eligibility: plain epilogue, M <= 64)
        const bool af = a.flags & MMSUM_GEMM_A_F32;
        if (a.M <= 32) return af ? launch_skinny_one<1, MMSUM_EPI_NONE, 4, true, true>(a, stream) : launch_skinny_one<1, MMSUM_EPI_NONE, 4, false, true>(a, stream);
        return af ? launch_skinny_one<2, MMSUM_EPI_NONE, 8, true, true>(a, stream) : launch_skinny_one<2, MMSUM_EPI_NONE, 8, false, true>(a, stream);
    }
    if (a.N <= 4096 && a.M <= 96 && a.K % 256 == 0) {          // 16-column workgroups: more CUs on the weight stream
        const bool w8 = a.K >= 2048 || a.M > 32;
#define SK16(MB)                                                                                                          \
        do {                                                                                                              \
            if (w8) return epi == MMSUM_EPI_GELU ? launch_skinny16_one<MB, MMSUM_EPI_GELU, 8>(a, stream)                  \
                                                 : launch_skinny16_one<MB, MMSUM_EPI_NONE, 8>(a, stream);                 \
            return epi == MMSUM_EPI_GELU ? launch_skinny16_one<MB, MMSUM_EPI_GELU, 4>(a, stream)                          \
                                         : launch_skinny16_one<MB, MMSUM_EPI_NONE, 4>(a, stream);                         \
        } while (0)
        if (a.M <= 16) SK16(1);
        else if (a.M <= 32) SK16(2);
        else if (a.M <= 48) SK16(3);
        else if (a.M <= 64) SK16(4);
        else if (a.M <= 80) SK16(5);
        else SK16(6);
#undef SK16
    }
    // eight waves for a long reduction or more than 32 rows (K % 256 == 0 then; eligibility guarantees K % 128)
    const bool wide = (a.K >= 2048 || a.M > 32) && a.K % 256 == 0 && a.M <= 96;
    // (sixteen waves for K = 4096 measured the same 16.7 us as eight: 32 workgroups stream the 8 MB of weights at ~16 GB/s per CU,
    // which is what bounds that product, not the number of loads in flight)
#define SKINNY(MT)                                                                                                     \
    do {                                                                                                               \
        if (wide) return epi == MMSUM_EPI_GELU ? launch_skinny_one<MT, MMSUM_EPI_GELU, 8>(a, stream)                   \
                                               : launch_skinny_one<MT, MMSUM_EPI_NONE, 8>(a, stream);                  \
        return epi == MMSUM_EPI_GELU ? launch_skinny_one<MT, MMSUM_EPI_GELU, 4>(a, stream)                             \
                                     : launch_skinny_one<MT, MMSUM_EPI_NONE, 4>(a, stream);                            \
    } while (0)
    if (a.M <= 32) SKINNY(1);
    else if (a.M <= 64) SKINNY(2);
    else if (a.M <= 96) SKINNY(3);
    else SKINNY(4);
#undef SKINNY
}
