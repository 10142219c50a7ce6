// Skinny bf16 GEMM for the single-token decode step of generation (multimodalsum_amd/generation.py):
//   out[M, N] = epi(alpha * x[M, K] . W[N, K]^T + bias),  M <= 128 hypothesis rows, N, K = model dimensions.
// The product is a weight stream: W (2 N K bytes) is read once, x (<= 64 rows) stays in L2.  The tiled kernels put
// one workgroup on a 128- or 256-column tile, i.e. 8 workgroups for N = 1024 -- 3 % of the chip pulling the
// weights (measured 26 us per product, half of a decode step).  Here a workgroup owns 32 output columns, its four
// waves split K four ways and read both operands straight from global memory in the MFMA operand layout (two 16-byte
// loads per lane per 32-deep slab and operand, no LDS in the loop), the partial accumulators meet in LDS and wave 0
// applies bias / GELU and stores.  N = 1024 gives 32 workgroups of 4 waves, 128 waves streaming.
#include "gemm_common.h"

namespace {

template <int MT, int EPI>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs p) {
    __shared__ float red[3][MT][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 32;
    const bf16_t* A = static_cast<const bf16_t*>(p.A);
    const bf16_t* A2 = static_cast<const bf16_t*>(p.A2);
    const bf16_t* B = static_cast<const bf16_t*>(p.B);
    const int nslab = p.K / 32, per = nslab / 4;
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
            const Frag a = global_frag<bf16_t>(Ab + (long)(m < p.M ? m : 0) * lda + k0, lane, m < p.M);
            mma_slab<bf16_t>(acc[mt], a, b);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave - 1][mt][r][lane] = acc[mt][r];
    }
    __syncthreads();
    if (wave == 0) {
        const bool col_ok = nrow < p.N;
        const float bv = ((p.flags & MMSUM_GEMM_BIAS) && col_ok) ? p.bias[nrow] : 0.f;
        bf16_t* C = static_cast<bf16_t*>(p.C);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[mt][r] + red[0][mt][r][lane] + red[1][mt][r][lane] + red[2][mt][r][lane];
                v = v * p.alpha + bv;
                if constexpr (EPI == MMSUM_EPI_GELU) v = gelu_fast_f(v);
                const int m = mt * 32 + acc_row(r, lane);
                if (col_ok && m < p.M) C[(long)m * p.ldc + nrow] = (bf16_t)v;
            }
    }
}

}  // namespace

bool gemm_skinny_eligible(int dtype, const GemmArgs& a) {
    if (dtype != MMSUM_BF16 || a.M > 128 || a.splitk != 1 || a.live != nullptr || a.alpha_dev != nullptr) return false;
    if (a.flags & (MMSUM_GEMM_A_T | MMSUM_GEMM_B_T | MMSUM_GEMM_ACCUM | MMSUM_GEMM_OUT_F32 | MMSUM_GEMM_SLABS | MMSUM_GEMM_COLSUM)) return false;
    const int epi = (a.flags >> 3) & 7;
    if (!(epi == MMSUM_EPI_NONE || (epi == MMSUM_EPI_GELU && a.aux == nullptr))) return false;
    if (a.K % 128 || (a.A2 && a.ksplit % 32)) return false;
    if (a.N < 256) return false;                 // tiny outputs: nothing to gain
    return true;
}

int launch_gemm_skinny(const GemmArgs& a, hipStream_t stream) {
    const int epi = (a.flags >> 3) & 7;
    const dim3 grid((a.N + 31) / 32), block(256);
#define SKINNY(MT)                                                                                         \
    do {                                                                                                   \
        if (epi == MMSUM_EPI_GELU) gemm_skinny_kernel<MT, MMSUM_EPI_GELU><<<grid, block, 0, stream>>>(a);   \
        else gemm_skinny_kernel<MT, MMSUM_EPI_NONE><<<grid, block, 0, stream>>>(a);                         \
    } while (0)
    if (a.M <= 32) SKINNY(1);
    else if (a.M <= 64) SKINNY(2);
    else if (a.M <= 96) SKINNY(3);
    else SKINNY(4);
#undef SKINNY
    return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
}
