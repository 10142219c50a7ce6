// Skinny bf16 GEMM for the single-token decode step of generation (multimodalsum_amd/generation.py):
//   out[M, N] = epi(alpha * x[M, K] . W[N, K]^T + bias),  M <= 128 hypothesis rows, N, K = model dimensions.
// The product is a weight stream: W (2 N K bytes) is read once, x (<= 64 rows) stays in L2.  The tiled kernels put
// one workgroup on a 128- or 256-column tile, i.e. 8 workgroups for N = 1024 -- 3 % of the chip pulling the
// weights (measured 26 us per product, half of a decode step).  Here a workgroup owns 32 output columns, its four
// waves split K four ways and read both operands straight from global memory in the MFMA operand layout (two 16-byte
// loads per lane per 32-deep slab and operand, no LDS in the loop), the partial accumulators meet in LDS and the waves
// share the bias / GELU / store work.  N = 1024 gives 32 workgroups of 4 (or 8, see below) waves streaming.
#include "gemm_common.h"
#include <type_traits>

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
    // slabs in batches of NSB with every load of a batch issued before its first MFMA (see skinny16_body); rows past M re-read row 0
    auto batch = [&](int s, auto nsb_c) {
        constexpr int NSB = decltype(nsb_c)::value;
        Frag b[NSB];
#pragma unroll
        for (int i = 0; i < NSB; ++i) b[i] = global_frag<bf16_t>(brow + (s + i) * 32, lane, true);
        if constexpr (AF32) {
            f32x4_t xa[MT][NSB][2][2];                          // [row block][slab][16-byte chunk of the fragment][half]: 8 floats per chunk
#pragma unroll
            for (int i = 0; i < NSB; ++i)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int m = mt * 32 + ln;
                    const float* arow = static_cast<const float*>(p.A) + (long)(m < p.M ? m : 0) * p.lda + (s + i) * 32;
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int q = 0; q < 2; ++q) xa[mt][i][c][q] = *reinterpret_cast<const f32x4_t*>(arow + lane_chunk<bf16_t>(lane >> 5, c) * 8 + 4 * q);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NSB; ++i)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    Frag hi, lo;
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        bf16_t h[8], l[8];
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float v = xa[mt][i][c][q][e];
                                h[4 * q + e] = (bf16_t)v;
                                l[4 * q + e] = (bf16_t)(v - (float)h[4 * q + e]);
                            }
                        __builtin_memcpy(&hi.c[c], h, 16);
                        __builtin_memcpy(&lo.c[c], l, 16);
                    }
                    mma_slab<bf16_t>(acc[mt], hi, b[i]);
                    mma_slab<bf16_t>(acc[mt], lo, b[i]);
                }
        } else {
            Frag a[MT][NSB];
#pragma unroll
            for (int i = 0; i < NSB; ++i) {
                int k0 = (s + i) * 32;
                const bf16_t* Ab = A;
                long lda = p.lda;
                if (A2 != nullptr && k0 >= p.ksplit) { Ab = A2; lda = p.lda2; k0 -= p.ksplit; }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int m = mt * 32 + ln;
                    a[mt][i] = global_frag<bf16_t>(Ab + (long)(m < p.M ? m : 0) * lda + k0, lane, true);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NSB; ++i)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) mma_slab<bf16_t>(acc[mt], a[mt][i], b[i]);
        }
    };
    {
        constexpr int NSMAX = (AF32 || MT > 1) ? 4 : 8;
        int s = s0;
        for (; s + NSMAX <= s1; s += NSMAX) batch(s, std::integral_constant<int, NSMAX>{});
        if constexpr (NSMAX == 8) if (s + 4 <= s1) { batch(s, std::integral_constant<int, 4>{}); s += 4; }
        if (s + 2 <= s1) { batch(s, std::integral_constant<int, 2>{}); s += 2; }
        if (s < s1) batch(s, std::integral_constant<int, 1>{});
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
__device__ __forceinline__ void skinny16_body(const GemmArgs& p, char* smem_raw) {
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
    // A wave's slabs in BATCHES: every load of a batch (NSB of the weights, MB x NSB of x) is issued before its first MFMA, and
    // unconditionally -- rows past M re-read row 0 (they only feed output rows that are never stored).  Round 5: the guarded
    // one-slab-at-a-time form this replaces compiled to load, wait, MFMA per slab (`#pragma unroll` did not apply to it), i.e. one
    // dependent memory round trip per slab: the "7 us floor" of these products was eight round trips, not one.
    auto batch = [&](int s, auto nsb_c) {
        constexpr int NSB = decltype(nsb_c)::value;
        u32x4_t b[NSB], a[MB][NSB];
#pragma unroll
        for (int i = 0; i < NSB; ++i) b[i] = *reinterpret_cast<const u32x4_t*>(brow + (s + i) * 32);
#pragma unroll
        for (int i = 0; i < NSB; ++i) {
            int k0 = (s + i) * 32;
            const bf16_t* Ab = A;
            long lda = p.lda;
            if (A2 != nullptr && k0 >= p.ksplit) { Ab = A2; lda = p.lda2; k0 -= p.ksplit; }       // (uniform: scalar selects)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int m = mb * 16 + lr;
                a[mb][i] = *reinterpret_cast<const u32x4_t*>(Ab + (long)(m < p.M ? m : 0) * lda + k0 + kg);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NSB; ++i)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
                acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[mb][i]), __builtin_bit_cast(bf16x8_t, b[i]), acc[mb], 0, 0, 0);
    };
    {
        constexpr int NSMAX = MB <= 2 ? 8 : 4;                 // (MB + 1) x NSB 16-byte loads in flight per lane
        int s = s0;
        for (; s + NSMAX <= s1; s += NSMAX) batch(s, std::integral_constant<int, NSMAX>{});
        if constexpr (NSMAX == 8) if (s + 4 <= s1) { batch(s, std::integral_constant<int, 4>{}); s += 4; }
        if (s + 2 <= s1) { batch(s, std::integral_constant<int, 2>{}); s += 2; }
        if (s < s1) batch(s, std::integral_constant<int, 1>{});
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
template <int MB, int EPI, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_skinny16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    skinny16_body<MB, EPI, NW>(p, smem_raw);
}
// Two INDEPENDENT products of one shape in one launch (grid.y picks the product): the decode step's alpha and beta projections
// ([yt ; ytab] W_alpha^T and [yt ; yimg] W_beta^T, :738-739) -- each alone is 64 workgroups on a ~7 us floor.
template <int MB, int EPI, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_skinny16_pair_kernel(GemmArgs p0, GemmArgs p1) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    skinny16_body<MB, EPI, NW>(blockIdx.y ? p1 : p0, smem_raw);
}

// The f32 compute mode's decode step (the mode whose generated token ids are held to the reference's, tests/test_generation_gpu.py): f32 x,
// f32 weights, f32 result.  The generic f32 kernel puts one workgroup on a 128-column tile -- 8 workgroups pull the 4 MB of an N = K = 1024
// weight matrix (measured: 15.7 ms per decode step, 2 % of its HBM roofline).  Same scheme as gemm_skinny16_kernel: 16 output columns per
// workgroup, the NW waves split K, both operands straight from global memory in the MFMA operand layout -- v_mfma_f32_16x16x4_f32, lane
// (r = l & 15, g = l >> 4) holds A[r][k = g] / B[k = g][r].  A lane loads FOUR consecutive k of its row with one 16-byte load; MFMA j of a
// 16-k slab takes element j of every lane, i.e. the k set {j, 4 + j, 8 + j, 12 + j} -- the same set on both operands, and a sum over k does
// not care about the order.  f32 products, f32 accumulation (as the generic f32 kernel's v_mfma_f32_32x32x2_f32).
template <int MB, int EPI, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_f32_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float (*red)[MB][4][64] = reinterpret_cast<float (*)[MB][4][64]>(smem_raw);       // [NW][MB][4][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * 16;
    const float* A = static_cast<const float*>(p.A);
    const float* A2 = static_cast<const float*>(p.A2);
    const float* B = static_cast<const float*>(p.B);
    // a wave's share of K in BATCHES of NS 16-k slabs: every load of a batch (NS of the weights, MB x NS of x) is issued before its first
    // MFMA and unconditionally (rows past M re-read row 0: they only feed output rows that are never stored) -- a loop of load, wait,
    // four MFMAs per slab (what the compiler made of the guarded form) is one dependent memory round trip per slab: 18 us per product
    constexpr int NS = MB <= 3 ? 8 : 4;                              // (MB x NS + NS) 16-byte loads in flight per lane: 96 rows x 8 slabs would not fit the registers
    const int nslab = p.K / 16, per = nslab / NW;                  // per % NS == 0 (eligibility: K % (16 * NS * NW) == 0)
    const int s0 = wave * per;
    const int lr = lane & 15, kg = (lane >> 4) * 4;
    const int nrow = n0 + lr;
    const float* brow = B + (long)(nrow < p.N ? nrow : p.N - 1) * p.ldb + kg;
    f32x4_t acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int bt = 0; bt < per / NS; ++bt) {
        int k0 = (s0 + bt * NS) * 16;
        f32x4_t b[NS], a[MB][NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) b[s] = *reinterpret_cast<const f32x4_t*>(brow + k0 + s * 16);
        const float* Ab = A;
        long lda = p.lda;
        if (A2 != nullptr && k0 >= p.ksplit) { Ab = A2; lda = p.lda2; k0 -= p.ksplit; }      // a batch lies in ONE of the two tensors (ksplit % 128 == 0)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int m = mb * 16 + lr;
            const float* ar = Ab + (long)(m < p.M ? m : 0) * lda + k0 + kg;
#pragma unroll
            for (int s = 0; s < NS; ++s) a[mb][s] = *reinterpret_cast<const f32x4_t*>(ar + s * 16);
        }
        __builtin_amdgcn_sched_barrier(0);           // every load above is issued before the first MFMA below (the scheduler sank most of them)
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb][s][j], b[s][j], acc[mb], 0, 0, 0);
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave][mb][e][lane] = acc[mb][e];
    __syncthreads();
    const bool col_ok = nrow < p.N;
    const float bv = ((p.flags & MMSUM_GEMM_BIAS) && col_ok) ? p.bias[nrow] : 0.f;
    float* C = static_cast<float*>(p.C);
    for (int i = wave; i < MB * 4; i += NW) {
        const int mb = i / 4, e = i % 4;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w][mb][e][lane];                       // fixed order: bit-reproducible
        v = v * p.alpha + bv;
        if constexpr (EPI == MMSUM_EPI_GELU) v = gelu_f(v);                           // the parity mode keeps erff
        const int m = mb * 16 + 4 * (lane >> 4) + e;
        if (col_ok && m < p.M) C[(long)m * p.ldc + nrow] = v;
    }
}

}  // namespace

bool gemm_skinny_f32_eligible(int dtype, const GemmArgs& a) {
    if (dtype != MMSUM_F32 || a.M > 96 || a.splitk != 1 || a.live != nullptr || a.alpha_dev != nullptr) return false;
    if (a.flags & (MMSUM_GEMM_A_T | MMSUM_GEMM_B_T | MMSUM_GEMM_ACCUM | MMSUM_GEMM_SLABS | MMSUM_GEMM_COLSUM | MMSUM_GEMM_A_F32)) return false;
    const int epi = (a.flags >> 3) & 7;
    if (!(epi == MMSUM_EPI_NONE || (epi == MMSUM_EPI_GELU && a.aux == nullptr))) return false;
    if (a.K % 1024 || (a.A2 && a.ksplit % 128) || a.N < 256) return false;          // eight waves x batches of eight 16-k slabs
    if ((((uintptr_t)a.A) | ((uintptr_t)a.B) | ((uintptr_t)a.A2)) & 15) return false;
    if ((a.lda & 3) || (a.ldb & 3) || (a.A2 && (a.lda2 & 3))) return false;
    return true;
}

template <int MB, int EPI, int NW>
int launch_skinny_f32_one(const GemmArgs& a, hipStream_t stream) {
    const size_t lds = (size_t)NW * MB * 4 * 64 * sizeof(float);
    gemm_skinny_f32_kernel<MB, EPI, NW><<<dim3((a.N + 15) / 16), dim3(NW * 64), lds, stream>>>(a);
    return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
}

int launch_gemm_skinny_f32(const GemmArgs& a, hipStream_t stream) {
    const int epi = (a.flags >> 3) & 7;
#define SKF(MB)                                                                                                           \
    do {                                                                                                                  \
        return epi == MMSUM_EPI_GELU ? launch_skinny_f32_one<MB, MMSUM_EPI_GELU, 8>(a, stream)                            \
                                     : launch_skinny_f32_one<MB, MMSUM_EPI_NONE, 8>(a, stream);                           \
    } while (0)
    if (a.M <= 16) SKF(1);
    else if (a.M <= 32) SKF(2);
    else if (a.M <= 48) SKF(3);
    else if (a.M <= 64) SKF(4);
    else if (a.M <= 80) SKF(5);
    else SKF(6);                 // 96 rows: the three modalities' head outputs of 32 hypotheses through out_proj in one product
#undef SKF
}

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

// ---------------------------------------------------------------------------------------------
// mmsum_dec_gemm: the decode step's products with the REDUCTION split over workgroups.
//
// The kernels above put one workgroup on 16 or 32 output columns and split K over its waves: an N = 1024 product is 32 .. 64
// workgroups, and a workgroup streams its weights at ~16 GB/s whatever its loads in flight (latency-bound), so the product takes
// 7 .. 17 us for 2 .. 8 MB.  Here a workgroup is ONE wave = (16 output columns) x (one K slice of 256 .. 1024): 256 .. 512 workgroups,
// every CU pulling weights, every load of a wave in flight at once (16 bytes per lane and 32-deep step straight into the
// v_mfma_f32_16x16x32_bf16 operand layout), i.e. the whole matrix is requested within one memory round trip.  The slices of a column
// tile meet through an f32 slab each + an arrival ticket; the LAST arriver adds the slabs (in slice order: bit-reproducible) and runs
// the epilogue (bias, GELU, residual add, f32 or bf16 store) -- the guide's in-launch split-K reduction in its write-through form
// (sc1 slab stores, vmcnt(0), relaxed agent-scope ticket; sc1 loads in the reducer): correct for any placement of the slices.  The
// reducer puts the ticket word back to 0, so the workspace serves every product of a stream in turn (kernels of one stream do not
// overlap) and a captured graph needs no memset node per product.
// ---------------------------------------------------------------------------------------------
namespace {

struct DecGemmArgs {
    const void* x; const void* x2; const bf16_t* W; const float* bias; const bf16_t* res; void* out;
    float* slabs; unsigned* tickets;
    long ldx, ldx2, ldw, ldres, ldo;
    int M, N, K, ksplit, sk, ksl, ns16_ok;
};

// NS = 32-deep steps per batch: a batch's loads (NS of the weights, MB x NS of x) are all issued before its first MFMA, so a wave
// that is alone on its SIMD (the small products: 256 .. 512 one-wave workgroups on 256 CUs) still has its whole slice in flight.
template <int MB, int EPI, bool AF32, bool CF32, int NS>
__global__ __launch_bounds__(64) void dec_gemm_kernel(DecGemmArgs p) {
    const int lane = threadIdx.x;
    const int total = gridDim.x;
    const int id = xcd_remap(blockIdx.x, total);               // the slices of a tile get consecutive ids: one XCD (speed only)
    const int tile = id / p.sk, slice = id % p.sk;
    const int lr = lane & 15, kg = (lane >> 4) * 8;
    const int col = tile * 16 + lr;
    const bf16_t* wrow = p.W + (long)(col < p.N ? col : p.N - 1) * p.ldw + kg;
    const int k_beg = slice * p.ksl, nbatch = p.ksl / (32 * NS);
    f32x4_t acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int bt = 0; bt < nbatch; ++bt) {
        int k0 = k_beg + bt * (32 * NS);
        u32x4_t b[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) b[s] = *reinterpret_cast<const u32x4_t*>(wrow + k0 + s * 32);
        const void* xb = p.x;
        long ldx = p.ldx;
        if (p.x2 != nullptr && k0 >= p.ksplit) { xb = p.x2; ldx = p.ldx2; k0 -= p.ksplit; }      // (a batch lies in one of the two tensors: the host checks)
        if constexpr (AF32) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const int m = mb * 16 + lr;
                    u32x4_t hi, lo;
                    split_hi_lo(static_cast<const float*>(xb) + (long)(m < p.M ? m : 0) * ldx + k0 + s * 32 + kg, m < p.M, hi, lo);
                    acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, hi), __builtin_bit_cast(bf16x8_t, b[s]), acc[mb], 0, 0, 0);
                    acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, lo), __builtin_bit_cast(bf16x8_t, b[s]), acc[mb], 0, 0, 0);
                }
        } else {
            u32x4_t a[MB][NS];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int m = mb * 16 + lr;
                const bf16_t* xr = static_cast<const bf16_t*>(xb) + (long)(m < p.M ? m : 0) * ldx + k0 + kg;
#pragma unroll
                for (int s = 0; s < NS; ++s) a[mb][s] = (m < p.M) ? *reinterpret_cast<const u32x4_t*>(xr + s * 32) : u32x4_t{0, 0, 0, 0};
            }
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
                    acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[mb][s]), __builtin_bit_cast(bf16x8_t, b[s]), acc[mb], 0, 0, 0);
        }
    }
    if (p.sk > 1) {
        // this slice's slab: [MB][4][64 lanes] f32, lane-contiguous (256-byte rows).  The hand-off is the guide's write-through form:
        // every slab word is stored sc1 (a relaxed agent-scope atomic store IS a write-through store: it reaches memory past the XCD's
        // L2, so no release fence -- an agent-scope release writes the whole L2's dirty lines back, microseconds per workgroup with
        // 256 .. 512 of them at it), the wave drains its stores (vmcnt(0)), one lane takes the ticket with a relaxed agent-scope add,
        // and the last arriver reads EVERY slab word with an sc1 load (relaxed agent-scope atomic load: it bypasses this CU's L1, which
        // no other CU's store refreshes), so no acquire either.  Placement-independent.
        float* mine = p.slabs + ((long)tile * p.sk + slice) * (MB * 4 * 64);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int e = 0; e < 4; ++e) __hip_atomic_store(mine + (mb * 4 + e) * 64 + lane, acc[mb][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned ticket = 0;
        if (lane == 0) ticket = __hip_atomic_fetch_add(p.tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        if (ticket != (unsigned)(p.sk - 1)) return;            // not the last slice of this tile to arrive
        asm volatile("" ::: "memory");
        // the slabs are added in SLICE order whichever slice arrived last (its own partial takes its place in the order): the result
        // does not depend on the arrival order, i.e. it is bit-reproducible run to run
        const float* base = p.slabs + (long)tile * p.sk * (MB * 4 * 64);
        f32x4_t own[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) { own[mb] = acc[mb]; acc[mb] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
        // (the slabs of up to eight slices are requested together: one slice per loop iteration was one dependent memory round trip
        // each -- fc2's eight slices: 12.0 -> 10.4 us per launch, profiles/r06_decode_ffn_one_launch_stamps.txt)
        for (int s0 = 0; s0 < p.sk; s0 += 8) {
            float other[8][MB][4];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int s = (s0 + j < p.sk) ? s0 + j : p.sk - 1;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        other[j][mb][e] = __hip_atomic_load(base + ((long)s * MB * 4 + mb * 4 + e) * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int s = s0 + j;
                if (s < p.sk) {
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[mb][e] += (s == slice) ? own[mb][e] : other[j][mb][e];
                }
            }
        }
        if (lane == 0) __hip_atomic_store(p.tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the stream's next product
    }
    const bool col_ok = col < p.N;
    const float bv = (p.bias != nullptr && col_ok) ? p.bias[col] : 0.f;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = mb * 16 + 4 * (lane >> 4) + e;        // 16x16 result: row 4 (l >> 4) + e, column l & 15
            if (!col_ok || m >= p.M) continue;
            float v = acc[mb][e] + bv;
            if constexpr (EPI == MMSUM_EPI_GELU) v = gelu_fast_f(v);
            if (p.res != nullptr) v += to_f32(p.res[(long)m * p.ldres + col]);
            if constexpr (CF32) static_cast<float*>(p.out)[(long)m * p.ldo + col] = v;
            else static_cast<bf16_t*>(p.out)[(long)m * p.ldo + col] = (bf16_t)v;
        }
}

// How many ways the reduction is split: enough (column tiles x slices) workgroups to cover the CUs about twice, slices of at
// least 256 k (8 loads per lane and operand) that divide K into whole 32-deep steps.
inline int dec_gemm_splitk(int N, int K) {
    const int tiles = (N + 15) / 16;
    int sk = 1;
    while (tiles * sk < 512 && K % (sk * 2 * 256) == 0) sk *= 2;
    return sk;
}

template <int MB, int EPI>
int launch_dec_gemm(const DecGemmArgs& a, bool af32, bool cf32, hipStream_t s) {
    const dim3 grid(((a.N + 15) / 16) * a.sk), block(64);
    // 16 steps per batch where the slice is a multiple of 512 k and the x fragments fit the registers beside them (MB <= 2), else 8
    const bool ns16 = MB <= 2 && a.ksl % 512 == 0 && a.ns16_ok;
    if (af32) { if (ns16) dec_gemm_kernel<MB, EPI, true, true, 16><<<grid, block, 0, s>>>(a); else dec_gemm_kernel<MB, EPI, true, true, 8><<<grid, block, 0, s>>>(a); }
    else if (cf32) { if (ns16) dec_gemm_kernel<MB, EPI, false, true, 16><<<grid, block, 0, s>>>(a); else dec_gemm_kernel<MB, EPI, false, true, 8><<<grid, block, 0, s>>>(a); }
    else { if (ns16) dec_gemm_kernel<MB, EPI, false, false, 16><<<grid, block, 0, s>>>(a); else dec_gemm_kernel<MB, EPI, false, false, 8><<<grid, block, 0, s>>>(a); }
    return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
}

}  // namespace

// Workspace: [tickets: one word per column tile, a FIXED 16 KB region -- products of different N share the workspace one after the other,
// and the slabs of one must never land on the ticket words of another][slabs: tiles x sk x (MB x 4 x 64) floats]
constexpr long DEC_TICKET_BYTES = 16384;             // 4,096 column tiles: N <= 65,536

extern "C" long mmsum_dec_gemm_workspace(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const long tiles = (N + 15) / 16, mb = (M + 15) / 16;
    const long sk = dec_gemm_splitk(N, K);
    return DEC_TICKET_BYTES + tiles * sk * mb * 4 * 64 * (long)sizeof(float);
}

extern "C" int mmsum_dec_gemm(const void* x, long ldx, const void* x2, long ldx2, int ksplit, const void* W, long ldw, const float* bias,
                              const void* residual, long ldres, void* out, long ldo, int M, int N, int K, int flags, void* workspace,
                              void* stream) {
    if (M <= 0 || M > 96 || N <= 0 || K <= 0 || K % 256) return MMSUM_ERR_BAD_SHAPE;          // slices are whole batches of eight 32-deep steps
    const bool af32 = flags & MMSUM_GEMM_A_F32, cf32 = flags & MMSUM_GEMM_OUT_F32;
    const int epi = (flags >> 3) & 7;
    if (flags & ~(MMSUM_GEMM_A_F32 | MMSUM_GEMM_OUT_F32 | MMSUM_GEMM_EPI(7))) return MMSUM_ERR_BAD_SHAPE;
    if (!(epi == MMSUM_EPI_NONE || epi == MMSUM_EPI_GELU)) return MMSUM_ERR_BAD_SHAPE;
    if (af32 && (!cf32 || x2 != nullptr || epi != MMSUM_EPI_NONE)) return MMSUM_ERR_BAD_SHAPE;
    if (x2 != nullptr && (ksplit <= 0 || ksplit >= K || ksplit % 256)) return MMSUM_ERR_BAD_SHAPE;
    const long esx = af32 ? 4 : 2;
    if ((((uintptr_t)x | (uintptr_t)x2 | (uintptr_t)W) & 15) || ((ldx * esx) & 15) || ((ldx2 * esx) & 15) || ((ldw * 2) & 15)) return MMSUM_ERR_BAD_ALIGN;
    if (workspace == nullptr || ((uintptr_t)workspace & 15)) return MMSUM_ERR_WORKSPACE;
    const int tiles = (N + 15) / 16;
    if ((long)tiles * 4 > DEC_TICKET_BYTES) return MMSUM_ERR_BAD_SHAPE;
    DecGemmArgs a;
    a.x = x; a.x2 = x2; a.W = static_cast<const bf16_t*>(W); a.bias = bias; a.res = static_cast<const bf16_t*>(residual); a.out = out;
    a.tickets = static_cast<unsigned*>(workspace);
    a.slabs = reinterpret_cast<float*>(static_cast<char*>(workspace) + DEC_TICKET_BYTES);
    a.ldx = ldx; a.ldx2 = ldx2; a.ldw = ldw; a.ldres = ldres; a.ldo = ldo;
    a.M = M; a.N = N; a.K = K; a.ksplit = ksplit;
    a.sk = dec_gemm_splitk(N, K);
    a.ksl = K / a.sk;                        // a multiple of 256
    a.ns16_ok = (x2 == nullptr || ksplit % 512 == 0) ? 1 : 0;       // a batch of steps must lie in ONE of the two operand tensors
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int mb = (M + 15) / 16;
#define DEC_CASE(MBV)                                                                                              \
    if (mb == MBV) return epi == MMSUM_EPI_GELU ? launch_dec_gemm<MBV, MMSUM_EPI_GELU>(a, af32, cf32, s)           \
                                                : launch_dec_gemm<MBV, MMSUM_EPI_NONE>(a, af32, cf32, s);
    DEC_CASE(1) DEC_CASE(2) DEC_CASE(3) DEC_CASE(4) DEC_CASE(5) DEC_CASE(6)
#undef DEC_CASE
    return MMSUM_ERR_BAD_SHAPE;
}


// ---------------------------------------------------------------------------------------------
// mmsum_gemm_pair: two independent weight-streaming products of the same shape in ONE launch.
// ---------------------------------------------------------------------------------------------
namespace {
template <int MB>
int launch_skinny16_pair(const GemmArgs& a0, const GemmArgs& a1, hipStream_t stream) {
    const bool w8 = a0.K >= 2048 || a0.M > 32;
    const dim3 grid((a0.N + 15) / 16, 2);
    if (w8) {
        const size_t lds = (size_t)8 * MB * 4 * 64 * sizeof(float);
        gemm_skinny16_pair_kernel<MB, MMSUM_EPI_NONE, 8><<<grid, dim3(8 * 64), lds, stream>>>(a0, a1);
    } else {
        const size_t lds = (size_t)4 * MB * 4 * 64 * sizeof(float);
        gemm_skinny16_pair_kernel<MB, MMSUM_EPI_NONE, 4><<<grid, dim3(4 * 64), lds, stream>>>(a0, a1);
    }
    return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
}
}  // namespace

extern "C" int mmsum_gemm_pair(const mmsum_gemm_operands* ops, int M, int N, int K, int ksplit, void* stream) {
    if (!ops || M <= 0 || M > 64 || N < 16 || N > 4096 || K <= 0 || K % 256 || ksplit < 0 || ksplit >= K || ksplit % 32) return MMSUM_ERR_BAD_SHAPE;
    GemmArgs a[2];
    for (int i = 0; i < 2; ++i) {
        const mmsum_gemm_operands& o = ops[i];
        if ((ksplit > 0) != (o.A2 != nullptr)) return MMSUM_ERR_BAD_SHAPE;
        if ((((uintptr_t)o.A | (uintptr_t)o.A2 | (uintptr_t)o.B) & 15) || ((o.lda | o.lda2 | o.ldb) & 7)) return MMSUM_ERR_BAD_ALIGN;
        a[i] = GemmArgs{o.A, o.A2, o.B, o.C, o.bias, nullptr, M, N, K, o.lda, o.lda2, o.ldb, o.ldc, 0, ksplit, 1.f, o.bias ? MMSUM_GEMM_BIAS : 0, 1, nullptr, nullptr};
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (M <= 16) return launch_skinny16_pair<1>(a[0], a[1], s);
    if (M <= 32) return launch_skinny16_pair<2>(a[0], a[1], s);
    if (M <= 48) return launch_skinny16_pair<3>(a[0], a[1], s);
    return launch_skinny16_pair<4>(a[0], a[1], s);
}
