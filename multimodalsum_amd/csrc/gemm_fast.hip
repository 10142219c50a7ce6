// bf16 "NT" GEMM with direct global->LDS DMA (global_load_lds_dwordx4) for gfx950.
//
//   C[m][n] = epi(alpha * sum_k A[m][k] * B[n][k] + bias[n]) (+C),   A [M,K], B [N,K] both K-contiguous.
//
// Every product of the training step is routed here in bf16 mode: forward (x, W), dgrad (dy, W^T
// from the transposed weight shadow) and wgrad (dy^T, x^T from the activation transposer).
//
// Tile BM x BN x 64, WAVES_M x WAVES_N waves, each wave (BM/WAVES_M) x (BN/WAVES_N) as MFMA 32x32x16
// tiles.  Operands live in LDS in the k-slab format of mmsum_device.h; the DMA writes LDS linearly
// (wave-uniform base + lane*16 B), so the XOR swizzle is applied to the per-lane SOURCE address and
// again on the fragment read (guide rule 21: linear destination + swizzled source + swizzled read).
// Two LDS stages: the DMA of K-tile t+1 is in flight while the MFMAs of tile t run; one
// vmcnt(0)+barrier per K-tile.  Rows past M/N are clamped (their results are never stored).
#include "gemm_common.h"
#include <stdlib.h>

namespace {

template <int BM, int BN, int WAVES_M, int WAVES_N>
struct FastCfg {
    static constexpr int NW = WAVES_M * WAVES_N;
    static constexpr int THREADS = NW * 64;
    static constexpr int TM = BM / WAVES_M / 32, TN = BN / WAVES_N / 32;
    static constexpr int A_BYTES = BM * 2 * SLAB_BYTES, B_BYTES = BN * 2 * SLAB_BYTES;
    static constexpr int STAGE = A_BYTES + B_BYTES;
    static constexpr int PA = BM / 16 * 2, PB = BN / 16 * 2;            // 1-KiB DMA pieces per operand tile
    static constexpr int PPW = (PA + PB) / NW;                          // pieces per wave per K-tile
    static_assert(PA % NW == 0 && PB % NW == 0, "pieces must split evenly over the waves");
};

__device__ __forceinline__ void dma16(const bf16_t* gsrc, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// One 1-KiB piece = 16 rows x 64 B of one slab.  `pidx` (wave-uniform) indexes pieces of a ROWS-row
// tile: slab = pidx / (ROWS/16), row block = pidx % (ROWS/16).
template <int ROWS>
__device__ __forceinline__ void dma_piece(char* tile, const bf16_t* __restrict__ g, long ld, int row0, int R, int k0, int pidx, int lane) {
    const int slab = pidx / (ROWS / 16), rb = pidx % (ROWS / 16);
    const int row = rb * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((row >> 2) & 3);               // logical chunk that belongs at this physical slot
    int grow = row0 + row;
    grow = grow < R ? grow : R - 1;
    dma16(g + (long)grow * ld + k0 + slab * 32 + c * 8, tile + slab * (ROWS * SLAB_BYTES) + rb * 1024);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, int OUT>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_nt_glds_kernel(GemmArgs p) {
    using Cfg = FastCfg<BM, BN, WAVES_M, WAVES_N>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles = tiles_m * tiles_n;
    const int ks = wg / tiles;
    const int t = wg % tiles;
    int tm, tn;
    tile_coords(t, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    const int ktiles = p.K / 64;
    const int per = (ktiles + p.splitk - 1) / p.splitk;
    const int kt_beg = ks * per, kt_end = min(ktiles, kt_beg + per);

    const bf16_t* A = static_cast<const bf16_t*>(p.A);
    const bf16_t* A2 = static_cast<const bf16_t*>(p.A2);
    const bf16_t* B = static_cast<const bf16_t*>(p.B);

    f32x16_t acc[Cfg::TM][Cfg::TN];
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j) acc[i][j] = zero_acc();

    auto stage = [&](int buf, int kt) {
        char* As = smem + buf * Cfg::STAGE;
        char* Bs = As + Cfg::A_BYTES;
        int k0 = kt * 64;
        const bf16_t* Ab = A;
        long lda = p.lda;
        if (A2 != nullptr && k0 >= p.ksplit) { Ab = A2; lda = p.lda2; k0 -= p.ksplit; }
        const int kb = kt * 64;
#pragma unroll
        for (int i = 0; i < Cfg::PPW; ++i) {
            if (i * Cfg::NW < Cfg::PA) dma_piece<BM>(As, Ab, lda, m0, p.M, k0, i * Cfg::NW + wave, lane);
            else dma_piece<BN>(Bs, B, p.ldb, n0, p.N, kb, i * Cfg::NW - Cfg::PA + wave, lane);
        }
    };

    if (kt_beg < kt_end) {
        stage(0, kt_beg);
        __syncthreads();
        int cur = 0;
        for (int kt = kt_beg; kt < kt_end; ++kt) {
            if (kt + 1 < kt_end) stage(cur ^ 1, kt + 1);
            const char* As = smem + cur * Cfg::STAGE;
            const char* Bs = As + Cfg::A_BYTES;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                Frag b[Cfg::TN];
#pragma unroll
                for (int j = 0; j < Cfg::TN; ++j) b[j] = lds_frag<bf16_t>(Bs + s * (BN * SLAB_BYTES), wn * (Cfg::TN * 32) + j * 32, lane);
#pragma unroll
                for (int i = 0; i < Cfg::TM; ++i) {
                    const Frag a = lds_frag<bf16_t>(As + s * (BM * SLAB_BYTES), wm * (Cfg::TM * 32) + i * 32, lane);
#pragma unroll
                    for (int j = 0; j < Cfg::TN; ++j) mma_slab<bf16_t>(acc[i][j], a, b[j]);
                }
            }
            __syncthreads();      // drains the DMA of tile kt+1 (vmcnt(0)) and fences the LDS reads of tile kt
            cur ^= 1;
        }
    }
    gemm_epilogue<bf16_t, Cfg::TM, Cfg::TN, EPI, OUT>(p, acc, m0 + wm * (Cfg::TM * 32), n0 + wn * (Cfg::TN * 32), ks, lane);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, int OUT>
int launch_one(const GemmArgs& a, hipStream_t stream) {
    using Cfg = FastCfg<BM, BN, WAVES_M, WAVES_N>;
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    const size_t lds = 2 * Cfg::STAGE;
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_glds_kernel<BM, BN, WAVES_M, WAVES_N, EPI, OUT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        once = true;
    }
    gemm_nt_glds_kernel<BM, BN, WAVES_M, WAVES_N, EPI, OUT><<<dim3(tiles * a.splitk), dim3(Cfg::THREADS), lds, stream>>>(a);
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

inline double tile_score(int M, int N, int splitk, int bm, int bn, double eff) {
    const long tiles = (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn) * splitk;
    const long rounds = (tiles + 255) / 256;
    const double fill = (double)tiles / (double)(rounds * 256);
    const double waste = ((double)M * N) / ((double)((M + bm - 1) / bm * bm) * ((N + bn - 1) / bn * bn));
    return fill * waste * eff;
}

}  // namespace

bool gemm_glds_eligible(int dtype, const GemmArgs& a) {
    if (dtype != MMSUM_BF16) return false;
    if (a.flags & (MMSUM_GEMM_A_T | MMSUM_GEMM_B_T)) return false;
    if (a.K % 64) return false;
    if (a.A2 && (a.ksplit % 64)) return false;
    const int epi = (a.flags >> 3) & 7, out = out_mode_of(a);
    if (epi != MMSUM_EPI_NONE && out != OUT_T) return false;     // rare combinations stay on the generic kernel
    return true;
}

int launch_gemm_glds(const GemmArgs& a, hipStream_t stream) {
    // pick the tile shape that keeps the 256 CUs busiest for this problem
    // relative main-loop efficiency of the tile shapes: operand bytes per FLOP halve from 128^2 to
    // 256^2 and the L2 -> LDS DMA rate, not the MFMA rate, bounds the small tiles
    static const double e128 = getenv("MMSUM_E128") ? atof(getenv("MMSUM_E128")) : 0.50;
    static const double e2x1 = getenv("MMSUM_E2X1") ? atof(getenv("MMSUM_E2X1")) : 0.72;
    const double s256 = tile_score(a.M, a.N, a.splitk, 256, 256, 1.00);
    const double s128 = tile_score(a.M, a.N, a.splitk, 128, 128, e128);
    const double s2x1 = tile_score(a.M, a.N, a.splitk, 256, 128, e2x1);
    if (s256 >= s128 && s256 >= s2x1) return launch_cfg<256, 256, 2, 4>(a, stream);
    if (s2x1 >= s128) return launch_cfg<256, 128, 4, 2>(a, stream);
    return launch_cfg<128, 128, 2, 2>(a, stream);
}
