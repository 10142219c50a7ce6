// MFMA GEMM for gfx950:  C[m][n] = epi(alpha * sum_k A(m,k) * B(n,k) + bias[n])  (+ C)
//
// One kernel covers the three products of a linear layer by choosing how each operand is staged
// into the common k-slab LDS format (mmsum_device.h):
//   forward  y = x W^T        : A natural  (x  [M,K]),   B natural    (W [N,K])
//   dgrad    dx = dy W        : A natural  (dy [M,N']),  B transposed (W [N',K'] read as B(n=k',k=n'))
//   wgrad    dW = dy^T x      : A transposed (dy [M',N]), B transposed (x [M',K'])
// Block tile 128x128, 4 waves in a 2x2 grid, each wave 64x64 = 2x2 MFMA 32x32 tiles; K step =
// two 64-byte slabs (64 bf16 / 32 f32); register-staged double buffering (global loads for tile
// t+1 are issued before the MFMAs of tile t, the LDS writes after them); one barrier per K step.
// Workgroup ids are remapped so that consecutive tiles (which share an A panel) land on one XCD.
#include "gemm_common.h"

namespace {

constexpr int BM = 128, BN = 128, NSLAB = 2, THREADS = 256;
constexpr int TILE_BYTES = BM * NSLAB * SLAB_BYTES;  // 16 KiB per operand per stage

template <typename T> struct NatRegs { u32x4_t v[4]; };

template <typename T>
__device__ __forceinline__ void nat_load(NatRegs<T>& r, const T* __restrict__ g, long ld, int row_lo, int R, int k0, int K, int tid) {
    constexpr int CPR = NSLAB * 4;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int id = tid + it * THREADS;
        const int row = id / CPR, cc = id % CPR;
        const int grow = row_lo + row, gk = k0 + cc * ElemTraits<T>::kPerChunk;
        r.v[it] = u32x4_t{0, 0, 0, 0};
        if (grow < R && gk < K) r.v[it] = *reinterpret_cast<const u32x4_t*>(g + (long)grow * ld + gk);
    }
}
template <typename T>
__device__ __forceinline__ void nat_store(char* lds, const NatRegs<T>& r, int tid) {
    constexpr int CPR = NSLAB * 4;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int id = tid + it * THREADS;
        const int row = id / CPR, cc = id % CPR;
        *reinterpret_cast<u32x4_t*>(lds + (cc >> 2) * (BM * SLAB_BYTES) + slab_off(row, cc & 3)) = r.v[it];
    }
}

template <typename T, bool AT, bool BT, int EPI, int OUT>
__global__ __launch_bounds__(THREADS, 2) void gemm_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KC = ElemTraits<T>::kPerChunk;
    constexpr int BK = NSLAB * ElemTraits<T>::kPerSlab;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    int m_cap;                                  // the M the host sized the grid (and the split-K slabs) for
    apply_live_rows(p, m_cap);
    // ---- tile assignment (bijective XCD remap: blocks b and b+8 share an XCD) -------------
    const int tiles_m = (m_cap + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
    const int wg = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    const int tiles = tiles_m * tiles_n;
    const int ks = wg / tiles;                 // split-K slice
    if (p.flags & MMSUM_GEMM_SLABS) p.C = static_cast<float*>(p.C) + (long)ks * m_cap * p.ldc;
    const int t = wg % tiles;
    int tm, tn;
    tile_coords(t, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    if (m0 >= p.M) return;                      // tile of rows past the live count

    // ---- K range of this slice ---------------------------------------------------------------
    const int ktiles = (p.K + BK - 1) / BK;
    const int per = (ktiles + p.splitk - 1) / p.splitk;
    const int kt_beg = ks * per, kt_end = min(ktiles, kt_beg + per);

    const T* A = static_cast<const T*>(p.A);
    const T* A2 = static_cast<const T*>(p.A2);
    const T* B = static_cast<const T*>(p.B);

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = zero_acc();

    NatRegs<T> an, bn;
    TBlock<T> at, bt;
    const int t_rg = tid & 31, t_kg = tid >> 5;   // transposed staging: 32 row groups x 8 k groups

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        if constexpr (AT) {
            load_tblock<T>(at, A, p.lda, m0 + t_rg * 4, p.M, k0 + t_kg * KC, p.K);
        } else {
            if (A2 != nullptr && k0 >= p.ksplit) nat_load<T>(an, A2, p.lda2, m0, p.M, k0 - p.ksplit, p.K - p.ksplit, tid);
            else nat_load<T>(an, A, p.lda, m0, p.M, k0, (A2 != nullptr) ? p.ksplit : p.K, tid);
        }
        if constexpr (BT) load_tblock<T>(bt, B, p.ldb, n0 + t_rg * 4, p.N, k0 + t_kg * KC, p.K);
        else nat_load<T>(bn, B, p.ldb, n0, p.N, k0, p.K, tid);
    };
    auto store_tile = [&](int buf) {
        char* As = smem + buf * 2 * TILE_BYTES;
        char* Bs = As + TILE_BYTES;
        if constexpr (AT) store_tblock<T>(As, BM, at, t_rg * 4, t_kg, m0 + t_rg * 4, p.M);
        else nat_store<T>(As, an, tid);
        if constexpr (BT) store_tblock<T>(Bs, BN, bt, t_rg * 4, t_kg, n0 + t_rg * 4, p.N);
        else nat_store<T>(Bs, bn, tid);
    };

    if (kt_beg < kt_end) {
        load_tile(kt_beg);
        store_tile(0);
        __syncthreads();
        int cur = 0;
        for (int kt = kt_beg; kt < kt_end; ++kt) {
            const bool more = (kt + 1 < kt_end);
            if (more) load_tile(kt + 1);
            const char* As = smem + cur * 2 * TILE_BYTES;
            const char* Bs = As + TILE_BYTES;
#pragma unroll
            for (int s = 0; s < NSLAB; ++s) {
                Frag a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = lds_frag<T>(As + s * (BM * SLAB_BYTES), wm * 64 + i * 32, lane);
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = lds_frag<T>(Bs + s * (BN * SLAB_BYTES), wn * 64 + j * 32, lane);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) mma_slab<T>(acc[i][j], a[i], b[j]);
            }
            if (more) store_tile(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }

    gemm_epilogue<T, 2, 2, EPI, OUT>(p, acc, m0 + wm * 64, n0 + wn * 64, ks, lane);
}

template <typename T, bool AT, bool BT>
int launch_layout(const GemmArgs& a, hipStream_t stream) {
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    const dim3 grid(tiles * a.splitk), block(THREADS);
    const size_t lds = 4 * TILE_BYTES;
    const int epi = (a.flags >> 3) & 7, out = out_mode_of(a);
#define GEMM_CASE(E, O) if (epi == E && out == O) { gemm_kernel<T, AT, BT, E, O><<<grid, block, lds, stream>>>(a); return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP; }
    GEMM_CASE(MMSUM_EPI_NONE, OUT_T) GEMM_CASE(MMSUM_EPI_NONE, OUT_T_ACC) GEMM_CASE(MMSUM_EPI_NONE, OUT_F32_ACC)
    GEMM_CASE(MMSUM_EPI_NONE, OUT_F32_ATOMIC) GEMM_CASE(MMSUM_EPI_NONE, OUT_F32)
    GEMM_CASE(MMSUM_EPI_GELU, OUT_T) GEMM_CASE(MMSUM_EPI_GELU_BWD, OUT_T) GEMM_CASE(MMSUM_EPI_RELU, OUT_T) GEMM_CASE(MMSUM_EPI_RELU_BWD, OUT_T)
    GEMM_CASE(MMSUM_EPI_GELU, OUT_F32) GEMM_CASE(MMSUM_EPI_GELU_BWD, OUT_F32) GEMM_CASE(MMSUM_EPI_RELU, OUT_F32) GEMM_CASE(MMSUM_EPI_RELU_BWD, OUT_F32)
#undef GEMM_CASE
    return MMSUM_ERR_BAD_SHAPE;   // unsupported epilogue/output combination
}

template <typename T>
int launch_gemm(const GemmArgs& a, hipStream_t stream) {
    const bool at = a.flags & MMSUM_GEMM_A_T, bt = a.flags & MMSUM_GEMM_B_T;
    if (!at && !bt) return launch_layout<T, false, false>(a, stream);
    if (!at && bt) return launch_layout<T, false, true>(a, stream);
    if (at && bt) return launch_layout<T, true, true>(a, stream);
    return launch_layout<T, true, false>(a, stream);
}

}  // namespace

static int gemm_check(int dtype, const void* A, long lda, const void* A2, long lda2, int ksplit, const void* B, long ldb, const float* bias,
                      const void* C, long ldc, const void* aux, long ldaux, int M, int N, int K, int flags, int splitk, const int* live_rows) {
    if (M <= 0 || N <= 0 || K <= 0 || splitk < 1) return MMSUM_ERR_BAD_SHAPE;
    if (dtype != MMSUM_F32 && dtype != MMSUM_BF16) return MMSUM_ERR_BAD_DTYPE;
    const int kc = (dtype == MMSUM_BF16) ? 8 : 4;
    const int bk = (dtype == MMSUM_BF16) ? 64 : 32;
    const bool at = flags & MMSUM_GEMM_A_T, bt = flags & MMSUM_GEMM_B_T;
    // natural operands are read in 16-byte chunks along K: K must be a chunk multiple unless the row
    // is padded (ld >= K rounded up) -- the caller then guarantees the padding holds zeros.
    const long kround = ((long)K + kc - 1) / kc * kc;
    if (!at && (K % kc) && lda < kround) return MMSUM_ERR_BAD_SHAPE;
    if (!bt && (K % kc) && ldb < kround) return MMSUM_ERR_BAD_SHAPE;
    if (A2 && (at || ksplit % bk || ksplit <= 0 || ksplit >= K)) return MMSUM_ERR_BAD_SHAPE;
    if (splitk > 1 && !((flags & MMSUM_GEMM_OUT_F32) && ((flags & MMSUM_GEMM_ACCUM) || (flags & MMSUM_GEMM_SLABS)))) return MMSUM_ERR_BAD_SHAPE;
    if ((flags & MMSUM_GEMM_SLABS) && (flags & (MMSUM_GEMM_ACCUM | MMSUM_GEMM_BIAS))) return MMSUM_ERR_BAD_SHAPE;
    const size_t es = (dtype == MMSUM_BF16) ? 2 : 4;
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)A2) & 15) return MMSUM_ERR_BAD_ALIGN;
    if (flags & MMSUM_GEMM_A_F32) {        // f32 A beside bf16 weights: the weight-streaming kernel only (checked again by the caller)
        if (dtype != MMSUM_BF16) return MMSUM_ERR_BAD_DTYPE;
        if (at || A2 || (K % 8)) return MMSUM_ERR_BAD_SHAPE;
        if ((lda * 4) & 15) return MMSUM_ERR_BAD_ALIGN;
    } else
    if (!at && ((lda * es) & 15)) return MMSUM_ERR_BAD_ALIGN;
    if (!bt && ((ldb * es) & 15)) return MMSUM_ERR_BAD_ALIGN;
    if (at && ((lda * es) & (es == 2 ? 7 : 15))) return MMSUM_ERR_BAD_ALIGN;
    if (bt && ((ldb * es) & (es == 2 ? 7 : 15))) return MMSUM_ERR_BAD_ALIGN;
    if (live_rows && at && !bt) return MMSUM_ERR_BAD_SHAPE;     // a live row count needs a row-streamed operand
    if (flags & MMSUM_GEMM_COLSUM) {
        GemmArgs a{A, A2, B, const_cast<void*>(C), bias, const_cast<void*>(aux), M, N, K, lda, lda2, ldb, ldc, ldaux, ksplit, 1.f, flags, splitk, live_rows, nullptr};
        const int epi = (flags >> 3) & 7;
        if (at && bt) {                   // weight-gradient product: column sums of A (the bias gradient) inside the four-wave TN kernel
            if ((flags & (MMSUM_GEMM_BIAS | MMSUM_GEMM_ACCUM)) || epi != MMSUM_EPI_NONE || !gemm_tn_colsum_ok(dtype, a)) return MMSUM_ERR_BAD_SHAPE;
        } else if (!gemm_glds_eligible(dtype, a) || (flags & (MMSUM_GEMM_BIAS | MMSUM_GEMM_OUT_F32 | MMSUM_GEMM_SLABS | MMSUM_GEMM_ACCUM)) || splitk != 1 ||
                   !bias || !(epi == MMSUM_EPI_NONE || epi == MMSUM_EPI_GELU_BWD)) {
            return MMSUM_ERR_BAD_SHAPE;   // epilogue column sums exist on the LDS-DMA NT path with a bf16 result only
        }
    }
    return MMSUM_OK;
}

extern "C" int mmsum_gemm(int dtype, const void* A, long lda, const void* A2, long lda2, int ksplit, const void* B, long ldb,
                          void* C, long ldc, const float* bias, void* aux, long ldaux, int M, int N, int K, float alpha,
                          const float* alpha_dev, int flags, int splitk, const int* live_rows, void* workspace, long workspace_bytes,
                          void* stream) {
    const int rc = gemm_check(dtype, A, lda, A2, lda2, ksplit, B, ldb, bias, C, ldc, aux, ldaux, M, N, K, flags, splitk, live_rows);
    if (rc != MMSUM_OK) return rc;
    if (workspace != nullptr && ((((uintptr_t)workspace) & 15) || workspace_bytes < 0)) return MMSUM_ERR_WORKSPACE;
    GemmArgs a{A, A2, B, C, bias, aux, M, N, K, lda, lda2, ldb, ldc, ldaux, ksplit, alpha, flags, splitk, live_rows, alpha_dev};
    a.split_ws = static_cast<float*>(workspace);
    a.split_ws_bytes = workspace ? workspace_bytes : 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (gemm_skinny_eligible(dtype, a)) return launch_gemm_skinny(a, s);
    if (gemm_skinny_f32_eligible(dtype, a)) return launch_gemm_skinny_f32(a, s);
    if (flags & MMSUM_GEMM_A_F32) return MMSUM_ERR_BAD_DTYPE;          // no other kernel reads an f32 A beside bf16 weights
    if (gemm_glds_eligible(dtype, a)) return launch_gemm_glds(a, s);
    if (gemm_tn_eligible(dtype, a)) return launch_gemm_tn(a, s);
    return dtype == MMSUM_BF16 ? launch_gemm<bf16_t>(a, s) : launch_gemm<float>(a, s);
}

// 3x3 convolution, stride 1, padding 1, as an IMPLICIT GEMM on the LDS-DMA NT kernels: the operand rows are read from the activations
// in the zero-bordered padded NHWC layout (gemm_common.h: conv_row / conv_koff); no im2col matrix exists.
extern "C" int mmsum_conv3x3_gemm(const void* xp, const void* w, long ldw, void* y, long ldy, float* stats, int n, int H, int W, int C,
                                  int Cout, const int* live_rows, void* stream) {
    if (n <= 0 || H <= 0 || W <= 0 || Cout <= 0 || C < 64 || (C & (C - 1))) return MMSUM_ERR_BAD_SHAPE;      // C: a power of two >= 64 (stages of one tap)
    if ((long)n * (H + 2) * (W + 2) * C * 2 >= 0x7fffffffL || (long)n * H * W >= 0x7fffffffL) return MMSUM_ERR_BAD_SHAPE;   // 32-bit buffer offsets
    if ((((uintptr_t)xp | (uintptr_t)w | (uintptr_t)y) & 15) || ((ldw * 2) & 15) || ((ldy * 2) & 15)) return MMSUM_ERR_BAD_ALIGN;
    if (ldw < 9L * C) return MMSUM_ERR_BAD_SHAPE;
    GemmArgs a{xp, nullptr, w, y, stats, nullptr, n * H * W, Cout, 9 * C, (long)C, 0, ldw, ldy, 0, 0, 1.f,
               stats ? (MMSUM_GEMM_COLSUM | MMSUM_GEMM_COLSUM2) : 0, 1, live_rows, nullptr};
    a.conv_wp = W + 2; a.conv_w = W; a.conv_hw = H * W; a.conv_hpwp = (H + 2) * (W + 2);
    a.conv_cshift = 0;
    while ((1 << a.conv_cshift) < C) ++a.conv_cshift;
    if (!gemm_glds_eligible(MMSUM_BF16, a)) return MMSUM_ERR_BAD_SHAPE;
    return launch_gemm_glds(a, static_cast<hipStream_t>(stream));
}

// Weight gradient of the same convolution, again without an im2col matrix: dW[co][(ky, kx, c)] = sum over the pixels of
// dy[pixel][co] * x[pixel + (ky - 1, kx - 1)][c].  BOTH operands in the zero-bordered padded layout (dyp [n, H+2, W+2, Cout], xp [n, H+2, W+2, C]):
// the reduction then runs over ALL padded positions k (border positions of dyp are zero and add nothing), the x row of tap (ky, kx) is
// row k + (ky - 1) (W + 2) + (kx - 1) of the same image matrix, and a 256-column tile of the [Cout, 9 C] result is one tap's channels,
// i.e. the four-wave TN kernel with a per-tile row shift of its B operand.  The first and last W + 3 positions are border positions:
// the reduction skips them, so every shifted row exists.  out: f32 [splitk][Cout][ldo] slabs (splitk > 1) or f32 [Cout][ldo].
extern "C" int mmsum_conv3x3_wgrad(const void* dyp, const void* xp, float* out, long ldo, int n, int H, int W, int C, int Cout, int splitk,
                                   const int* live_positions, void* stream) {
    if (n <= 0 || H <= 0 || W <= 0 || Cout <= 0 || C < 256 || (C & (C - 1)) || (Cout & 7) || splitk < 1) return MMSUM_ERR_BAD_SHAPE;   // a tile = channels of one tap
    const long Kp = (long)n * (H + 2) * (W + 2), skip = W + 3;
    if (Kp * C * 2 >= 0x7fffffffL * 16 || Kp - 2 * skip <= 0) return MMSUM_ERR_BAD_SHAPE;
    if ((((uintptr_t)dyp | (uintptr_t)xp | (uintptr_t)out) & 15) || ((ldo * 4) & 15)) return MMSUM_ERR_BAD_ALIGN;
    if (ldo < 9L * C) return MMSUM_ERR_BAD_SHAPE;
    const bf16_t* A = static_cast<const bf16_t*>(dyp) + skip * Cout;
    const bf16_t* B = static_cast<const bf16_t*>(xp) + skip * C;
    GemmArgs a{A, nullptr, B, out, nullptr, nullptr, Cout, 9 * C, (int)(Kp - 2 * skip), (long)Cout, 0, (long)C, ldo, 0, 0, 1.f,
               MMSUM_GEMM_A_T | MMSUM_GEMM_B_T | MMSUM_GEMM_OUT_F32 | (splitk > 1 ? MMSUM_GEMM_SLABS : 0), splitk, live_positions, nullptr};
    a.conv_wp = W + 2;
    a.conv_cshift = 0;
    while ((1 << a.conv_cshift) < C) ++a.conv_cshift;
    return launch_gemm_tn_w4(a, static_cast<hipStream_t>(stream));
}

// Which kernel, tile and grid mmsum_gemm would launch for these arguments (no device work, no pointers dereferenced):
// plan[0] = MMSUM_PLAN_* kernel family, plan[1] x plan[2] = block tile, plan[3] = workgroups launched (< tiles * splitk
// means persistent workgroups walking the tile list).  Lets tests assert that a shape reaches the kernel they mean to cover.
extern "C" int mmsum_gemm_plan(int dtype, const void* A, long lda, const void* A2, long lda2, int ksplit, const void* B, long ldb,
                               const void* C, long ldc, const float* bias, const void* aux, long ldaux, int M, int N, int K, int flags,
                               int splitk, const int* live_rows, const float* alpha_dev, long workspace_bytes, int* plan) {
    const int rc = gemm_check(dtype, A, lda, A2, lda2, ksplit, B, ldb, bias, C, ldc, aux, ldaux, M, N, K, flags, splitk, live_rows);
    if (rc != MMSUM_OK) return rc;
    GemmArgs a{A, A2, B, const_cast<void*>(C), bias, const_cast<void*>(aux), M, N, K, lda, lda2, ldb, ldc, ldaux, ksplit, 1.f, flags, splitk, live_rows, alpha_dev};
    a.split_ws = workspace_bytes > 0 ? reinterpret_cast<float*>(16) : nullptr;         // never dereferenced here: "a workspace is lent"
    a.split_ws_bytes = workspace_bytes > 0 ? workspace_bytes : 0;
    GemmPlan g;
    if (gemm_skinny_eligible(dtype, a)) g = GemmPlan{MMSUM_PLAN_SKINNY, a.M, 32, (a.N + 31) / 32};
    else if (gemm_skinny_f32_eligible(dtype, a)) g = GemmPlan{MMSUM_PLAN_SKINNY, a.M, 16, (a.N + 15) / 16};
    else if (flags & MMSUM_GEMM_A_F32) return MMSUM_ERR_BAD_DTYPE;
    else if (gemm_glds_eligible(dtype, a)) g = plan_gemm_glds(a);
    else if (gemm_tn_eligible(dtype, a)) g = plan_gemm_tn(a);
    else g = GemmPlan{MMSUM_PLAN_GENERIC, BM, BN, ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN) * a.splitk};
    plan[0] = g.kernel; plan[1] = g.bm; plan[2] = g.bn; plan[3] = g.grid;
    return MMSUM_OK;
}

namespace {
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ ws, int nslabs, int rows, int cols, float* __restrict__ out,
                                                          long ldo, int accumulate) {
    const int cv = cols >> 2;
    const long total = (long)rows * cv;
    const long slab = (long)rows * cols;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / cv), c = (int)(i % cv) * 4;
        // four slabs in flight per thread (the slab count is a run-time value: without this the loop is one dependent load at a time);
        // the order of the additions is fixed, so the result does not depend on the launch geometry
        const float* w0 = ws + (long)r * cols + c;
        f32x4_t s = *reinterpret_cast<const f32x4_t*>(w0);
        int k = 1;
        for (; k + 3 < nslabs; k += 4) {
            const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(w0 + k * slab), a1 = *reinterpret_cast<const f32x4_t*>(w0 + (k + 1) * slab);
            const f32x4_t a2 = *reinterpret_cast<const f32x4_t*>(w0 + (k + 2) * slab), a3 = *reinterpret_cast<const f32x4_t*>(w0 + (k + 3) * slab);
            s = (((s + a0) + a1) + a2) + a3;
        }
        for (; k < nslabs; ++k) s = s + *reinterpret_cast<const f32x4_t*>(w0 + k * slab);
        float* o = out + (long)r * ldo + c;
        if (accumulate) s = s + *reinterpret_cast<const f32x4_t*>(o);
        *reinterpret_cast<f32x4_t*>(o) = s;
    }
}
}  // namespace

extern "C" int mmsum_slab_reduce(const float* ws, int nslabs, int rows, int cols, float* out, long ldo, int accumulate, void* stream) {
    if (nslabs < 1 || rows <= 0 || cols <= 0 || (cols & 3) || (ldo & 3)) return MMSUM_ERR_BAD_SHAPE;
    if ((((uintptr_t)ws) | ((uintptr_t)out)) & 15) return MMSUM_ERR_BAD_ALIGN;
    const long total = (long)rows * (cols >> 2);
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    slab_reduce_kernel<<<dim3((int)blocks), dim3(256), 0, (hipStream_t)stream>>>(ws, nslabs, rows, cols, out, ldo, accumulate);
    return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
}
