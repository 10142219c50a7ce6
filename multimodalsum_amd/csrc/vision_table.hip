// ResNet101 helpers (NHWC im2col / col2im, BatchNorm statistics and apply, max-pool, weight
// layout permutes) and the table-encoder gather.  All HBM-bound; vector accesses along the
// channel / feature axis, which is the contiguous one in every layout used here.
#include "mmsum_device.h"
#include "mmsum_kernels.h"

namespace {

template <typename T> __device__ __forceinline__ f32x4_t ld4(const T* p);
template <> __device__ __forceinline__ f32x4_t ld4<float>(const float* p) { return *reinterpret_cast<const f32x4_t*>(p); }
template <> __device__ __forceinline__ f32x4_t ld4<bf16_t>(const bf16_t* p) {
    const bf16x4_t v = *reinterpret_cast<const bf16x4_t*>(p);
    return f32x4_t{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <typename T> __device__ __forceinline__ void st4(T* p, f32x4_t v);
template <> __device__ __forceinline__ void st4<float>(float* p, f32x4_t v) { *reinterpret_cast<f32x4_t*>(p) = v; }
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, f32x4_t v) {
    *reinterpret_cast<bf16x4_t*>(p) = bf16x4_t{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
}

inline int grid_for(long items, int per_block, int cap = 4096) {
    long b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return (int)(b > cap ? cap : b);
}
inline int ok() { return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP; }

// ---- live-image window ------------------------------------------------------------------------------
// The image branch of the fused step runs on the images that need running only (mmsum_image_plan): the filled slots first, then ONE
// representative of the empty (masked, all-zero) slots -- identical inputs give identical activations in every layer, so the
// representative stands for all `mult` of them with a multiplicity in the BatchNorm sums and in the weight gradients.  The window is
// device-resident (one captured graph serves every batch): images[0] = images to process, images[1] = index of the representative
// among them (-1: none), images[2] = its multiplicity.  NULL = every image, no representative.
__device__ __forceinline__ int img_count(const int* __restrict__ images, int N) {
    return images != nullptr ? min(N, max(0, images[0])) : N;
}
struct ImgWin { int rows, rep0; float mult; };
__device__ __forceinline__ ImgWin img_window(const int* __restrict__ images, int rpi, int R) {
    ImgWin w{R, R, 1.f};
    if (images != nullptr) {
        w.rows = min(R, max(0, images[0]) * rpi);
        if (images[1] >= 0) { w.rep0 = images[1] * rpi; w.mult = (float)images[2]; }
    }
    return w;
}

// ---- im2col / col2im ---------------------------------------------------------------------------
// A tap (kh, kw) of an output pixel is C contiguous elements on both sides: copied as raw 16-byte vectors (EV elements).
template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const T* __restrict__ x, T* __restrict__ col, int N, int H, int W, int C,
                                                     int KH, int KW, int stride, int pad, int Ho, int Wo, int Kpad,
                                                     const int* __restrict__ images) {
    constexpr int EV = 16 / sizeof(T);
    N = img_count(images, N);
    const int cv = C / EV;
    const long total = (long)N * Ho * Wo * KH * KW * cv;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv) * EV;
        long t = i / cv;
        const int kw = (int)(t % KW); t /= KW;
        const int kh = (int)(t % KH); t /= KH;
        const long row = t;
        const int wo = (int)(row % Wo);
        const int ho = (int)((row / Wo) % Ho);
        const int n = (int)(row / ((long)Wo * Ho));
        const int h = ho * stride - pad + kh, w = wo * stride - pad + kw;
        const bool in = h >= 0 && h < H && w >= 0 && w < W;
        u32x4_t v = u32x4_t{0, 0, 0, 0};
        if (in) v = *reinterpret_cast<const u32x4_t*>(x + (((long)n * H + h) * W + w) * C + c);
        *reinterpret_cast<u32x4_t*>(col + row * Kpad + (kh * KW + kw) * C + c) = v;
    }
    const int K = KH * KW * C;
    if (Kpad > K) {
        const int tail = Kpad - K;
        const long tt = (long)N * Ho * Wo * tail;
        for (long i = blockIdx.x * 256L + threadIdx.x; i < tt; i += (long)gridDim.x * 256)
            col[(i / tail) * Kpad + K + (i % tail)] = from_f32<T>(0.f);
    }
}
// Few channels (the 7x7 stem, C = 3): a thread gathers EV consecutive columns of a row (several taps) and stores them as
// one 16-byte vector, zero padding of the K tail included.
template <typename T>
__global__ __launch_bounds__(256) void im2col_few_channels_kernel(const T* __restrict__ x, T* __restrict__ col, int N, int H, int W, int C,
                                                                  int KH, int KW, int stride, int pad, int Ho, int Wo, int Kpad,
                                                                  const int* __restrict__ images) {
    constexpr int EV = 16 / sizeof(T);
    N = img_count(images, N);
    const int kv = Kpad / EV, K = KH * KW * C;
    const long total = (long)N * Ho * Wo * kv;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k0 = (int)(i % kv) * EV;
        const long row = i / kv;
        const int wo = (int)(row % Wo);
        const int ho = (int)((row / Wo) % Ho);
        const int n = (int)(row / ((long)Wo * Ho));
        T v[EV];
#pragma unroll
        for (int e = 0; e < EV; ++e) {
            const int k = k0 + e, tap = k / C, c = k - tap * C;
            const int kh = tap / KW, kw = tap - kh * KW;
            const int h = ho * stride - pad + kh, w = wo * stride - pad + kw;
            const bool in = k < K && h >= 0 && h < H && w >= 0 && w < W;
            v[e] = in ? x[(((long)n * H + h) * W + w) * C + c] : from_f32<T>(0.f);
        }
        u32x4_t pk;
        __builtin_memcpy(&pk, v, 16);
        *reinterpret_cast<u32x4_t*>(col + row * Kpad + k0) = pk;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void col2im_kernel(const T* __restrict__ dcol, T* __restrict__ dx, int N, int H, int W, int C,
                                                     int KH, int KW, int stride, int pad, int Ho, int Wo, int Kpad,
                                                     const int* __restrict__ images) {
    constexpr int EV = 16 / sizeof(T);
    N = img_count(images, N);
    const int cv = C / EV;
    const long total = (long)N * H * W * cv;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv) * EV;
        long t = i / cv;
        const int w = (int)(t % W); t /= W;
        const int h = (int)(t % H);
        const int n = (int)(t / H);
        float acc[EV];
#pragma unroll
        for (int e = 0; e < EV; ++e) acc[e] = 0.f;
        for (int kh = 0; kh < KH; ++kh) {
            const int hh = h + pad - kh;
            if (hh < 0 || hh % stride) continue;
            const int ho = hh / stride;
            if (ho >= Ho) continue;
            for (int kw = 0; kw < KW; ++kw) {
                const int ww = w + pad - kw;
                if (ww < 0 || ww % stride) continue;
                const int wo = ww / stride;
                if (wo >= Wo) continue;
                const T* src = dcol + (((long)n * Ho + ho) * Wo + wo) * Kpad + (kh * KW + kw) * C + c;
                const u32x4_t raw = *reinterpret_cast<const u32x4_t*>(src);
                T v[EV];
                __builtin_memcpy(v, &raw, 16);
#pragma unroll
                for (int e = 0; e < EV; ++e) acc[e] += to_f32(v[e]);
            }
        }
        T o[EV];
#pragma unroll
        for (int e = 0; e < EV; ++e) o[e] = from_f32<T>(acc[e]);
        u32x4_t pk;
        __builtin_memcpy(&pk, o, 16);
        *reinterpret_cast<u32x4_t*>(dx + (((long)n * H + h) * W + w) * C + c) = pk;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void weight_to_matrix_kernel(T* __restrict__ mat, const float* __restrict__ w, int Cout, int Cin,
                                                               int KH, int KW, int Kpad) {
    const long total = (long)Cout * Kpad;
    const int K = KH * KW * Cin;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int col = (int)(i % Kpad), co = (int)(i / Kpad);
        float v = 0.f;
        if (col < K) {
            const int c = col % Cin, kk = col / Cin;        // kk = kh*KW + kw
            v = w[((long)co * Cin + c) * (KH * KW) + kk];
        }
        mat[i] = from_f32<T>(v);
    }
}
// The input-gradient operand of a stride-1 convolution: mat[ci][((KH-1-kh) KW + (KW-1-kw)) Cout + co] = w[co][ci][kh][kw] -- the weights
// rotated by 180 degrees with the channel roles exchanged, so that dx = conv(dy, mat) with the forward's own kernel (mmsum_conv3x3_gemm).
template <typename T>
__global__ __launch_bounds__(256) void weight_to_dgrad_matrix_kernel(T* __restrict__ mat, const float* __restrict__ w, int Cout, int Cin,
                                                                     int KH, int KW, int Kpad) {
    const long total = (long)Cin * Kpad;
    const int K = KH * KW * Cout;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int col = (int)(i % Kpad), ci = (int)(i / Kpad);
        float v = 0.f;
        if (col < K) {
            const int co = col % Cout, tap = col / Cout;
            const int kh = KH - 1 - tap / KW, kw = KW - 1 - tap % KW;
            v = w[(((long)co * Cin + ci) * KH + kh) * KW + kw];
        }
        mat[i] = from_f32<T>(v);
    }
}
__global__ __launch_bounds__(256) void matrix_to_weight_grad_kernel(const float* __restrict__ mat, float* __restrict__ dw, int Cout,
                                                                    int Cin, int KH, int KW, int Kpad, int accumulate) {
    const long total = (long)Cout * Cin * KH * KW;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int kk = (int)(i % (KH * KW));
        const int c = (int)((i / (KH * KW)) % Cin);
        const int co = (int)(i / ((long)KH * KW * Cin));
        const float v = mat[(long)co * Kpad + kk * Cin + c];
        dw[i] = accumulate ? dw[i] + v : v;
    }
}

// ---- BatchNorm ----------------------------------------------------------------------------------
// All four kernels give a thread a FIXED group of 8 (bf16) / 4 (f32) consecutive channels and let it walk rows: the
// per-channel constants are loop invariant registers, every access is one 16-byte vector, and several independent rows are
// in flight per thread.  block = CG channel groups x (256 / CG) row lanes, CG = min(C / VEC, 32): a wave reads whole
// 128..512-byte row pieces.  C % 64 == 0 in every ResNet layer (64 .. 1024); other C take VEC = 4 / scalar-free paths below.
constexpr int BN_BLOCKS = 2048;          // workgroups a statistics pass aims for (channel blocks x row splits)
template <typename T> struct BnVec { static constexpr int N = 16 / sizeof(T); };
template <typename T, int N> __device__ __forceinline__ void ldv(const T* p, float (&v)[N]) {
    if constexpr (sizeof(T) == 2) {
        const bf16x8_t t = *reinterpret_cast<const bf16x8_t*>(p);
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = (float)t[j];
    } else {
        const f32x4_t t = *reinterpret_cast<const f32x4_t*>(p);
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = t[j];
    }
}
template <typename T, int N> __device__ __forceinline__ void stv(T* p, const float (&v)[N]) {
    if constexpr (sizeof(T) == 2) {
        bf16x8_t t;
#pragma unroll
        for (int j = 0; j < N; ++j) t[j] = (bf16_t)v[j];
        *reinterpret_cast<bf16x8_t*>(p) = t;
    } else {
        *reinterpret_cast<f32x4_t*>(p) = f32x4_t{v[0], v[1], v[2], v[3]};
    }
}
// Row of pixel r = (n, y, x) of an [n, pH, pW] image in the zero-bordered PADDED layout [n, pH + 2, pW + 2] (the operand layout of the
// implicit 3x3 convolution, mmsum_conv3x3_gemm); pW == 0: the compact layout, row r itself.
// floor(a / b) for 0 <= a < 2^24 and b > 0 without the integer-division sequence (~40 instructions per call; the padded-layout kernels
// call this once per 16-byte access): a float quotient, corrected by at most one.
__device__ __forceinline__ int fast_div(int a, int b, float inv_b) {
    int q = (int)((float)a * inv_b);
    const int r = a - q * b;
    q += (r >= b) - (r < 0);
    return q;
}
__device__ __forceinline__ long bn_yrow(int r, int pH, int pW) {
    if (pW == 0) return r;
    const int hw = pH * pW;
    int n, yy;
    if (r < (1 << 24)) {
        n = fast_div(r, hw, __frcp_rn((float)hw));
        const int rem = r - n * hw;
        yy = fast_div(rem, pW, __frcp_rn((float)pW));
    } else {
        n = r / hw;
        yy = (r - n * hw) / pW;
    }
    const int xx = r - n * hw - yy * pW;
    return (long)n * (pH + 2) * (pW + 2) + (long)(yy + 1) * (pW + 2) + xx + 1;
}
// thread -> (channel group, row lane) of a block that spans `cgb` channel groups
struct BnMap { int cg, rl, lanes; };
__device__ __forceinline__ BnMap bn_map(int cgb) { return BnMap{(int)threadIdx.x % cgb, (int)threadIdx.x / cgb, 256 / cgb}; }
// {sum x, sum x^2} over R rows -> {mean, biased variance}, and the running-statistics update: ONE spelling with explicit fused operations,
// shared by every kernel that does either (the apply kernel's in-launch form must equal mmsum_bn_stats_from_sums to the last bit; left to
// the compiler's contraction, the same source expression came out differently in different kernels)
__device__ __forceinline__ void bn_mean_var(float sum, float sumsq, int R, float& mean, float& var) {
    const float invR = 1.f / (float)R;
    mean = sum * invR;
    var = fmaxf(fmaf(-mean, mean, sumsq * invR), 0.f);
}
__device__ __forceinline__ void bn_running_update(float& rm, float& rv, float mean, float var, int R, float momentum) {
    const float unbiased = R > 1 ? var * ((float)R / (float)(R - 1)) : var;
    rm = fmaf(momentum, mean, (1.f - momentum) * rm);
    rv = fmaf(momentum, unbiased, (1.f - momentum) * rv);
}
inline int bn_cgb(int C, int vec) { const int g = C / vec; return g >= 32 ? 32 : (g >= 16 ? 16 : (g >= 8 ? 8 : (g >= 4 ? 4 : (g >= 2 ? 2 : 1)))); }

// MODE 0: sums of (x - pivot, (x - pivot)^2), pivot = row 0 (var = E[(x-p)^2] - E[x-p]^2 does not cancel catastrophically when
// |mean| >> std).  MODE 1: sums of (dy', dy' * (x - mean)) for the backward pass (dy' = dy where the ReLU passed).
// grid (channel blocks, row splits); part[split][2][C].
template <typename T, int MODE, bool RELU = false>
__global__ __launch_bounds__(256) void bn_partial_kernel(const T* __restrict__ a, const T* __restrict__ y, const T* __restrict__ x,
                                                         const float* __restrict__ sums, int R, int C, int cgb, int relu,
                                                         float* __restrict__ part, int pH = 0, int pW = 0,
                                                         const int* __restrict__ images = nullptr, int rpi = 0) {
    constexpr int V = BnVec<T>::N;
    __shared__ float red[2][256 * V];
    const BnMap m = bn_map(cgb);
    const int col = (blockIdx.x * cgb + m.cg) * V;
    const int rows_per = (R + gridDim.y - 1) / gridDim.y;
    // live-image window: rows past it are not read; MODE 0 counts the representative's rows `mult` times (MODE 1 needs no weights: the
    // representative's gradient rows arrive already multiplied, see bn_bwd_apply_kernel)
    const ImgWin win = img_window(images, rpi, R);
    const int r0 = blockIdx.y * rows_per, r1 = min(win.rows, r0 + rows_per);
    float s0[V], s1[V], ref[V];
#pragma unroll
    for (int j = 0; j < V; ++j) s0[j] = s1[j] = ref[j] = 0.f;
    const bool live = col < C;
    if (live) {
        if (MODE == 0) ldv<T, V>(a + col, ref);
        else {
#pragma unroll
            for (int j = 0; j < V; ++j) ref[j] = sums[col + j];
        }
        constexpr int UN = 4;
        int r = r0 + m.rl;
        for (; r + (UN - 1) * m.lanes < r1; r += UN * m.lanes) {
            float av[UN][V], yv[UN][V], xv[UN][V];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const long o = (long)(r + u * m.lanes) * C + col;
                ldv<T, V>(a + o, av[u]);
                if (MODE == 1) {
                    ldv<T, V>(x + o, xv[u]);
                    if constexpr (RELU) ldv<T, V>(y + bn_yrow(r + u * m.lanes, pH, pW) * C + col, yv[u]);     // (a run-time test here serialises the rows' loads)
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u)
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    if (MODE == 0) {
                        const float d = av[u][j] - ref[j];
                        const float wd = (r + u * m.lanes >= win.rep0) ? win.mult * d : d;
                        s0[j] += wd;
                        s1[j] = fmaf(wd, d, s1[j]);
                    } else {
                        const float g = (RELU && !(yv[u][j] > 0.f)) ? 0.f : av[u][j];
                        s0[j] += g;
                        s1[j] = fmaf(g, xv[u][j] - ref[j], s1[j]);
                    }
                }
        }
        for (; r < r1; r += m.lanes) {
            float av[V], yv[V], xv[V];
            const long o = (long)r * C + col;
            ldv<T, V>(a + o, av);
            if (MODE == 1) {
                ldv<T, V>(x + o, xv);
                if constexpr (RELU) ldv<T, V>(y + bn_yrow(r, pH, pW) * C + col, yv);
            }
#pragma unroll
            for (int j = 0; j < V; ++j) {
                if (MODE == 0) {
                    const float d = av[j] - ref[j];
                    const float wd = (r >= win.rep0) ? win.mult * d : d;
                    s0[j] += wd;
                    s1[j] = fmaf(wd, d, s1[j]);
                } else {
                    const float g = (RELU && !(yv[j] > 0.f)) ? 0.f : av[j];
                    s0[j] += g;
                    s1[j] = fmaf(g, xv[j] - ref[j], s1[j]);
                }
            }
        }
    }
    // fold the row lanes: red[w][rl][cg][j]
#pragma unroll
    for (int j = 0; j < V; ++j) {
        red[0][(m.rl * cgb + m.cg) * V + j] = s0[j];
        red[1][(m.rl * cgb + m.cg) * V + j] = s1[j];
    }
    __syncthreads();
    const int ncol = cgb * V;                               // channels of this block
    for (int i = threadIdx.x; i < 2 * ncol; i += 256) {
        const int w = i / ncol, c = i % ncol;
        if (blockIdx.x * ncol + c < C) {
            float t = 0.f;
            for (int k = 0; k < m.lanes; ++k) t += red[w][k * ncol + c];
            part[(long)blockIdx.y * 2 * C + w * C + blockIdx.x * ncol + c] = t;
        }
    }
}
// Finishing passes: block = 16 columns x 16 split lanes (the partials of a column are 2 C floats apart: 16 independent
// loads per thread and round trip).
// backward: dsums = {sum dy', sum dy' * xhat}: the second partial carries (x - mean), rstd is applied here
__global__ __launch_bounds__(256) void bn_finish_kernel(const float* __restrict__ part, int splits, int C, const float* __restrict__ sums,
                                                        float eps, float* __restrict__ out) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + cl;                   // over 2 C
    float s = 0.f;
    if (col < 2 * C) {
#pragma unroll 8
        for (int k = sl; k < splits; k += 16) s += part[(long)k * 2 * C + col];
    }
    red[sl][cl] = s;
    __syncthreads();
    if (sl == 0 && col < 2 * C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cl];
        out[col] = col < C ? t : t * rsqrtf(sums[col] + eps);        // sums[C + c] = variance of channel c
    }
}

// forward statistics: shifted partial sums -> {mean, biased variance}
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_finish_kernel(const float* __restrict__ part, int splits, int C, int R, const T* __restrict__ x,
                                                              float* __restrict__ out) {
    __shared__ float red[2][16][17];
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + cl;
    float s0 = 0.f, s1 = 0.f;
    if (col < C) {
#pragma unroll 8
        for (int k = sl; k < splits; k += 16) { s0 += part[(long)k * 2 * C + col]; s1 += part[(long)k * 2 * C + C + col]; }
    }
    red[0][sl][cl] = s0; red[1][sl][cl] = s1;
    __syncthreads();
    if (sl == 0 && col < C) {
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { t0 += red[0][k][cl]; t1 += red[1][k][cl]; }
        const float m = t0 / R;
        out[col] = to_f32(x[col]) + m;
        out[C + col] = fmaxf(t1 / R - m * m, 0.f);
    }
}

// y = relu?((x - mean) rstd gamma + beta (+ residual)) = x * sc + sh (+ residual); grid (channel blocks, row splits)
// raw != nullptr (training): the statistics arrive as plain column sums {sum x, sum x^2} left by the convolution's GEMM epilogue
// (MMSUM_GEMM_COLSUM | COLSUM2); every thread turns the sums of its channels into {mean, biased variance} itself, and the first row
// split's first row lane also WRITES them to `sums` (the backward pass reads them there) and updates the running statistics: nobody
// reads those three arrays in this launch, so there is no race and no separate statistics launch.
// RES (a residual is added) is a template parameter: as a run-time test around the residual's load it made every row of the
// "four rows in flight" loop its own load -> wait round (the compiler waits at the join of a branch that holds a load): 3.3 TB/s
template <typename T, bool RES>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, float* __restrict__ sums, const float* __restrict__ raw,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const T* __restrict__ residual, T* __restrict__ y,
                                                       float* __restrict__ running_mean, float* __restrict__ running_var, int R, int C,
                                                       int cgb, float eps, float momentum, int relu, int training, int pH, int pW,
                                                       const int* __restrict__ images, int rpi) {
    constexpr int V = BnVec<T>::N;
    const BnMap m = bn_map(cgb);
    const int col = (blockIdx.x * cgb + m.cg) * V;
    if (col >= C) return;
    const int Rl = img_window(images, rpi, R).rows;          // rows of the images that run; the statistics keep all R rows as their count
    // per-channel constants of this thread's V channels: every array as 16-byte vector loads issued together (element by element inside
    // the raw / training branches, the compiler had made V dependent load -> wait -> store rounds of this prologue)
    float sc[V], sh[V], mean[V], var[V], ga[V], be[V];
    auto ldf = [](const float* p, float (&v)[V]) {
#pragma unroll
        for (int q4 = 0; q4 < V / 4; ++q4) {
            const f32x4_t t = *reinterpret_cast<const f32x4_t*>(p + 4 * q4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * q4 + e] = t[e];
        }
    };
    auto stf = [](float* p, const float (&v)[V]) {
#pragma unroll
        for (int q4 = 0; q4 < V / 4; ++q4) *reinterpret_cast<f32x4_t*>(p + 4 * q4) = f32x4_t{v[4 * q4], v[4 * q4 + 1], v[4 * q4 + 2], v[4 * q4 + 3]};
    };
    ldf(gamma + col, ga);
    ldf(beta + col, be);
    if (raw != nullptr) {
        ldf(raw + col, mean);
        ldf(raw + C + col, var);
#pragma unroll
        for (int j = 0; j < V; ++j) bn_mean_var(mean[j], var[j], R, mean[j], var[j]);
        if (blockIdx.y == 0 && m.rl == 0) {
            stf(sums + col, mean);
            stf(sums + C + col, var);
            if (running_mean != nullptr && running_var != nullptr) {
                float rm[V], rv[V];
                ldf(running_mean + col, rm);
                ldf(running_var + col, rv);
#pragma unroll
                for (int j = 0; j < V; ++j) bn_running_update(rm[j], rv[j], mean[j], var[j], R, momentum);
                stf(running_mean + col, rm);
                stf(running_var + col, rv);
            }
        }
    } else {
        ldf((training ? sums : running_mean) + col, mean);
        ldf((training ? sums + C : running_var) + col, var);
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
        sc[j] = rsqrtf(var[j] + eps) * ga[j];
        sh[j] = be[j] - mean[j] * sc[j];
    }
    const int step = m.lanes * gridDim.y;
    constexpr int UN = 4;
    int r = blockIdx.y * m.lanes + m.rl;
    for (; r + (UN - 1) * step < Rl; r += UN * step) {
        float xv[UN][V], rv[UN][V];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long o = (long)(r + u * step) * C + col;
            ldv<T, V>(x + o, xv[u]);
            if constexpr (RES) ldv<T, V>(residual + o, rv[u]);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float o = fmaf(xv[u][j], sc[j], sh[j]);
                if constexpr (RES) o += rv[u][j];
                xv[u][j] = relu ? fmaxf(o, 0.f) : o;
            }
            stv<T, V>(y + bn_yrow(r + u * step, pH, pW) * C + col, xv[u]);
        }
    }
    for (; r < Rl; r += step) {
        float xv[V], rv[V];
        const long o = (long)r * C + col;
        ldv<T, V>(x + o, xv);
        if constexpr (RES) ldv<T, V>(residual + o, rv);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float t = fmaf(xv[j], sc[j], sh[j]);
            if constexpr (RES) t += rv[j];
            xv[j] = relu ? fmaxf(t, 0.f) : t;
        }
        stv<T, V>(y + bn_yrow(r, pH, pW) * C + col, xv);
    }
}
// running stats update, separate launch so the apply kernel never races with it
__global__ void bn_running_kernel(const float* __restrict__ sums, float* __restrict__ running_mean, float* __restrict__ running_var,
                                  int R, int C, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float rm = running_mean[c], rv = running_var[c];
    bn_running_update(rm, rv, sums[c], sums[C + c], R, momentum);
    running_mean[c] = rm;
    running_var[c] = rv;
}

// Statistics that arrive as plain column sums (the convolution's GEMM epilogue leaves raw = {sum x, sum x^2} of the values it
// stored: MMSUM_GEMM_COLSUM | MMSUM_GEMM_COLSUM2): -> {mean, biased variance} and the running-statistics update in one launch.
__global__ void bn_stats_from_sums_kernel(const float* __restrict__ raw, int R, int C, float* __restrict__ sums, float* __restrict__ running_mean,
                                          float* __restrict__ running_var, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float mean, var;
    bn_mean_var(raw[c], raw[C + c], R, mean, var);
    sums[c] = mean;
    sums[C + c] = var;
    if (running_mean != nullptr && running_var != nullptr) {
        float rm = running_mean[c], rv = running_var[c];
        bn_running_update(rm, rv, mean, var, R, momentum);
        running_mean[c] = rm;
        running_var[c] = rv;
    }
}

// dx = gamma rstd (g - sum(g)/R - xhat sum(g xhat)/R) = k g + kx x + k0 with per-channel constants; dresidual = g
template <typename T, bool RELU, bool DRES>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ x,
                                                           const float* __restrict__ sums, const float* __restrict__ dsums,
                                                           const float* __restrict__ gamma, T* __restrict__ dx, T* __restrict__ dresidual,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int R, int C, int cgb,
                                                           float eps, int relu, int pH, int pW, int dxH, int dxW,
                                                           const int* __restrict__ images, int rpi) {
    constexpr int V = BnVec<T>::N;
    const BnMap m = bn_map(cgb);
    const int col = (blockIdx.x * cgb + m.cg) * V;
    // live-image window.  The representative of the empty slots carries its gradient rows MULTIPLIED by its multiplicity through the whole
    // backward pass (every consumer is linear in them: the weight gradients then hold the sum over all `mult` identical images, the
    // statistics sums too); what this kernel adds to a row -- the two batch-mean terms -- is therefore multiplied as well.
    const ImgWin win = img_window(images, rpi, R);
    if (col < C) {
        const float invR = 1.f / R;
        float k[V], kx[V], k0[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float mean = sums[col + j], rstd = rsqrtf(sums[C + col + j] + eps);
            k[j] = gamma[col + j] * rstd;
            kx[j] = -k[j] * rstd * dsums[C + col + j] * invR;
            k0[j] = -k[j] * dsums[col + j] * invR - kx[j] * mean;
        }
        const int step = m.lanes * gridDim.y;
        constexpr int UN = 2;
        int r = blockIdx.y * m.lanes + m.rl;
        for (; r + (UN - 1) * step < win.rows; r += UN * step) {
            float g[UN][V], yv[UN][V], xv[UN][V];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const long o = (long)(r + u * step) * C + col;
                ldv<T, V>(dy + o, g[u]);
                ldv<T, V>(x + o, xv[u]);
                if constexpr (RELU) ldv<T, V>(y + bn_yrow(r + u * step, pH, pW) * C + col, yv[u]);
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const long o = (long)(r + u * step) * C + col;
                const float wm = (r + u * step >= win.rep0) ? win.mult : 1.f;
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    if constexpr (RELU) if (!(yv[u][j] > 0.f)) g[u][j] = 0.f;
                    xv[u][j] = fmaf(k[j], g[u][j], wm * fmaf(kx[j], xv[u][j], k0[j]));
                }
                stv<T, V>(dx + bn_yrow(r + u * step, dxH, dxW) * C + col, xv[u]);
                if constexpr (DRES) stv<T, V>(dresidual + o, g[u]);
            }
        }
        for (; r < win.rows; r += step) {
            float g[V], yv[V], xv[V];
            const long o = (long)r * C + col;
            ldv<T, V>(dy + o, g);
            ldv<T, V>(x + o, xv);
            if constexpr (RELU) ldv<T, V>(y + bn_yrow(r, pH, pW) * C + col, yv);
            const float wm = (r >= win.rep0) ? win.mult : 1.f;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                if constexpr (RELU) if (!(yv[j] > 0.f)) g[j] = 0.f;
                xv[j] = fmaf(k[j], g[j], wm * fmaf(kx[j], xv[j], k0[j]));
            }
            stv<T, V>(dx + bn_yrow(r, dxH, dxW) * C + col, xv);
            if constexpr (DRES) stv<T, V>(dresidual + o, g);
        }
    }
    // parameter gradients: dbeta = sum dy', dgamma = sum dy'*xhat (accumulate into the f32 arena)
    if (blockIdx.x == 0 && blockIdx.y == 0 && dgamma != nullptr)
        for (int c = threadIdx.x; c < C; c += 256) { dbeta[c] += dsums[c]; dgamma[c] += dsums[C + c]; }
}

// The representative's share of the statistics that arrive as plain column sums from the convolution's GEMM epilogue: the epilogue counted its
// rows once, raw += (mult - 1) * {sum y, sum y^2} over them (of the values as stored, like the epilogue's own sums).  grid (channel blocks,
// row splits); returns at once when the batch has no representative.
template <typename T>
__global__ __launch_bounds__(256) void bn_rep_fix_kernel(const T* __restrict__ y, float* __restrict__ raw, int R, int C, int cgb,
                                                         const int* __restrict__ images, int rpi) {
    constexpr int V = BnVec<T>::N;
    __shared__ float red[2][256 * V];
    const ImgWin win = img_window(images, rpi, R);
    if (win.rep0 >= win.rows || !(win.mult > 1.f)) return;
    const BnMap m = bn_map(cgb);
    const int col = (blockIdx.x * cgb + m.cg) * V;
    const int r1 = min(win.rows, win.rep0 + rpi);
    float s0[V], s1[V];
#pragma unroll
    for (int j = 0; j < V; ++j) s0[j] = s1[j] = 0.f;
    if (col < C) {
        for (int r = win.rep0 + blockIdx.y * m.lanes + m.rl; r < r1; r += m.lanes * gridDim.y) {
            float v[V];
            ldv<T, V>(y + (long)r * C + col, v);
#pragma unroll
            for (int j = 0; j < V; ++j) { s0[j] += v[j]; s1[j] = fmaf(v[j], v[j], s1[j]); }
        }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
        red[0][(m.rl * cgb + m.cg) * V + j] = s0[j];
        red[1][(m.rl * cgb + m.cg) * V + j] = s1[j];
    }
    __syncthreads();
    const int ncol = cgb * V;
    const float extra = win.mult - 1.f;
    for (int i = threadIdx.x; i < 2 * ncol; i += 256) {
        const int w = i / ncol, c = i % ncol;
        if (blockIdx.x * ncol + c < C) {
            float t = 0.f;
            for (int k = 0; k < m.lanes; ++k) t += red[w][k * ncol + c];
            atomicAdd(raw + (long)w * C + blockIdx.x * ncol + c, extra * t);
        }
    }
}

// ---- image plan: which slots of the batch the image branch runs (mmsum_image_plan) ------------------------------------------------
// empty[i] = 1 when slot i is masked (mask[i] == 0) AND every element of its image is zero: only then are its activations those of
// every other empty slot and its output gradient zero (a masked key).  One workgroup per slot; unmasked slots are not read.
__global__ __launch_bounds__(256) void image_empty_kernel(const float* __restrict__ img, long elems, const uint8_t* __restrict__ mask, int n,
                                                          int* __restrict__ empty) {
    const int i = blockIdx.x;
    if (mask[i] != 0) {
        if (threadIdx.x == 0) empty[i] = 0;
        return;
    }
    const float* p = img + (long)i * elems;
    int nz = 0;
    if ((elems & 3) == 0 && (((uintptr_t)p) & 15) == 0) {
        const f32x4_t* q = reinterpret_cast<const f32x4_t*>(p);
        for (long k = threadIdx.x; k < elems / 4; k += 256) {
            const f32x4_t v = q[k];
            nz |= (v[0] != 0.f) | (v[1] != 0.f) | (v[2] != 0.f) | (v[3] != 0.f);
        }
    } else {
        for (long k = threadIdx.x; k < elems; k += 256) nz |= (p[k] != 0.f);
    }
    nz = __syncthreads_or(nz);
    if (threadIdx.x == 0) empty[i] = nz ? 0 : 1;
}
struct ImgRpi { int v[8], adj[8]; };
constexpr int IMG_PLAN_MAX = 8192;
// One workgroup.  Run order = the non-empty slots in batch order, then the first empty slot as the representative of all `mult` empty ones.
//   plan      = {images that run, index of the representative among them (-1: none), mult, non-empty slots, (images that run) * rpi[k] ...}
//   src[r]    = slot whose image runs as image r (0 past the images that run)
//   slot_rows = row of the run-order matrix [n * positions, .] that slot row (slot, p) takes its result from
//   run_rows  = slot row whose output gradient run-order row (r, p) receives; -1 (zeros) for the representative and past the images that run
__global__ __launch_bounds__(256) void image_plan_kernel(const int* __restrict__ empty, int n, int positions, ImgRpi rpi, int n_rpi,
                                                         int* __restrict__ plan, int* __restrict__ src, int64_t* __restrict__ slot_rows,
                                                         int64_t* __restrict__ run_rows) {
    __shared__ int run_of[IMG_PLAN_MAX];
    __shared__ int s_live, s_mult, s_first;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        int live = 0, mult = 0, first = -1;
        for (int base = 0; base < n; base += 64) {
            const int i = base + lane;
            const bool in = i < n;
            const bool e = in && empty[i] != 0;
            const unsigned long long lv = __ballot(in && !e), em = __ballot(e);
            if (in && !e) {
                const int r = live + __popcll(lv & ((1ull << lane) - 1ull));
                run_of[i] = r;
                src[r] = i;
            }
            if (first < 0 && em != 0) first = base + __ffsll((long long)em) - 1;
            live += __popcll(lv);
            mult += __popcll(em);
        }
        if (lane == 0) { s_live = live; s_mult = mult; s_first = first; }
    }
    __syncthreads();
    const int live = s_live, mult = s_mult;
    const int rep = mult > 0 ? live : -1, nrun = live + (mult > 0 ? 1 : 0);
    for (int r = nrun + threadIdx.x; r < n; r += 256) src[r] = 0;
    if (threadIdx.x == 0) {
        if (rep >= 0) src[rep] = s_first;
        plan[0] = nrun; plan[1] = rep; plan[2] = mult > 0 ? mult : 1; plan[3] = live;
        for (int k = 0; k < n_rpi; ++k) plan[4 + k] = max(0, nrun * rpi.v[k] + rpi.adj[k]);
    }
    __syncthreads();
    const long total = (long)n * positions;
    for (long e = threadIdx.x; e < total; e += 256) {
        const int a = (int)(e / positions), p = (int)(e - (long)a * positions);
        const int run = empty[a] != 0 ? rep : run_of[a];
        slot_rows[e] = (long)run * positions + p;
        run_rows[e] = a < live ? (long)src[a] * positions + p : -1;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int Ho, int Wo,
                                                      const int* __restrict__ images) {
    N = img_count(images, N);
    const int cv = C / 4;
    const long total = (long)N * Ho * Wo * cv;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv) * 4;
        long t = i / cv;
        const int wo = (int)(t % Wo); t /= Wo;
        const int ho = (int)(t % Ho);
        const int n = (int)(t / Ho);
        f32x4_t m = f32x4_t{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int kh = 0; kh < 3; ++kh) {
            const int h = ho * 2 - 1 + kh;
            if (h < 0 || h >= H) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int w = wo * 2 - 1 + kw;
                if (w < 0 || w >= W) continue;
                const f32x4_t v = ld4<T>(x + (((long)n * H + h) * W + w) * C + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) m[j] = fmaxf(m[j], v[j]);
            }
        }
        st4<T>(y + i * 4, m);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int N, int C, int H, int W,
                                                           const int* __restrict__ images, const int* __restrict__ src) {
    N = img_count(images, N);
    const long total = (long)N * C * H * W;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long t = i / C;
        const int w = (int)(t % W); t /= W;
        const int h = (int)(t % H);
        const int n = (int)(t / H);
        const int ns = src != nullptr ? src[n] : n;                // image of the caller's batch that runs as image n
        y[i] = from_f32<T>(x[(((long)ns * C + c) * H + h) * W + w]);
    }
}

// ---- table encoder gather --------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void table_gather_kernel(const T* __restrict__ E, const int64_t* __restrict__ field,
                                                           const int64_t* __restrict__ name, const int64_t* __restrict__ category,
                                                           const int64_t* __restrict__ str_cat, const int64_t* __restrict__ str_bool,
                                                           const int64_t* __restrict__ rating, const int64_t* __restrict__ hours,
                                                           const T* __restrict__ w_rating, const T* __restrict__ w_hours,
                                                           T* __restrict__ out, uint8_t* __restrict__ mask, int D, int pad_id) {
    const int f = blockIdx.x, b = blockIdx.y;
    T* orow = out + ((long)b * 47 + f) * 2 * D;
    // mask (table_encoder.py:75-82)
    if (threadIdx.x == 0) {
        uint8_t m = 1;
        if (f == 1) m = category[(long)b * 72] != pad_id;
        else if (f >= 2 && f <= 6) m = str_cat[((long)b * 5 + (f - 2)) * 3] != pad_id;
        else if (f >= 7 && f <= 38) m = str_bool[(long)b * 32 + (f - 7)] != pad_id;
        else if (f >= 40) {
            long s = 0;
            for (int k = 0; k < 4; ++k) s += hours[((long)b * 7 + (f - 40)) * 4 + k];
            m = s != 0;
        }
        mask[(long)b * 47 + f] = m;
    }
    for (int dv = threadIdx.x * 4; dv < D; dv += 256 * 4) {
        // field-name half: masked sum of the 6 name tokens (table_encoder.py:28-31)
        f32x4_t nm = f32x4_t{0, 0, 0, 0};
        for (int j = 0; j < 6; ++j) {
            const long id = field[f * 6 + j];
            if (id != pad_id) nm = nm + ld4<T>(E + id * D + dv);
        }
        st4<T>(orow + dv, nm);
        f32x4_t v = f32x4_t{0, 0, 0, 0};
        if (f == 0) {
            for (int j = 0; j < 24; ++j) {
                const long id = name[(long)b * 24 + j];
                if (id != pad_id) v = v + ld4<T>(E + id * D + dv);
            }
        } else if (f == 1) {
            float nvalid = 0.f;
            for (int r = 0; r < 6; ++r) {
                bool any = false;
                for (int j = 0; j < 12; ++j) {
                    const long id = category[((long)b * 6 + r) * 12 + j];
                    if (id != pad_id) { v = v + ld4<T>(E + id * D + dv); any = true; }
                }
                nvalid += any ? 1.f : 0.f;
            }
            const float inv = 1.f / (nvalid + 1e-6f);
            v = v * inv;
        } else if (f <= 6) {
            for (int j = 0; j < 3; ++j) {
                const long id = str_cat[((long)b * 5 + (f - 2)) * 3 + j];
                if (id != pad_id) v = v + ld4<T>(E + id * D + dv);
            }
        } else if (f <= 38) {
            const long id = str_bool[(long)b * 32 + (f - 7)];
            if (id != pad_id) v = ld4<T>(E + id * D + dv);
        } else {
            const int64_t* bits = (f == 39) ? rating + (long)b * 4 : hours + ((long)b * 7 + (f - 40)) * 4;
            const T* w = (f == 39) ? w_rating : w_hours;
            for (int k = 0; k < 4; ++k) {
                const float x = (float)bits[k];
                if (x != 0.f)
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += x * to_f32(w[(long)(dv + j) * 4 + k]);
            }
        }
        st4<T>(orow + D + dv, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void table_gather_bwd_kernel(const T* __restrict__ dall, const int64_t* __restrict__ rating,
                                                               const int64_t* __restrict__ hours, float* __restrict__ dw_rating,
                                                               float* __restrict__ dw_hours, int B, int D) {
    const int dcol = blockIdx.x * 256 + threadIdx.x;
    if (dcol >= D) return;
    float gr[4] = {0, 0, 0, 0}, gh[4] = {0, 0, 0, 0};
    for (int b = 0; b < B; ++b) {
        const float g = to_f32(dall[((long)b * 47 + 39) * 2 * D + D + dcol]);
        for (int k = 0; k < 4; ++k) gr[k] += (float)rating[(long)b * 4 + k] * g;
        for (int j = 0; j < 7; ++j) {
            const float gj = to_f32(dall[((long)b * 47 + 40 + j) * 2 * D + D + dcol]);
            for (int k = 0; k < 4; ++k) gh[k] += (float)hours[((long)b * 7 + j) * 4 + k] * gj;
        }
    }
    for (int k = 0; k < 4; ++k) { dw_rating[(long)dcol * 4 + k] += gr[k]; dw_hours[(long)dcol * 4 + k] += gh[k]; }
}

// ---- Amazon table encoder gather (table_encoder.py:86-167): 133 positions = price, rating, brand, name, category, 128 description tokens
template <typename T>
__global__ __launch_bounds__(256) void amazon_gather_kernel(const T* __restrict__ E, const int64_t* __restrict__ field,
                                                            const int64_t* __restrict__ price, const int64_t* __restrict__ rating,
                                                            const int64_t* __restrict__ brand, const int64_t* __restrict__ name,
                                                            const int64_t* __restrict__ category, const int64_t* __restrict__ description,
                                                            const T* __restrict__ w_price, const T* __restrict__ w_rating,
                                                            T* __restrict__ out, uint8_t* __restrict__ mask, int D, int pad_id) {
    const int f = blockIdx.x, b = blockIdx.y;
    T* orow = out + ((long)b * 133 + f) * 2 * D;
    if (threadIdx.x == 0) {                                   // masks (:160-166)
        uint8_t m = 1;
        if (f == 0) {
            long s = 0;
            for (int k = 0; k < 11; ++k) s += price[(long)b * 11 + k];
            m = s != 0;
        } else if (f == 2) m = brand[(long)b * 12] != pad_id;
        else if (f == 3) m = name[(long)b * 32] != pad_id;
        else if (f >= 5) m = description[(long)b * 128 + (f - 5)] != pad_id;
        mask[(long)b * 133 + f] = m;
    }
    const long fid = field[f < 5 ? f : 5];                    // single-token field names, the last one repeated (:109-111)
    for (int dv = threadIdx.x * 4; dv < D; dv += 256 * 4) {
        st4<T>(orow + dv, ld4<T>(E + fid * D + dv));
        f32x4_t v = f32x4_t{0, 0, 0, 0};
        if (f <= 1) {                                         // price / rating: Linear(11 | 4 -> D, no bias) (:116-117)
            const int nb = f == 0 ? 11 : 4;
            const int64_t* bits = f == 0 ? price + (long)b * 11 : rating + (long)b * 4;
            const T* w = f == 0 ? w_price : w_rating;
            for (int k = 0; k < nb; ++k) {
                const float x = (float)bits[k];
                if (x != 0.f)
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += x * to_f32(w[(long)(dv + j) * nb + k]);
            }
        } else if (f == 2 || f == 3) {                        // brand / name: masked token sum (:120-129)
            const int nt = f == 2 ? 12 : 32;
            const int64_t* ids = f == 2 ? brand + (long)b * 12 : name + (long)b * 32;
            for (int j = 0; j < nt; ++j)
                if (ids[j] != pad_id) v = v + ld4<T>(E + ids[j] * D + dv);
        } else if (f == 4) {                                  // category [3][8][12]: token sum, mean over valid rows, mean over valid groups (:132-145)
            float ngroups = 0.f;
            for (int g = 0; g < 3; ++g) {
                f32x4_t gv = f32x4_t{0, 0, 0, 0};
                float nrows = 0.f;
                for (int r = 0; r < 8; ++r) {
                    bool any = false;
                    for (int j = 0; j < 12; ++j) {
                        const long id = category[(((long)b * 3 + g) * 8 + r) * 12 + j];
                        if (id != pad_id) { gv = gv + ld4<T>(E + id * D + dv); any = true; }
                    }
                    nrows += any ? 1.f : 0.f;
                }
                if (nrows > 0.f) {
                    v = v + gv * (1.f / (nrows + 1e-6f));
                    ngroups += 1.f;
                }
            }
            v = v * (1.f / (ngroups + 1e-6f));
        } else {                                              // description tokens, NOT masked here (:148-150)
            v = ld4<T>(E + description[(long)b * 128 + (f - 5)] * D + dv);
        }
        st4<T>(orow + D + dv, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void amazon_gather_bwd_kernel(const T* __restrict__ dall, const int64_t* __restrict__ price,
                                                                const int64_t* __restrict__ rating, float* __restrict__ dw_price,
                                                                float* __restrict__ dw_rating, int B, int D) {
    const int dcol = blockIdx.x * 256 + threadIdx.x;
    if (dcol >= D) return;
    float gp[11], gr[4];
    for (int k = 0; k < 11; ++k) gp[k] = 0.f;
    for (int k = 0; k < 4; ++k) gr[k] = 0.f;
    for (int b = 0; b < B; ++b) {
        const float g0 = to_f32(dall[((long)b * 133 + 0) * 2 * D + D + dcol]);
        const float g1 = to_f32(dall[((long)b * 133 + 1) * 2 * D + D + dcol]);
        for (int k = 0; k < 11; ++k) gp[k] += (float)price[(long)b * 11 + k] * g0;
        for (int k = 0; k < 4; ++k) gr[k] += (float)rating[(long)b * 4 + k] * g1;
    }
    for (int k = 0; k < 11; ++k) dw_price[(long)dcol * 11 + k] += gp[k];
    for (int k = 0; k < 4; ++k) dw_rating[(long)dcol * 4 + k] += gr[k];
}

}  // namespace

#define DT_SWITCH(dtype, CALL_BF16, CALL_F32)               \
    do {                                                    \
        if ((dtype) == MMSUM_BF16) { CALL_BF16; }           \
        else if ((dtype) == MMSUM_F32) { CALL_F32; }        \
        else return MMSUM_ERR_BAD_DTYPE;                    \
    } while (0)

extern "C" int mmsum_im2col(int dtype, const void* x, void* col, int N, int H, int W, int C, int KH, int KW, int stride, int pad,
                            int Ho, int Wo, int Kpad, const int* images, void* stream) {
    const int ev = dtype == MMSUM_BF16 ? 8 : 4;
    if (N <= 0 || Kpad < KH * KW * C || Kpad % ev) return MMSUM_ERR_BAD_SHAPE;
    if (((uintptr_t)x | (uintptr_t)col) & 15) return MMSUM_ERR_BAD_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const dim3 block(256);
    if (C % ev == 0) {
        const dim3 grid(grid_for((long)N * Ho * Wo * KH * KW * (C / ev), 256, 16384));
        DT_SWITCH(dtype, (im2col_kernel<bf16_t><<<grid, block, 0, s>>>((const bf16_t*)x, (bf16_t*)col, N, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad, images)),
                  (im2col_kernel<float><<<grid, block, 0, s>>>((const float*)x, (float*)col, N, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad, images)));
    } else {
        const dim3 grid(grid_for((long)N * Ho * Wo * (Kpad / ev), 256, 16384));
        DT_SWITCH(dtype, (im2col_few_channels_kernel<bf16_t><<<grid, block, 0, s>>>((const bf16_t*)x, (bf16_t*)col, N, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad, images)),
                  (im2col_few_channels_kernel<float><<<grid, block, 0, s>>>((const float*)x, (float*)col, N, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad, images)));
    }
    return ok();
}

extern "C" int mmsum_col2im(int dtype, const void* dcol, void* dx, int N, int H, int W, int C, int KH, int KW, int stride, int pad,
                            int Ho, int Wo, int Kpad, const int* images, void* stream) {
    const int ev = dtype == MMSUM_BF16 ? 8 : 4;
    if (N <= 0 || C % ev || Kpad % ev) return MMSUM_ERR_BAD_SHAPE;
    if (((uintptr_t)dcol | (uintptr_t)dx) & 15) return MMSUM_ERR_BAD_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(grid_for((long)N * H * W * C / ev, 256, 16384)), block(256);
    DT_SWITCH(dtype, (col2im_kernel<bf16_t><<<grid, block, 0, s>>>((const bf16_t*)dcol, (bf16_t*)dx, N, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad, images)),
              (col2im_kernel<float><<<grid, block, 0, s>>>((const float*)dcol, (float*)dx, N, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad, images)));
    return ok();
}

extern "C" int mmsum_conv_weight_permute(int dtype, void* matrix, float* weight, int Cout, int Cin, int KH, int KW, int Kpad,
                                         int to_matrix, int accumulate, void* stream) {
    if (Cout <= 0 || Kpad < (to_matrix == 2 ? Cout : Cin) * KH * KW) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    if (to_matrix == 2) {
        const dim3 grid(grid_for((long)Cin * Kpad, 256)), block(256);
        DT_SWITCH(dtype, (weight_to_dgrad_matrix_kernel<bf16_t><<<grid, block, 0, s>>>((bf16_t*)matrix, weight, Cout, Cin, KH, KW, Kpad)),
                  (weight_to_dgrad_matrix_kernel<float><<<grid, block, 0, s>>>((float*)matrix, weight, Cout, Cin, KH, KW, Kpad)));
    } else if (to_matrix) {
        const dim3 grid(grid_for((long)Cout * Kpad, 256)), block(256);
        DT_SWITCH(dtype, (weight_to_matrix_kernel<bf16_t><<<grid, block, 0, s>>>((bf16_t*)matrix, weight, Cout, Cin, KH, KW, Kpad)),
                  (weight_to_matrix_kernel<float><<<grid, block, 0, s>>>((float*)matrix, weight, Cout, Cin, KH, KW, Kpad)));
    } else {
        const dim3 grid(grid_for((long)Cout * Cin * KH * KW, 256)), block(256);
        matrix_to_weight_grad_kernel<<<grid, block, 0, s>>>((const float*)matrix, weight, Cout, Cin, KH, KW, Kpad, accumulate);
    }
    return ok();
}

// splits * C <= BN_BLOCKS * (channels of one block = 32 groups x 8): partial sums of both statistics
extern "C" long mmsum_bn_workspace(int C) { return (long)2 * sizeof(float) * ((long)BN_BLOCKS * 256 + 2L * C); }

// grid.y of the streaming kernels: enough row blocks to fill the chip (about 8 workgroups per CU) while a thread still
// walks >= 16 rows (its per-channel constants cost ~40 instructions to set up)
inline int bn_row_blocks(int R, int lanes, int col_blocks) {
    int want = (BN_BLOCKS + col_blocks - 1) / col_blocks;
    const int most = (R + lanes * 16 - 1) / (lanes * 16);
    if (want > most) want = most;
    return want < 1 ? 1 : want;
}
// row splits of the statistics passes: BN_BLOCKS workgroups in all, at least 4 rows per row lane
inline int bn_splits(int R, int lanes, int col_blocks) {
    int want = (BN_BLOCKS / 2 + col_blocks - 1) / col_blocks;
    if (want > 512) want = 512;                             // the finishing pass walks the splits
    const int most = (R + lanes * 4 - 1) / (lanes * 4);
    if (want > most) want = most;
    return want < 1 ? 1 : want;
}

// images / rows_per_image of the BatchNorm entry points: the live-image window (NULL: every row, no representative)
static bool img_args_ok(const int* images, int rpi, int R) { return images == nullptr || (rpi > 0 && R % rpi == 0); }

extern "C" int mmsum_bn_reduce(int dtype, const void* x, int R, int C, float* sums, void* workspace, const int* images, int rows_per_image,
                               void* stream) {
    const int vec = dtype == MMSUM_BF16 ? 8 : 4;
    if (R <= 0 || C <= 0 || C % vec || !img_args_ok(images, rows_per_image, R)) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int cgb = bn_cgb(C, vec), cblocks = (C / vec + cgb - 1) / cgb;
    const int splits = bn_splits(R, 256 / cgb, cblocks);
    const dim3 grid(cblocks, splits), block(256);
    float* part = (float*)workspace;
    DT_SWITCH(dtype, (bn_partial_kernel<bf16_t, 0><<<grid, block, 0, s>>>((const bf16_t*)x, nullptr, nullptr, nullptr, R, C, cgb, 0, part, 0, 0, images, rows_per_image)),
              (bn_partial_kernel<float, 0><<<grid, block, 0, s>>>((const float*)x, nullptr, nullptr, nullptr, R, C, cgb, 0, part, 0, 0, images, rows_per_image)));
    DT_SWITCH(dtype, (bn_stats_finish_kernel<bf16_t><<<dim3((C + 15) / 16), dim3(256), 0, s>>>(part, splits, C, R, (const bf16_t*)x, sums)),
              (bn_stats_finish_kernel<float><<<dim3((C + 15) / 16), dim3(256), 0, s>>>(part, splits, C, R, (const float*)x, sums)));
    return ok();
}

extern "C" int mmsum_bn_stats_from_sums(const float* raw, int R, int C, float* sums, float* running_mean, float* running_var, float momentum,
                                        void* stream) {
    if (R <= 0 || C <= 0 || raw == nullptr || sums == nullptr) return MMSUM_ERR_BAD_SHAPE;
    bn_stats_from_sums_kernel<<<dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(raw, R, C, sums, running_mean, running_var, momentum);
    return ok();
}

extern "C" int mmsum_bn_apply(int dtype, const void* x, float* sums, const float* raw, const float* gamma, const float* beta, const void* residual,
                              void* y, float* running_mean, float* running_var, int R, int C, float eps, float momentum, int relu,
                              int training, int pad_H, int pad_W, const int* images, int rows_per_image, void* stream) {
    const int vec = dtype == MMSUM_BF16 ? 8 : 4;
    if (R <= 0 || C % vec || (raw != nullptr && !training) || !img_args_ok(images, rows_per_image, R)) return MMSUM_ERR_BAD_SHAPE;
    if (pad_W != 0 && (pad_H <= 0 || pad_W < 0 || R % (pad_H * pad_W))) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int cgb = bn_cgb(C, vec), cblocks = (C / vec + cgb - 1) / cgb;
    const dim3 grid(cblocks, bn_row_blocks(R, 256 / cgb, cblocks)), block(256);
    if (residual != nullptr) {
        DT_SWITCH(dtype, (bn_apply_kernel<bf16_t, true><<<grid, block, 0, s>>>((const bf16_t*)x, sums, raw, gamma, beta, (const bf16_t*)residual, (bf16_t*)y, running_mean, running_var, R, C, cgb, eps, momentum, relu, training, pad_H, pad_W, images, rows_per_image)),
                  (bn_apply_kernel<float, true><<<grid, block, 0, s>>>((const float*)x, sums, raw, gamma, beta, (const float*)residual, (float*)y, running_mean, running_var, R, C, cgb, eps, momentum, relu, training, pad_H, pad_W, images, rows_per_image)));
    } else {
        DT_SWITCH(dtype, (bn_apply_kernel<bf16_t, false><<<grid, block, 0, s>>>((const bf16_t*)x, sums, raw, gamma, beta, (const bf16_t*)residual, (bf16_t*)y, running_mean, running_var, R, C, cgb, eps, momentum, relu, training, pad_H, pad_W, images, rows_per_image)),
                  (bn_apply_kernel<float, false><<<grid, block, 0, s>>>((const float*)x, sums, raw, gamma, beta, (const float*)residual, (float*)y, running_mean, running_var, R, C, cgb, eps, momentum, relu, training, pad_H, pad_W, images, rows_per_image)));
    }
    if (training && raw == nullptr && running_mean && running_var)
        bn_running_kernel<<<dim3((C + 255) / 256), dim3(256), 0, s>>>(sums, running_mean, running_var, R, C, momentum);
    return ok();
}

extern "C" int mmsum_bn_bwd_reduce(int dtype, const void* dy, const void* y, const void* x, const float* sums, int R, int C, float eps,
                                   int relu, float* dsums, void* workspace, int pad_H, int pad_W, const int* images, int rows_per_image,
                                   void* stream) {
    if (pad_W != 0 && (pad_H <= 0 || pad_W < 0 || R % (pad_H * pad_W))) return MMSUM_ERR_BAD_SHAPE;
    if (!img_args_ok(images, rows_per_image, R)) return MMSUM_ERR_BAD_SHAPE;
    const int vec = dtype == MMSUM_BF16 ? 8 : 4;
    if (R <= 0 || C <= 0 || C % vec) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int cgb = bn_cgb(C, vec), cblocks = (C / vec + cgb - 1) / cgb;
    const int splits = bn_splits(R, 256 / cgb, cblocks);
    const dim3 grid(cblocks, splits), block(256);
    float* part = (float*)workspace;
    if (relu) {
        DT_SWITCH(dtype, (bn_partial_kernel<bf16_t, 1, true><<<grid, block, 0, s>>>((const bf16_t*)dy, (const bf16_t*)y, (const bf16_t*)x, sums, R, C, cgb, relu, part, pad_H, pad_W, images, rows_per_image)),
                  (bn_partial_kernel<float, 1, true><<<grid, block, 0, s>>>((const float*)dy, (const float*)y, (const float*)x, sums, R, C, cgb, relu, part, pad_H, pad_W, images, rows_per_image)));
    } else {
        DT_SWITCH(dtype, (bn_partial_kernel<bf16_t, 1, false><<<grid, block, 0, s>>>((const bf16_t*)dy, (const bf16_t*)y, (const bf16_t*)x, sums, R, C, cgb, relu, part, pad_H, pad_W, images, rows_per_image)),
                  (bn_partial_kernel<float, 1, false><<<grid, block, 0, s>>>((const float*)dy, (const float*)y, (const float*)x, sums, R, C, cgb, relu, part, pad_H, pad_W, images, rows_per_image)));
    }
    bn_finish_kernel<<<dim3((2 * C + 15) / 16), dim3(256), 0, s>>>(part, splits, C, sums, eps, dsums);
    return ok();
}

extern "C" int mmsum_bn_bwd_apply(int dtype, const void* dy, const void* y, const void* x, const float* sums, const float* dsums,
                                  const float* gamma, void* dx, void* dresidual, float* dgamma, float* dbeta, int R, int C, float eps,
                                  int relu, int pad_H, int pad_W, int dx_pad_H, int dx_pad_W, const int* images, int rows_per_image,
                                  void* stream) {
    if (pad_W != 0 && (pad_H <= 0 || pad_W < 0 || R % (pad_H * pad_W))) return MMSUM_ERR_BAD_SHAPE;
    if (!img_args_ok(images, rows_per_image, R)) return MMSUM_ERR_BAD_SHAPE;
    if (dx_pad_W != 0 && (dx_pad_H <= 0 || dx_pad_W < 0 || R % (dx_pad_H * dx_pad_W))) return MMSUM_ERR_BAD_SHAPE;
    const int vec = dtype == MMSUM_BF16 ? 8 : 4;
    if (R <= 0 || C % vec) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int cgb = bn_cgb(C, vec), cblocks = (C / vec + cgb - 1) / cgb;
    const dim3 grid(cblocks, bn_row_blocks(R, 256 / cgb, cblocks)), block(256);
#define BN_BWD_APPLY(RL, DR)                                                                                                                          \
    DT_SWITCH(dtype, (bn_bwd_apply_kernel<bf16_t, RL, DR><<<grid, block, 0, s>>>((const bf16_t*)dy, (const bf16_t*)y, (const bf16_t*)x, sums, dsums, gamma, (bf16_t*)dx, (bf16_t*)dresidual, dgamma, dbeta, R, C, cgb, eps, relu, pad_H, pad_W, dx_pad_H, dx_pad_W, images, rows_per_image)), \
              (bn_bwd_apply_kernel<float, RL, DR><<<grid, block, 0, s>>>((const float*)dy, (const float*)y, (const float*)x, sums, dsums, gamma, (float*)dx, (float*)dresidual, dgamma, dbeta, R, C, cgb, eps, relu, pad_H, pad_W, dx_pad_H, dx_pad_W, images, rows_per_image)))
    if (relu) { if (dresidual) { BN_BWD_APPLY(true, true); } else { BN_BWD_APPLY(true, false); } }
    else { if (dresidual) { BN_BWD_APPLY(false, true); } else { BN_BWD_APPLY(false, false); } }
#undef BN_BWD_APPLY
    return ok();
}

extern "C" int mmsum_bn_rep_fix(int dtype, const void* y, float* raw, int R, int C, const int* images, int rows_per_image, void* stream) {
    const int vec = dtype == MMSUM_BF16 ? 8 : 4;
    if (R <= 0 || C <= 0 || C % vec || images == nullptr || raw == nullptr || !img_args_ok(images, rows_per_image, R)) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int cgb = bn_cgb(C, vec), cblocks = (C / vec + cgb - 1) / cgb, lanes = 256 / cgb;
    int splits = (rows_per_image + lanes * 4 - 1) / (lanes * 4);
    splits = splits < 1 ? 1 : (splits > 64 ? 64 : splits);
    const dim3 grid(cblocks, splits), block(256);
    DT_SWITCH(dtype, (bn_rep_fix_kernel<bf16_t><<<grid, block, 0, s>>>((const bf16_t*)y, raw, R, C, cgb, images, rows_per_image)),
              (bn_rep_fix_kernel<float><<<grid, block, 0, s>>>((const float*)y, raw, R, C, cgb, images, rows_per_image)));
    return ok();
}

extern "C" long mmsum_image_plan_workspace(int n) { return (long)sizeof(int) * (n > 0 ? n : 1); }

extern "C" int mmsum_image_plan(const float* img, long elems_per_image, const uint8_t* mask, int n, int positions, const int* rows_per_image,
                                const int* row_adjust, int n_rpi, int* plan, int* src, int64_t* slot_rows, int64_t* run_rows, void* workspace, void* stream) {
    if (n <= 0 || n > IMG_PLAN_MAX || positions <= 0 || elems_per_image <= 0 || n_rpi < 0 || n_rpi > 8 || (n_rpi > 0 && rows_per_image == nullptr))
        return MMSUM_ERR_BAD_SHAPE;
    if (img == nullptr || mask == nullptr || plan == nullptr || src == nullptr || slot_rows == nullptr || run_rows == nullptr || workspace == nullptr)
        return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    ImgRpi rpi{};
    for (int k = 0; k < n_rpi; ++k) { rpi.v[k] = rows_per_image[k]; rpi.adj[k] = row_adjust != nullptr ? row_adjust[k] : 0; }
    int* empty = (int*)workspace;
    image_empty_kernel<<<dim3(n), dim3(256), 0, s>>>(img, elems_per_image, mask, n, empty);
    image_plan_kernel<<<dim3(1), dim3(256), 0, s>>>(empty, n, positions, rpi, n_rpi, plan, src, slot_rows, run_rows);
    return ok();
}

extern "C" int mmsum_maxpool3x3s2(int dtype, const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, const int* images,
                                  void* stream) {
    if (N <= 0 || C % 4) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(grid_for((long)N * Ho * Wo * C / 4, 256)), block(256);
    DT_SWITCH(dtype, (maxpool_kernel<bf16_t><<<grid, block, 0, s>>>((const bf16_t*)x, (bf16_t*)y, N, H, W, C, Ho, Wo, images)),
              (maxpool_kernel<float><<<grid, block, 0, s>>>((const float*)x, (float*)y, N, H, W, C, Ho, Wo, images)));
    return ok();
}

extern "C" int mmsum_nchw_to_nhwc(int dtype, const float* x, void* y, int N, int C, int H, int W, const int* images, const int* src,
                                  void* stream) {
    if (N <= 0) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(grid_for((long)N * C * H * W, 256)), block(256);
    DT_SWITCH(dtype, (nchw_to_nhwc_kernel<bf16_t><<<grid, block, 0, s>>>(x, (bf16_t*)y, N, C, H, W, images, src)),
              (nchw_to_nhwc_kernel<float><<<grid, block, 0, s>>>(x, (float*)y, N, C, H, W, images, src)));
    return ok();
}

extern "C" int mmsum_table_gather(int dtype, const void* E, const int64_t* field, const int64_t* name, const int64_t* category,
                                  const int64_t* str_cat, const int64_t* str_bool, const int64_t* rating, const int64_t* hours,
                                  const void* w_rating, const void* w_hours, void* out, uint8_t* mask, int B, int D, int pad_id,
                                  void* stream) {
    if (B <= 0 || D % 4) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(47, B), block(256);
    DT_SWITCH(dtype, (table_gather_kernel<bf16_t><<<grid, block, 0, s>>>((const bf16_t*)E, field, name, category, str_cat, str_bool, rating, hours, (const bf16_t*)w_rating, (const bf16_t*)w_hours, (bf16_t*)out, mask, D, pad_id)),
              (table_gather_kernel<float><<<grid, block, 0, s>>>((const float*)E, field, name, category, str_cat, str_bool, rating, hours, (const float*)w_rating, (const float*)w_hours, (float*)out, mask, D, pad_id)));
    return ok();
}

extern "C" int mmsum_table_gather_bwd(int dtype, const void* dall, const int64_t* rating, const int64_t* hours, float* dw_rating,
                                      float* dw_hours, int B, int D, void* stream) {
    if (B <= 0) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((D + 255) / 256), block(256);
    DT_SWITCH(dtype, (table_gather_bwd_kernel<bf16_t><<<grid, block, 0, s>>>((const bf16_t*)dall, rating, hours, dw_rating, dw_hours, B, D)),
              (table_gather_bwd_kernel<float><<<grid, block, 0, s>>>((const float*)dall, rating, hours, dw_rating, dw_hours, B, D)));
    return ok();
}

extern "C" int mmsum_amazon_table_gather(int dtype, const void* E, const int64_t* field, const int64_t* price, const int64_t* rating,
                                         const int64_t* brand, const int64_t* name, const int64_t* category, const int64_t* description,
                                         const void* w_price, const void* w_rating, void* out, uint8_t* mask, int B, int D, int pad_id,
                                         void* stream) {
    if (B <= 0 || D % 4) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(133, B), block(256);
    DT_SWITCH(dtype, (amazon_gather_kernel<bf16_t><<<grid, block, 0, s>>>((const bf16_t*)E, field, price, rating, brand, name, category, description, (const bf16_t*)w_price, (const bf16_t*)w_rating, (bf16_t*)out, mask, D, pad_id)),
              (amazon_gather_kernel<float><<<grid, block, 0, s>>>((const float*)E, field, price, rating, brand, name, category, description, (const float*)w_price, (const float*)w_rating, (float*)out, mask, D, pad_id)));
    return ok();
}

extern "C" int mmsum_amazon_table_gather_bwd(int dtype, const void* dall, const int64_t* price, const int64_t* rating, float* dw_price,
                                             float* dw_rating, int B, int D, void* stream) {
    if (B <= 0) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((D + 255) / 256), block(256);
    DT_SWITCH(dtype, (amazon_gather_bwd_kernel<bf16_t><<<grid, block, 0, s>>>((const bf16_t*)dall, price, rating, dw_price, dw_rating, B, D)),
              (amazon_gather_bwd_kernel<float><<<grid, block, 0, s>>>((const float*)dall, price, rating, dw_price, dw_rating, B, D)));
    return ok();
}
