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

// ---- im2col / col2im ---------------------------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(256) void im2col_kernel(const T* __restrict__ x, T* __restrict__ col, int N, int H, int W, int C,
                                                     int KH, int KW, int stride, int pad, int Ho, int Wo, int Kpad) {
    const int cv = C / VEC;
    const long total = (long)N * Ho * Wo * KH * KW * cv;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv) * VEC;
        long t = i / cv;
        const int kw = (int)(t % KW); t /= KW;
        const int kh = (int)(t % KH); t /= KH;
        const long row = t;
        const int wo = (int)(row % Wo);
        const int ho = (int)((row / Wo) % Ho);
        const int n = (int)(row / ((long)Wo * Ho));
        const int h = ho * stride - pad + kh, w = wo * stride - pad + kw;
        T* dst = col + row * Kpad + (kh * KW + kw) * C + c;
        const bool in = h >= 0 && h < H && w >= 0 && w < W;
        const T* src = x + (((long)n * H + h) * W + w) * C + c;
        if constexpr (VEC == 4) {
            st4<T>(dst, in ? ld4<T>(src) : f32x4_t{0, 0, 0, 0});
        } else {
            dst[0] = in ? src[0] : from_f32<T>(0.f);
        }
    }
    const int K = KH * KW * C;
    if (Kpad > K) {
        const int tail = Kpad - K;
        const long tt = (long)N * Ho * Wo * tail;
        for (long i = blockIdx.x * 256L + threadIdx.x; i < tt; i += (long)gridDim.x * 256)
            col[(i / tail) * Kpad + K + (i % tail)] = from_f32<T>(0.f);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void col2im_kernel(const T* __restrict__ dcol, T* __restrict__ dx, int N, int H, int W, int C,
                                                     int KH, int KW, int stride, int pad, int Ho, int Wo, int Kpad) {
    const int cv = C / 4;
    const long total = (long)N * H * W * cv;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv) * 4;
        long t = i / cv;
        const int w = (int)(t % W); t /= W;
        const int h = (int)(t % H);
        const int n = (int)(t / H);
        f32x4_t acc = f32x4_t{0, 0, 0, 0};
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
                acc = acc + ld4<T>(dcol + (((long)n * Ho + ho) * Wo + wo) * Kpad + (kh * KW + kw) * C + c);
            }
        }
        st4<T>(dx + (((long)n * H + h) * W + w) * C + c, acc);
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
constexpr int BN_SPLITS = 128;
// MODE 0: sums of (x, x^2).  MODE 1: sums of (dy', dy'*xhat) for the backward pass.
// block = 16 column groups (4 channels, vector loads) x 16 row lanes; grid (C/64, splits)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(const T* __restrict__ a, const T* __restrict__ y, const T* __restrict__ x,
                                                         const float* __restrict__ sums, int R, int C, float eps, int relu,
                                                         float* __restrict__ part) {
    __shared__ float red[2][16][64];
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int col = blockIdx.x * 64 + cg * 4;
    const int rows_per = (R + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * rows_per, r1 = min(R, r0 + rows_per);
    f32x4_t s0 = f32x4_t{0, 0, 0, 0}, s1 = f32x4_t{0, 0, 0, 0};
    if (col < C) {
        f32x4_t mean = f32x4_t{0, 0, 0, 0}, rstd = f32x4_t{0, 0, 0, 0};
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mean[j] = sums[col + j];
                rstd[j] = rsqrtf(sums[C + col + j] + eps);
            }
        }
        // MODE 0 accumulates around a per-channel pivot (row 0) so that var = E[(x-p)^2] - E[x-p]^2 does not
        // cancel catastrophically when |mean| >> std
        const f32x4_t pivot = (MODE == 0) ? ld4<T>(a + col) : f32x4_t{0, 0, 0, 0};
        for (int r = r0 + rl; r < r1; r += 16) {
            const long o = (long)r * C + col;
            if (MODE == 0) {
                const f32x4_t v = ld4<T>(a + o) - pivot;
                s0 = s0 + v;
                s1 = s1 + v * v;
            } else {
                f32x4_t g = ld4<T>(a + o);
                if (relu) {
                    const f32x4_t yv = ld4<T>(y + o);
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (!(yv[j] > 0.f)) g[j] = 0.f;
                }
                s0 = s0 + g;
                s1 = s1 + g * (ld4<T>(x + o) - mean) * rstd;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[0][rl][cg * 4 + j] = s0[j]; red[1][rl][cg * 4 + j] = s1[j]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int w = threadIdx.x >> 6, c = threadIdx.x & 63;
        if (blockIdx.x * 64 + c < C) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += red[w][k][c];
            part[(long)blockIdx.y * 2 * C + w * C + blockIdx.x * 64 + c] = t;
        }
    }
}
__global__ __launch_bounds__(256) void bn_finish_kernel(const float* __restrict__ part, int splits, int C2, float* __restrict__ out) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (col < C2) {
#pragma unroll 4
        for (int k = sl; k < splits; k += 4) s += part[(long)k * C2 + col];
    }
    red[sl][cl] = s;
    __syncthreads();
    if (sl == 0 && col < C2) out[col] = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
}

// forward statistics: shifted partial sums -> {mean, biased variance}
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_finish_kernel(const float* __restrict__ part, int splits, int C, int R, const T* __restrict__ x,
                                                              float* __restrict__ out) {
    __shared__ float red[2][4][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    float s0 = 0.f, s1 = 0.f;
    if (col < C) {
#pragma unroll 4
        for (int k = sl; k < splits; k += 4) { s0 += part[(long)k * 2 * C + col]; s1 += part[(long)k * 2 * C + C + col]; }
    }
    red[0][sl][cl] = s0; red[1][sl][cl] = s1;
    __syncthreads();
    if (sl == 0 && col < C) {
        const float t0 = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
        const float t1 = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
        const float m = t0 / R;
        out[col] = to_f32(x[col]) + m;
        out[C + col] = fmaxf(t1 / R - m * m, 0.f);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ sums, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const T* __restrict__ residual, T* __restrict__ y,
                                                       float* __restrict__ running_mean, float* __restrict__ running_var, int R, int C,
                                                       float eps, float momentum, int relu, int training) {
    const int cv = C / 4;
    const long total = (long)R * cv;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv) * 4;
        const long o = (i / cv) * C + c;
        const f32x4_t xv = ld4<T>(x + o);
        f32x4_t out;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float mean, var;
            if (training) {
                mean = sums[c + j];
                var = sums[C + c + j];
            } else {
                mean = running_mean[c + j];
                var = running_var[c + j];
            }
            out[j] = (xv[j] - mean) * rsqrtf(var + eps) * gamma[c + j] + beta[c + j];
        }
        if (residual) out = out + ld4<T>(residual + o);
        if (relu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) out[j] = fmaxf(out[j], 0.f);
        }
        st4<T>(y + o, out);
    }
}
// running stats update, separate launch so the apply kernel never races with it
__global__ void bn_running_kernel(const float* __restrict__ sums, float* __restrict__ running_mean, float* __restrict__ running_var,
                                  int R, int C, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float mean = sums[c];
    const float var = sums[C + c];
    const float unbiased = R > 1 ? var * ((float)R / (float)(R - 1)) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ x,
                                                           const float* __restrict__ sums, const float* __restrict__ dsums,
                                                           const float* __restrict__ gamma, T* __restrict__ dx, T* __restrict__ dresidual,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int R, int C, float eps,
                                                           int relu) {
    const int cv = C / 4;
    const long total = (long)R * cv;
    const float invR = 1.f / R;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv) * 4;
        const long o = (i / cv) * C + c;
        f32x4_t g = ld4<T>(dy + o);
        if (relu) {
            const f32x4_t yv = ld4<T>(y + o);
#pragma unroll
            for (int j = 0; j < 4; ++j) if (!(yv[j] > 0.f)) g[j] = 0.f;
        }
        const f32x4_t xv = ld4<T>(x + o);
        f32x4_t out;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float mean = sums[c + j];
            const float rstd = rsqrtf(sums[C + c + j] + eps);
            const float xh = (xv[j] - mean) * rstd;
            out[j] = gamma[c + j] * rstd * (g[j] - dsums[c + j] * invR - xh * dsums[C + c + j] * invR);
        }
        st4<T>(dx + o, out);
        if (dresidual) st4<T>(dresidual + o, g);
    }
    // parameter gradients: dbeta = sum dy', dgamma = sum dy'*xhat (accumulate into the f32 arena)
    if (blockIdx.x == 0 && dgamma != nullptr)
        for (int c = threadIdx.x; c < C; c += 256) { dbeta[c] += dsums[c]; dgamma[c] += dsums[C + c]; }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int Ho, int Wo) {
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
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int N, int C, int H, int W) {
    const long total = (long)N * C * H * W;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long t = i / C;
        const int w = (int)(t % W); t /= W;
        const int h = (int)(t % H);
        const int n = (int)(t / H);
        y[i] = from_f32<T>(x[(((long)n * C + c) * H + h) * W + w]);
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
                            int Ho, int Wo, int Kpad, void* stream) {
    if (N <= 0 || Kpad < KH * KW * C) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = (C % 4 == 0);
    const long items = (long)N * Ho * Wo * KH * KW * (v4 ? C / 4 : C);
    const dim3 grid(grid_for(items, 256)), block(256);
    if (v4) DT_SWITCH(dtype, (im2col_kernel<bf16_t, 4><<<grid, block, 0, s>>>((const bf16_t*)x, (bf16_t*)col, N, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad)),
                      (im2col_kernel<float, 4><<<grid, block, 0, s>>>((const float*)x, (float*)col, N, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad)));
    else DT_SWITCH(dtype, (im2col_kernel<bf16_t, 1><<<grid, block, 0, s>>>((const bf16_t*)x, (bf16_t*)col, N, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad)),
                   (im2col_kernel<float, 1><<<grid, block, 0, s>>>((const float*)x, (float*)col, N, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad)));
    return ok();
}

extern "C" int mmsum_col2im(int dtype, const void* dcol, void* dx, int N, int H, int W, int C, int KH, int KW, int stride, int pad,
                            int Ho, int Wo, int Kpad, void* stream) {
    if (N <= 0 || C % 4) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(grid_for((long)N * H * W * C / 4, 256)), block(256);
    DT_SWITCH(dtype, (col2im_kernel<bf16_t><<<grid, block, 0, s>>>((const bf16_t*)dcol, (bf16_t*)dx, N, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad)),
              (col2im_kernel<float><<<grid, block, 0, s>>>((const float*)dcol, (float*)dx, N, H, W, C, KH, KW, stride, pad, Ho, Wo, Kpad)));
    return ok();
}

extern "C" int mmsum_conv_weight_permute(int dtype, void* matrix, float* weight, int Cout, int Cin, int KH, int KW, int Kpad,
                                         int to_matrix, int accumulate, void* stream) {
    if (Cout <= 0 || Kpad < Cin * KH * KW) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    if (to_matrix) {
        const dim3 grid(grid_for((long)Cout * Kpad, 256)), block(256);
        DT_SWITCH(dtype, (weight_to_matrix_kernel<bf16_t><<<grid, block, 0, s>>>((bf16_t*)matrix, weight, Cout, Cin, KH, KW, Kpad)),
                  (weight_to_matrix_kernel<float><<<grid, block, 0, s>>>((float*)matrix, weight, Cout, Cin, KH, KW, Kpad)));
    } else {
        const dim3 grid(grid_for((long)Cout * Cin * KH * KW, 256)), block(256);
        matrix_to_weight_grad_kernel<<<grid, block, 0, s>>>((const float*)matrix, weight, Cout, Cin, KH, KW, Kpad, accumulate);
    }
    return ok();
}

extern "C" long mmsum_bn_workspace(int C) { return (long)BN_SPLITS * 2 * C * sizeof(float); }

extern "C" int mmsum_bn_reduce(int dtype, const void* x, int R, int C, float* sums, void* workspace, void* stream) {
    if (R <= 0 || C <= 0) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int splits = R < BN_SPLITS * 16 ? max(1, R / 16) : BN_SPLITS;
    const dim3 grid((C + 63) / 64, splits), block(256);
    float* part = (float*)workspace;
    DT_SWITCH(dtype, (bn_partial_kernel<bf16_t, 0><<<grid, block, 0, s>>>((const bf16_t*)x, nullptr, nullptr, nullptr, R, C, 0.f, 0, part)),
              (bn_partial_kernel<float, 0><<<grid, block, 0, s>>>((const float*)x, nullptr, nullptr, nullptr, R, C, 0.f, 0, part)));
    DT_SWITCH(dtype, (bn_stats_finish_kernel<bf16_t><<<dim3((C + 63) / 64), dim3(256), 0, s>>>(part, splits, C, R, (const bf16_t*)x, sums)),
              (bn_stats_finish_kernel<float><<<dim3((C + 63) / 64), dim3(256), 0, s>>>(part, splits, C, R, (const float*)x, sums)));
    return ok();
}

extern "C" int mmsum_bn_apply(int dtype, const void* x, const float* sums, const float* gamma, const float* beta, const void* residual,
                              void* y, float* running_mean, float* running_var, int R, int C, float eps, float momentum, int relu,
                              int training, void* stream) {
    if (R <= 0 || C % 4) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(grid_for((long)R * C / 4, 256)), block(256);
    DT_SWITCH(dtype, (bn_apply_kernel<bf16_t><<<grid, block, 0, s>>>((const bf16_t*)x, sums, gamma, beta, (const bf16_t*)residual, (bf16_t*)y, running_mean, running_var, R, C, eps, momentum, relu, training)),
              (bn_apply_kernel<float><<<grid, block, 0, s>>>((const float*)x, sums, gamma, beta, (const float*)residual, (float*)y, running_mean, running_var, R, C, eps, momentum, relu, training)));
    if (training && running_mean && running_var)
        bn_running_kernel<<<dim3((C + 255) / 256), dim3(256), 0, s>>>(sums, running_mean, running_var, R, C, momentum);
    return ok();
}

extern "C" int mmsum_bn_bwd_reduce(int dtype, const void* dy, const void* y, const void* x, const float* sums, int R, int C, float eps,
                                   int relu, float* dsums, void* workspace, void* stream) {
    if (R <= 0 || C <= 0) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int splits = R < BN_SPLITS * 16 ? max(1, R / 16) : BN_SPLITS;
    const dim3 grid((C + 63) / 64, splits), block(256);
    float* part = (float*)workspace;
    DT_SWITCH(dtype, (bn_partial_kernel<bf16_t, 1><<<grid, block, 0, s>>>((const bf16_t*)dy, (const bf16_t*)y, (const bf16_t*)x, sums, R, C, eps, relu, part)),
              (bn_partial_kernel<float, 1><<<grid, block, 0, s>>>((const float*)dy, (const float*)y, (const float*)x, sums, R, C, eps, relu, part)));
    bn_finish_kernel<<<dim3((2 * C + 63) / 64), dim3(256), 0, s>>>(part, splits, 2 * C, dsums);
    return ok();
}

extern "C" int mmsum_bn_bwd_apply(int dtype, const void* dy, const void* y, const void* x, const float* sums, const float* dsums,
                                  const float* gamma, void* dx, void* dresidual, float* dgamma, float* dbeta, int R, int C, float eps,
                                  int relu, void* stream) {
    if (R <= 0 || C % 4) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(grid_for((long)R * C / 4, 256)), block(256);
    DT_SWITCH(dtype, (bn_bwd_apply_kernel<bf16_t><<<grid, block, 0, s>>>((const bf16_t*)dy, (const bf16_t*)y, (const bf16_t*)x, sums, dsums, gamma, (bf16_t*)dx, (bf16_t*)dresidual, dgamma, dbeta, R, C, eps, relu)),
              (bn_bwd_apply_kernel<float><<<grid, block, 0, s>>>((const float*)dy, (const float*)y, (const float*)x, sums, dsums, gamma, (float*)dx, (float*)dresidual, dgamma, dbeta, R, C, eps, relu)));
    return ok();
}

extern "C" int mmsum_maxpool3x3s2(int dtype, const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, void* stream) {
    if (N <= 0 || C % 4) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(grid_for((long)N * Ho * Wo * C / 4, 256)), block(256);
    DT_SWITCH(dtype, (maxpool_kernel<bf16_t><<<grid, block, 0, s>>>((const bf16_t*)x, (bf16_t*)y, N, H, W, C, Ho, Wo)),
              (maxpool_kernel<float><<<grid, block, 0, s>>>((const float*)x, (float*)y, N, H, W, C, Ho, Wo)));
    return ok();
}

extern "C" int mmsum_nchw_to_nhwc(int dtype, const float* x, void* y, int N, int C, int H, int W, void* stream) {
    if (N <= 0) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(grid_for((long)N * C * H * W, 256)), block(256);
    DT_SWITCH(dtype, (nchw_to_nhwc_kernel<bf16_t><<<grid, block, 0, s>>>(x, (bf16_t*)y, N, C, H, W)),
              (nchw_to_nhwc_kernel<float><<<grid, block, 0, s>>>(x, (float*)y, N, C, H, W)));
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
