// HBM-bound row-wise kernels: LayerNorm (+residual, +dropout), embedding+LN, gated fusion,
// label-smoothing loss, column sums, grad-norm, AdamW, casts.  All are one-pass-over-the-row
// kernels with 8/16-byte vector accesses; a row of D <= 1024 lives in one wave's registers.
#include "mmsum_device.h"
#include "mmsum_kernels.h"
#include <type_traits>

namespace {

template <typename T> __device__ __forceinline__ f32x4_t load4(const T* p);
template <> __device__ __forceinline__ f32x4_t load4<float>(const float* p) { return *reinterpret_cast<const f32x4_t*>(p); }
template <> __device__ __forceinline__ f32x4_t load4<bf16_t>(const bf16_t* p) {
    const bf16x4_t v = *reinterpret_cast<const bf16x4_t*>(p);
    return f32x4_t{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <typename T> __device__ __forceinline__ void store4(T* p, f32x4_t v);
template <> __device__ __forceinline__ void store4<float>(float* p, f32x4_t v) { *reinterpret_cast<f32x4_t*>(p) = v; }
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, f32x4_t v) {
    *reinterpret_cast<bf16x4_t*>(p) = bf16x4_t{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
}

// Dropout salt: an optional device-resident step counter mixed into every dropout seed.  A captured HIP graph bakes
// the seed ARGUMENT into its kernel nodes; bumping the salt (mmsum_bump_u64, itself a graph node) gives every replay
// fresh masks while forward and backward of one step still agree.  NULL = seeds used as passed.  The salt is an
// ARGUMENT of every dropout entry point (the library keeps no state of its own).
__device__ __forceinline__ uint64_t salted_seed(uint64_t seed, const uint64_t* salt) {
    return salt != nullptr ? seed + *salt * 0x9E3779B97F4A7C15ull : seed;
}
__global__ void bump_u64_kernel(uint64_t* p, uint64_t inc) { *p += inc; }
// Device-resident live row count (NULL = all R rows): rows at and past it are neither read nor written, so one captured
// HIP graph sized for the row capacity serves every batch.
__device__ __forceinline__ int live_rows_of(int R, const int* __restrict__ live) {
    return live != nullptr ? min(R, max(0, *live)) : R;
}

__device__ __forceinline__ uint32_t keep_threshold(float p_drop) {
    if (p_drop <= 0.f) return 0xFFFFFFFFu;
    const double t = (1.0 - (double)p_drop) * 4294967296.0;
    return t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
}

// --------------------------------------------------------------------------------------------
// LayerNorm family.  One wave per row; lane owns columns  (i*64 + lane)*4 .. +3, i < VPL.
// --------------------------------------------------------------------------------------------
template <int VPL>
__device__ __forceinline__ void ln_stats(const f32x4_t (&z)[VPL], int D, float eps, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) s += z[i][0] + z[i][1] + z[i][2] + z[i][3];
    mean = warp_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = z[i][j] - mean; q += d * d; }
    rstd = rsqrtf(warp_sum(q) / D + eps);
}

template <typename T, int VPL>
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         T* __restrict__ y, float* __restrict__ mean_out,
                                                         float* __restrict__ rstd_out, int R, int D, float eps,
                                                         float p_drop, uint64_t seed, const uint64_t* __restrict__ salt,
                                                         const int* __restrict__ live, float* __restrict__ y32) {
    seed = salted_seed(seed, salt);
    R = live_rows_of(R, live);
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const uint32_t thr = keep_threshold(p_drop);
    const float dscale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < R; row += gridDim.x * wpb) {
        f32x4_t z[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * 64 + lane) * 4;
            f32x4_t xv = load4<T>(x + (long)row * D + c);
            const f32x4_t rv = load4<T>(res + (long)row * D + c);
            if (p_drop > 0.f) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    xv[j] = dropout_keep(seed, (uint64_t)row * D + c + j, thr) ? xv[j] * dscale : 0.f;
            }
            z[i] = xv + rv;
        }
        float mean, rstd;
        ln_stats<VPL>(z, D, eps, mean, rstd);
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * 64 + lane) * 4;
            const f32x4_t g = load4<float>(gamma + c), b = load4<float>(beta + c);
            f32x4_t o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (z[i][j] - mean) * rstd * g[j] + b[j];
            store4<T>(y + (long)row * D + c, o);
            if (y32 != nullptr) store4<float>(y32 + (long)row * D + c, o);
        }
        if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    }
}

// The decode step's gated fusion + residual + LayerNorm in one pass (generation only: nothing is saved for a backward):
//   y = LN(res + yt + relu(tanh(pa)) [table present] ytab + relu(tanh(pb)) [image present] yimg)      (:732-744, :474-477)
// One row per wave.  The gated sum is not rounded to the compute dtype on the way (the two-kernel form stores it).
template <typename T, int VPL>
__global__ __launch_bounds__(256) void gate_add_ln_fwd_kernel(const T* __restrict__ pa, const T* __restrict__ pb, const T* __restrict__ yt,
                                                              const T* __restrict__ ytab, const T* __restrict__ yimg,
                                                              const uint8_t* __restrict__ no_table, const uint8_t* __restrict__ no_img,
                                                              const T* __restrict__ res, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, T* __restrict__ y, int R, int D, int rows_per_b, float eps) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < R; row += gridDim.x * wpb) {
        const int b = row / rows_per_b;
        const float ma = no_table[b] ? 0.f : 1.f, mb = no_img[b] ? 0.f : 1.f;
        f32x4_t z[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const long o = (long)row * D + (i * 64 + lane) * 4;
            const f32x4_t a = load4<T>(pa + o), bb = load4<T>(pb + o), t = load4<T>(yt + o), tb = load4<T>(ytab + o), im = load4<T>(yimg + o);
            const f32x4_t rv = load4<T>(res + o);
#pragma unroll
            for (int j = 0; j < 4; ++j) z[i][j] = rv[j] + t[j] + ma * fmaxf(tanhf(a[j]), 0.f) * tb[j] + mb * fmaxf(tanhf(bb[j]), 0.f) * im[j];
        }
        float mean, rstd;
        ln_stats<VPL>(z, D, eps, mean, rstd);
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * 64 + lane) * 4;
            const f32x4_t g = load4<float>(gamma + c), be = load4<float>(beta + c);
            f32x4_t o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (z[i][j] - mean) * rstd * g[j] + be[j];
            store4<T>(y + (long)row * D + c, o);
        }
    }
}

// dz = rstd * (g*dy - mean(g*dy) - xhat*mean(g*dy*xhat)); block accumulates dgamma/dbeta over its rows
// in registers, reduces across its waves through LDS and issues one f32 atomic per column.
template <typename T, int VPL>
__global__ __launch_bounds__(256) void add_ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                         const T* __restrict__ res, const float* __restrict__ gamma,
                                                         const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                         T* __restrict__ dx, T* __restrict__ dres, int accumulate_dres,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta, int R, int D,
                                                         float p_drop, uint64_t seed, const uint64_t* __restrict__ salt,
                                                         float* __restrict__ dxsum, const int* __restrict__ live) {
    seed = salted_seed(seed, salt);
    R = live_rows_of(R, live);
    __shared__ float red[4][VPL * 256 * 3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;
    const uint32_t thr = keep_threshold(p_drop);
    const float dscale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    f32x4_t ag[VPL], ab[VPL], ax[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) { ag[i] = f32x4_t{0, 0, 0, 0}; ab[i] = f32x4_t{0, 0, 0, 0}; ax[i] = f32x4_t{0, 0, 0, 0}; }
    for (int row = blockIdx.x * wpb + wave; row < R; row += gridDim.x * wpb) {
        const float mean = mean_in[row], rstd = rstd_in[row];
        f32x4_t xh[VPL], gdy[VPL];
        bool keep[VPL][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * 64 + lane) * 4;
            f32x4_t xv = load4<T>(x + (long)row * D + c);
            const f32x4_t rv = load4<T>(res + (long)row * D + c);
            const f32x4_t dv = load4<T>(dy + (long)row * D + c);
            const f32x4_t g = load4<float>(gamma + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                keep[i][j] = p_drop > 0.f ? dropout_keep(seed, (uint64_t)row * D + c + j, thr) : true;
                const float zz = (keep[i][j] ? xv[j] * dscale : 0.f) + rv[j];
                xh[i][j] = (zz - mean) * rstd;
                gdy[i][j] = g[j] * dv[j];
                s1 += gdy[i][j];
                s2 += gdy[i][j] * xh[i][j];
                ag[i][j] += dv[j] * xh[i][j];
                ab[i][j] += dv[j];
            }
        }
        s1 = warp_sum(s1) / D;
        s2 = warp_sum(s2) / D;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * 64 + lane) * 4;
            f32x4_t dz, dxv;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dz[j] = rstd * (gdy[i][j] - s1 - xh[i][j] * s2);
                dxv[j] = keep[i][j] ? dz[j] * dscale : 0.f;
            }
            ax[i] = ax[i] + dxv;                       // column sums of dx = bias gradient of the linear layer that produced x
            store4<T>(dx + (long)row * D + c, dxv);
            if (accumulate_dres) dz = dz + load4<T>(dres + (long)row * D + c);
            store4<T>(dres + (long)row * D + c, dz);
        }
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = (i * 64 + lane) * 4 + j;
            red[wave][c] = ag[i][j];
            red[wave][VPL * 256 + c] = ab[i][j];
            red[wave][2 * VPL * 256 + c] = ax[i][j];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        float g = 0.f, b = 0.f, xs = 0.f;
        for (int w = 0; w < wpb; ++w) { g += red[w][c]; b += red[w][VPL * 256 + c]; xs += red[w][2 * VPL * 256 + c]; }
        atomicAdd(dgamma + c, g);
        atomicAdd(dbeta + c, b);
        if (dxsum != nullptr) atomicAdd(dxsum + c, xs);
    }
}

// y = dropout(LN(E[id] + P[t + off] + rd[seq] * rvec))
template <typename T, int VPL>
__global__ __launch_bounds__(256) void embed_ln_fwd_kernel(const int64_t* __restrict__ ids, const T* __restrict__ E,
                                                           const T* __restrict__ P, const float* __restrict__ rd,
                                                           const T* __restrict__ rvec, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, T* __restrict__ y,
                                                           float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                           int R, int T_len, int D, int pos_offset, float eps, float p_drop,
                                                           uint64_t seed, const uint64_t* __restrict__ salt) {
    seed = salted_seed(seed, salt);
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const uint32_t thr = keep_threshold(p_drop);
    const float dscale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < R; row += gridDim.x * wpb) {
        const long id = ids[row];
        const int t = row % T_len, seq = row / T_len;
        const float r = rd ? rd[seq] : 0.f;
        f32x4_t z[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * 64 + lane) * 4;
            z[i] = load4<T>(E + id * D + c) + load4<T>(P + (long)(t + pos_offset) * D + c);
            if (rd) z[i] = z[i] + load4<T>(rvec + c) * r;
        }
        float mean, rstd;
        ln_stats<VPL>(z, D, eps, mean, rstd);
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * 64 + lane) * 4;
            const f32x4_t g = load4<float>(gamma + c), b = load4<float>(beta + c);
            f32x4_t o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = (z[i][j] - mean) * rstd * g[j] + b[j];
                if (p_drop > 0.f) o[j] = dropout_keep(seed, (uint64_t)row * D + c + j, thr) ? o[j] * dscale : 0.f;
            }
            store4<T>(y + (long)row * D + c, o);
        }
        if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    }
}

// Block = one position t (all sequences): dP[t+off] is written without atomics, dE rows get f32
// atomics (pad rows skipped: nn.Embedding(padding_idx)), drvec/dgamma/dbeta one atomic per column.
template <typename T, int VPL>
__global__ __launch_bounds__(256) void embed_ln_bwd_kernel(const T* __restrict__ dy, const int64_t* __restrict__ ids,
                                                           const T* __restrict__ E, const T* __restrict__ P,
                                                           const float* __restrict__ rd, const T* __restrict__ rvec,
                                                           const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                           const float* __restrict__ rstd_in, float* __restrict__ dE,
                                                           float* __restrict__ dP, float* __restrict__ drvec,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int nseq,
                                                           int T_len, int D, int pos_offset, int pad_id, float p_drop,
                                                           uint64_t seed, const uint64_t* __restrict__ salt) {
    seed = salted_seed(seed, salt);
    __shared__ float red[4][VPL * 256 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;
    const int t = blockIdx.x;
    const uint32_t thr = keep_threshold(p_drop);
    const float dscale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
    f32x4_t ag[VPL], ab[VPL], ap[VPL], ar[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) { ag[i] = ab[i] = ap[i] = ar[i] = f32x4_t{0, 0, 0, 0}; }
    // grid = (positions, sequence chunks): a position's sequences are shared out over gridDim.y blocks (one block per position left
    // half the CUs idle and 252 rows in a row per wave); every per-position / per-column sum below leaves through atomics anyway
    for (int seq = blockIdx.y * wpb + wave; seq < nseq; seq += wpb * gridDim.y) {
        const int row = seq * T_len + t;
        const long id = ids[row];
        const float r = rd ? rd[seq] : 0.f;
        const float mean = mean_in[row], rstd = rstd_in[row];
        f32x4_t xh[VPL], gdy[VPL], dv[VPL];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * 64 + lane) * 4;
            f32x4_t z = load4<T>(E + id * D + c) + load4<T>(P + (long)(t + pos_offset) * D + c);
            if (rd) z = z + load4<T>(rvec + c) * r;
            dv[i] = load4<T>(dy + (long)row * D + c);
            const f32x4_t g = load4<float>(gamma + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (p_drop > 0.f) dv[i][j] = dropout_keep(seed, (uint64_t)row * D + c + j, thr) ? dv[i][j] * dscale : 0.f;
                xh[i][j] = (z[j] - mean) * rstd;
                gdy[i][j] = g[j] * dv[i][j];
                s1 += gdy[i][j];
                s2 += gdy[i][j] * xh[i][j];
                ag[i][j] += dv[i][j] * xh[i][j];
                ab[i][j] += dv[i][j];
            }
        }
        s1 = warp_sum(s1) / D;
        s2 = warp_sum(s2) / D;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * 64 + lane) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dz = rstd * (gdy[i][j] - s1 - xh[i][j] * s2);
                ap[i][j] += dz;
                ar[i][j] += dz * r;
                if (id != pad_id) atomicAdd(dE + id * D + c + j, dz);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = (i * 64 + lane) * 4 + j;
            red[wave][c] = ag[i][j];
            red[wave][VPL * 256 + c] = ab[i][j];
            red[wave][2 * VPL * 256 + c] = ap[i][j];
            red[wave][3 * VPL * 256 + c] = ar[i][j];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        float g = 0.f, b = 0.f, pp = 0.f, rr = 0.f;
        for (int w = 0; w < wpb; ++w) {
            g += red[w][c]; b += red[w][VPL * 256 + c]; pp += red[w][2 * VPL * 256 + c]; rr += red[w][3 * VPL * 256 + c];
        }
        atomicAdd(dgamma + c, g);
        atomicAdd(dbeta + c, b);
        atomicAdd(dP + (long)(t + pos_offset) * D + c, pp);
        if (rd) atomicAdd(drvec + c, rr);
    }
}

// --------------------------------------------------------------------------------------------
// Gated fusion (elementwise)
// --------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gate_fwd_kernel(const T* __restrict__ pa, const T* __restrict__ pb, const T* __restrict__ yt,
                                                       const T* __restrict__ ytab, const T* __restrict__ yimg,
                                                       const uint8_t* __restrict__ no_table, const uint8_t* __restrict__ no_img,
                                                       T* __restrict__ out, long n4, int D, int rows_per_b) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long e = i * 4;
        const int b = (int)((e / D) / rows_per_b);
        const float ma = no_table[b] ? 0.f : 1.f, mb = no_img[b] ? 0.f : 1.f;
        const f32x4_t a = load4<T>(pa + e), bb = load4<T>(pb + e), t = load4<T>(yt + e), tb = load4<T>(ytab + e), im = load4<T>(yimg + e);
        f32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = t[j] + ma * fmaxf(tanhf(a[j]), 0.f) * tb[j] + mb * fmaxf(tanhf(bb[j]), 0.f) * im[j];
        store4<T>(out + e, o);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ pa, const T* __restrict__ pb,
                                                       const T* __restrict__ ytab, const T* __restrict__ yimg,
                                                       const uint8_t* __restrict__ no_table, const uint8_t* __restrict__ no_img,
                                                       T* __restrict__ dpa, T* __restrict__ dpb, T* __restrict__ dyt,
                                                       T* __restrict__ dytab, T* __restrict__ dyimg, long n4, int D, int rows_per_b) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long e = i * 4;
        const int b = (int)((e / D) / rows_per_b);
        const float ma = no_table[b] ? 0.f : 1.f, mb = no_img[b] ? 0.f : 1.f;
        const f32x4_t g = load4<T>(dout + e), a = load4<T>(pa + e), bb = load4<T>(pb + e), tb = load4<T>(ytab + e), im = load4<T>(yimg + e);
        f32x4_t oa, ob, otb, oim;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float ta = tanhf(a[j]), tbv = tanhf(bb[j]);
            const float al = ma * fmaxf(ta, 0.f), be = mb * fmaxf(tbv, 0.f);
            otb[j] = g[j] * al;
            oim[j] = g[j] * be;
            oa[j] = (ta > 0.f) ? ma * g[j] * tb[j] * (1.f - ta * ta) : 0.f;
            ob[j] = (tbv > 0.f) ? mb * g[j] * im[j] * (1.f - tbv * tbv) : 0.f;
        }
        store4<T>(dpa + e, oa);
        store4<T>(dpb + e, ob);
        store4<T>(dyt + e, g);
        store4<T>(dytab + e, otb);
        store4<T>(dyimg + e, oim);
    }
}

// The same with the two bias gradients that are column sums of the gate's own outputs taken on the way (the separate
// column-sum passes re-read 264 MB per decoder layer): sum_dpa += column sums of dpa (alpha_proj.bias), sum_dpb += of dpb
// (beta_proj.bias).  (out_proj.bias is the column sum of dyt / dytab / dyimg only AFTER the alpha / beta input gradients have
// been accumulated into them: it comes out of out_proj's weight-gradient kernel instead.)  A thread owns column groups
// threadIdx.x + 256 v of D / 4 in every row it visits (rows blockIdx.x, + gridDim.x, ...), keeps its sums in registers -- of
// the values as stored, i.e. rounded to T -- and issues one f32 atomic per column at the end (a few hundred blocks per launch).
template <typename T, int VPL>
__global__ __launch_bounds__(256) void gate_bwd_sums_kernel(const T* __restrict__ dout, const T* __restrict__ pa, const T* __restrict__ pb,
                                                            const T* __restrict__ ytab, const T* __restrict__ yimg,
                                                            const uint8_t* __restrict__ no_table, const uint8_t* __restrict__ no_img,
                                                            T* __restrict__ dpa, T* __restrict__ dpb, T* __restrict__ dyt,
                                                            T* __restrict__ dytab, T* __restrict__ dyimg, int R, int D, int rows_per_b,
                                                            float* __restrict__ sum_dpa, float* __restrict__ sum_dpb) {
    const int cg = D >> 2;
    f32x4_t sa[VPL], sb[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) { sa[v] = f32x4_t{0, 0, 0, 0}; sb[v] = f32x4_t{0, 0, 0, 0}; }
    auto rnd = [](float x) { return to_f32(from_f32<T>(x)); };
    for (int row = blockIdx.x; row < R; row += gridDim.x) {
        const int b = row / rows_per_b;
        const float ma = no_table[b] ? 0.f : 1.f, mb = no_img[b] ? 0.f : 1.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int c = threadIdx.x + 256 * v;
            if (c >= cg) break;
            const long e = (long)row * D + 4 * c;
            const f32x4_t g = load4<T>(dout + e), a = load4<T>(pa + e), bb = load4<T>(pb + e), tb = load4<T>(ytab + e), im = load4<T>(yimg + e);
            f32x4_t oa, ob, otb, oim;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float ta = tanhf(a[j]), tbv = tanhf(bb[j]);
                const float al = ma * fmaxf(ta, 0.f), be = mb * fmaxf(tbv, 0.f);
                otb[j] = g[j] * al;
                oim[j] = g[j] * be;
                oa[j] = (ta > 0.f) ? ma * g[j] * tb[j] * (1.f - ta * ta) : 0.f;
                ob[j] = (tbv > 0.f) ? mb * g[j] * im[j] * (1.f - tbv * tbv) : 0.f;
                sa[v][j] += rnd(oa[j]);
                sb[v][j] += rnd(ob[j]);
            }
            store4<T>(dpa + e, oa);
            store4<T>(dpb + e, ob);
            store4<T>(dyt + e, g);
            store4<T>(dytab + e, otb);
            store4<T>(dyimg + e, oim);
        }
    }
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        const int c = threadIdx.x + 256 * v;
        if (c >= cg) break;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            atomicAdd(sum_dpa + 4 * c + j, sa[v][j]);
            atomicAdd(sum_dpb + 4 * c + j, sb[v][j]);
        }
    }
}

// --------------------------------------------------------------------------------------------
// Label-smoothing loss: one block per row, online max/sum in one sweep, second sweep writes the
// gradient in place.
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ void online_merge(float& m, float& s, float m2, float s2) {
    const float mn = fmaxf(m, m2);
    if (mn == -INFINITY) { s = 0.f; m = mn; return; }
    s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
    m = mn;
}

// 16-byte chunks of a logits row (8 bf16 / 4 f32); rows start 16-byte aligned when ld * sizeof(T) is a multiple of 16.
template <typename T> struct RowChunk;
template <> struct RowChunk<bf16_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
        const u32x4_t w = *reinterpret_cast<const u32x4_t*>(p);
        bf16_t t[8];
        __builtin_memcpy(t, &w, 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)t[j];
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
        bf16_t t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = (bf16_t)v[j];
        u32x4_t w;
        __builtin_memcpy(&w, t, 16);
        *reinterpret_cast<u32x4_t*>(p) = w;
    }
};
template <> struct RowChunk<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
        const f32x4_t w = *reinterpret_cast<const f32x4_t*>(p);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = w[j];
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) { *reinterpret_cast<f32x4_t*>(p) = f32x4_t{v[0], v[1], v[2], v[3]}; }
};

// One workgroup per row, 16-byte accesses (VEC) or scalar ones (unaligned rows); pass 1: online max / sum-exp / sum of
// logits; pass 2 (the row is L2-resident by then): dlogits in place.
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void ls_loss_kernel(T* __restrict__ logits, long ld, const int64_t* __restrict__ target,
                                                      float* __restrict__ row_loss, int V, float smoothing, float gscale,
                                                      int write_grad) {
    __shared__ float sm[4], ss[4], sx[4];
    constexpr int CN = RowChunk<T>::N;
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    T* x = logits + (long)row * ld;
    float m = -INFINITY, s = 0.f, sumx = 0.f;
    if constexpr (VEC) {
        for (int c = tid * CN; c < V; c += 256 * CN) {
            float v[CN];
            RowChunk<T>::load(x + c, v);
            float cm = -INFINITY;
#pragma unroll
            for (int j = 0; j < CN; ++j)
                if (c + j < V) { cm = fmaxf(cm, v[j]); sumx += v[j]; }
            if (cm > m) { s *= __expf(m - cm); m = cm; }
#pragma unroll
            for (int j = 0; j < CN; ++j)
                if (c + j < V) s += __expf(v[j] - m);
        }
    } else {
        for (int c = tid; c < V; c += 256) {
            const float v = to_f32(x[c]);
            sumx += v;
            if (v > m) { s = s * __expf(m - v) + 1.f; m = v; } else { s += __expf(v - m); }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o), s2 = __shfl_xor(s, o);
        online_merge(m, s, m2, s2);
        sumx += __shfl_xor(sumx, o);
    }
    if (lane == 0) { sm[wave] = m; ss[wave] = s; sx[wave] = sumx; }
    __syncthreads();
    m = sm[0]; s = ss[0]; sumx = sx[0];
    for (int w = 1; w < 4; ++w) { online_merge(m, s, sm[w], ss[w]); sumx += sx[w]; }
    const float lse = m + logf(s);
    const long y = target[row];
    const float eps_p = (V > 1) ? smoothing / (float)(V - 1) : 0.f;
    const float conf = 1.f - smoothing;
    if (tid == 0) {
        const float logp_y = to_f32(x[y]) - lse;
        const float sum_logp = sumx - (float)V * lse;
        row_loss[row] = -(conf * logp_y + eps_p * (sum_logp - logp_y));
    }
    if (write_grad) {
        __syncthreads();
        if constexpr (VEC) {
            for (int c = tid * CN; c < (int)ld; c += 256 * CN) {
                float v[CN];
                if (c < V) RowChunk<T>::load(x + c, v);
#pragma unroll
                for (int j = 0; j < CN; ++j)
                    v[j] = (c + j < V) ? gscale * (__expf(v[j] - lse) - ((c + j) == y ? conf : eps_p)) : 0.f;
                RowChunk<T>::store(x + c, v);
            }
        } else {
            for (int c = tid; c < (int)ld; c += 256) {
                float g = 0.f;
                if (c < V) {
                    const float pr = __expf(to_f32(x[c]) - lse);
                    g = gscale * (pr - (c == y ? conf : eps_p));
                }
                x[c] = from_f32<T>(g);
            }
        }
    }
}

// The same loss with the row held in REGISTERS between the two passes (bf16, 16-byte rows, ld <= 256 * 8 * MAXC): a thread keeps its
// MAXC chunks of the row packed as they were loaded (4 registers each: 100 of 50,265 logits per thread at MAXC = 25), so the row is read
// from memory exactly once -- the second pass of ls_loss_kernel re-reads 14.8 GB per step from L2 / Infinity Cache -- and the maximum is
// known before the first exponential (no online rescaling).  Columns past V are patched to -inf when they are loaded: no pass carries a
// per-element column test (200 lane masks per thread otherwise).
template <int MAXC>
__global__ __launch_bounds__(256, 2) void ls_loss_reg_kernel(bf16_t* __restrict__ logits, long ld, const int64_t* __restrict__ target,
                                                             float* __restrict__ row_loss, int V, float smoothing, float gscale, int write_grad) {
    __shared__ float sm[4], ss[4], sx[4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bf16_t* x = logits + (long)row * ld;
    const uint32_t NINF2 = 0xff80ff80u;                        // two bf16 -inf
    u32x4_t reg[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = (i * 256 + tid) * 8;
        reg[i] = u32x4_t{NINF2, NINF2, NINF2, NINF2};
        if (c < V) reg[i] = *reinterpret_cast<const u32x4_t*>(x + c);
        if (c < V && c + 8 > V) {                              // the chunk that straddles V (one thread of the row)
            bf16_t t[8];
            __builtin_memcpy(t, &reg[i], 16);
            for (int k = 0; k < 8; ++k)
                if (c + k >= V) t[k] = (bf16_t)(-INFINITY);
            __builtin_memcpy(&reg[i], t, 16);
        }
    }
    float m = -INFINITY, sumx = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        bf16_t t[8];
        __builtin_memcpy(t, &reg[i], 16);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float v = (float)t[k];
            m = fmaxf(m, v);
            sumx += (v == -INFINITY) ? 0.f : v;
        }
        __builtin_amdgcn_sched_barrier(0);                     // chunk by chunk: do not hold 200 converted floats
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { m = fmaxf(m, __shfl_xor(m, o)); sumx += __shfl_xor(sumx, o); }
    if (lane == 0) { sm[wave] = m; sx[wave] = sumx; }
    __syncthreads();
    m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    sumx = (sx[0] + sx[1]) + (sx[2] + sx[3]);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        bf16_t t[8];
        __builtin_memcpy(t, &reg[i], 16);
#pragma unroll
        for (int k = 0; k < 8; ++k) s += __expf((float)t[k] - m);       // -inf -> 0
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) ss[wave] = s;
    __syncthreads();
    s = (ss[0] + ss[1]) + (ss[2] + ss[3]);
    const float lse = m + logf(s);
    const int y = (int)target[row];
    const float eps_p = (V > 1) ? smoothing / (float)(V - 1) : 0.f;
    const float conf = 1.f - smoothing;
    if (tid == 0) {
        const float logp_y = to_f32(x[y]) - lse;
        const float sum_logp = sumx - (float)V * lse;
        row_loss[row] = -(conf * logp_y + eps_p * (sum_logp - logp_y));
    }
    if (write_grad) {
        __syncthreads();                                       // x[y] above is read before anybody overwrites the row
        const float ge = gscale * eps_p;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = (i * 256 + tid) * 8;
            if (c < (int)ld) {
                bf16_t t[8];
                __builtin_memcpy(t, &reg[i], 16);
                float g[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float v = (float)t[k];
                    g[k] = (v == -INFINITY) ? 0.f : fmaf(gscale, __expf(v - lse), -ge);
                }
                if (y >= c && y < c + 8) {                     // the target's column: (p - conf) instead of (p - eps)
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (c + k == y) g[k] -= gscale * (conf - eps_p);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) t[k] = (bf16_t)g[k];
                u32x4_t w;
                __builtin_memcpy(&w, t, 16);
                *reinterpret_cast<u32x4_t*>(x + c) = w;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

__global__ void segment_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int seg, float scale) {
    __shared__ double red[256];
    const float* p = x + (long)blockIdx.x * seg;
    double s = 0.0;
    for (int i = threadIdx.x; i < seg; i += 256) s += (double)p[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = (float)(red[0] * (double)scale);
}

// --------------------------------------------------------------------------------------------
// Column sums: grid (C/64 column groups, row splits); partials in workspace, then a finish pass.
// --------------------------------------------------------------------------------------------
constexpr int CS_SPLITS = 256;           // most row splits of a column-sum pass (workspace = CS_SPLITS x C floats)
// block = 16 column groups (4 columns each, one 8/16-byte load) x 16 row lanes; grid (C/64, splits)
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ X, long ld, int R, int C, float* __restrict__ part,
                                                             const int* __restrict__ live) {
    R = live_rows_of(R, live);
    __shared__ float red[16][64];
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int col = blockIdx.x * 64 + cg * 4;
    const int rows_per = (R + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * rows_per, r1 = min(R, r0 + rows_per);
    f32x4_t s = f32x4_t{0, 0, 0, 0};
    if (col + 4 <= C) {
        for (int r = r0 + rl; r < r1; r += 16) s = s + load4<T>(X + (long)r * ld + col);
    } else if (col < C) {
        for (int r = r0 + rl; r < r1; r += 16)
            for (int j = 0; j < 4 && col + j < C; ++j) s[j] += to_f32(X[(long)r * ld + col + j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[rl][cg * 4 + j] = s[j];
    __syncthreads();
    if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
        part[(long)blockIdx.y * C + blockIdx.x * 64 + threadIdx.x] = t;
    }
}
// bf16 with 16-byte addressable rows: block = 16 column groups (8 columns, one 16-byte load) x 16 row lanes, four rows in
// flight per thread; grid (C/128, splits)
__global__ __launch_bounds__(256) void colsum_partial_wide_kernel(const bf16_t* __restrict__ X, long ld, int R, int C, float* __restrict__ part,
                                                                  const int* __restrict__ live) {
    R = live_rows_of(R, live);
    __shared__ float red[16][128];
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int col = blockIdx.x * 128 + cg * 8;
    const int rows_per = (R + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * rows_per, r1 = min(R, r0 + rows_per);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (col < C) {                                           // C % 8 == 0 on this path
        int r = r0 + rl;
        for (; r + 48 < r1; r += 64) {
            bf16x8_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const bf16x8_t*>(X + (long)(r + 16 * u) * ld + col);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] += (float)v[u][j];
        }
        for (; r < r1; r += 16) {
            const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(X + (long)r * ld + col);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += (float)v[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[rl][cg * 8 + j] = s[j];
    __syncthreads();
    if (threadIdx.x < 128 && blockIdx.x * 128 + threadIdx.x < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
        part[(long)blockIdx.y * C + blockIdx.x * 128 + threadIdx.x] = t;
    }
}
// block = 16 columns x 16 split lanes: 16 independent loads per thread and round trip
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ part, int splits, int C, float* __restrict__ out, int accumulate) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + cl;
    float s = 0.f;
    if (col < C) {
#pragma unroll 8
        for (int k = sl; k < splits; k += 16) s += part[(long)k * C + col];
    }
    red[sl][cl] = s;
    __syncthreads();
    if (sl == 0 && col < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cl];
        out[col] = accumulate ? out[col] + t : t;
    }
}

// --------------------------------------------------------------------------------------------
// Optimiser kernels over the flat f32 arena
// --------------------------------------------------------------------------------------------
constexpr int L2_BLOCKS = 1024;
__global__ __launch_bounds__(256) void l2_partial_kernel(const float* __restrict__ g, long n, double* __restrict__ part) {
    __shared__ double red[256];
    double s = 0.0;
    const long n4 = n >> 2;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(g + i * 4);
        s += (double)(v[0] * v[0] + v[1] * v[1]) + (double)(v[2] * v[2] + v[3] * v[3]);
    }
    if (blockIdx.x == 0)
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) s += (double)g[i] * g[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ void l2_finish_kernel(const double* __restrict__ part, int nparts, float* __restrict__ out, int accumulate) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + (float)red[0] : (float)red[0];
}

__device__ __forceinline__ float clip_coef(const float* norm_sq, float max_norm) {
    if (norm_sq == nullptr || max_norm <= 0.f) return 1.f;
    const float c = max_norm / (sqrtf(norm_sq[0]) + 1e-6f);
    return c < 1.f ? c : 1.f;
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ shadow, long n,
                                                    const float* __restrict__ hyper, const float* __restrict__ norm_sq,
                                                    float beta1, float beta2, float eps) {
    const float step_size = hyper[0], lr_wd = hyper[1];
    const float cc = clip_coef(norm_sq, hyper[2]);
    const long n4 = n >> 2;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4_t pv = *reinterpret_cast<f32x4_t*>(p + i * 4);
        const f32x4_t gv = *reinterpret_cast<const f32x4_t*>(g + i * 4) * cc;
        f32x4_t mv = *reinterpret_cast<f32x4_t*>(m + i * 4), vv = *reinterpret_cast<f32x4_t*>(v + i * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            mv[j] = mv[j] * beta1 + gv[j] * (1.f - beta1);
            vv[j] = vv[j] * beta2 + gv[j] * gv[j] * (1.f - beta2);
            pv[j] = pv[j] - step_size * (mv[j] / (sqrtf(vv[j]) + eps));
            pv[j] = pv[j] - lr_wd * pv[j];
        }
        *reinterpret_cast<f32x4_t*>(p + i * 4) = pv;
        *reinterpret_cast<f32x4_t*>(m + i * 4) = mv;
        *reinterpret_cast<f32x4_t*>(v + i * 4) = vv;
        if (shadow) *reinterpret_cast<bf16x4_t*>(shadow + i * 4) = bf16x4_t{(bf16_t)pv[0], (bf16_t)pv[1], (bf16_t)pv[2], (bf16_t)pv[3]};
    }
    if (blockIdx.x == 0)
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
            const float gg = g[i] * cc;
            const float mm = m[i] * beta1 + gg * (1.f - beta1);
            const float vv = v[i] * beta2 + gg * gg * (1.f - beta2);
            float pp = p[i] - step_size * (mm / (sqrtf(vv) + eps));
            pp -= lr_wd * pp;
            p[i] = pp; m[i] = mm; v[i] = vv;
            if (shadow) shadow[i] = (bf16_t)pp;
        }
}

__global__ __launch_bounds__(256) void scale_by_clip_kernel(float* __restrict__ g, long n, const float* __restrict__ norm_sq, float max_norm) {
    const float cc = clip_coef(norm_sq, max_norm);
    if (cc >= 1.f) return;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) g[i] *= cc;
}

template <typename TD, typename TS>
__global__ __launch_bounds__(256) void cast_kernel(TD* __restrict__ dst, const TS* __restrict__ src, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = from_f32<TD>(to_f32(src[i]));
}

inline int grid_for(long work_items, int per_block, int cap = 2048) {
    long b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return (int)(b > cap ? cap : b);
}
inline int ok() { return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP; }

template <typename F>
int dispatch_vpl(int D, F&& f) {
    switch (D) {
        case 256: f(std::integral_constant<int, 1>{}); break;
        case 512: f(std::integral_constant<int, 2>{}); break;
        case 768: f(std::integral_constant<int, 3>{}); break;
        case 1024: f(std::integral_constant<int, 4>{}); break;
        default: return MMSUM_ERR_BAD_SHAPE;
    }
    return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
}

template <typename T>
int add_ln_fwd_t(const void* x, const void* res, const void* gamma, const void* beta, void* y, float* mean, float* rstd, int R,
                 int D, float eps, float p_drop, uint64_t seed, const uint64_t* salt, const int* live, float* y32, hipStream_t s) {
    const int grid = (R + 3) / 4 > 2048 ? 2048 : (R + 3) / 4;
    return dispatch_vpl(D, [&](auto vpl) {
        constexpr int VPL = decltype(vpl)::value;
        add_ln_fwd_kernel<T, VPL><<<dim3(grid), dim3(256), 0, s>>>((const T*)x, (const T*)res, (const float*)gamma, (const float*)beta,
                                                                  (T*)y, mean, rstd, R, D, eps, p_drop, seed, salt, live, y32);
    });
}
template <typename T>
int add_ln_bwd_t(const void* dy, const void* x, const void* res, const void* gamma, const float* mean, const float* rstd, void* dx,
                 void* dres, int accumulate_dres, float* dgamma, float* dbeta, int R, int D, float p_drop, uint64_t seed,
                 const uint64_t* salt, float* dxsum, const int* live, hipStream_t s) {
    int grid = (R + 15) / 16;
    grid = grid > 1024 ? 1024 : grid;
    return dispatch_vpl(D, [&](auto vpl) {
        constexpr int VPL = decltype(vpl)::value;
        add_ln_bwd_kernel<T, VPL><<<dim3(grid), dim3(256), 0, s>>>((const T*)dy, (const T*)x, (const T*)res, (const float*)gamma, mean,
                                                                  rstd, (T*)dx, (T*)dres, accumulate_dres, dgamma, dbeta, R, D, p_drop,
                                                                  seed, salt, dxsum, live);
    });
}
template <typename T>
int embed_ln_fwd_t(const int64_t* ids, const void* E, const void* P, const float* rd, const void* rvec, const void* gamma,
                   const void* beta, void* y, float* mean, float* rstd, int R, int T_len, int D, int pos_offset, float eps,
                   float p_drop, uint64_t seed, const uint64_t* salt, hipStream_t s) {
    const int grid = (R + 3) / 4 > 2048 ? 2048 : (R + 3) / 4;
    return dispatch_vpl(D, [&](auto vpl) {
        constexpr int VPL = decltype(vpl)::value;
        embed_ln_fwd_kernel<T, VPL><<<dim3(grid), dim3(256), 0, s>>>(ids, (const T*)E, (const T*)P, rd, (const T*)rvec,
                                                                    (const float*)gamma, (const float*)beta, (T*)y, mean, rstd, R,
                                                                    T_len, D, pos_offset, eps, p_drop, seed, salt);
    });
}
template <typename T>
int embed_ln_bwd_t(const void* dy, const int64_t* ids, const void* E, const void* P, const float* rd, const void* rvec,
                   const void* gamma, const float* mean, const float* rstd, float* dE, float* dP, float* drvec, float* dgamma,
                   float* dbeta, int nseq, int T_len, int D, int pos_offset, int pad_id, float p_drop, uint64_t seed, const uint64_t* salt,
                   hipStream_t s) {
    return dispatch_vpl(D, [&](auto vpl) {
        constexpr int VPL = decltype(vpl)::value;
        const int chunks = nseq >= 256 ? 8 : nseq >= 64 ? 4 : 1;
        embed_ln_bwd_kernel<T, VPL><<<dim3(T_len, chunks), dim3(256), 0, s>>>((const T*)dy, ids, (const T*)E, (const T*)P, rd, (const T*)rvec,
                                                                     (const float*)gamma, mean, rstd, dE, dP, drvec, dgamma, dbeta,
                                                                     nseq, T_len, D, pos_offset, pad_id, p_drop, seed, salt);
    });
}

}  // namespace

extern "C" int mmsum_abi_version(void) { return MMSUM_ABI_VERSION; }

extern "C" int mmsum_add_ln_fwd(int dtype, const void* x, const void* res, const void* gamma, const void* beta, void* y,
                                float* mean, float* rstd, int R, int D, float eps, float p_drop, uint64_t seed, const void* salt,
                                const int* live_rows, float* y_f32, void* stream) {
    if (R <= 0) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const uint64_t* sp = static_cast<const uint64_t*>(salt);
    if (dtype == MMSUM_BF16) return add_ln_fwd_t<bf16_t>(x, res, gamma, beta, y, mean, rstd, R, D, eps, p_drop, seed, sp, live_rows, y_f32, s);
    if (dtype == MMSUM_F32) return add_ln_fwd_t<float>(x, res, gamma, beta, y, mean, rstd, R, D, eps, p_drop, seed, sp, live_rows, y_f32, s);
    return MMSUM_ERR_BAD_DTYPE;
}

extern "C" int mmsum_add_ln_bwd(int dtype, const void* dy, const void* x, const void* res, const void* gamma, const float* mean,
                                const float* rstd, void* dx, void* dres, int accumulate_dres, float* dgamma, float* dbeta, int R,
                                int D, float p_drop, uint64_t seed, const void* salt, float* dxsum, const int* live_rows, void* stream) {
    if (R <= 0) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const uint64_t* sp = static_cast<const uint64_t*>(salt);
    if (dtype == MMSUM_BF16) return add_ln_bwd_t<bf16_t>(dy, x, res, gamma, mean, rstd, dx, dres, accumulate_dres, dgamma, dbeta, R, D, p_drop, seed, sp, dxsum, live_rows, s);
    if (dtype == MMSUM_F32) return add_ln_bwd_t<float>(dy, x, res, gamma, mean, rstd, dx, dres, accumulate_dres, dgamma, dbeta, R, D, p_drop, seed, sp, dxsum, live_rows, s);
    return MMSUM_ERR_BAD_DTYPE;
}

extern "C" int mmsum_embed_ln_fwd(int dtype, const int64_t* ids, const void* E, const void* P, const float* rating_diff,
                                  const void* rvec, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                                  int nseq, int T, int D, int pos_offset, float eps, float p_drop, uint64_t seed, const void* salt,
                                  void* stream) {
    const int R = nseq * T;
    if (R <= 0) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const uint64_t* sp = static_cast<const uint64_t*>(salt);
    if (dtype == MMSUM_BF16) return embed_ln_fwd_t<bf16_t>(ids, E, P, rating_diff, rvec, gamma, beta, y, mean, rstd, R, T, D, pos_offset, eps, p_drop, seed, sp, s);
    if (dtype == MMSUM_F32) return embed_ln_fwd_t<float>(ids, E, P, rating_diff, rvec, gamma, beta, y, mean, rstd, R, T, D, pos_offset, eps, p_drop, seed, sp, s);
    return MMSUM_ERR_BAD_DTYPE;
}

extern "C" int mmsum_embed_ln_bwd(int dtype, const void* dy, const int64_t* ids, const void* E, const void* P,
                                  const float* rating_diff, const void* rvec, const void* gamma, const float* mean,
                                  const float* rstd, float* dE, float* dP, float* drvec, float* dgamma, float* dbeta, int nseq,
                                  int T, int D, int pos_offset, int pad_id, float p_drop, uint64_t seed, const void* salt, void* stream) {
    if (nseq <= 0 || T <= 0) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const uint64_t* sp = static_cast<const uint64_t*>(salt);
    if (dtype == MMSUM_BF16) return embed_ln_bwd_t<bf16_t>(dy, ids, E, P, rating_diff, rvec, gamma, mean, rstd, dE, dP, drvec, dgamma, dbeta, nseq, T, D, pos_offset, pad_id, p_drop, seed, sp, s);
    if (dtype == MMSUM_F32) return embed_ln_bwd_t<float>(dy, ids, E, P, rating_diff, rvec, gamma, mean, rstd, dE, dP, drvec, dgamma, dbeta, nseq, T, D, pos_offset, pad_id, p_drop, seed, sp, s);
    return MMSUM_ERR_BAD_DTYPE;
}

extern "C" int mmsum_gate_add_ln_fwd(int dtype, const void* pa, const void* pb, const void* yt, const void* ytab, const void* yimg,
                                     const uint8_t* no_table, const uint8_t* no_img, const void* res, const void* gamma, const void* beta,
                                     void* y, int R, int D, int rows_per_b, float eps, void* stream) {
    if (R <= 0 || rows_per_b <= 0) return MMSUM_ERR_BAD_SHAPE;
    if (dtype != MMSUM_BF16 && dtype != MMSUM_F32) return MMSUM_ERR_BAD_DTYPE;
    hipStream_t s = (hipStream_t)stream;
    const int grid = (R + 3) / 4 > 2048 ? 2048 : (R + 3) / 4;
    return dispatch_vpl(D, [&](auto vpl) {
        constexpr int VPL = decltype(vpl)::value;
        if (dtype == MMSUM_BF16)
            gate_add_ln_fwd_kernel<bf16_t, VPL><<<dim3(grid), dim3(256), 0, s>>>((const bf16_t*)pa, (const bf16_t*)pb, (const bf16_t*)yt, (const bf16_t*)ytab,
                (const bf16_t*)yimg, no_table, no_img, (const bf16_t*)res, (const float*)gamma, (const float*)beta, (bf16_t*)y, R, D, rows_per_b, eps);
        else
            gate_add_ln_fwd_kernel<float, VPL><<<dim3(grid), dim3(256), 0, s>>>((const float*)pa, (const float*)pb, (const float*)yt, (const float*)ytab,
                (const float*)yimg, no_table, no_img, (const float*)res, (const float*)gamma, (const float*)beta, (float*)y, R, D, rows_per_b, eps);
    });
}

extern "C" int mmsum_gate_fwd(int dtype, const void* pa, const void* pb, const void* yt, const void* ytab, const void* yimg,
                              const uint8_t* no_table, const uint8_t* no_img, void* out, int R, int D, int rows_per_b, void* stream) {
    if (R <= 0 || D % 4) return MMSUM_ERR_BAD_SHAPE;
    const long n4 = (long)R * D / 4;
    hipStream_t s = (hipStream_t)stream;
    const int grid = grid_for(n4, 256);
    if (dtype == MMSUM_BF16) hipLaunchKernelGGL((gate_fwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, (const bf16_t*)pa, (const bf16_t*)pb, (const bf16_t*)yt, (const bf16_t*)ytab, (const bf16_t*)yimg, no_table, no_img, (bf16_t*)out, n4, D, rows_per_b);
    else if (dtype == MMSUM_F32) hipLaunchKernelGGL((gate_fwd_kernel<float>), dim3(grid), dim3(256), 0, s, (const float*)pa, (const float*)pb, (const float*)yt, (const float*)ytab, (const float*)yimg, no_table, no_img, (float*)out, n4, D, rows_per_b);
    else return MMSUM_ERR_BAD_DTYPE;
    return ok();
}

template <typename T>
static int gate_bwd_sums_t(const void* dout, const void* pa, const void* pb, const void* ytab, const void* yimg, const uint8_t* no_table,
                           const uint8_t* no_img, void* dpa, void* dpb, void* dyt, void* dytab, void* dyimg, int R, int D, int rows_per_b,
                           float* s0, float* s1, hipStream_t s) {
    const int grid = R < 512 ? R : 512;            // 512 blocks x 2 D atomics: the additions stay far below the kernel's own time
    const int vpl = (D / 4 + 255) / 256;
#define GATE_CASE(V) if (vpl == V) { hipLaunchKernelGGL((gate_bwd_sums_kernel<T, V>), dim3(grid), dim3(256), 0, s, (const T*)dout, (const T*)pa, (const T*)pb, \
        (const T*)ytab, (const T*)yimg, no_table, no_img, (T*)dpa, (T*)dpb, (T*)dyt, (T*)dytab, (T*)dyimg, R, D, rows_per_b, s0, s1); return ok(); }
    GATE_CASE(1) GATE_CASE(2) GATE_CASE(3) GATE_CASE(4)
#undef GATE_CASE
    return MMSUM_ERR_BAD_SHAPE;
}

extern "C" int mmsum_gate_bwd(int dtype, const void* dout, const void* pa, const void* pb, const void* ytab, const void* yimg,
                              const uint8_t* no_table, const uint8_t* no_img, void* dpa, void* dpb, void* dyt, void* dytab,
                              void* dyimg, int R, int D, int rows_per_b, float* sum_dpa, float* sum_dpb, void* stream) {
    if (R <= 0 || D % 4) return MMSUM_ERR_BAD_SHAPE;
    const long n4 = (long)R * D / 4;
    hipStream_t s = (hipStream_t)stream;
    if (sum_dpa != nullptr || sum_dpb != nullptr) {
        if (!(sum_dpa && sum_dpb) || D > 4096) return MMSUM_ERR_BAD_SHAPE;                 // both or none
        if (dtype == MMSUM_BF16) return gate_bwd_sums_t<bf16_t>(dout, pa, pb, ytab, yimg, no_table, no_img, dpa, dpb, dyt, dytab, dyimg, R, D, rows_per_b, sum_dpa, sum_dpb, s);
        if (dtype == MMSUM_F32) return gate_bwd_sums_t<float>(dout, pa, pb, ytab, yimg, no_table, no_img, dpa, dpb, dyt, dytab, dyimg, R, D, rows_per_b, sum_dpa, sum_dpb, s);
        return MMSUM_ERR_BAD_DTYPE;
    }
    const int grid = grid_for(n4, 256);
    if (dtype == MMSUM_BF16) hipLaunchKernelGGL((gate_bwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, (const bf16_t*)dout, (const bf16_t*)pa, (const bf16_t*)pb, (const bf16_t*)ytab, (const bf16_t*)yimg, no_table, no_img, (bf16_t*)dpa, (bf16_t*)dpb, (bf16_t*)dyt, (bf16_t*)dytab, (bf16_t*)dyimg, n4, D, rows_per_b);
    else if (dtype == MMSUM_F32) hipLaunchKernelGGL((gate_bwd_kernel<float>), dim3(grid), dim3(256), 0, s, (const float*)dout, (const float*)pa, (const float*)pb, (const float*)ytab, (const float*)yimg, no_table, no_img, (float*)dpa, (float*)dpb, (float*)dyt, (float*)dytab, (float*)dyimg, n4, D, rows_per_b);
    else return MMSUM_ERR_BAD_DTYPE;
    return ok();
}

extern "C" int mmsum_ls_loss(int dtype, void* logits, long ld, const int64_t* target, float* row_loss, int R, int V,
                             float smoothing, float gscale, int write_grad, void* stream) {
    if (R <= 0 || V <= 0 || ld < V) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const size_t es = dtype == MMSUM_BF16 ? 2 : 4;
    // 16-byte path: aligned rows whose padded length is a whole number of chunks (the engine pads V to 128 columns)
    const bool vec = (((uintptr_t)logits) & 15) == 0 && ((ld * es) & 15) == 0;
    if (dtype == MMSUM_BF16) {
        constexpr int MAXC = 25;                               // 256 threads x 8 logits x 25 chunks = 51,200 >= the 50,304-column rows of BART's vocabulary
        if (vec && ld <= 256L * 8 * MAXC) hipLaunchKernelGGL((ls_loss_reg_kernel<MAXC>), dim3(R), dim3(256), 0, s, (bf16_t*)logits, ld, target, row_loss, V, smoothing, gscale, write_grad);
        else if (vec) hipLaunchKernelGGL((ls_loss_kernel<bf16_t, true>), dim3(R), dim3(256), 0, s, (bf16_t*)logits, ld, target, row_loss, V, smoothing, gscale, write_grad);
        else hipLaunchKernelGGL((ls_loss_kernel<bf16_t, false>), dim3(R), dim3(256), 0, s, (bf16_t*)logits, ld, target, row_loss, V, smoothing, gscale, write_grad);
    } else if (dtype == MMSUM_F32) {
        if (vec) hipLaunchKernelGGL((ls_loss_kernel<float, true>), dim3(R), dim3(256), 0, s, (float*)logits, ld, target, row_loss, V, smoothing, gscale, write_grad);
        else hipLaunchKernelGGL((ls_loss_kernel<float, false>), dim3(R), dim3(256), 0, s, (float*)logits, ld, target, row_loss, V, smoothing, gscale, write_grad);
    } else return MMSUM_ERR_BAD_DTYPE;
    return ok();
}

extern "C" int mmsum_segment_sum(const float* x, float* out, int nseg, int seg, float scale, void* stream) {
    if (nseg <= 0 || seg <= 0) return MMSUM_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(segment_sum_kernel, dim3(nseg), dim3(256), 0, (hipStream_t)stream, x, out, seg, scale);
    return ok();
}

extern "C" long mmsum_colsum_workspace(int C) { return (long)CS_SPLITS * C * sizeof(float); }
extern "C" int mmsum_colsum(int dtype, const void* X, long ld, int R, int C, float* out, int accumulate, void* workspace,
                            const int* live_rows, void* stream) {
    if (R <= 0 || C <= 0) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    float* part = (float*)workspace;
    const bool wide = dtype == MMSUM_BF16 && C % 8 == 0 && ((uintptr_t)X & 15) == 0 && (ld * 2) % 16 == 0;
    // about 2048 workgroups in all, at least 64 rows per split
    const int cblocks = wide ? (C + 127) / 128 : (C + 63) / 64;
    int splits = (2048 + cblocks - 1) / cblocks;
    if (splits > CS_SPLITS) splits = CS_SPLITS;
    if (splits > (R + 63) / 64) splits = (R + 63) / 64;
    if (splits < 1) splits = 1;
    const dim3 grid(cblocks, splits);
    if (wide) hipLaunchKernelGGL(colsum_partial_wide_kernel, grid, dim3(256), 0, s, (const bf16_t*)X, ld, R, C, part, live_rows);
    else if (dtype == MMSUM_BF16) hipLaunchKernelGGL((colsum_partial_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)X, ld, R, C, part, live_rows);
    else if (dtype == MMSUM_F32) hipLaunchKernelGGL((colsum_partial_kernel<float>), grid, dim3(256), 0, s, (const float*)X, ld, R, C, part, live_rows);
    else return MMSUM_ERR_BAD_DTYPE;
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((C + 15) / 16), dim3(256), 0, s, part, splits, C, out, accumulate);
    return ok();
}

extern "C" long mmsum_l2_workspace(void) { return (long)L2_BLOCKS * sizeof(double); }
extern "C" int mmsum_l2norm_sq(const float* g, long n, float* out, int accumulate, void* workspace, void* stream) {
    if (n <= 0) return MMSUM_ERR_BAD_SHAPE;
    if ((uintptr_t)g & 15) return MMSUM_ERR_BAD_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const int blocks = grid_for(n / 4 + 1, 256, L2_BLOCKS);
    hipLaunchKernelGGL(l2_partial_kernel, dim3(blocks), dim3(256), 0, s, g, n, (double*)workspace);
    hipLaunchKernelGGL(l2_finish_kernel, dim3(1), dim3(256), 0, s, (const double*)workspace, blocks, out, accumulate);
    return ok();
}

extern "C" int mmsum_adamw(float* p, const float* g, float* m, float* v, void* shadow_bf16, long n, const float* hyper,
                           const float* norm_sq, float beta1, float beta2, float eps, void* stream) {
    if (n <= 0) return MMSUM_ERR_BAD_SHAPE;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return MMSUM_ERR_BAD_ALIGN;
    if ((uintptr_t)shadow_bf16 & 7) return MMSUM_ERR_BAD_ALIGN;
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)shadow_bf16, n, hyper, norm_sq, beta1, beta2, eps);
    return ok();
}

extern "C" int mmsum_scale_by_clip(float* g, long n, const float* norm_sq, float max_norm, void* stream) {
    if (n <= 0) return MMSUM_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(scale_by_clip_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, g, n, norm_sq, max_norm);
    return ok();
}

extern "C" int mmsum_cast(int dtype_dst, void* dst, int dtype_src, const void* src, long n, void* stream) {
    if (n <= 0) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int grid = grid_for(n, 256);
    if (dtype_dst == MMSUM_BF16 && dtype_src == MMSUM_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(grid), dim3(256), 0, s, (bf16_t*)dst, (const float*)src, n);
    else if (dtype_dst == MMSUM_F32 && dtype_src == MMSUM_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(grid), dim3(256), 0, s, (float*)dst, (const bf16_t*)src, n);
    else if (dtype_dst == MMSUM_F32 && dtype_src == MMSUM_F32) hipLaunchKernelGGL((cast_kernel<float, float>), dim3(grid), dim3(256), 0, s, (float*)dst, (const float*)src, n);
    else if (dtype_dst == MMSUM_BF16 && dtype_src == MMSUM_BF16) hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(grid), dim3(256), 0, s, (bf16_t*)dst, (const bf16_t*)src, n);
    else return MMSUM_ERR_BAD_DTYPE;
    return ok();
}

namespace {
// Row compaction / expansion for the padding-free text encoder: dst[i] = src[map[i]] (zeros where map[i] < 0), 16-byte pieces.
__global__ __launch_bounds__(256) void rows_gather_kernel(const char* __restrict__ src, long src_pitch, char* __restrict__ dst, long dst_pitch,
                                                          const int64_t* __restrict__ map, int nrows, int row_bytes, int src_rows,
                                                          const int* __restrict__ live) {
    nrows = live_rows_of(nrows, live);
    const int chunks = row_bytes >> 4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < (long)nrows * chunks; i += (long)gridDim.x * 256) {
        const int r = (int)(i / chunks), c = (int)(i % chunks);
        const long m = map[r];
        u32x4_t v = u32x4_t{0, 0, 0, 0};
        if (m >= 0 && m < src_rows) v = *reinterpret_cast<const u32x4_t*>(src + m * src_pitch + (long)c * 16);
        *reinterpret_cast<u32x4_t*>(dst + (long)r * dst_pitch + (long)c * 16) = v;
    }
}
}  // namespace

/* dst[i, :] = map[i] >= 0 ? src[map[i], :] : 0 for i < nrows; rows are row_bytes long (multiple of 16), pitches in bytes.
 * One kernel serves both directions: compact (map = compact -> padded row) and expand (map = padded -> compact row). */
extern "C" int mmsum_rows_gather(const void* src, long src_pitch, int src_rows, void* dst, long dst_pitch, const int64_t* map, int nrows,
                                 int row_bytes, const int* live_rows, void* stream) {
    if (nrows <= 0 || row_bytes <= 0 || (row_bytes & 15) || (src_pitch & 15) || (dst_pitch & 15)) return MMSUM_ERR_BAD_SHAPE;
    if ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) return MMSUM_ERR_BAD_ALIGN;
    const long items = (long)nrows * (row_bytes >> 4);
    long blocks = (items + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    rows_gather_kernel<<<dim3((int)blocks), dim3(256), 0, (hipStream_t)stream>>>((const char*)src, src_pitch, (char*)dst, dst_pitch, map, nrows,
                                                                              row_bytes, src_rows, live_rows);
    return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
}

extern "C" int mmsum_bump_u64(void* dev_u64, unsigned long long inc, void* stream) {
    if (dev_u64 == nullptr || (((uintptr_t)dev_u64) & 7)) return MMSUM_ERR_BAD_ALIGN;
    bump_u64_kernel<<<dim3(1), dim3(1), 0, (hipStream_t)stream>>>(static_cast<uint64_t*>(dev_u64), (uint64_t)inc);
    return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
}
