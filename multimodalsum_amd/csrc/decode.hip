// Beam-search decode-step kernels: what one step of
// _generate_beam_search (modeling_multimodalsum.py:2857-3010) does on [rows, V] logits and on the self-attention caches.
//
//   beam_topk_rows / beam_topk_merge : adjust_logits (forced BOS / EOS, :3084-3089) -> log_softmax (:2874) -> min-length EOS
//       ban and no-repeat-n-gram bans (generation_utils.py:57-98, 848-868) -> + beam score -> top 2*num_beams over the
//       num_beams * V candidates of a business (:2925), as two launches that read the logits twice from L2 and write
//       2 * num_beams (score, index) pairs per business.  The reference materialises four [rows, V] f32 tensors per step.
//   decode_self_attn : single-query self-attention over the K/V caches THROUGH an ancestor table, so the beam reorder of the
//       reference (_reorder_cache :3104-3115: index_select of every layer's cache, every step) is a copy of the table
//       (rows * max_length int32) instead of 2 * layers copies of [rows, max_length, D].
#include "mmsum_device.h"
#include "mmsum_kernels.h"

namespace {

constexpr int TK_MAX = 16;            // 2 * num_beams, num_beams <= 8
constexpr int TK_THREADS = 256;

struct Cand { float v; int tok; };
// ordering of candidates: higher value first, lower token index first among equal values (deterministic; the reference's
// torch.topk leaves ties unspecified)
__device__ __forceinline__ bool better(float v, int tok, float v2, int tok2) { return v > v2 || (v == v2 && tok < tok2); }

template <typename T> __device__ __forceinline__ float ldf(const T* p, long i) { return to_f32(p[i]); }

// One block per hypothesis row.  out_v/out_t [rows, K]: the row's K best (log-prob + beam score, token), best first.
template <typename T>
__global__ __launch_bounds__(TK_THREADS) void beam_topk_rows_kernel(T* __restrict__ logits, long ld, int V, const float* __restrict__ beam_scores,
                                                                    const int* __restrict__ banned, int nban, int force_token, int ban_token, int K,
                                                                    float* __restrict__ out_v, int* __restrict__ out_t) {
    __shared__ float red_m[TK_THREADS / 64], red_s[TK_THREADS / 64];
    __shared__ float win_v[TK_THREADS / 64];
    __shared__ int win_t[TK_THREADS / 64], win_w[TK_THREADS / 64];
    __shared__ float s_lse;
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    T* x = logits + (long)row * ld;
    const float bs = beam_scores[row];
    if (force_token >= 0) {
        // every other logit is -inf: log_softmax gives 0 at the forced token and -inf elsewhere (:3084-3089 then :2874)
        if (tid < K) {
            int tok = tid == 0 ? force_token : (tid - 1 < force_token ? tid - 1 : tid);
            out_v[(long)row * K + tid] = tid == 0 ? 0.f + bs : -INFINITY;
            out_t[(long)row * K + tid] = tok;
        }
        return;
    }
    // ---- pass 1: log-sum-exp of the row (online max / sum per thread, then across the block)
    float m = -INFINITY, s = 0.f;
    for (int i = tid; i < V; i += TK_THREADS) {
        const float v = ldf(x, i);
        if (v > m) { s = s * __expf(m - v) + 1.f; m = v; }
        else if (v != -INFINITY) s += __expf(v - m);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o), s2 = __shfl_xor(s, o);
        const float mm = fmaxf(m, m2);
        s = (m == -INFINITY ? 0.f : s * __expf(m - mm)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mm));
        m = mm;
    }
    if (lane == 0) { red_m[wave] = m; red_s[wave] = s; }
    __syncthreads();
    if (tid == 0) {
        float mm = red_m[0], ss = red_s[0];
#pragma unroll
        for (int w = 1; w < TK_THREADS / 64; ++w) {
            const float m2 = red_m[w], s2 = red_s[w], mx = fmaxf(mm, m2);
            ss = (mm == -INFINITY ? 0.f : ss * __expf(mm - mx)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mx));
            mm = mx;
        }
        s_lse = mm + __logf(ss);
    }
    // ---- bans apply AFTER the normalisation (:2880-2900): the banned tokens' mass stays in the log-sum-exp
    if (ban_token >= 0 && tid == 0) x[ban_token] = from_f32<T>(-INFINITY);
    if (banned != nullptr)
        for (int i = tid; i < nban; i += TK_THREADS) {
            const int t = banned[(long)row * nban + i];
            if (t >= 0 && t < V) x[t] = from_f32<T>(-INFINITY);
        }
    __threadfence_block();
    __syncthreads();
    const float lse = s_lse;
    // ---- pass 2: every thread keeps its K best (sorted, best first), then K rounds of block-wide arg-best
    Cand best[TK_MAX];
#pragma unroll
    for (int k = 0; k < TK_MAX; ++k) best[k] = Cand{-INFINITY, 0x7fffffff};
    for (int i = tid; i < V; i += TK_THREADS) {
        const float v = ldf(x, i);
        Cand last = best[0];
#pragma unroll
        for (int k = 1; k < TK_MAX; ++k) if (k == K - 1) last = best[k];       // best[K - 1] without a runtime register index
        if (better(v, i, last.v, last.tok)) {
            Cand c{v, i};
#pragma unroll
            for (int k = 0; k < TK_MAX; ++k) {           // insertion into the sorted list (the tail past K is never read)
                if (k < K && better(c.v, c.tok, best[k].v, best[k].tok)) { const Cand t = best[k]; best[k] = c; c = t; }
            }
        }
    }
    int head = 0;
    for (int r = 0; r < K; ++r) {
        Cand c = Cand{-INFINITY, 0x7fffffff};
#pragma unroll
        for (int k = 0; k < TK_MAX; ++k) if (k == head) c = best[k];          // runtime index without scratch
        float bv = c.v;
        int bt = c.tok;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(bv, o);
            const int t2 = __shfl_xor(bt, o);
            if (better(v2, t2, bv, bt)) { bv = v2; bt = t2; }
        }
        if (lane == 0) { win_v[wave] = bv; win_t[wave] = bt; }
        __syncthreads();
        float gv = win_v[0];
        int gt = win_t[0];
#pragma unroll
        for (int w = 1; w < TK_THREADS / 64; ++w) if (better(win_v[w], win_t[w], gv, gt)) { gv = win_v[w]; gt = win_t[w]; }
        if (c.tok == gt && gt != 0x7fffffff) ++head;                        // the owner of the winner moves on
        if (tid == 0) {
            out_v[(long)row * K + r] = (gv - lse) + bs;                       // log_softmax, then + beam score (:2874, :2917)
            out_t[(long)row * K + r] = gt == 0x7fffffff ? 0 : gt;
        }
        __syncthreads();
    }
}

// One thread block per business: the 2*num_beams best of its num_beams * K row candidates, by (score desc, flat index asc);
// flat index = beam * V + token as in next_scores.view(batch, num_beams * vocab) (:2920-2925).
__global__ __launch_bounds__(64) void beam_topk_merge_kernel(const float* __restrict__ cv, const int* __restrict__ ct, int num_beams, int K, int V,
                                                             float* __restrict__ out_scores, long long* __restrict__ out_ids) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = num_beams * K;                                            // <= 8 * 16 = 128 candidates: two per lane
    float v[2];
    long long id[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = lane + 64 * j;
        if (c < n) {
            const int beam = c / K;
            v[j] = cv[((long)b * num_beams + beam) * K + (c % K)];
            id[j] = (long long)beam * V + ct[((long)b * num_beams + beam) * K + (c % K)];
        } else {
            v[j] = -INFINITY;
            id[j] = 0x7fffffffffffffffLL;
        }
    }
    for (int r = 0; r < K; ++r) {
        int pick = (v[1] > v[0] || (v[1] == v[0] && id[1] < id[0])) ? 1 : 0;
        float bv = v[pick];
        long long bi = id[pick];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(bv, o);
            const long long i2 = __shfl_xor(bi, o);
            if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
        }
        if (id[pick] == bi) { v[pick] = -INFINITY; id[pick] = 0x7fffffffffffffffLL; }     // the owner retires the winner
        if (lane == 0) { out_scores[(long)b * K + r] = bv; out_ids[(long)b * K + r] = bi; }
    }
}

// One wave per (hypothesis row, head).  Key s of row r lives at cache row anc[r * Tmax + s] * Tmax + s.
template <typename T>
__global__ __launch_bounds__(64) void decode_self_attn_kernel(const T* __restrict__ q, long ldq, const T* __restrict__ kc, const T* __restrict__ vc,
                                                              long ldc, const int* __restrict__ anc, T* __restrict__ out, long ldo, int len, int Tmax,
                                                              float scale) {
    constexpr int HD = 64;
    const int r = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const T* qrow = q + (long)r * ldq + h * HD;
    float qv[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) qv[d] = to_f32(qrow[d]) * scale;        // uniform loads: every lane holds the query (the :783 scaling)
    // scores: lane owns keys lane, lane + 64, ... (Tmax <= 256: at most 4); it keeps their cache rows for the second phase
    float sc[4];
    long prow[4];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int s = lane + 64 * j;
        sc[j] = -INFINITY;
        prow[j] = 0;
        if (s < len) {
            prow[j] = ((long)anc[(long)r * Tmax + s] * Tmax + s) * ldc + h * HD;
            const T* krow = kc + prow[j];
            float acc = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc = fmaf(qv[d], to_f32(krow[d]), acc);
            sc[j] = acc;
            m = fmaxf(m, acc);
        }
    }
    m = warp_max(m);
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { sc[j] = (sc[j] == -INFINITY) ? 0.f : __expf(sc[j] - m); l += sc[j]; }
    l = warp_sum(l);
    const float inv = l > 0.f ? 1.f / l : 0.f;
    // output: lane owns dimension `lane`; probability and cache row of key s are broadcast from their owner lane (no memory
    // access on the address path), eight keys' V rows in flight at a time
    float o = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int base = 64 * j;
        if (base >= len) break;
        const int n = min(64, len - base);
        for (int s0 = 0; s0 < n; s0 += 8) {
            float p[8], v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int src = min(s0 + u, n - 1);
                p[u] = (s0 + u < n) ? __shfl(sc[j], src) : 0.f;
                const long row = __shfl(prow[j], src);
                v[u] = to_f32(vc[row + lane]);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) o = fmaf(p[u], v[u], o);
        }
    }
    out[(long)r * ldo + h * HD + lane] = from_f32<T>(o * inv);
}

}  // namespace

extern "C" int mmsum_beam_topk(int dtype, void* logits, long ld, int V, const float* beam_scores, const int* banned, int nban, int force_token,
                               int ban_token, int rows, int num_beams, float* row_scores, int* row_tokens, float* out_scores,
                               long long* out_ids, void* stream) {
    const int K = 2 * num_beams;
    if (rows <= 0 || V <= 0 || ld < V || num_beams < 1 || K > TK_MAX || rows % num_beams || V < K || nban < 0) return MMSUM_ERR_BAD_SHAPE;
    if (force_token >= V || ban_token >= V) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MMSUM_BF16)
        beam_topk_rows_kernel<bf16_t><<<dim3(rows), dim3(TK_THREADS), 0, s>>>((bf16_t*)logits, ld, V, beam_scores, banned, nban, force_token, ban_token, K,
                                                                              row_scores, row_tokens);
    else if (dtype == MMSUM_F32)
        beam_topk_rows_kernel<float><<<dim3(rows), dim3(TK_THREADS), 0, s>>>((float*)logits, ld, V, beam_scores, banned, nban, force_token, ban_token, K,
                                                                             row_scores, row_tokens);
    else return MMSUM_ERR_BAD_DTYPE;
    beam_topk_merge_kernel<<<dim3(rows / num_beams), dim3(64), 0, s>>>(row_scores, row_tokens, num_beams, K, V, out_scores, out_ids);
    return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
}

extern "C" int mmsum_decode_self_attn(int dtype, const void* q, long ldq, const void* k_cache, const void* v_cache, long ld_cache, const int* ancestors,
                                      void* out, long ldo, int rows, int H, int len, int Tmax, float scale, void* stream) {
    if (rows <= 0 || H <= 0 || len <= 0 || len > Tmax || Tmax > 256) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(rows, H);
    if (dtype == MMSUM_BF16)
        decode_self_attn_kernel<bf16_t><<<grid, dim3(64), 0, s>>>((const bf16_t*)q, ldq, (const bf16_t*)k_cache, (const bf16_t*)v_cache, ld_cache, ancestors,
                                                                  (bf16_t*)out, ldo, len, Tmax, scale);
    else if (dtype == MMSUM_F32)
        decode_self_attn_kernel<float><<<grid, dim3(64), 0, s>>>((const float*)q, ldq, (const float*)k_cache, (const float*)v_cache, ld_cache, ancestors,
                                                                 (float*)out, ldo, len, Tmax, scale);
    else return MMSUM_ERR_BAD_DTYPE;
    return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
}
