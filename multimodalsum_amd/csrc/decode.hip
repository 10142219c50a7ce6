// Beam-search decode-step kernels: what one step of
// _generate_beam_search (modeling_multimodalsum.py:2857-3010) does on [rows, V] logits and on the self-attention caches.
//
//   beam_topk_rows / beam_topk_merge : adjust_logits (forced BOS / EOS, :3084-3089) -> log_softmax (:2874) -> min-length EOS
//       ban and no-repeat-n-gram bans (generation_utils.py:57-98, 848-868) -> + beam score -> top 2*num_beams over the
//       num_beams * V candidates of a business (:2925), as two launches that read the logits once (rows x 8 chunk blocks) and write
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

// Stage 1: grid (rows, TK_CHUNKS).  A block takes one chunk of a row's logits into registers (read ONCE), and leaves
//   part_ms [row][chunk][2] : running (max, sum of exp) of the chunk's raw logits (the bans come after the normalisation, so a
//                             banned token's mass stays in the log-sum-exp, :2880-2900),
//   part_v / part_t [row][chunk][K] : the chunk's K best (raw logit, token) after the bans, best first.
// The final 2 * num_beams of a business can take at most K = 2 * num_beams candidates from any one chunk, so nothing is lost.
// 32 rows x 8 chunks fill the chip; the one-block-per-row form of round 2's first version ran on 32 CUs for 185 us.
constexpr int TK_CHUNKS = 8;
constexpr int TK_NPT = 32;            // logits per thread: a chunk holds at most TK_THREADS * TK_NPT = 8,192 of them

template <typename T>
__global__ __launch_bounds__(TK_THREADS) void beam_topk_chunk_kernel(T* __restrict__ logits, long ld, int V, int chunk_len, const int* __restrict__ banned,
                                                                     int nban, int ban_token, int K, float* __restrict__ part_ms,
                                                                     float* __restrict__ part_v, int* __restrict__ part_t) {
    __shared__ float red_m[TK_THREADS / 64], red_s[TK_THREADS / 64];
    __shared__ float win_v[TK_THREADS / 64];
    __shared__ int win_t[TK_THREADS / 64];
    const int row = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    T* x = logits + (long)row * ld;
    const int c0 = ch * chunk_len, c1 = min(V, c0 + chunk_len);
    float val[TK_NPT];
    float m = -INFINITY, s = 0.f;
#pragma unroll
    for (int k = 0; k < TK_NPT; ++k) {
        const int i = c0 + tid + k * TK_THREADS;
        val[k] = i < c1 ? to_f32(x[i]) : -INFINITY;
        const float v = val[k];
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
        part_ms[((long)row * TK_CHUNKS + ch) * 2] = mm;
        part_ms[((long)row * TK_CHUNKS + ch) * 2 + 1] = ss;
    }
    // ---- bans, after the statistics: the owner thread drops the value it holds and writes -inf into the logits (the contract)
    auto ban = [&](int t) {
        if (t >= c0 && t < c1 && ((t - c0) % TK_THREADS) == tid) {
            const int kk = (t - c0) / TK_THREADS;
#pragma unroll
            for (int k = 0; k < TK_NPT; ++k) if (k == kk) val[k] = -INFINITY;
            x[t] = from_f32<T>(-INFINITY);
        }
    };
    if (ban_token >= 0) ban(ban_token);
    if (banned != nullptr)
        for (int i = 0; i < nban; ++i) {
            const int t = banned[(long)row * nban + i];             // uniform load: every thread looks at every ban, the owner acts
            if (t >= 0 && t < V) ban(t);
        }
    // ---- K rounds of block-wide arg-best over the values in registers (higher value first, lower token among equals)
    for (int r = 0; r < K; ++r) {
        float bv = -INFINITY;
        int bt = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < TK_NPT; ++k) {
            const int i = c0 + tid + k * TK_THREADS;
            if (i < c1 && better(val[k], i, bv, bt)) { bv = val[k]; bt = i; }          // a retired value is NaN: never better
        }
        float gv = bv;
        int gt = bt;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(gv, o);
            const int t2 = __shfl_xor(gt, o);
            if (better(v2, t2, gv, gt)) { gv = v2; gt = t2; }
        }
        if (lane == 0) { win_v[wave] = gv; win_t[wave] = gt; }
        __syncthreads();
        gv = win_v[0];
        gt = win_t[0];
#pragma unroll
        for (int w = 1; w < TK_THREADS / 64; ++w) if (better(win_v[w], win_t[w], gv, gt)) { gv = win_v[w]; gt = win_t[w]; }
        if (gt != 0x7fffffff && gt >= c0 && ((gt - c0) % TK_THREADS) == tid) {     // the owner retires the winner: it leaves the chunk for good
            const int kk = (gt - c0) / TK_THREADS;
#pragma unroll
            for (int k = 0; k < TK_NPT; ++k) if (k == kk) val[k] = __builtin_nanf("");
        }
        if (tid == 0) {
            part_v[((long)row * TK_CHUNKS + ch) * K + r] = gv;
            part_t[((long)row * TK_CHUNKS + ch) * K + r] = gt;
        }
        __syncthreads();
    }
}

// Stage 2: one wave per business: log-sum-exp of every hypothesis row from its chunks' (max, sum) pairs, then the 2 * num_beams best
// of the business's num_beams * TK_CHUNKS * K candidates by (log-prob + beam score desc, flat index asc); flat index = beam * V + token
// as in next_scores.view(batch, num_beams * vocab) (:2920-2925).  force_token >= 0 (adjust_logits_during_generation :3084-3089: every
// other logit is -inf, so log_softmax is 0 at the forced token): the candidates are built here and stage 1 is not launched.
constexpr int TK_MAXC = 16;           // candidates per lane: num_beams * TK_CHUNKS * K / 64 <= 8 * 8 * 16 / 64
__global__ __launch_bounds__(64) void beam_topk_merge_kernel(const float* __restrict__ part_ms, const float* __restrict__ part_v,
                                                             const int* __restrict__ part_t, const float* __restrict__ beam_scores, int num_beams, int K,
                                                             int V, int force_token, float* __restrict__ out_scores, long long* __restrict__ out_ids) {
    __shared__ float lse[8];
    const int b = blockIdx.x, lane = threadIdx.x;
    if (force_token < 0 && lane < num_beams) {
        const long row = (long)b * num_beams + lane;
        float mm = -INFINITY, ss = 0.f;
        for (int c = 0; c < TK_CHUNKS; ++c) {
            const float m2 = part_ms[(row * TK_CHUNKS + c) * 2], s2 = part_ms[(row * TK_CHUNKS + c) * 2 + 1];
            const float mx = fmaxf(mm, m2);
            ss = (mm == -INFINITY ? 0.f : ss * __expf(mm - mx)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mx));
            mm = mx;
        }
        lse[lane] = mm + __logf(ss);
    }
    __syncthreads();
    const int per_beam = force_token >= 0 ? K : TK_CHUNKS * K;
    const int n = num_beams * per_beam;
    float v[TK_MAXC];
    long long id[TK_MAXC];
#pragma unroll
    for (int j = 0; j < TK_MAXC; ++j) {
        const int c = lane + 64 * j;
        v[j] = -INFINITY;
        id[j] = 0x7fffffffffffffffLL;
        if (c < n) {
            const int beam = c / per_beam, k = c % per_beam;
            const long row = (long)b * num_beams + beam;
            if (force_token >= 0) {            // the row's K candidates: the forced token at 0 + beam score, then the lowest other tokens at -inf
                const int tok = k == 0 ? force_token : (k - 1 < force_token ? k - 1 : k);
                v[j] = k == 0 ? beam_scores[row] : -INFINITY;
                id[j] = (long long)beam * V + tok;
            } else {
                const int tok = part_t[row * TK_CHUNKS * K + k];
                if (tok != 0x7fffffff) {
                    v[j] = (part_v[row * TK_CHUNKS * K + k] - lse[beam]) + beam_scores[row];      // log_softmax, then + beam score (:2874, :2917)
                    id[j] = (long long)beam * V + tok;
                }
            }
        }
    }
    for (int r = 0; r < K; ++r) {
        int pick = 0;
#pragma unroll
        for (int j = 1; j < TK_MAXC; ++j) if (v[j] > v[pick] || (v[j] == v[pick] && id[j] < id[pick])) pick = j;
        float bv = -INFINITY;
        long long bi = 0x7fffffffffffffffLL;
#pragma unroll
        for (int j = 0; j < TK_MAXC; ++j) if (j == pick) { bv = v[j]; bi = id[j]; }
        const long long mine = bi;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(bv, o);
            const long long i2 = __shfl_xor(bi, o);
            if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
        }
        if (mine == bi && bi != 0x7fffffffffffffffLL) {                     // the owner retires the winner
#pragma unroll
            for (int j = 0; j < TK_MAXC; ++j) if (j == pick) { v[j] = -INFINITY; id[j] = 0x7fffffffffffffffLL; }
        }
        if (lane == 0) { out_scores[(long)b * K + r] = bv; out_ids[(long)b * K + r] = bi == 0x7fffffffffffffffLL ? 0 : bi; }
    }
}

// One wave per (hypothesis row, head).  Key s of row r lives at cache row anc[r * Tmax + s] * Tmax + s.
// k_new / v_new (optional): this step's projections [rows, H*64] for position len - 1.  The wave stores its 64-element slices into
// the caches (row r itself: ancestors[r, len - 1] == r by construction) and reads THAT position from k_new / v_new, so the two
// copy launches per layer and step that used to fill the caches are gone and no store -> load ordering is needed.
template <typename T>
__global__ __launch_bounds__(64) void decode_self_attn_kernel(const T* __restrict__ q, long ldq, T* __restrict__ kc, T* __restrict__ vc,
                                                              long ldc, const int* __restrict__ anc, T* __restrict__ out, long ldo, int len, int Tmax,
                                                              float scale, const T* __restrict__ k_new, const T* __restrict__ v_new, long ldn) {
    constexpr int HD = 64;
    const int r = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const T* qrow = q + (long)r * ldq + h * HD;
    const bool fresh = k_new != nullptr;
    const long newrow = ((long)r * Tmax + (len - 1)) * ldc + h * HD;       // where position len - 1 of this row lives in the caches
    if (fresh) {
        kc[newrow + lane] = k_new[(long)r * ldn + h * HD + lane];
        vc[newrow + lane] = v_new[(long)r * ldn + h * HD + lane];
    }
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
            const T* krow = (fresh && s == len - 1) ? k_new + (long)r * ldn + h * HD : kc + prow[j];
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
    const T* vfresh = fresh ? v_new + (long)r * ldn + h * HD : nullptr;
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
                v[u] = (fresh && base + src == len - 1) ? to_f32(vfresh[lane]) : to_f32(vc[row + lane]);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) o = fmaf(p[u], v[u], o);
        }
    }
    out[(long)r * ldo + h * HD + lane] = from_f32<T>(o * inv);
}

}  // namespace

extern "C" long mmsum_beam_topk_workspace(int rows, int num_beams) {
    const long K = 2L * num_beams;
    return (long)rows * TK_CHUNKS * (2 * sizeof(float) + K * (sizeof(float) + sizeof(int)));
}

extern "C" int mmsum_beam_topk(int dtype, void* logits, long ld, int V, const float* beam_scores, const int* banned, int nban, int force_token,
                               int ban_token, int rows, int num_beams, void* workspace, float* out_scores, long long* out_ids, void* stream) {
    const int K = 2 * num_beams;
    if (rows <= 0 || V <= 0 || ld < V || num_beams < 1 || num_beams > 8 || K > TK_MAX || rows % num_beams || V < K || nban < 0) return MMSUM_ERR_BAD_SHAPE;
    if (force_token >= V || ban_token >= V) return MMSUM_ERR_BAD_SHAPE;
    const int chunk_len = ((V + TK_CHUNKS - 1) / TK_CHUNKS + 7) & ~7;
    if (chunk_len > TK_THREADS * TK_NPT) return MMSUM_ERR_BAD_SHAPE;          // V <= 65,536
    if (workspace == nullptr) return MMSUM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* part_ms = static_cast<float*>(workspace);
    float* part_v = part_ms + (long)rows * TK_CHUNKS * 2;
    int* part_t = reinterpret_cast<int*>(part_v + (long)rows * TK_CHUNKS * K);
    if (force_token < 0) {
        const dim3 grid(rows, TK_CHUNKS);
        if (dtype == MMSUM_BF16)
            beam_topk_chunk_kernel<bf16_t><<<grid, dim3(TK_THREADS), 0, s>>>((bf16_t*)logits, ld, V, chunk_len, banned, nban, ban_token, K, part_ms, part_v, part_t);
        else if (dtype == MMSUM_F32)
            beam_topk_chunk_kernel<float><<<grid, dim3(TK_THREADS), 0, s>>>((float*)logits, ld, V, chunk_len, banned, nban, ban_token, K, part_ms, part_v, part_t);
        else return MMSUM_ERR_BAD_DTYPE;
    } else if (dtype != MMSUM_BF16 && dtype != MMSUM_F32) {
        return MMSUM_ERR_BAD_DTYPE;
    }
    beam_topk_merge_kernel<<<dim3(rows / num_beams), dim3(64), 0, s>>>(part_ms, part_v, part_t, beam_scores, num_beams, K, V, force_token, out_scores, out_ids);
    return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
}

extern "C" int mmsum_decode_self_attn(int dtype, const void* q, long ldq, void* k_cache, void* v_cache, long ld_cache, const int* ancestors,
                                      void* out, long ldo, int rows, int H, int len, int Tmax, float scale, const void* k_new, const void* v_new,
                                      long ld_new, void* stream) {
    if (rows <= 0 || H <= 0 || len <= 0 || len > Tmax || Tmax > 256) return MMSUM_ERR_BAD_SHAPE;
    if ((k_new == nullptr) != (v_new == nullptr)) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(rows, H);
    if (dtype == MMSUM_BF16)
        decode_self_attn_kernel<bf16_t><<<grid, dim3(64), 0, s>>>((const bf16_t*)q, ldq, (bf16_t*)k_cache, (bf16_t*)v_cache, ld_cache, ancestors,
                                                                  (bf16_t*)out, ldo, len, Tmax, scale, (const bf16_t*)k_new, (const bf16_t*)v_new, ld_new);
    else if (dtype == MMSUM_F32)
        decode_self_attn_kernel<float><<<grid, dim3(64), 0, s>>>((const float*)q, ldq, (float*)k_cache, (float*)v_cache, ld_cache, ancestors,
                                                                 (float*)out, ldo, len, Tmax, scale, (const float*)k_new, (const float*)v_new, ld_new);
    else return MMSUM_ERR_BAD_DTYPE;
    return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
}
