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
constexpr int TK_MAX_CAND = 64;       // an explicit candidate count (sampling: the top_k best of a row)
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

// Repetition penalty (enforce_repetition_penalty_, generation_utils.py:47-55; round 6): `pen` [rows, npen] lists a row's distinct previous
// tokens (-1 ends a list); each listed score s becomes s * penalty when negative, s / penalty otherwise -- BEFORE the bans, as
// postprocess_next_token_scores orders them.  pen_mode 1: on the raw logits (greedy decoding post-processes the logits themselves);
// pen_mode 2: on the log-probabilities (beam search: after log_softmax), which needs the row's log-sum-exp first: the host launches this
// kernel twice, `phase` 1 = statistics only, `phase` 2 = read them back (every chunk's pair: the row's log-sum-exp), penalise, ban, select;
// a penalised candidate leaves as (lp' + lse) so that stage 2's subtraction of lse gives lp'.  phase 0: one launch, as before.
template <typename T>
__global__ __launch_bounds__(TK_THREADS) void beam_topk_chunk_kernel(T* __restrict__ logits, long ld, int V, int chunk_len, const int* __restrict__ banned,
                                                                     int nban, int ban_token, int K, float* __restrict__ part_ms,
                                                                     float* __restrict__ part_v, int* __restrict__ part_t,
                                                                     const int* __restrict__ pen, int npen, float penalty, int pen_mode, int phase) {
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
    if (tid == 0 && phase != 2) {
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
    if (phase == 1) return;                                    // statistics only (the log-prob penalty's first launch)
    // ---- repetition penalty, before the bans
    if (pen != nullptr && pen_mode != 0) {
        float lse = 0.f;
        if (pen_mode == 2) {                                   // the row's log-sum-exp from the statistics launch (uniform loads)
            float mm = -INFINITY, ss = 0.f;
#pragma unroll
            for (int c = 0; c < TK_CHUNKS; ++c) {
                const float m2 = part_ms[((long)row * TK_CHUNKS + c) * 2], s2 = part_ms[((long)row * TK_CHUNKS + c) * 2 + 1];
                const float mx = fmaxf(mm, m2);
                ss = (mm == -INFINITY ? 0.f : ss * __expf(mm - mx)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mx));
                mm = mx;
            }
            lse = mm + __logf(ss);
        }
        for (int i = 0; i < npen; ++i) {
            const int t = pen[(long)row * npen + i];
            if (t < 0) break;
            if (t >= c0 && t < c1 && ((t - c0) % TK_THREADS) == tid) {
                const int kk = (t - c0) / TK_THREADS;
#pragma unroll
                for (int k = 0; k < TK_NPT; ++k)
                    if (k == kk) {
                        const float sc = val[k] - lse;         // mode 1: lse = 0, the raw logit
                        val[k] = (sc < 0.f ? sc * penalty : sc / penalty) + lse;
                    }
            }
        }
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
    if (banned != nullptr) {
        // the list is filled from the front and -1 padded (generation._banned_ngram_table): stop at the first -1.  Each entry is a
        // uniform, dependent load -- walking all nban (= max_length) slots cost ~0.3 us apiece, most of this kernel's 37 us
        for (int i = 0; i < nban; ++i) {
            const int t = banned[(long)row * nban + i];             // uniform load: every thread looks at every ban, the owner acts
            if (t < 0) break;
            if (t < V) ban(t);
        }
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
// (flat index = beam * V + token < 8 * 65,536: 32 bits carry it; every lane keeps its candidates SORTED, best first, so a round is one
// wave-wide arg-best over the lanes' heads -- two dwords through six exchange steps -- and a pop on the winning lane)
__device__ __forceinline__ bool cand_better(float v, int id, float v2, int id2) { return v > v2 || (v == v2 && id < id2); }
template <int MAXC>
__global__ __launch_bounds__(64) void beam_topk_merge_kernel(const float* __restrict__ part_ms, const float* __restrict__ part_v,
                                                             const int* __restrict__ part_t, const float* __restrict__ beam_scores, int num_beams, int K,
                                                             int V, int force_token, float* __restrict__ out_scores, long long* __restrict__ out_ids) {
    __shared__ float lse[8];
    const int b = blockIdx.x, lane = threadIdx.x;
    if (force_token < 0 && lane < num_beams) {
        const long row = (long)b * num_beams + lane;
        float mm = -INFINITY, ss = 0.f;
#pragma unroll
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
    constexpr int NONE = 0x7fffffff;
    float v[MAXC];
    int id[MAXC];
#pragma unroll
    for (int j = 0; j < MAXC; ++j) {
        const int c = lane + 64 * j;
        v[j] = -INFINITY;
        id[j] = NONE;
        if (c < n) {
            const int beam = c / per_beam, k = c % per_beam;
            const long row = (long)b * num_beams + beam;
            if (force_token >= 0) {            // the row's K candidates: the forced token at 0 + beam score, then the lowest other tokens at -inf
                const int tok = k == 0 ? force_token : (k - 1 < force_token ? k - 1 : k);
                v[j] = k == 0 ? beam_scores[row] : -INFINITY;
                id[j] = beam * V + tok;
            } else {
                const int tok = part_t[row * TK_CHUNKS * K + k];
                if (tok != NONE) {
                    v[j] = (part_v[row * TK_CHUNKS * K + k] - lse[beam]) + beam_scores[row];      // log_softmax, then + beam score (:2874, :2917)
                    id[j] = beam * V + tok;
                }
            }
        }
    }
    // sort this lane's candidates, best first (insertion sort on registers: MAXC is 4 for the usual 4 beams)
#pragma unroll
    for (int i = 1; i < MAXC; ++i)
#pragma unroll
        for (int j = i; j > 0; --j) {
            const bool sw = cand_better(v[j], id[j], v[j - 1], id[j - 1]);
            const float tv = sw ? v[j - 1] : v[j];
            const int ti = sw ? id[j - 1] : id[j];
            v[j - 1] = sw ? v[j] : v[j - 1];
            id[j - 1] = sw ? id[j] : id[j - 1];
            v[j] = tv;
            id[j] = ti;
        }
    for (int r = 0; r < K; ++r) {
        float bv = v[0];
        int bi = id[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(bv, o);
            const int i2 = __shfl_xor(bi, o);
            if (cand_better(v2, i2, bv, bi)) { bv = v2; bi = i2; }
        }
        if (bi != NONE && id[0] == bi) {                          // the owner pops its head (ids are unique: one lane matches)
#pragma unroll
            for (int j = 0; j + 1 < MAXC; ++j) { v[j] = v[j + 1]; id[j] = id[j + 1]; }
            v[MAXC - 1] = -INFINITY;
            id[MAXC - 1] = NONE;
        }
        if (lane == 0) { out_scores[(long)b * K + r] = bv; out_ids[(long)b * K + r] = bi == NONE ? 0 : (long long)bi; }
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

extern "C" long mmsum_beam_topk_workspace(int rows, int num_beams, int ncand) {
    const long K = ncand > 0 ? ncand : 2L * num_beams;
    return (long)rows * TK_CHUNKS * (2 * sizeof(float) + K * (sizeof(float) + sizeof(int)));
}

extern "C" int mmsum_beam_topk(int dtype, void* logits, long ld, int V, const float* beam_scores, const int* banned, int nban, int force_token,
                               int ban_token, int rows, int num_beams, void* workspace, float* out_scores, long long* out_ids,
                               const int* penalized, int npen, float penalty, int penalty_on_logits, int ncand, void* stream) {
    const int K = ncand > 0 ? ncand : 2 * num_beams;
    // (candidates per lane of the merge wave: num_beams * TK_CHUNKS * K / 64 <= TK_MAXC -- 2 * num_beams of <= 8 beams, or up to 64 of one row)
    if (rows <= 0 || V <= 0 || ld < V || num_beams < 1 || num_beams > 8 || ncand < 0 || K > (ncand > 0 ? TK_MAX_CAND : TK_MAX) ||
        num_beams * TK_CHUNKS * K > TK_MAXC * 64 || rows % num_beams || V < K || nban < 0) return MMSUM_ERR_BAD_SHAPE;
    if (npen < 0 || (penalized != nullptr && (npen == 0 || !(penalty > 0.f)))) return MMSUM_ERR_BAD_SHAPE;
    const int pen_mode = (penalized == nullptr || penalty == 1.f) ? 0 : (penalty_on_logits ? 1 : 2);
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
        // (a penalty on the log-probabilities needs the row's log-sum-exp before it can rank: a statistics launch first)
        for (int phase = (pen_mode == 2 ? 1 : 0); phase <= (pen_mode == 2 ? 2 : 0); ++phase) {
            if (dtype == MMSUM_BF16)
                beam_topk_chunk_kernel<bf16_t><<<grid, dim3(TK_THREADS), 0, s>>>((bf16_t*)logits, ld, V, chunk_len, banned, nban, ban_token, K, part_ms, part_v, part_t,
                                                                                  penalized, npen, penalty, pen_mode, phase);
            else if (dtype == MMSUM_F32)
                beam_topk_chunk_kernel<float><<<grid, dim3(TK_THREADS), 0, s>>>((float*)logits, ld, V, chunk_len, banned, nban, ban_token, K, part_ms, part_v, part_t,
                                                                                 penalized, npen, penalty, pen_mode, phase);
            else return MMSUM_ERR_BAD_DTYPE;
        }
    } else if (dtype != MMSUM_BF16 && dtype != MMSUM_F32) {
        return MMSUM_ERR_BAD_DTYPE;
    }
    if ((long)num_beams * V >= 0x7fffffffL) return MMSUM_ERR_BAD_SHAPE;
    if (num_beams * TK_CHUNKS * K <= 4 * 64)
        beam_topk_merge_kernel<4><<<dim3(rows / num_beams), dim3(64), 0, s>>>(part_ms, part_v, part_t, beam_scores, num_beams, K, V, force_token, out_scores, out_ids);
    else
        beam_topk_merge_kernel<TK_MAXC><<<dim3(rows / num_beams), dim3(64), 0, s>>>(part_ms, part_v, part_t, beam_scores, num_beams, K, V, force_token, out_scores, out_ids);
    return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------
// mmsum_decode_cross_attn: the decode step's per-entity cross-attention + entity mean over the cached K / V of ALL modalities in
// one launch (modeling_multimodalsum.py:819-869 with T = 1 per hypothesis; the hypotheses of a business share its memory).
//
// The training kernel this replaces in the decode step runs one workgroup per (business, head) that walks the business's entities
// one after the other with 4 live query rows in a 128-row tile: 128 workgroups, 19 + 15 + 5 us per layer for 61 MB of K / V (three
// launches).  The step is a pure stream of the cached K / V -- 4 queries per business cost nothing -- so here ONE workgroup takes
// ONE (entity, head): 8 x (8 + 1 + 4) x 16 = 1,664 workgroups, every row of an entity's K and V requested up front (eight lanes
// fetch one 128-byte row: one full line per 8 lanes), scores by VALU dot products reduced over the eight lanes of a row, softmax per
// query by one wave each, P V per lane over its rows and a tree over the row slots.  The entity mean crosses workgroups: each
// leaves its normalised [beams, 64] output as f32 and takes a ticket on its (modality, business, head); the last arriver adds the
// valid entities' outputs (in entity order), divides by their count and writes the bf16 head slice (write-through hand-off as in
// mmsum_dec_gemm).
// ---------------------------------------------------------------------------------------------
namespace {

constexpr int XA_THREADS = 256;
constexpr int XA_MAXQ = 8;            // hypotheses per business (num_beams <= 8)
constexpr int XA_MAXROWS = 7;         // key rows per thread: S <= 224
struct XaMod { const bf16_t* k; const bf16_t* v; const uint8_t* pad; const uint8_t* null_entity; int N, S, ent0; };   // ent0: first global entity index of the modality
struct XaArgs {
    XaMod mod[3];
    const bf16_t* q; bf16_t* out; float* part; unsigned* tickets;
    long ldq, ldkv, ldo;
    int nmod, B, H, qpb, R;          // R = B * qpb rows of q; out rows: modality * R + row
    float scale;
};

// Sum over the 8 lanes that share a key row (lane & 7 = the 8-dim chunk): DPP adds, no LDS crossbar (ds_bpermute) on the way.
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row8_sum(float v) {
    v = dpp_add<0xB1>(v);            // quad_perm [1,0,3,2]: lane ^ 1
    v = dpp_add<0x4E>(v);            // quad_perm [2,3,0,1]: lane ^ 2
    return dpp_add<0x141>(v);        // row_half_mirror: lane i <-> 7 - i of each 8 (the other quad's sum)
}

// Layout of a workgroup's work (256 threads = 32 key-row slots x 8 chunks of 16 bytes):
//   K  : row s = 32 i + slot, chunk c -> registers (one 16-byte load per row; eight lanes fetch one 128-byte row);
//   V  : the same rows by LDS-DMA (global_load_lds, 16 bytes per lane: a wave's instruction lands as 8 rows x 128 bytes, contiguous)
//        -- no registers, the tile [S][64] bf16 sits in LDS for the P V pass;
//   scores: per lane 8 dims x QPB queries, summed over the 8 lanes of a row by DPP; softmax: one wave per query;
//   P V: thread = (query, dim): walks the S keys with p broadcast from LDS and V read conflict-free (64 lanes x 2 bytes of one row).
template <int QPB>
__global__ __launch_bounds__(XA_THREADS) void decode_cross_attn_kernel(XaArgs a) {
    __shared__ __attribute__((aligned(16))) char vl[XA_MAXROWS * 32 * 128];     // V tile
    __shared__ float sc[QPB][XA_MAXROWS * 32];                                  // scores, then probabilities, [query][key]
    __shared__ unsigned last;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.x;
    // the LAST entities first: the modalities are listed text, table, image, and the 196-key image entities are the longest workgroups --
    // started last they are the launch's tail (longest-processing-time-first)
    int e = gridDim.y - 1 - blockIdx.y, m = 0;                 // global entity index -> (modality, business, entity)
    while (m + 1 < a.nmod && e >= a.mod[m + 1].ent0) ++m;
    const XaMod M = a.mod[m];
    const int le = e - M.ent0, b = le / M.N;
    const int S = M.S;
    float* mypart = a.part + ((long)e * a.H + h) * (QPB * 64);
    unsigned* ticket = a.tickets + ((long)m * a.B + b) * a.H + h;
    const bool is_null = M.null_entity != nullptr && M.null_entity[le] != 0;
    if (!is_null) {
        const int kslot = tid >> 3, dch = tid & 7;             // key row slot 0..31, 8-dim chunk of the head
        const long row0 = (long)le * S;
        const bf16_t* kb = M.k + row0 * a.ldkv + h * 64 + dch * 8;
        const bf16_t* vb = M.v + row0 * a.ldkv + h * 64 + dch * 8;
        const int nrow = (S + 31) >> 5;                        // rows per slot (<= XA_MAXROWS)
        // the key mask of this thread's rows, requested with the K / V rows (inside the score loop each byte was a load + wait of its own)
        int padv[XA_MAXROWS];
        {
            // every load unconditional, from a clamped position of a buffer that always exists (a NULL mask reads the null-entity flags'
            // first byte and is ignored below): a guarded load is a branch, and the compiler waits at every join
            const uint8_t* pm = M.pad != nullptr ? M.pad + row0 : M.null_entity;
            const int lim = M.pad != nullptr ? S - 1 : 0;
#pragma unroll
            for (int i = 0; i < XA_MAXROWS; ++i) padv[i] = pm != nullptr ? (int)__builtin_nontemporal_load(pm + min(i * 32 + kslot, lim)) : 0;
            if (M.pad == nullptr) {
#pragma unroll
                for (int i = 0; i < XA_MAXROWS; ++i) padv[i] = 0;
            }
        }
        u32x4_t kreg[XA_MAXROWS];
#pragma unroll
        for (int i = 0; i < XA_MAXROWS; ++i) {
            const int s = i * 32 + kslot;
            kreg[i] = u32x4_t{0, 0, 0, 0};
            if (i < nrow && s < S) {
                kreg[i] = *reinterpret_cast<const u32x4_t*>(kb + (long)s * a.ldkv);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vb + (long)s * a.ldkv),
                                                 (__attribute__((address_space(3))) void*)(vl + (i * 32 + wave * 8) * 128), 16, 0, 0);
            }
        }
        float qv[QPB][8];
#pragma unroll
        for (int qi = 0; qi < QPB; ++qi) {
            const u32x4_t raw = *reinterpret_cast<const u32x4_t*>(a.q + (long)(b * QPB + qi) * a.ldq + h * 64 + dch * 8);
            bf16_t qq[8];
            __builtin_memcpy(qq, &raw, 16);
#pragma unroll
            for (int j = 0; j < 8; ++j) qv[qi][j] = to_f32(qq[j]) * a.scale;
        }
        // ---- scores
#pragma unroll
        for (int i = 0; i < XA_MAXROWS; ++i) {
            if (i >= nrow) break;
            bf16_t kk[8];
            __builtin_memcpy(kk, &kreg[i], 16);
            float part[QPB];
#pragma unroll
            for (int qi = 0; qi < QPB; ++qi) {
                float x = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) x = fmaf(qv[qi][j], to_f32(kk[j]), x);
                part[qi] = row8_sum(x);
            }
            const int key = i * 32 + kslot;
            if (dch == 0 && key < S) {
                const bool masked = padv[i] != 0;
#pragma unroll
                for (int qi = 0; qi < QPB; ++qi) sc[qi][key] = masked ? -65536.0f : part[qi];      // masked_fill(-2^16), :841-845
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's V rows have landed in LDS
        __syncthreads();
        // ---- softmax: one wave per query
        for (int qi = wave; qi < QPB; qi += XA_THREADS / 64) {
            float mx = -INFINITY;
            for (int s = lane; s < S; s += 64) mx = fmaxf(mx, sc[qi][s]);
            mx = warp_max(mx);
            float l = 0.f;
            for (int s = lane; s < S; s += 64) { const float p = __expf(sc[qi][s] - mx); sc[qi][s] = p; l += p; }
            l = warp_sum(l);
            const float inv = 1.f / l;
            for (int s = lane; s < S; s += 64) sc[qi][s] *= inv;
        }
        __syncthreads();
        // ---- P V: thread = (query, dim)
        for (int idx = tid; idx < QPB * 64; idx += XA_THREADS) {
            const int qi = idx >> 6, dd = idx & 63;
            const bf16_t* vcol = reinterpret_cast<const bf16_t*>(vl) + dd;
            float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
            int s = 0;
            for (; s + 3 < S; s += 4) {
                o0 = fmaf(sc[qi][s], to_f32(vcol[s * 64]), o0);
                o1 = fmaf(sc[qi][s + 1], to_f32(vcol[(s + 1) * 64]), o1);
                o2 = fmaf(sc[qi][s + 2], to_f32(vcol[(s + 2) * 64]), o2);
                o3 = fmaf(sc[qi][s + 3], to_f32(vcol[(s + 3) * 64]), o3);
            }
            for (; s < S; ++s) o0 = fmaf(sc[qi][s], to_f32(vcol[s * 64]), o0);
            __hip_atomic_store(mypart + qi * 64 + dd, (o0 + o1) + (o2 + o3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // sc1: write-through past the XCD's L2
        }
    }
    // ---- the entity mean across workgroups: ticket on (modality, business, head); the last arriver adds the valid entities.
    // Write-through hand-off (guide, Guideline 16): sc1 payload stores, every storing wave drains them, the workgroup's barrier, one
    // lane's relaxed agent-scope ticket; the reducer reads every payload word with an sc1 load.  No release / acquire fence: an
    // agent-scope release writes back the whole L2's dirty lines, and 1,664 workgroups of a launch would each pay it.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (t == (unsigned)(M.N - 1)) ? 1u : 0u;
        if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // ready for the next layer's launch
    }
    __syncthreads();
    if (!last) return;
    for (int idx = tid; idx < QPB * 64; idx += XA_THREADS) {
        const int qi = idx >> 6, dd = idx & 63;
        float x = 0.f;
        int cnt = 0;
        // the entities' outputs (and null flags) eight at a time, every load unconditional from a clamped entity: a load per loop
        // iteration behind a `continue` was one dependent memory round trip per entity on the launch's critical path (round 6)
        for (int n0 = 0; n0 < M.N; n0 += 8) {
            float pv[8];
            int nul[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int nn = min(n0 + j, M.N - 1);
                nul[j] = M.null_entity != nullptr ? (int)__builtin_nontemporal_load(M.null_entity + b * M.N + nn) : 0;      // null entities are dropped from the mean (:856-866)
                pv[j] = __hip_atomic_load(a.part + ((long)(M.ent0 + b * M.N + nn) * a.H + h) * (QPB * 64) + qi * 64 + dd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (n0 + j < M.N && nul[j] == 0) { x += pv[j]; ++cnt; }
        }
        x = cnt > 0 ? x / (float)cnt : 0.f;                    // every entity null: zeros (the out_proj bias still enters, :884-885)
        a.out[((long)m * a.R + b * QPB + qi) * a.ldo + h * 64 + dd] = (bf16_t)x;
    }
}

// The same launch in the f32 compute mode (the mode whose generated ids are held to the reference's): f32 q / K / V / out.  Same
// decomposition (one workgroup per (entity, head), 32 key-row slots x 8 chunks of 8 dims, DPP row sums, one wave per query for the
// softmax, thread = (query, dim) for P V, the entity mean through the last arriver).  A chunk is 32 bytes here: K in two 16-byte
// loads per row, V through registers into a row-major [S][64] f32 LDS tile (a row = 256 bytes = all 64 banks: the P V pass reads
// one row per step, conflict-free).  Replaces three launches of the training kernel with 4 live query rows in a 128-row tile
// (attn_fwd_pipe_kernel<float>: 150 us per layer).
struct XaModF { const float* k; const float* v; const uint8_t* pad; const uint8_t* null_entity; int N, S, ent0; };
struct XaArgsF {
    XaModF mod[3];
    const float* q; float* out; float* part; unsigned* tickets;
    long ldq, ldkv, ldo;
    int nmod, B, H, qpb, R;
    float scale;
};
template <int QPB>
__global__ __launch_bounds__(XA_THREADS) void decode_cross_attn_f32_kernel(XaArgsF a) {
    extern __shared__ __attribute__((aligned(16))) char xa_smem[];
    float* vl = reinterpret_cast<float*>(xa_smem);                                       // V tile [XA_MAXROWS * 32][64]
    float (*sc)[XA_MAXROWS * 32] = reinterpret_cast<float (*)[XA_MAXROWS * 32]>(xa_smem + XA_MAXROWS * 32 * 256);   // [QPB][keys]
    __shared__ unsigned last;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.x;
    int e = gridDim.y - 1 - blockIdx.y, m = 0;
    while (m + 1 < a.nmod && e >= a.mod[m + 1].ent0) ++m;
    const XaModF M = a.mod[m];
    const int le = e - M.ent0, b = le / M.N;
    const int S = M.S;
    float* mypart = a.part + ((long)e * a.H + h) * (QPB * 64);
    unsigned* ticket = a.tickets + ((long)m * a.B + b) * a.H + h;
    const bool is_null = M.null_entity != nullptr && M.null_entity[le] != 0;
    if (!is_null) {
        const int kslot = tid >> 3, dch = tid & 7;
        const long row0 = (long)le * S;
        const float* kb = M.k + row0 * a.ldkv + h * 64 + dch * 8;
        const float* vb = M.v + row0 * a.ldkv + h * 64 + dch * 8;
        const int nrow = (S + 31) >> 5;
        int padv[XA_MAXROWS];
        {
            // every load unconditional, from a clamped position of a buffer that always exists (a NULL mask reads the null-entity flags'
            // first byte and is ignored below): a guarded load is a branch, and the compiler waits at every join
            const uint8_t* pm = M.pad != nullptr ? M.pad + row0 : M.null_entity;
            const int lim = M.pad != nullptr ? S - 1 : 0;
#pragma unroll
            for (int i = 0; i < XA_MAXROWS; ++i) padv[i] = pm != nullptr ? (int)__builtin_nontemporal_load(pm + min(i * 32 + kslot, lim)) : 0;
            if (M.pad == nullptr) {
#pragma unroll
                for (int i = 0; i < XA_MAXROWS; ++i) padv[i] = 0;
            }
        }
        f32x4_t kreg[XA_MAXROWS][2], vreg[XA_MAXROWS][2];
#pragma unroll
        for (int i = 0; i < XA_MAXROWS; ++i) {
            const int s = i * 32 + kslot;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) kreg[i][hh] = vreg[i][hh] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (i < nrow && s < S) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    kreg[i][hh] = *reinterpret_cast<const f32x4_t*>(kb + (long)s * a.ldkv + 4 * hh);
                    vreg[i][hh] = *reinterpret_cast<const f32x4_t*>(vb + (long)s * a.ldkv + 4 * hh);
                }
            }
        }
        float qv[QPB][8];
#pragma unroll
        for (int qi = 0; qi < QPB; ++qi) {
            const float* qp = a.q + (long)(b * QPB + qi) * a.ldq + h * 64 + dch * 8;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const f32x4_t t = *reinterpret_cast<const f32x4_t*>(qp + 4 * hh);
#pragma unroll
                for (int j = 0; j < 4; ++j) qv[qi][4 * hh + j] = t[j] * a.scale;
            }
        }
#pragma unroll
        for (int i = 0; i < XA_MAXROWS; ++i) {
            if (i >= nrow) break;
            const int key = i * 32 + kslot;
            float part[QPB];
#pragma unroll
            for (int qi = 0; qi < QPB; ++qi) {
                float x = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) x = fmaf(qv[qi][j], kreg[i][j >> 2][j & 3], x);
                part[qi] = row8_sum(x);
            }
            if (key < S) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) *reinterpret_cast<f32x4_t*>(vl + key * 64 + dch * 8 + 4 * hh) = vreg[i][hh];
                if (dch == 0) {
                    const bool masked = padv[i] != 0;
#pragma unroll
                    for (int qi = 0; qi < QPB; ++qi) sc[qi][key] = masked ? -65536.0f : part[qi];      // masked_fill(-2^16), :841-845
                }
            }
        }
        __syncthreads();
        for (int qi = wave; qi < QPB; qi += XA_THREADS / 64) {
            float mx = -INFINITY;
            for (int s = lane; s < S; s += 64) mx = fmaxf(mx, sc[qi][s]);
            mx = warp_max(mx);
            float l = 0.f;
            for (int s = lane; s < S; s += 64) { const float p = expf(sc[qi][s] - mx); sc[qi][s] = p; l += p; }
            l = warp_sum(l);
            const float inv = 1.f / l;
            for (int s = lane; s < S; s += 64) sc[qi][s] *= inv;
        }
        __syncthreads();
        for (int idx = tid; idx < QPB * 64; idx += XA_THREADS) {
            const int qi = idx >> 6, dd = idx & 63;
            const float* vcol = vl + dd;
            float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
            int s = 0;
            for (; s + 3 < S; s += 4) {
                o0 = fmaf(sc[qi][s], vcol[s * 64], o0);
                o1 = fmaf(sc[qi][s + 1], vcol[(s + 1) * 64], o1);
                o2 = fmaf(sc[qi][s + 2], vcol[(s + 2) * 64], o2);
                o3 = fmaf(sc[qi][s + 3], vcol[(s + 3) * 64], o3);
            }
            for (; s < S; ++s) o0 = fmaf(sc[qi][s], vcol[s * 64], o0);
            __hip_atomic_store(mypart + qi * 64 + dd, (o0 + o1) + (o2 + o3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (t == (unsigned)(M.N - 1)) ? 1u : 0u;
        if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!last) return;
    for (int idx = tid; idx < QPB * 64; idx += XA_THREADS) {
        const int qi = idx >> 6, dd = idx & 63;
        float x = 0.f;
        int cnt = 0;
        // the entities' outputs (and null flags) eight at a time, every load unconditional from a clamped entity: a load per loop
        // iteration behind a `continue` was one dependent memory round trip per entity on the launch's critical path (round 6)
        for (int n0 = 0; n0 < M.N; n0 += 8) {
            float pv[8];
            int nul[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int nn = min(n0 + j, M.N - 1);
                nul[j] = M.null_entity != nullptr ? (int)__builtin_nontemporal_load(M.null_entity + b * M.N + nn) : 0;      // null entities are dropped from the mean (:856-866)
                pv[j] = __hip_atomic_load(a.part + ((long)(M.ent0 + b * M.N + nn) * a.H + h) * (QPB * 64) + qi * 64 + dd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (n0 + j < M.N && nul[j] == 0) { x += pv[j]; ++cnt; }
        }
        x = cnt > 0 ? x / (float)cnt : 0.f;
        a.out[((long)m * a.R + b * QPB + qi) * a.ldo + h * 64 + dd] = x;
    }
}

// Single-query self-attention over the caches through the ancestor table, bf16, in the layout of the cross-attention kernel above:
// 256 threads = 32 key slots x 8 chunks; every K row of the hypothesis requested at once into registers and every V row by LDS-DMA
// (the one-wave kernel further up walks the V rows eight at a time: up to 16 dependent round trips at 128 keys, 13 us per layer).
// grid = (rows, H).  Key s of row r lives at cache row anc[r][s] * Tmax + s; position len - 1 comes from k_new / v_new and is
// appended to the caches by the threads that hold it.
__global__ __launch_bounds__(XA_THREADS) void decode_self_attn_bf16_kernel(const bf16_t* __restrict__ q, long ldq, bf16_t* __restrict__ kc, bf16_t* __restrict__ vc,
                                                                           long ldc, const int* __restrict__ anc, bf16_t* __restrict__ out, long ldo, int len,
                                                                           int Tmax, float scale, const bf16_t* __restrict__ k_new,
                                                                           const bf16_t* __restrict__ v_new, long ldn) {
    constexpr int MAXROWS = 8;                                  // Tmax <= 256
    __shared__ __attribute__((aligned(16))) char vl[MAXROWS * 32 * 128];
    __shared__ float sc[MAXROWS * 32];
    __shared__ float red[XA_THREADS / 64][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = blockIdx.x, h = blockIdx.y;
    const int kslot = tid >> 3, dch = tid & 7;
    const bool fresh = k_new != nullptr;
    const int nrow = (len + 31) >> 5;
    const long col = h * 64 + dch * 8;
    // the cache row of every key this thread holds comes from the ancestor table: ALL of its entries are requested first (unconditionally,
    // from a clamped position) -- looked up inside the loop below, each K / V request waited for its own index (round 5: the ISA showed
    // load, wait, load per row slot: eight dependent round trips instead of two)
    int ancv[MAXROWS];
#pragma unroll
    for (int i = 0; i < MAXROWS; ++i) ancv[i] = anc[(long)r * Tmax + min(i * 32 + kslot, len - 1)];
    __builtin_amdgcn_sched_barrier(0);
    u32x4_t kreg[MAXROWS];
#pragma unroll
    for (int i = 0; i < MAXROWS; ++i) {
        const int s = i * 32 + kslot;
        kreg[i] = u32x4_t{0, 0, 0, 0};
        if (i < nrow && s < len) {
            const bool isnew = fresh && s == len - 1;
            const long crow = ((long)ancv[i] * Tmax + s) * ldc + col;
            const bf16_t* ksrc = isnew ? k_new + (long)r * ldn + col : kc + crow;
            const bf16_t* vsrc = isnew ? v_new + (long)r * ldn + col : vc + crow;
            kreg[i] = *reinterpret_cast<const u32x4_t*>(ksrc);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)vsrc,
                                             (__attribute__((address_space(3))) void*)(vl + (i * 32 + wave * 8) * 128), 16, 0, 0);
        }
    }
    // (this step's K / V are appended to the caches further down, once the loads above have landed: stored from inside the loop, the
    // store of the freshly loaded K row made every wave wait for its loads row slot by row slot)
    float qv[8];
    {
        const u32x4_t raw = *reinterpret_cast<const u32x4_t*>(q + (long)r * ldq + col);
        bf16_t qq[8];
        __builtin_memcpy(qq, &raw, 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) qv[j] = to_f32(qq[j]) * scale;
    }
#pragma unroll
    for (int i = 0; i < MAXROWS; ++i) {
        if (i >= nrow) break;
        bf16_t kk[8];
        __builtin_memcpy(kk, &kreg[i], 16);
        float x = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) x = fmaf(qv[j], to_f32(kk[j]), x);
        x = row8_sum(x);
        const int key = i * 32 + kslot;
        if (dch == 0 && key < len) sc[key] = x;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (fresh) {                                                // append this step's K / V to the caches (row r itself: anc[r][len - 1] == r)
        const int sn = len - 1;
        if (kslot == (sn & 31)) {
            const long nrow_off = ((long)r * Tmax + sn) * ldc + col;
            u32x4_t kn = kreg[0];
#pragma unroll
            for (int i = 1; i < MAXROWS; ++i) if ((sn >> 5) == i) kn = kreg[i];
            *reinterpret_cast<u32x4_t*>(kc + nrow_off) = kn;
            *reinterpret_cast<u32x4_t*>(vc + nrow_off) = *reinterpret_cast<const u32x4_t*>(v_new + (long)r * ldn + col);
        }
    }
    __syncthreads();
    if (wave == 0) {
        float mx = -INFINITY;
        for (int s = lane; s < len; s += 64) mx = fmaxf(mx, sc[s]);
        mx = warp_max(mx);
        float l = 0.f;
        for (int s = lane; s < len; s += 64) { const float p = __expf(sc[s] - mx); sc[s] = p; l += p; }
        l = warp_sum(l);
        const float inv = 1.f / l;
        for (int s = lane; s < len; s += 64) sc[s] *= inv;
    }
    __syncthreads();
    // P V: thread = (key quarter, dim)
    {
        const bf16_t* vcol = reinterpret_cast<const bf16_t*>(vl) + lane;
        float o = 0.f;
        for (int s = wave; s < len; s += XA_THREADS / 64) o = fmaf(sc[s], to_f32(vcol[s * 64]), o);
        red[wave][lane] = o;
    }
    __syncthreads();
    if (tid < 64) out[(long)r * ldo + h * 64 + tid] = (bf16_t)((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
}

}  // namespace

extern "C" long mmsum_decode_cross_attn_workspace(int n_entities, int H, int qpb, int B, int nmod) {
    if (n_entities <= 0 || H <= 0 || qpb <= 0 || B <= 0 || nmod <= 0) return 0;
    // [tickets: nmod * B * H words, padded to 256 bytes][per-entity outputs: n_entities * H * qpb * 64 floats]
    return (((long)nmod * B * H * 4 + 255) / 256) * 256 + (long)n_entities * H * qpb * 64 * (long)sizeof(float);
}

template <int QPB>
static int launch_xattn_f32(const XaArgsF& a, dim3 grid, hipStream_t s) {
    const size_t lds = (size_t)XA_MAXROWS * 32 * 256 + (size_t)QPB * XA_MAXROWS * 32 * sizeof(float);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(decode_cross_attn_f32_kernel<QPB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return MMSUM_ERR_HIP;
    decode_cross_attn_f32_kernel<QPB><<<grid, dim3(XA_THREADS), lds, s>>>(a);
    return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
}

static int decode_cross_attn_f32(const void* q, long ldq, const mmsum_xattn_memory* mods, int nmod, long ldkv, void* out, long ldo,
                                 int B, int qpb, int H, float scale, void* workspace, void* stream) {
    if ((((uintptr_t)q) & 15) || (ldq & 3) || (ldkv & 3)) return MMSUM_ERR_BAD_ALIGN;
    XaArgsF a;
    int ent = 0;
    for (int m = 0; m < 3; ++m) {
        if (m < nmod) {
            if (mods[m].N <= 0 || mods[m].N > 32 || mods[m].S <= 0 || mods[m].S > XA_MAXROWS * 32) return MMSUM_ERR_BAD_SHAPE;
            if ((((uintptr_t)mods[m].k | (uintptr_t)mods[m].v) & 15)) return MMSUM_ERR_BAD_ALIGN;
            a.mod[m] = XaModF{static_cast<const float*>(mods[m].k), static_cast<const float*>(mods[m].v), mods[m].pad, mods[m].null_entity, mods[m].N, mods[m].S, ent};
            ent += B * mods[m].N;
        } else {
            a.mod[m] = XaModF{nullptr, nullptr, nullptr, nullptr, 1, 1, 1 << 30};
        }
    }
    a.q = static_cast<const float*>(q); a.out = static_cast<float*>(out);
    a.tickets = static_cast<unsigned*>(workspace);
    a.part = reinterpret_cast<float*>(static_cast<char*>(workspace) + (((long)nmod * B * H * 4 + 255) / 256) * 256);
    a.ldq = ldq; a.ldkv = ldkv; a.ldo = ldo;
    a.nmod = nmod; a.B = B; a.H = H; a.qpb = qpb; a.R = B * qpb; a.scale = scale;
    const dim3 grid(H, ent);
    hipStream_t s = (hipStream_t)stream;
    switch (qpb) {
        case 1: return launch_xattn_f32<1>(a, grid, s);
        case 2: return launch_xattn_f32<2>(a, grid, s);
        case 3: return launch_xattn_f32<3>(a, grid, s);
        case 4: return launch_xattn_f32<4>(a, grid, s);
        case 5: return launch_xattn_f32<5>(a, grid, s);
        case 6: return launch_xattn_f32<6>(a, grid, s);
        case 7: return launch_xattn_f32<7>(a, grid, s);
        default: return launch_xattn_f32<8>(a, grid, s);
    }
}

extern "C" int mmsum_decode_cross_attn(int dtype, const void* q, long ldq, const mmsum_xattn_memory* mods, int nmod, long ldkv, void* out, long ldo,
                                       int B, int qpb, int H, float scale, void* workspace, void* stream) {
    if (nmod < 1 || nmod > 3 || B <= 0 || H <= 0 || qpb < 1 || qpb > XA_MAXQ || !mods) return MMSUM_ERR_BAD_SHAPE;
    if (workspace == nullptr) return MMSUM_ERR_WORKSPACE;
    if (dtype == MMSUM_F32) return decode_cross_attn_f32(q, ldq, mods, nmod, ldkv, out, ldo, B, qpb, H, scale, workspace, stream);
    if (dtype != MMSUM_BF16) return MMSUM_ERR_BAD_DTYPE;
    if ((((uintptr_t)q) & 1) || ((ldkv * 2) & 15)) return MMSUM_ERR_BAD_ALIGN;
    XaArgs a;
    int ent = 0;
    for (int m = 0; m < 3; ++m) {
        if (m < nmod) {
            if (mods[m].N <= 0 || mods[m].N > 32 || mods[m].S <= 0 || mods[m].S > XA_MAXROWS * 32) return MMSUM_ERR_BAD_SHAPE;
            if ((((uintptr_t)mods[m].k | (uintptr_t)mods[m].v) & 15)) return MMSUM_ERR_BAD_ALIGN;
            a.mod[m] = XaMod{static_cast<const bf16_t*>(mods[m].k), static_cast<const bf16_t*>(mods[m].v), mods[m].pad, mods[m].null_entity, mods[m].N, mods[m].S, ent};
            ent += B * mods[m].N;
        } else {
            a.mod[m] = XaMod{nullptr, nullptr, nullptr, nullptr, 1, 1, 1 << 30};
        }
    }
    a.q = static_cast<const bf16_t*>(q); a.out = static_cast<bf16_t*>(out);
    a.tickets = static_cast<unsigned*>(workspace);
    a.part = reinterpret_cast<float*>(static_cast<char*>(workspace) + (((long)nmod * B * H * 4 + 255) / 256) * 256);
    a.ldq = ldq; a.ldkv = ldkv; a.ldo = ldo;
    a.nmod = nmod; a.B = B; a.H = H; a.qpb = qpb; a.R = B * qpb; a.scale = scale;
    const dim3 grid(H, ent), block(XA_THREADS);
    hipStream_t s = (hipStream_t)stream;
    switch (qpb) {
        case 1: decode_cross_attn_kernel<1><<<grid, block, 0, s>>>(a); break;
        case 2: decode_cross_attn_kernel<2><<<grid, block, 0, s>>>(a); break;
        case 3: decode_cross_attn_kernel<3><<<grid, block, 0, s>>>(a); break;
        case 4: decode_cross_attn_kernel<4><<<grid, block, 0, s>>>(a); break;
        case 5: decode_cross_attn_kernel<5><<<grid, block, 0, s>>>(a); break;
        case 6: decode_cross_attn_kernel<6><<<grid, block, 0, s>>>(a); break;
        case 7: decode_cross_attn_kernel<7><<<grid, block, 0, s>>>(a); break;
        default: decode_cross_attn_kernel<8><<<grid, block, 0, s>>>(a); break;
    }
    return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
}

extern "C" int mmsum_decode_self_attn(int dtype, const void* q, long ldq, void* k_cache, void* v_cache, long ld_cache, const int* ancestors,
                                      void* out, long ldo, int rows, int H, int len, int Tmax, float scale, const void* k_new, const void* v_new,
                                      long ld_new, void* stream) {
    if (rows <= 0 || H <= 0 || len <= 0 || len > Tmax || Tmax > 256) return MMSUM_ERR_BAD_SHAPE;
    if ((k_new == nullptr) != (v_new == nullptr)) return MMSUM_ERR_BAD_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(rows, H);
    const bool al16 = !((((uintptr_t)q | (uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)k_new | (uintptr_t)v_new) & 15) || ((ldq | ld_cache | ld_new) & 7));
    if (dtype == MMSUM_BF16 && al16)        // 16-byte rows: the 256-thread kernel (every K / V row of the hypothesis in flight at once)
        decode_self_attn_bf16_kernel<<<grid, dim3(XA_THREADS), 0, s>>>((const bf16_t*)q, ldq, (bf16_t*)k_cache, (bf16_t*)v_cache, ld_cache, ancestors,
                                                                       (bf16_t*)out, ldo, len, Tmax, scale, (const bf16_t*)k_new, (const bf16_t*)v_new, ld_new);
    else if (dtype == MMSUM_BF16)
        decode_self_attn_kernel<bf16_t><<<grid, dim3(64), 0, s>>>((const bf16_t*)q, ldq, (bf16_t*)k_cache, (bf16_t*)v_cache, ld_cache, ancestors,
                                                                  (bf16_t*)out, ldo, len, Tmax, scale, (const bf16_t*)k_new, (const bf16_t*)v_new, ld_new);
    else if (dtype == MMSUM_F32)
        decode_self_attn_kernel<float><<<grid, dim3(64), 0, s>>>((const float*)q, ldq, (float*)k_cache, (float*)v_cache, ld_cache, ancestors,
                                                                 (float*)out, ldo, len, Tmax, scale, (const float*)k_new, (const float*)v_new, ld_new);
    else return MMSUM_ERR_BAD_DTYPE;
    return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
}
