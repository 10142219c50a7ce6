// Entity attention for gfx950: encoder self-attention, causal decoder self-attention and the
// reference's per-entity cross-attention with entity mean (modeling_multimodalsum.py:752-875),
// forward and backward, f32 softmax / accumulation.  head_dim = 64.
//
// Two kernel families share the decomposition -- forward / dQ: one workgroup = (query block of <= 128 rows, head), 4 waves x 32
// query rows, scores computed TRANSPOSED (S^T = K Q^T: keys on accumulator rows, the query on the lane) so the row softmax is
// register-local; dK/dV: one workgroup = (entity, head), waves own 32-key blocks and sweep the query blocks that attend to the
// entity (8 of the 9 leave-one-out passes for a review), so dK/dV need no atomics; softmax statistics (log-sum-exp, delta) go
// from the dQ kernel to the dK/dV kernel via `stats`; the [N,B,H,T,hd] per-entity outputs of the reference are never
// materialised (1/(count*l) is folded in and every entity accumulates into the same registers):
//   * bf16 (the step): attn_tr_* in the second half of this file -- transposing LDS reads, P / dS kept in registers as MFMA
//     operands, compact operands through row maps;
//   * f32 (parity mode): attn_*_pipe_kernel / attn_*_kernel -- tiles prefetched into registers, P / dS through an LDS image,
//     transposed tiles staged separately.
#include "mmsum_device.h"
#include "mmsum_kernels.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int HD = 64;
constexpr int ATT_THREADS = 256;
#ifndef MMSUM_ATTN_W64
#define MMSUM_ATTN_W64 0                  // 1 (make EXTRA="-DMMSUM_ATTN_W64=1 -mllvm -amdgpu-mfma-vgpr-form"): the round-6 one-wave-per-SIMD forward, measured 24 % SLOWER than the
#endif                                    // two-waves-per-SIMD kernel it was to replace (profiles/r06_attention_w64_fwd_pmc.txt, tools/r6_attn_ab.sh); kept for the record

template <typename T> struct AttnTraits {
    static constexpr int kSlabsHD = HD * sizeof(T) / SLAB_BYTES;      // slabs covering head_dim (2 bf16 / 4 f32)
    static constexpr int kSlabsPer32 = 32 * sizeof(T) / SLAB_BYTES;   // slabs covering 32 reduction elements (1 / 2)
};

// Bit n = entity n of business b takes part (not the excluded one, not all-padding).  Lane n reads entity n's flag and
// the wave ballots: ONE global round trip (the per-entity loop it replaces was N dependent load -> wait steps, and the
// dK/dV kernel ran it in every query-chunk iteration).  N <= 32.
__device__ __forceinline__ uint32_t valid_entities(const mmsum_attn_desc& d, int b, int excl) {
    const int n = threadIdx.x & 63;
    bool ok = n < d.N && n != excl;
    if (ok && d.null_entity) ok = d.null_entity[b * d.N + n] == 0;
    return (uint32_t)__ballot(ok);
}
__device__ __forceinline__ int count_valid(const mmsum_attn_desc& d, int b, int excl) { return __popc(valid_entities(d, b, excl)); }

// Retire loads into fragment registers HERE.  Without it the compiler places the wait at the first use inside the
// entity loop, where the only loads still in flight are the next entity's prefetch -- and vmcnt retires in order, so the
// prefetch would be waited for before the MFMAs it is supposed to hide under.
template <int N>
__device__ __forceinline__ void pin_frags(Frag (&f)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(f[i].c[0]), "+v"(f[i].c[1]));
}

// ---------------------------------------------------------------------------------------------
// Writing a wave's 32 x 64 result tile (two 32x32 accumulators side by side).  In the accumulator layout a lane owns one
// column and 16 scattered rows: a direct store is 32 two-byte stores per lane (and 32 two-byte loads when the result
// accumulates into memory).  Staged through LDS as f32 [32][64] instead, a lane owns half a row: four 16-byte
// stores (bf16), the accumulate operand fetched by four 16-byte loads issued before the staging.
// `stg` is this wave's OUT_STAGE_BYTES of LDS (the operand tiles are dead by then).  ADD: the staging area already holds a
// partial sum in the same layout (key-split dQ kernel).
// ---------------------------------------------------------------------------------------------
constexpr int OUT_STAGE_LD = 68;                               // floats per staged row
constexpr int OUT_STAGE_BYTES = 32 * OUT_STAGE_LD * 4;

template <typename T, bool ADD = false>
__device__ __forceinline__ void flush_tile(float* stg, const f32x16_t (&acc)[2], T* g, long ld, int nrows, bool accumulate, int lane) {
    if (nrows <= 0) return;                                    // wave-uniform
    constexpr int EPV = 16 / sizeof(T), NV = 32 / EPV;         // elements per 16-byte vector, vectors per half row
    const bool vec = ((((uintptr_t)g) | (uintptr_t)(ld * (long)sizeof(T))) & 15) == 0;
    const int row = lane >> 1, half = lane & 1;
    const bool mine = row < nrows;
    T* o = g + (long)row * ld + half * 32;
    u32x4_t prev[NV];
    if (vec && accumulate && mine) {
#pragma unroll
        for (int v = 0; v < NV; ++v) prev[v] = *reinterpret_cast<const u32x4_t*>(o + v * EPV);
    }
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float* p = stg + acc_row(r, lane) * OUT_STAGE_LD + db * 32 + (lane & 31);
            *p = ADD ? *p + acc[db][r] : acc[db][r];
        }
    __builtin_amdgcn_wave_barrier();
    if (mine) {
        const float* srow = stg + row * OUT_STAGE_LD + half * 32;
        if (vec) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                float x[EPV];
#pragma unroll
                for (int q4 = 0; q4 < EPV / 4; ++q4) {
                    const f32x4_t t = *reinterpret_cast<const f32x4_t*>(srow + v * EPV + 4 * q4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[4 * q4 + e] = t[e];
                }
                T w[EPV];
                if (accumulate) {
                    __builtin_memcpy(w, &prev[v], 16);
#pragma unroll
                    for (int e = 0; e < EPV; ++e) x[e] += to_f32(w[e]);
                }
#pragma unroll
                for (int e = 0; e < EPV; ++e) w[e] = from_f32<T>(x[e]);
                u32x4_t pk;
                __builtin_memcpy(&pk, w, 16);
                *reinterpret_cast<u32x4_t*>(o + v * EPV) = pk;
            }
        } else {
            float pv[32];
#pragma unroll
            for (int e = 0; e < 32; ++e) pv[e] = accumulate ? to_f32(o[e]) : 0.f;
#pragma unroll
            for (int e = 0; e < 32; ++e) o[e] = from_f32<T>(srow[e] + pv[e]);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// Stage the key mask of one entity into LDS: 1 = masked (padded key, or key index >= S).
__device__ __forceinline__ void stage_mask(uint8_t* m, const uint8_t* pad, long ent, int S, int spad, int tid) {
    for (int s = tid; s < spad; s += ATT_THREADS) m[s] = (s >= S) ? 1 : (pad ? pad[ent * S + s] : 0);
}

// Scores of one entity for this wave's 32 queries, transposed: sacc[kb][reg] = q . k[key], key =
// kb*32 + acc_row(reg).  Then scale + mask + softmax statistics.  On return sacc holds
// exp(s - m) (0 for masked keys); *m_out, *l_out are the row max and sum for the lane's query.
template <typename T, int NKB>
__device__ __forceinline__ void scores_softmax(f32x16_t (&sacc)[NKB], const char* ktile, int spad, const Frag* qf,
                                               const uint8_t* maskb, int S, float scale, bool causal, int qpos, int lane,
                                               float& m_out, float& l_out) {
    constexpr int NS = AttnTraits<T>::kSlabsHD;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        sacc[kb] = zero_acc();
        if (kb * 32 < S) {
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) {
                const Frag a = lds_frag<T>(ktile + sl * (spad * SLAB_BYTES), kb * 32, lane);
                mma_slab<T>(sacc[kb], a, qf[sl]);
            }
        }
    }
    float m = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + acc_row(r, lane);
            const bool masked = (kb * 32 >= S) || maskb[key] || (causal && key > qpos);
            const float s = masked ? -INFINITY : sacc[kb][r] * scale;
            sacc[kb][r] = s;
            m = fmaxf(m, s);
        }
    }
    m = wave_half_max(m);
    float l = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = (m == -INFINITY) ? 0.f : __expf(sacc[kb][r] - m);
            sacc[kb][r] = p;
            l += p;
        }
    }
    l = wave_half_sum(l);
    m_out = m;
    l_out = l;
}

// Leaner variant used by the pipelined kernels.  Works in the log2 domain (one FMA folds scale*log2(e) and
// the key mask, which is an additive 0 / -inf vector read from LDS four keys at a time), exponentials
// are v_exp_f32 (2^x), fragment addresses are hoisted, and for causal self-attention the key blocks above
// the wave's diagonal block are skipped altogether.  Returns the row max / sum in the log2 domain.
// Number of keys of the staged entity up to its last unmasked one (all waves agree after the next barrier): the key blocks
// past it are pure padding (probability exactly 0) and their matrix products are skipped -- a 75-token review uses 3 of 4.
__device__ __forceinline__ void publish_key_extent(int* slots, uint8_t masked, int S, int tid) {
    int e = (tid < S && !masked) ? tid + 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e = max(e, __shfl_xor(e, o));
    if ((tid & 63) == 0) slots[tid >> 6] = e;
}
template <int NWAVES>
__device__ __forceinline__ int read_key_extent(const int* slots) {
    int e = 0;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) e = max(e, slots[w]);
    return e;
}

#define LOG2E_F 1.4426950408889634f
#define LN2_F 0.6931471805599453f
template <typename T, int NKB, bool CAUSAL>
__device__ __forceinline__ void scores_softmax2(f32x16_t (&sacc)[NKB], const char* ktile, int spad, const Frag* qf, const float* biasf,
                                                int S, float scale, int qpos, int wave, int lane, const FragOff& fo, float& m_out,
                                                float& l_out) {
    constexpr int NS = AttnTraits<T>::kSlabsHD;
    const float c2 = scale * LOG2E_F;
    const int h = lane >> 5;
    float m = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const bool active = (kb * 32 < S) && (!CAUSAL || kb <= wave);
        sacc[kb] = zero_acc();
        if (active) {
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) {
                const Frag a = lds_frag_o(ktile + sl * (spad * SLAB_BYTES) + kb * 32 * SLAB_BYTES, fo);
                mma_slab<T>(sacc[kb], a, qf[sl]);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4_t bias = *reinterpret_cast<const f32x4_t*>(biasf + kb * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float s = fmaf(sacc[kb][4 * g + j], c2, bias[j]);
                    if (CAUSAL && kb == wave && (kb * 32 + 8 * g + 4 * h + j) > qpos) s = -INFINITY;
                    sacc[kb][4 * g + j] = s;
                    m = fmaxf(m, s);
                }
            }
        }
    }
    m = wave_half_max(m);
    const float ms = (m == -INFINITY) ? 0.f : m;
    float l = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const bool active = (kb * 32 < S) && (!CAUSAL || kb <= wave);
        if (active) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(sacc[kb][r] - ms);
                sacc[kb][r] = pv;
                l += pv;
            }
        }
    }
    l = wave_half_sum(l);
    m_out = ms;
    l_out = l;
}

// ---------------------------------------------------------------------------------------------
// Forward
// ---------------------------------------------------------------------------------------------
template <typename T, int NKB>
__global__ __launch_bounds__(ATT_THREADS) void attn_fwd_kernel(mmsum_attn_desc d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SPAD = NKB * 32;
    constexpr int NS = AttnTraits<T>::kSlabsHD;
    constexpr int TILE = SPAD * HD * sizeof(T);
    char* tile = smem;
    char* img = smem + TILE + (threadIdx.x >> 6) * ImageTraits<T>::kBytes;
    uint8_t* maskb = reinterpret_cast<uint8_t*>(smem + TILE + 4 * ImageTraits<T>::kBytes);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, qb = blockIdx.y;
    const int b = qb / d.qpb;
    const int excl = d.exclude_self ? (qb % d.qpb) : -1;
    const int cnt = count_valid(d, b, excl);
    const float inv_cnt = cnt > 0 ? 1.f / (float)cnt : 0.f;

    const T* Q = static_cast<const T*>(d.q);
    const T* K = static_cast<const T*>(d.k);
    const T* V = static_cast<const T*>(d.v);
    T* O = static_cast<T*>(d.out);

    const int qpos = wave * 32 + (lane & 31);
    const bool qvalid = qpos < d.T;
    Frag qf[NS];
    {
        const T* qrow = Q + ((long)qb * d.T + qpos) * d.ldq + h * HD;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) qf[sl] = global_frag<T>(qrow + sl * ElemTraits<T>::kPerSlab, lane, qvalid);
    }
    f32x16_t oacc[2] = {zero_acc(), zero_acc()};

    for (int n = 0; n < d.N; ++n) {
        if (n == excl) continue;
        const long ent = (long)b * d.N + n;
        if (d.null_entity && d.null_entity[ent]) continue;
        const long row0 = ent * d.S;
        __syncthreads();
        stage_natural<T, SPAD, NS, ATT_THREADS>(tile, K + row0 * d.ldk + h * HD, d.ldk, 0, d.S, 0, HD, tid);
        stage_mask(maskb, d.pad, ent, d.S, SPAD, tid);
        __syncthreads();
        f32x16_t sacc[NKB];
        float m, l;
        scores_softmax<T, NKB>(sacc, tile, SPAD, qf, maskb, d.S, d.scale, d.causal, qpos + d.causal_q0, lane, m, l);
        const float norm = (l > 0.f) ? inv_cnt / l : 0.f;
        __syncthreads();
        // V^T: tile row = d (64 rows), reduction = key
        stage_transposed<T, HD, NKB * AttnTraits<T>::kSlabsPer32, ATT_THREADS>(tile, V + row0 * d.ldv + h * HD, d.ldv, 0, HD, 0, d.S, tid);
        __syncthreads();
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (kb * 32 < d.S) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[kb][r] *= norm;
                acc_to_image<T>(img, sacc[kb], lane);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    mma_image<T>(oacc[db], img, tile + kb * AttnTraits<T>::kSlabsPer32 * (HD * SLAB_BYTES), HD * SLAB_BYTES, db * 32, lane);
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        const int col = h * HD + db * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = wave * 32 + acc_row(r, lane);
            if (qq < d.T) O[((long)qb * d.T + qq) * d.ldo + col] = from_f32<T>(oacc[db][r]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Backward, kernel A: dQ (+ per-entity log-sum-exp and delta for kernel B)
// ---------------------------------------------------------------------------------------------
template <typename T, int NKB>
__global__ __launch_bounds__(ATT_THREADS) void attn_bwd_dq_kernel(mmsum_attn_desc d, const T* __restrict__ dO, long lddo,
                                                                  T* __restrict__ dQ, long lddq, int accumulate_dq,
                                                                  float* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SPAD = NKB * 32;
    constexpr int NS = AttnTraits<T>::kSlabsHD;
    constexpr int TILE = SPAD * HD * sizeof(T);
    char* tile = smem;
    char* img = smem + TILE + (threadIdx.x >> 6) * ImageTraits<T>::kBytes;
    uint8_t* maskb = reinterpret_cast<uint8_t*>(smem + TILE + 4 * ImageTraits<T>::kBytes);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, qb = blockIdx.y;
    const int b = qb / d.qpb;
    const int excl = d.exclude_self ? (qb % d.qpb) : -1;
    const int cnt = count_valid(d, b, excl);
    const float inv_cnt = cnt > 0 ? 1.f / (float)cnt : 0.f;

    const T* Q = static_cast<const T*>(d.q);
    const T* K = static_cast<const T*>(d.k);
    const T* V = static_cast<const T*>(d.v);

    const int qpos = wave * 32 + (lane & 31);
    const bool qvalid = qpos < d.T;
    Frag qf[NS], dof[NS];
    {
        const T* qrow = Q + ((long)qb * d.T + qpos) * d.ldq + h * HD;
        const T* drow = dO + ((long)qb * d.T + qpos) * lddo + h * HD;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            qf[sl] = global_frag<T>(qrow + sl * ElemTraits<T>::kPerSlab, lane, qvalid);
            dof[sl] = global_frag<T>(drow + sl * ElemTraits<T>::kPerSlab, lane, qvalid);
        }
    }
    f32x16_t dqacc[2] = {zero_acc(), zero_acc()};

    for (int n = 0; n < d.N; ++n) {
        if (n == excl) continue;
        const long ent = (long)b * d.N + n;
        if (d.null_entity && d.null_entity[ent]) continue;
        const long row0 = ent * d.S;
        __syncthreads();
        stage_natural<T, SPAD, NS, ATT_THREADS>(tile, K + row0 * d.ldk + h * HD, d.ldk, 0, d.S, 0, HD, tid);
        stage_mask(maskb, d.pad, ent, d.S, SPAD, tid);
        __syncthreads();
        f32x16_t p[NKB];
        float m, l;
        scores_softmax<T, NKB>(p, tile, SPAD, qf, maskb, d.S, d.scale, d.causal, qpos + d.causal_q0, lane, m, l);
        const float invl = (l > 0.f) ? 1.f / l : 0.f;
        __syncthreads();
        stage_natural<T, SPAD, NS, ATT_THREADS>(tile, V + row0 * d.ldv + h * HD, d.ldv, 0, d.S, 0, HD, tid);
        __syncthreads();
        // dP^T[key][q] = v[key] . dO[q]
        f32x16_t dp[NKB];
        float delta = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            dp[kb] = zero_acc();
            if (kb * 32 < d.S) {
#pragma unroll
                for (int sl = 0; sl < NS; ++sl) {
                    const Frag a = lds_frag<T>(tile + sl * (SPAD * SLAB_BYTES), kb * 32, lane);
                    mma_slab<T>(dp[kb], a, dof[sl]);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    p[kb][r] *= invl;                 // normalised probability
                    dp[kb][r] *= inv_cnt;             // dO_e = dO / count
                    delta += p[kb][r] * dp[kb][r];
                }
            }
        }
        delta = wave_half_sum(delta);
        if (lane < 32 && qvalid) {
            float* st = stats + ((((long)qb * d.N + n) * d.H + h) * d.T + qpos) * 2;
            st[0] = (m + __logf(l)) * LOG2E_F;   // log2 domain, as the dK/dV kernel expects
            st[1] = delta;
        }
        __syncthreads();
        // K^T: tile row = d, reduction = key
        stage_transposed<T, HD, NKB * AttnTraits<T>::kSlabsPer32, ATT_THREADS>(tile, K + row0 * d.ldk + h * HD, d.ldk, 0, HD, 0, d.S, tid);
        __syncthreads();
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (kb * 32 < d.S) {
#pragma unroll
                for (int r = 0; r < 16; ++r) p[kb][r] = p[kb][r] * (dp[kb][r] - delta) * d.scale;   // dS^T
                acc_to_image<T>(img, p[kb], lane);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    mma_image<T>(dqacc[db], img, tile + kb * AttnTraits<T>::kSlabsPer32 * (HD * SLAB_BYTES), HD * SLAB_BYTES, db * 32, lane);
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    // the accumulate operand is requested for all 32 elements first: a load inside the store loop would wait for the
    // previous store's acknowledgement every time (vmcnt retires in order)
    float prev[2][16];
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        const int col = h * HD + db * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = wave * 32 + acc_row(r, lane);
            prev[db][r] = (accumulate_dq && qq < d.T) ? to_f32(dQ[((long)qb * d.T + qq) * lddq + col]) : 0.f;
        }
    }
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        const int col = h * HD + db * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = wave * 32 + acc_row(r, lane);
            if (qq < d.T) dQ[((long)qb * d.T + qq) * lddq + col] = from_f32<T>(dqacc[db][r] + prev[db][r]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Backward, kernel B: dK, dV
// ---------------------------------------------------------------------------------------------
template <typename T, int NKB>
__global__ __launch_bounds__(ATT_THREADS) void attn_bwd_dkv_kernel(mmsum_attn_desc d, const T* __restrict__ dO, long lddo,
                                                                   T* __restrict__ dK, long lddk, T* __restrict__ dV, long lddv,
                                                                   const float* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NS = AttnTraits<T>::kSlabsHD;
    constexpr int TQ = 64;                                   // query rows staged per step
    constexpr int NOWN = (NKB + 3) / 4;                      // key blocks owned by a wave
    constexpr int QT_TILE = TQ * HD * sizeof(T);             // one staged tile
    constexpr int NSQ = TQ * sizeof(T) / SLAB_BYTES;         // slabs covering TQ queries (2 / 4)
    char* qn = smem;                  // Q natural  [TQ rows][64]
    char* don = smem + QT_TILE;       // dO natural
    char* qt = smem + 2 * QT_TILE;    // Q^T  [64 rows (d)][TQ]
    char* dot = smem + 3 * QT_TILE;   // dO^T
    char* imgP = smem + 4 * QT_TILE + (threadIdx.x >> 6) * 2 * ImageTraits<T>::kBytes;
    char* imgS = imgP + ImageTraits<T>::kBytes;
    float* st = reinterpret_cast<float*>(smem + 4 * QT_TILE + 8 * ImageTraits<T>::kBytes);   // [TQ][2]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x;
    const long ent = blockIdx.y;
    const int b = (int)(ent / d.N), n = (int)(ent % d.N);
    const long row0 = ent * d.S;
    const T* Q = static_cast<const T*>(d.q);
    const T* K = static_cast<const T*>(d.k);
    const T* V = static_cast<const T*>(d.v);
    const bool is_null = d.null_entity && d.null_entity[ent];

    Frag kf[NOWN][NS], vf[NOWN][NS];
    bool keymask[NOWN];
    f32x16_t dkacc[NOWN][2], dvacc[NOWN][2];
#pragma unroll
    for (int o = 0; o < NOWN; ++o) {
        const int key = (wave + 4 * o) * 32 + (lane & 31);
        const bool kvalid = key < d.S;
        keymask[o] = !kvalid || (d.pad && d.pad[ent * d.S + (kvalid ? key : 0)]);
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            kf[o][sl] = global_frag<T>(K + (row0 + key) * d.ldk + h * HD + sl * ElemTraits<T>::kPerSlab, lane, kvalid);
            vf[o][sl] = global_frag<T>(V + (row0 + key) * d.ldv + h * HD + sl * ElemTraits<T>::kPerSlab, lane, kvalid);
        }
        dkacc[o][0] = dkacc[o][1] = dvacc[o][0] = dvacc[o][1] = zero_acc();
    }

    if (!is_null) {
        for (int i = 0; i < d.qpb; ++i) {
            if (d.exclude_self && i == n) continue;
            const int qb = b * d.qpb + i;
            const int cnt = count_valid(d, b, d.exclude_self ? i : -1);
            const float inv_cnt = cnt > 0 ? 1.f / (float)cnt : 0.f;
            const T* qbase = Q + (long)qb * d.T * d.ldq + h * HD;
            const T* dobase = dO + (long)qb * d.T * lddo + h * HD;
            const float* sbase = stats + (((long)qb * d.N + n) * d.H + h) * d.T * 2;
            for (int qc = 0; qc < d.T; qc += TQ) {
                __syncthreads();
                stage_natural<T, TQ, NS, ATT_THREADS>(qn, qbase, d.ldq, qc, d.T, 0, HD, tid);
                stage_natural<T, TQ, NS, ATT_THREADS>(don, dobase, lddo, qc, d.T, 0, HD, tid);
                // transposed: tile row = d (64), reduction index = query (element (d, q) at base[q*ld + d])
                stage_transposed<T, HD, NSQ, ATT_THREADS>(qt, qbase, d.ldq, 0, HD, qc, d.T, tid);
                stage_transposed<T, HD, NSQ, ATT_THREADS>(dot, dobase, lddo, 0, HD, qc, d.T, tid);
                for (int j = tid; j < TQ * 2; j += ATT_THREADS) st[j] = (qc + (j >> 1) < d.T) ? sbase[(long)qc * 2 + j] : 0.f;
                __syncthreads();
#pragma unroll
                for (int o = 0; o < NOWN; ++o) {
                    const int kb = wave + 4 * o;
                    if (kb >= NKB || kb * 32 >= d.S) continue;
                    const int key = kb * 32 + (lane & 31);
#pragma unroll
                    for (int qq = 0; qq < TQ / 32; ++qq) {
                        if (qc + qq * 32 >= d.T) continue;
                        f32x16_t s = zero_acc(), dp = zero_acc();
#pragma unroll
                        for (int sl = 0; sl < NS; ++sl) {
                            const Frag aq = lds_frag<T>(qn + sl * (TQ * SLAB_BYTES), qq * 32, lane);
                            mma_slab<T>(s, aq, kf[o][sl]);
                            const Frag ad = lds_frag<T>(don + sl * (TQ * SLAB_BYTES), qq * 32, lane);
                            mma_slab<T>(dp, ad, vf[o][sl]);
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int ql = qq * 32 + acc_row(r, lane);      // query within the staged chunk
                            const int qg = qc + ql;                         // query position in the block
                            const bool masked = keymask[o] || qg >= d.T || (d.causal && key > qg + d.causal_q0);
                            const float pr = masked ? 0.f : __expf(s[r] * d.scale - st[ql * 2]);
                            dp[r] = pr * (dp[r] * inv_cnt - st[ql * 2 + 1]) * d.scale;   // dS
                            s[r] = pr * inv_cnt;                                        // P / count
                        }
                        acc_to_image<T>(imgP, s, lane);
                        acc_to_image<T>(imgS, dp, lane);
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int db = 0; db < 2; ++db) {
                            mma_image<T>(dvacc[o][db], imgP, dot + qq * AttnTraits<T>::kSlabsPer32 * (HD * SLAB_BYTES), HD * SLAB_BYTES, db * 32, lane);
                            mma_image<T>(dkacc[o][db], imgS, qt + qq * AttnTraits<T>::kSlabsPer32 * (HD * SLAB_BYTES), HD * SLAB_BYTES, db * 32, lane);
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < NOWN; ++o) {
        const int kb = wave + 4 * o;
        if (kb >= NKB) continue;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const int col = h * HD + db * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * 32 + acc_row(r, lane);
                if (key < d.S) {
                    dK[(row0 + key) * lddk + col] = from_f32<T>(dkacc[o][db][r]);
                    dV[(row0 + key) * lddv + col] = from_f32<T>(dvacc[o][db][r]);
                }
            }
        }
    }
}

// =============================================================================================
// Software-pipelined variants: the global loads of the NEXT entity (or query chunk) are issued into
// registers before the MFMA/softmax work of the current one and committed to LDS afterwards, so a
// workgroup pays one global round trip per kernel instead of two or three per entity.  K and V
// (forward), K, V and K^T (dQ), Q, dO, Q^T and dO^T (dK/dV) live in separate LDS regions.
// =============================================================================================
template <typename T, int NKB, bool CAUSAL>
__global__ __launch_bounds__(ATT_THREADS, (NKB <= 4 && sizeof(T) == 2) ? 2 : 1) void attn_fwd_pipe_kernel(mmsum_attn_desc d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SPAD = NKB * 32;
    constexpr int NS = AttnTraits<T>::kSlabsHD;
    constexpr int TILE = SPAD * HD * sizeof(T);
    char* ktile = smem;
    char* vtile = smem + TILE;
    char* img = smem + 2 * TILE + (threadIdx.x >> 6) * ImageTraits<T>::kBytes;
    float* biasf = reinterpret_cast<float*>(smem + 2 * TILE + 4 * ImageTraits<T>::kBytes);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const FragOff fo = frag_off<T>(lane);
    const int h = blockIdx.x, qb = blockIdx.y;
    const int b = qb / d.qpb;
    const int excl = d.exclude_self ? (qb % d.qpb) : -1;
    uint32_t rem = valid_entities(d, b, excl);
    const int cnt = __popc(rem);
    const float inv_cnt = cnt > 0 ? 1.f / (float)cnt : 0.f;

    const T* Q = static_cast<const T*>(d.q);
    const T* K = static_cast<const T*>(d.k);
    const T* V = static_cast<const T*>(d.v);
    T* O = static_cast<T*>(d.out);

    const int qpos = wave * 32 + (lane & 31);
    const int cwave = wave + (d.causal_q0 >> 5);      // causal: the key block of this wave's diagonal (a query block may start past key 0)
    const bool qvalid = qpos < d.T;
    Frag qf[NS];
    {
        const T* qrow = Q + ((long)qb * d.T + qpos) * d.ldq + h * HD;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) qf[sl] = global_frag<T>(qrow + sl * ElemTraits<T>::kPerSlab, lane, qvalid);
    }
    pin_frags(qf);
    f32x16_t oacc[2] = {zero_acc(), zero_acc()};

    NatTile<T, SPAD, NS, ATT_THREADS> kreg;
    TrTile<T, HD, NKB * AttnTraits<T>::kSlabsPer32, ATT_THREADS> vreg;
    uint8_t mreg = 1;
    auto prefetch = [&](int n) {
        const long ent = (long)b * d.N + n;
        const long row0 = ent * d.S;
        kreg.load(K + row0 * d.ldk + h * HD, d.ldk, 0, d.S, 0, HD, tid);
        vreg.load(V + row0 * d.ldv + h * HD, d.ldv, 0, HD, 0, d.S, tid);
        mreg = (tid >= d.S) ? 1 : (d.pad ? d.pad[ent * d.S + tid] : 0);
    };
    if (rem) prefetch(__builtin_ctz(rem));
    while (rem) {
        rem &= rem - 1;
        __syncthreads();
        kreg.commit(ktile, tid);
        vreg.commit(vtile, tid);
        if (tid < SPAD) biasf[tid] = mreg ? -INFINITY : 0.f;
        publish_key_extent(reinterpret_cast<int*>(biasf + SPAD), mreg, d.S, tid);
        __syncthreads();
        const int slen = read_key_extent<ATT_THREADS / 64>(reinterpret_cast<const int*>(biasf + SPAD));
        if (rem) prefetch(__builtin_ctz(rem));
        f32x16_t sacc[NKB];
        float m, l;
        scores_softmax2<T, NKB, CAUSAL>(sacc, ktile, SPAD, qf, biasf, slen, d.scale, qpos + d.causal_q0, cwave, lane, fo, m, l);
        const float norm = (l > 0.f) ? inv_cnt / l : 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (kb * 32 < slen && (!CAUSAL || kb <= cwave)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[kb][r] *= norm;
                acc_to_image<T>(img, sacc[kb], lane);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    mma_image_o<T>(oacc[db], img, vtile + kb * AttnTraits<T>::kSlabsPer32 * (HD * SLAB_BYTES) + db * 32 * SLAB_BYTES, HD * SLAB_BYTES, fo);
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __syncthreads();                                           // the tiles are dead: their LDS stages the output rows
    flush_tile<T>(reinterpret_cast<float*>(smem + wave * OUT_STAGE_BYTES), oacc, O + ((long)qb * d.T + wave * 32) * d.ldo + h * HD, d.ldo,
                  d.T - wave * 32, false, lane);
}

template <typename T, int NKB, bool CAUSAL>
__global__ __launch_bounds__(ATT_THREADS, (NKB <= 2 && sizeof(T) == 2) ? 2 : 1) void attn_bwd_dq_pipe_kernel(mmsum_attn_desc d, const T* __restrict__ dO, long lddo,
                                                                       T* __restrict__ dQ, long lddq, int accumulate_dq,
                                                                       float* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SPAD = NKB * 32;
    constexpr int NS = AttnTraits<T>::kSlabsHD;
    constexpr int TILE = SPAD * HD * sizeof(T);
    char* ktile = smem;
    char* vtile = smem + TILE;
    char* kttile = smem + 2 * TILE;
    char* img = smem + 3 * TILE + (threadIdx.x >> 6) * ImageTraits<T>::kBytes;
    float* biasf = reinterpret_cast<float*>(smem + 3 * TILE + 4 * ImageTraits<T>::kBytes);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const FragOff fo = frag_off<T>(lane);
    const int h = blockIdx.x, qb = blockIdx.y;
    const int b = qb / d.qpb;
    const int excl = d.exclude_self ? (qb % d.qpb) : -1;
    uint32_t rem = valid_entities(d, b, excl);
    const int cnt = __popc(rem);
    const float inv_cnt = cnt > 0 ? 1.f / (float)cnt : 0.f;

    const T* Q = static_cast<const T*>(d.q);
    const T* K = static_cast<const T*>(d.k);
    const T* V = static_cast<const T*>(d.v);

    const int qpos = wave * 32 + (lane & 31);
    const int cwave = wave + (d.causal_q0 >> 5);      // causal: the key block of this wave's diagonal (a query block may start past key 0)
    const bool qvalid = qpos < d.T;
    Frag qf[NS], dof[NS];
    {
        const T* qrow = Q + ((long)qb * d.T + qpos) * d.ldq + h * HD;
        const T* drow = dO + ((long)qb * d.T + qpos) * lddo + h * HD;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            qf[sl] = global_frag<T>(qrow + sl * ElemTraits<T>::kPerSlab, lane, qvalid);
            dof[sl] = global_frag<T>(drow + sl * ElemTraits<T>::kPerSlab, lane, qvalid);
        }
    }
    pin_frags(qf);
    pin_frags(dof);
    f32x16_t dqacc[2] = {zero_acc(), zero_acc()};

    NatTile<T, SPAD, NS, ATT_THREADS> kreg, vreg;
    TrTile<T, HD, NKB * AttnTraits<T>::kSlabsPer32, ATT_THREADS> ktreg;
    uint8_t mreg = 1;
    int cur_n = 0, next_n = 0;
    auto prefetch = [&](int n) {
        const long ent = (long)b * d.N + n;
        const long row0 = ent * d.S;
        kreg.load(K + row0 * d.ldk + h * HD, d.ldk, 0, d.S, 0, HD, tid);
        vreg.load(V + row0 * d.ldv + h * HD, d.ldv, 0, d.S, 0, HD, tid);
        ktreg.load(K + row0 * d.ldk + h * HD, d.ldk, 0, HD, 0, d.S, tid);
        mreg = (tid >= d.S) ? 1 : (d.pad ? d.pad[ent * d.S + tid] : 0);
        next_n = n;
    };
    if (rem) prefetch(__builtin_ctz(rem));
    while (rem) {
        rem &= rem - 1;
        __syncthreads();
        kreg.commit(ktile, tid);
        vreg.commit(vtile, tid);
        ktreg.commit(kttile, tid);
        if (tid < SPAD) biasf[tid] = mreg ? -INFINITY : 0.f;
        publish_key_extent(reinterpret_cast<int*>(biasf + SPAD), mreg, d.S, tid);
        cur_n = next_n;
        __syncthreads();
        const int slen = read_key_extent<ATT_THREADS / 64>(reinterpret_cast<const int*>(biasf + SPAD));
        if (rem) prefetch(__builtin_ctz(rem));
        f32x16_t p[NKB];
        float m, l;
        scores_softmax2<T, NKB, CAUSAL>(p, ktile, SPAD, qf, biasf, slen, d.scale, qpos + d.causal_q0, cwave, lane, fo, m, l);
        const float invl = (l > 0.f) ? 1.f / l : 0.f;
        // dP^T = V dO^T.  With two resident workgroups per CU (NKB <= 2) the registers to keep it for all key blocks
        // are not there, so it is computed twice (delta pass, then dS pass); the text entities (NKB 3..4, one
        // workgroup per CU) keep it.
        constexpr bool KEEP_DP = (sizeof(T) == 2 && NKB >= 3 && NKB <= 4);
        f32x16_t dpkeep[KEEP_DP ? NKB : 1];
        float delta = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (kb * 32 < slen && (!CAUSAL || kb <= cwave)) {
                f32x16_t dpk = zero_acc();
#pragma unroll
                for (int sl = 0; sl < NS; ++sl) {
                    const Frag a = lds_frag_o(vtile + sl * (SPAD * SLAB_BYTES) + kb * 32 * SLAB_BYTES, fo);
                    mma_slab<T>(dpk, a, dof[sl]);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    p[kb][r] *= invl;                 // normalised probability
                    delta += p[kb][r] * dpk[r];
                }
                if constexpr (KEEP_DP) dpkeep[kb] = dpk;
            }
        }
        delta = wave_half_sum(delta) * inv_cnt;       // dO_e = dO / count
        if (lane < 32 && qvalid) {
            float* st = stats + ((((long)qb * d.N + cur_n) * d.H + h) * d.T + qpos) * 2;
            st[0] = m + __log2f(l);            // log-sum-exp in the log2 domain (dK/dV kernel uses exp2)
            st[1] = delta;
        }
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (kb * 32 < slen && (!CAUSAL || kb <= cwave)) {
                f32x16_t dpk;
                if constexpr (KEEP_DP) {
                    dpk = dpkeep[kb];
                } else {
                    dpk = zero_acc();
#pragma unroll
                    for (int sl = 0; sl < NS; ++sl) {
                        const Frag a = lds_frag_o(vtile + sl * (SPAD * SLAB_BYTES) + kb * 32 * SLAB_BYTES, fo);
                        mma_slab<T>(dpk, a, dof[sl]);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) dpk[r] = p[kb][r] * (dpk[r] * inv_cnt - delta) * d.scale;   // dS^T
                acc_to_image<T>(img, dpk, lane);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    mma_image_o<T>(dqacc[db], img, kttile + kb * AttnTraits<T>::kSlabsPer32 * (HD * SLAB_BYTES) + db * 32 * SLAB_BYTES, HD * SLAB_BYTES, fo);
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __syncthreads();                                           // the tiles are dead: their LDS stages the output rows
    flush_tile<T>(reinterpret_cast<float*>(smem + wave * OUT_STAGE_BYTES), dqacc, dQ + ((long)qb * d.T + wave * 32) * lddq + h * HD, lddq,
                  d.T - wave * 32, accumulate_dq != 0, lane);
}

template <typename T, int NKB, bool CAUSAL>
__global__ __launch_bounds__(ATT_THREADS, (NKB <= 4 && sizeof(T) == 2) ? 2 : 1) void attn_bwd_dkv_pipe_kernel(mmsum_attn_desc d, const T* __restrict__ dO, long lddo,
                                                                        T* __restrict__ dK, long lddk, T* __restrict__ dV, long lddv,
                                                                        const float* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NS = AttnTraits<T>::kSlabsHD;
    constexpr int TQ = 64;
    constexpr int NOWN = (NKB + 3) / 4;
    constexpr int QT_TILE = TQ * HD * sizeof(T);
    constexpr int NSQ = TQ * sizeof(T) / SLAB_BYTES;
    char* qn = smem;
    char* don = smem + QT_TILE;
    char* qt = smem + 2 * QT_TILE;
    char* dot = smem + 3 * QT_TILE;
    char* imgP = smem + 4 * QT_TILE + (threadIdx.x >> 6) * 2 * ImageTraits<T>::kBytes;
    char* imgS = imgP + ImageTraits<T>::kBytes;
    float* st = reinterpret_cast<float*>(smem + 4 * QT_TILE + 8 * ImageTraits<T>::kBytes);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const FragOff fo = frag_off<T>(lane);
    const float c2 = d.scale * LOG2E_F;
    const int hh = lane >> 5;
    const int h = blockIdx.x;
    const long ent = blockIdx.y;
    const int b = (int)(ent / d.N), n = (int)(ent % d.N);
    const long row0 = ent * d.S;
    const T* Q = static_cast<const T*>(d.q);
    const T* K = static_cast<const T*>(d.k);
    const T* V = static_cast<const T*>(d.v);
    const bool is_null = d.null_entity && d.null_entity[ent];

    Frag kf[NOWN][NS], vf[NOWN][NS];
    bool keymask[NOWN];
    f32x16_t dkacc[NOWN][2], dvacc[NOWN][2];
#pragma unroll
    for (int o = 0; o < NOWN; ++o) {
        const int key = (wave + 4 * o) * 32 + (lane & 31);
        const bool kvalid = key < d.S;
        keymask[o] = !kvalid || (d.pad && d.pad[ent * d.S + (kvalid ? key : 0)]);
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            kf[o][sl] = global_frag<T>(K + (row0 + key) * d.ldk + h * HD + sl * ElemTraits<T>::kPerSlab, lane, kvalid);
            vf[o][sl] = global_frag<T>(V + (row0 + key) * d.ldv + h * HD + sl * ElemTraits<T>::kPerSlab, lane, kvalid);
        }
        dkacc[o][0] = dkacc[o][1] = dvacc[o][0] = dvacc[o][1] = zero_acc();
    }

#pragma unroll
    for (int o = 0; o < NOWN; ++o) { pin_frags(kf[o]); pin_frags(vf[o]); }
    const uint32_t live = valid_entities(d, b, -1);
    const int nchunks = (d.T + TQ - 1) / TQ;
    const int nqb = is_null ? 0 : (d.qpb - ((d.exclude_self && n < d.qpb) ? 1 : 0));
    const int n_it = nqb * nchunks;
    NatTile<T, TQ, NS, ATT_THREADS> qreg, doreg;
    TrTile<T, HD, NSQ, ATT_THREADS> qtreg, dotreg;
    float streg = 0.f;
    auto coords = [&](int it, int& qb, int& qc, int& i) {
        const int idx = it / nchunks;
        i = idx + ((d.exclude_self && idx >= n) ? 1 : 0);
        qb = b * d.qpb + i;
        qc = (it % nchunks) * TQ;
    };
    auto prefetch = [&](int it) {
        int qb, qc, i;
        coords(it, qb, qc, i);
        const T* qbase = Q + (long)qb * d.T * d.ldq + h * HD;
        const T* dobase = dO + (long)qb * d.T * lddo + h * HD;
        qreg.load(qbase, d.ldq, qc, d.T, 0, HD, tid);
        doreg.load(dobase, lddo, qc, d.T, 0, HD, tid);
        qtreg.load(qbase, d.ldq, 0, HD, qc, d.T, tid);
        dotreg.load(dobase, lddo, 0, HD, qc, d.T, tid);
        const float* sbase = stats + (((long)qb * d.N + n) * d.H + h) * d.T * 2;
        streg = (tid < TQ * 2 && qc + (tid >> 1) < d.T) ? sbase[(long)qc * 2 + tid] : 0.f;      // tid = 2*query + {0: lse, 1: delta}
    };
    if (n_it > 0) prefetch(0);
    for (int it = 0; it < n_it; ++it) {
        int qb, qc, i;
        coords(it, qb, qc, i);
        const int cnt = __popc(d.exclude_self ? (live & ~(1u << i)) : live);
        const float inv_cnt = cnt > 0 ? 1.f / (float)cnt : 0.f;
        __syncthreads();
        qreg.commit(qn, tid);
        doreg.commit(don, tid);
        qtreg.commit(qt, tid);
        dotreg.commit(dot, tid);
        if (tid < TQ * 2) st[(tid & 1) * TQ + (tid >> 1)] = streg;
        __syncthreads();
        if (it + 1 < n_it) prefetch(it + 1);
#pragma unroll
        for (int o = 0; o < NOWN; ++o) {
            const int kb = wave + 4 * o;
            if (kb >= NKB || kb * 32 >= d.S) continue;
            const int key = kb * 32 + (lane & 31);
#pragma unroll
            for (int qq = 0; qq < TQ / 32; ++qq) {
                if (qc + qq * 32 >= d.T) continue;
                if (CAUSAL && kb * 32 > d.causal_q0 + qc + qq * 32 + 31) continue;          // every key of the block lies above every query
                f32x16_t s = zero_acc(), dp = zero_acc();
#pragma unroll
                for (int sl = 0; sl < NS; ++sl) {
                    const Frag aq = lds_frag_o(qn + sl * (TQ * SLAB_BYTES) + qq * 32 * SLAB_BYTES, fo);
                    mma_slab<T>(s, aq, kf[o][sl]);
                    const Frag ad = lds_frag_o(don + sl * (TQ * SLAB_BYTES) + qq * 32 * SLAB_BYTES, fo);
                    mma_slab<T>(dp, ad, vf[o][sl]);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ql0 = qq * 32 + 8 * g + 4 * hh;
                    const f32x4_t lse4 = *reinterpret_cast<const f32x4_t*>(st + ql0);
                    const f32x4_t del4 = *reinterpret_cast<const f32x4_t*>(st + TQ + ql0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = 4 * g + j;
                        const int qg = qc + ql0 + j;
                        bool masked = keymask[o] || qg >= d.T;
                        if (CAUSAL) masked = masked || key > qg + d.causal_q0;
                        const float pr = masked ? 0.f : __builtin_amdgcn_exp2f(fmaf(s[r], c2, -lse4[j]));
                        dp[r] = pr * (dp[r] * inv_cnt - del4[j]) * d.scale;
                        s[r] = pr * inv_cnt;
                    }
                }
                acc_to_image<T>(imgP, s, lane);
                acc_to_image<T>(imgS, dp, lane);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    mma_image_o<T>(dvacc[o][db], imgP, dot + qq * AttnTraits<T>::kSlabsPer32 * (HD * SLAB_BYTES) + db * 32 * SLAB_BYTES, HD * SLAB_BYTES, fo);
                    mma_image_o<T>(dkacc[o][db], imgS, qt + qq * AttnTraits<T>::kSlabsPer32 * (HD * SLAB_BYTES) + db * 32 * SLAB_BYTES, HD * SLAB_BYTES, fo);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __syncthreads();                                           // the query tiles are dead: their LDS stages the output rows
    float* stg = reinterpret_cast<float*>(smem + wave * OUT_STAGE_BYTES);
#pragma unroll
    for (int o = 0; o < NOWN; ++o) {
        const int kb = wave + 4 * o;
        if (kb >= NKB) continue;
        flush_tile<T>(stg, dkacc[o], dK + (row0 + kb * 32) * lddk + h * HD, lddk, d.S - kb * 32, false, lane);
        flush_tile<T>(stg, dvacc[o], dV + (row0 + kb * 32) * lddv + h * HD, lddv, d.S - kb * 32, false, lane);
    }
}

// =============================================================================================
// bf16 kernels on the gfx950 transposing LDS read and accumulator-as-operand products.
//
// K, V (forward, dQ) and Q, dO (dK/dV) are staged ONCE, in the natural k-slab layout.  Every product that needs a staged
// tile transposed reads it with ds_read_b64_tr_b16 (a 16-lane group fetches 4 rows x 16 columns and lane i receives
// column i; four consecutive rows of a slab are 256 contiguous bytes, so the read is bank-conflict free), and every product
// whose operand is a freshly computed tile (P, dS) takes it straight from the accumulator registers: a 32x32 result has
// its column on the lane and its rows in the registers, which is the B-operand layout of a product that sums over the
// rows (guide: "an accumulator tile as the next MFMA's operand").  The k order inside such a 16-deep step is permuted
// (element j of lane half h = row 16 s + 8 (j >> 2) + 4 h + (j & 3)); tr_frag() reads the other operand in the same order.
//   forward : S^T = K Q^T,  O^T += V^T P^T          (queries on the lane: softmax statistics are register-local)
//   dQ      : S^T, dP^T = V dO^T, dQ^T += K^T dS^T
//   dK/dV   : S = Q K^T, dP = dO V^T (keys on the lane, K / V fragments live in registers), dV^T += dO^T P, dK^T += Q^T dS
// No probability / dS image ever goes through LDS and no transposed tile is staged.
// =============================================================================================
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

// Physical row of logical row i through an optional map; -1 = the row does not exist (or `valid` is false).
__device__ __forceinline__ long phys_row(const int* map, long i, bool valid) {
    if (!valid) return -1;
    return map ? (long)map[i] : i;
}

struct TrOff { int o0, o1; };
// Lane 4q+p of a 16-lane group supplies the address of row q, columns 4p..4p+3 of the group's 4 x 16 block; the groups of a
// wave cover d-halves (lane >> 4) & 1 and reduction-row halves lane >> 5.  o0 / o1: the two reads of one 16-deep step.
__device__ __forceinline__ TrOff tr_off(int lane) {
    const int h = lane >> 5, dhalf = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
    const int c = 2 * dhalf + (p >> 1);
    TrOff t;
    t.o0 = (4 * h + q) * SLAB_BYTES + (((c ^ h) & 3) << 4) + 8 * (p & 1);
    t.o1 = (8 + 4 * h + q) * SLAB_BYTES + (((c ^ (2 + h)) & 3) << 4) + 8 * (p & 1);
    return t;
}
// A-operand fragment of X^T (rows = 32 columns of slab `slab`, reduction = 16 rows of X starting at a multiple of 16):
// `p` = slab base + first row * 64.  EXEC must be all ones (call sites branch on wave-uniform values only).
__device__ __forceinline__ bf16x8_t tr_frag(const char* p, const TrOff& t) {
    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + t.o0));
    const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + t.o1));
    const s16x8_t c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, c);
}
// Registers 8s..8s+7 of an accumulator as the operand of 16-deep step s.
__device__ __forceinline__ bf16x8_t pack8(const f32x16_t& a, int s) {
    bf16x8_t r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (bf16_t)a[8 * s + j];
    return r;
}
__device__ __forceinline__ void mfma16(f32x16_t& acc, const bf16x8_t a, const bf16x8_t b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}

// A wave's TRANSPOSED 64 x 32 result (acc[db][reg]: row = column db*32 + acc_row(reg) of the output, lane & 31 = output
// row) written as `nrows` rows of 64 elements: the registers of a lane hold four consecutive output columns at a time, so
// the staging pass is eight 16-byte LDS stores, and the row pass is flush_tile's.
// g = column-offset matrix base; the wave's rows are logical rows row0 .. row0 + nrows - 1, physical rows through `map`
// (NULL = identity; -1 = the row does not exist and is skipped).
__device__ __forceinline__ void flush_tile_t(float* stg, const f32x16_t (&acc)[2], bf16_t* g, long ld, long row0, const int* map, int nrows,
                                             bool accumulate, int lane) {
    if (nrows <= 0) return;                                    // wave-uniform
    const bool vec = ((((uintptr_t)g) | (uintptr_t)(ld * 2)) & 15) == 0;
    const int row = lane >> 1, half = lane & 1;
    long prow = row0 + row;
    if (map != nullptr) prow = row < nrows ? (long)map[row0 + row] : -1;
    const bool mine = row < nrows && prow >= 0;
    bf16_t* o = g + (mine ? prow : 0) * ld + half * 32;
    u32x4_t prev[4];
    if (vec && accumulate && mine) {
#pragma unroll
        for (int v = 0; v < 4; ++v) prev[v] = *reinterpret_cast<const u32x4_t*>(o + v * 8);
    }
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            f32x4_t v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[db][4 * g4 + j];
            *reinterpret_cast<f32x4_t*>(stg + (lane & 31) * OUT_STAGE_LD + db * 32 + 8 * g4 + 4 * (lane >> 5)) = v;
        }
    __builtin_amdgcn_wave_barrier();
    if (mine) {
        const float* srow = stg + row * OUT_STAGE_LD + half * 32;
        if (vec) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                float x[8];
#pragma unroll
                for (int q4 = 0; q4 < 2; ++q4) {
                    const f32x4_t t = *reinterpret_cast<const f32x4_t*>(srow + v * 8 + 4 * q4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[4 * q4 + e] = t[e];
                }
                bf16_t w[8];
                if (accumulate) {
                    __builtin_memcpy(w, &prev[v], 16);
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] += to_f32(w[e]);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) w[e] = (bf16_t)x[e];
                u32x4_t pk;
                __builtin_memcpy(&pk, w, 16);
                *reinterpret_cast<u32x4_t*>(o + v * 8) = pk;
            }
        } else {
            float pv[32];
#pragma unroll
            for (int e = 0; e < 32; ++e) pv[e] = accumulate ? to_f32(o[e]) : 0.f;
#pragma unroll
            for (int e = 0; e < 32; ++e) o[e] = (bf16_t)(srow[e] + pv[e]);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// Key mask of a staged entity: additive bias (0 / -inf) per key, the key extent (index of the last unmasked key + 1: key
// blocks past it are pure padding and are skipped) and the index of the first masked key (key blocks before it need no
// bias at all).  slots: [0..3] extent per wave, [4..7] first masked key per wave.
__device__ __forceinline__ void publish_key_mask(float* biasf, int* slots, uint8_t masked, int S, int spad, int tid) {
    if (tid < spad) biasf[tid] = masked ? -INFINITY : 0.f;
    const unsigned long long open = __ballot(tid < S && !masked), shut = __ballot(tid < spad && masked);
    if ((tid & 63) == 0) {
        slots[tid >> 6] = open ? tid + 64 - __builtin_clzll(open) : 0;
        slots[4 + (tid >> 6)] = shut ? tid + __builtin_ctzll(shut) : spad;
    }
}
template <int NW = ATT_THREADS / 64>
__device__ __forceinline__ void read_key_mask(const int* slots, int& extent, int& first_masked) {
    int e = 0, f = 1 << 30;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        e = max(e, slots[w]);
        f = min(f, slots[4 + w]);
    }
    extent = __builtin_amdgcn_readfirstlane(e);
    first_masked = __builtin_amdgcn_readfirstlane(f);
}

// A ROWS x 64 bf16 tile requested into registers with bounds-checked buffer loads and committed to LDS in the k-slab
// layout later (same image as NatTile).  Without a row map the descriptor covers exactly the `rows` valid rows of the tile,
// so rows past them read as zeros in hardware: no per-load compare / select, and the per-thread offset is loop invariant (the
// step between a thread's rows is a scalar offset).  With a row map (compact matrices) the descriptor covers the matrix and
// every pass takes its physical row from RowIdx -- looked up one tile AHEAD, so the tile's loads do not wait for the map --
// with an out-of-range offset for rows that do not exist.  THREADS / 8 rows per pass.
// Slab 1 (columns 32..63) of a staged tile starts 64 bytes past a multiple of 128.  ds_write_b128 is served in groups of 8 contiguous
// lanes over 32 four-byte banks, and a group of commit() is one row: lanes 0-3 write its 64 bytes of slab 0, lanes 4-7 those of slab 1.
// With slab 1 at a multiple of 128 bytes from slab 0 the two halves of every group met in the same 16 banks (a 2-way conflict on every
// staging store: SQ_LDS_BANK_CONFLICT = 3.775e7 cycles per launch in forward AND dQ, 15 - 25 % of the LDS-array cycles,
// profiles/r05_attention_pmc_B128.txt); skewed by 64 bytes they cover all 32.  The reads address one slab at a time: a constant
// offset only rotates their banks.  A tile is padded to a multiple of 128 bytes again.
constexpr int SLAB_SKEW = 64, TR_TILE_PAD = 128;
__host__ __device__ constexpr int slab_stride(int rows) { return rows * SLAB_BYTES + SLAB_SKEW; }
__host__ __device__ constexpr int tr_tile_bytes(int rows) { return rows * HD * 2 + TR_TILE_PAD; }
template <int ROWS, int THREADS = ATT_THREADS>
struct RowIdx {
    static constexpr int NIT = ROWS * 8 / THREADS;
    int v[NIT];
    // map + first = the tile's first logical row; rows = valid logical rows of the tile
    __device__ __forceinline__ void load(const int* map, long first, int rows, int tid) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int lr = (tid >> 3) + it * (THREADS / 8);
            v[it] = lr < rows ? map[first + lr] : -1;
        }
    }
};
template <int ROWS, int THREADS = ATT_THREADS>
struct BufTile {
    static constexpr int NIT = ROWS * 8 / THREADS;
    u32x4_t v[NIT];
    // mat: the matrix, column offset applied; first: first logical row of the tile; ld in elements
    __device__ __forceinline__ void load(const bf16_t* mat, long ld, long first, int rows, int tid) {
        const int r = rows > 0 ? rows : 0;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(mat + first * ld), 0,
                                                                               r > 0 ? (int)((r - 1) * ld * 2 + 128) : 0, 0x00020000);
        const int voff = (tid >> 3) * (int)(ld * 2) + (tid & 7) * 16;
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            v[it] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, it * (THREADS / 8) * (int)(ld * 2), 0));
    }
    __device__ __forceinline__ void load_mapped(const bf16_t* mat, long ld, const RowIdx<ROWS, THREADS>& idx, int tid) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(mat), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int voff = idx.v[it] >= 0 ? idx.v[it] * (int)(ld * 2) + (tid & 7) * 16 : (int)0x80000000u;     // past the range: zeros
            v[it] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
        }
    }
    __device__ __forceinline__ void commit(char* lds, int tid) const {
        const int r = tid >> 3, cc = tid & 7;
        char* p = lds + (cc >> 2) * slab_stride(ROWS) + slab_off(r, cc & 3);
#pragma unroll
        for (int it = 0; it < NIT; ++it) *reinterpret_cast<u32x4_t*>(p + it * (THREADS / 8) * SLAB_BYTES) = v[it];     // (r + 32 it) >> 2 keeps r's swizzle (THREADS / 8 rows per pass: a multiple of 4)
    }
};

// Scores of one staged entity for this wave's 32 queries, transposed, then p = 2^(t - max) with t = s * scale * log2(e)
// (+ bias).  NACT = key blocks this wave works on (the entity's key extent; for causal attention up to the wave's
// diagonal block) and NFAST = how many leading ones hold no masked key: both compile-time, so the body is straight-line.
// A block without masked keys needs no bias: its maximum is taken over the raw products and scale, maximum and
// exponent are one FMA + v_exp per score.  Returns max (log2 domain) and sum.
template <int NKB, int NACT, int NFAST, bool CAUSAL>
__device__ __forceinline__ void scores_tr(f32x16_t (&sacc)[NKB], const char* ktile, const Frag* qf, const float* biasf,
                                          float c2, int qpos, int lane, const FragOff& fo, float& m_out, float& l_out,
                                          float m_floor = -INFINITY) {
    constexpr int SPAD = NKB * 32;
    const int h = lane >> 5;
#pragma unroll
    for (int kb = 0; kb < NACT; ++kb) {
        sacc[kb] = zero_acc();
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const Frag a = lds_frag_o(ktile + sl * slab_stride(SPAD) + kb * 32 * SLAB_BYTES, fo);
            mma_slab<bf16_t>(sacc[kb], a, qf[sl]);
        }
    }
    float mraw = -INFINITY, m = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NACT; ++kb) {
        if (kb < NFAST && !(CAUSAL && kb == NACT - 1)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mraw = fmaxf(mraw, sacc[kb][r]);
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4_t bias = *reinterpret_cast<const f32x4_t*>(biasf + kb * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = fmaf(sacc[kb][4 * g + j], c2, bias[j]);
                    if (CAUSAL && kb == NACT - 1 && (kb * 32 + 8 * g + 4 * h + j) > qpos) t = -INFINITY;     // earlier blocks lie below the diagonal
                    sacc[kb][4 * g + j] = t;
                    m = fmaxf(m, t);
                }
            }
        }
    }
    m = fmaxf(wave_half_max(fmaxf(m, mraw * c2)), m_floor);  // scale > 0; m_floor: the running maximum of the chunks walked so far
    const float ms = (m == -INFINITY) ? 0.f : m;
    // exponent arguments and the row sum on register PAIRS (packed f32: one issue slot per two scores for the FMA and for the
    // add; a single running sum is a serial chain the compiler may not re-associate into pairs by itself)
    typedef float pair_t __attribute__((ext_vector_type(2)));
    pair_t l2 = {0.f, 0.f};
    const pair_t c2v = {c2, c2}, msv = {ms, ms};
#pragma unroll
    for (int kb = 0; kb < NACT; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const pair_t a = {sacc[kb][r], sacc[kb][r + 1]};
            const pair_t t = (kb < NFAST && !(CAUSAL && kb == NACT - 1)) ? a * c2v - msv : a - msv;
            const pair_t pv = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
            sacc[kb][r] = pv.x;
            sacc[kb][r + 1] = pv.y;
            l2 += pv;
        }
    }
    m_out = ms;
    l_out = wave_half_sum(l2.x + l2.y);
}

// Calls body(integral_constant<NACT>, integral_constant<NFAST>) for the wave-uniform run-time counts: NACT in 1..NKB,
// NFAST = NACT - 1 when only the last active block can hold masked keys (trailing padding, the usual case), else 0.
template <int NKB, typename F>
__device__ __forceinline__ void dispatch_blocks(int nact, int nfull, F&& body) {
    const bool tail_only = nfull >= nact - 1;
#define MMSUM_CASE(N)                                                                                   \
    if constexpr (NKB >= N) if (nact == N) {                                                            \
        if (tail_only) body(std::integral_constant<int, N>{}, std::integral_constant<int, N - 1>{});    \
        else body(std::integral_constant<int, N>{}, std::integral_constant<int, 0>{});                  \
    }
    MMSUM_CASE(1) MMSUM_CASE(2) MMSUM_CASE(3) MMSUM_CASE(4) MMSUM_CASE(5) MMSUM_CASE(6) MMSUM_CASE(7)
#undef MMSUM_CASE
}
// Key blocks a wave works on: up to the entity's last unmasked key, and for causal attention up to the wave's own block.
template <bool CAUSAL>
__device__ __forceinline__ int active_blocks(int slen, int wave) {
    const int n = (slen + 31) >> 5;
    return CAUSAL ? min(n, wave + 1) : n;
}

// Forward and dQ share the staging scheme: the K and V tiles of an entity live in one of two LDS stages; the next entity's
// rows are requested into registers at the top of an iteration, written to the other stage in the middle of it (the loads
// have landed by then, and nobody reads that stage: everybody passed the barrier that ended the previous iteration), and ONE
// barrier ends the iteration.
template <int NKB> struct TrStage {
    static constexpr int SPAD = NKB * 32;
    static constexpr int TILE = tr_tile_bytes(SPAD);
    static constexpr int BYTES = 2 * TILE + SPAD * 4 + 32;
    char* base;
    __device__ __forceinline__ char* k() const { return base; }
    __device__ __forceinline__ char* v() const { return base + TILE; }
    __device__ __forceinline__ float* bias() const { return reinterpret_cast<float*>(base + 2 * TILE); }
    __device__ __forceinline__ int* slots() const { return reinterpret_cast<int*>(base + 2 * TILE + SPAD * 4); }
};

template <int NKB, bool CAUSAL, bool KVMAP>
__global__ __launch_bounds__(ATT_THREADS, NKB <= 4 ? 2 : 1) void attn_tr_fwd_kernel(mmsum_attn_desc d) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SPAD = NKB * 32;
    typedef TrStage<NKB> Stage;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const FragOff fo = frag_off<T>(lane);
    const TrOff tro = tr_off(lane);
    const int h = blockIdx.x, qb = blockIdx.y;
    const int b = qb / d.qpb;
    const int excl = d.exclude_self ? (qb % d.qpb) : -1;
    uint32_t rem = valid_entities(d, b, excl);
    const int cnt = __popc(rem);
    const float inv_cnt = cnt > 0 ? 1.f / (float)cnt : 0.f;
    const float c2 = d.scale * LOG2E_F;

    const T* Q = static_cast<const T*>(d.q);
    const T* K = static_cast<const T*>(d.k);
    const T* V = static_cast<const T*>(d.v);
    T* O = static_cast<T*>(d.out);

    const int qpos = wave * 32 + (lane & 31);
    Frag qf[2];
    {
        const long qr = phys_row(d.q_rows, (long)qb * d.T + qpos, qpos < d.T);
        const T* qrow = Q + (qr >= 0 ? qr : 0) * d.ldq + h * HD;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) qf[sl] = global_frag<T>(qrow + sl * 32, lane, qr >= 0);
    }
    pin_frags(qf);
    f32x16_t oacc[2] = {zero_acc(), zero_acc()};

    BufTile<SPAD> kreg, vreg;
    RowIdx<SPAD> kvidx;                                       // physical rows of the entity to request next (kv_rows only)
    uint8_t mreg = 1;
    auto lookup = [&](int n) { kvidx.load(d.kv_rows, ((long)b * d.N + n) * d.S, d.S, tid); };
    auto prefetch = [&](int n) {
        const long ent = (long)b * d.N + n;
        if constexpr (KVMAP) {
            kreg.load_mapped(K + h * HD, d.ldk, kvidx, tid);
            vreg.load_mapped(V + h * HD, d.ldv, kvidx, tid);
        } else {
            kreg.load(K + h * HD, d.ldk, ent * d.S, d.S, tid);
            vreg.load(V + h * HD, d.ldv, ent * d.S, d.S, tid);
        }
        mreg = (tid >= d.S) ? 1 : (d.pad ? d.pad[ent * d.S + tid] : 0);
    };
    auto commit = [&](const Stage& st) {
        kreg.commit(st.k(), tid);
        vreg.commit(st.v(), tid);
        publish_key_mask(st.bias(), st.slots(), mreg, d.S, SPAD, tid);
    };
    int cur = 0;
    if (rem) {
        if constexpr (KVMAP) lookup(__builtin_ctz(rem));
        prefetch(__builtin_ctz(rem));
        if constexpr (KVMAP) if (rem & (rem - 1)) lookup(__builtin_ctz(rem & (rem - 1)));
        commit(Stage{smem});
    }
    __syncthreads();
    while (rem) {
        rem &= rem - 1;
        const Stage st{smem + cur * Stage::BYTES}, nx{smem + (cur ^ 1) * Stage::BYTES};
        int slen, fmask;
        read_key_mask(st.slots(), slen, fmask);
        if (rem) {
            prefetch(__builtin_ctz(rem));
            if constexpr (KVMAP) if (rem & (rem - 1)) lookup(__builtin_ctz(rem & (rem - 1)));     // one entity ahead of the loads that use it
        }
        dispatch_blocks<NKB>(active_blocks<CAUSAL>(slen, wave + (d.causal_q0 >> 5)), fmask >> 5, [&](auto nact, auto nfast) {
            constexpr int NACT = decltype(nact)::value, NFAST = decltype(nfast)::value;
            f32x16_t sacc[NKB];
            float m, l;
            scores_tr<NKB, NACT, NFAST, CAUSAL>(sacc, st.k(), qf, st.bias(), c2, qpos + d.causal_q0, lane, fo, m, l);
            if (rem) commit(nx);
            const float norm = (l > 0.f) ? inv_cnt * __builtin_amdgcn_rcpf(l) : 0.f;
            f32x16_t tmp[2] = {zero_acc(), zero_acc()};
#pragma unroll
            for (int kb = 0; kb < NACT; ++kb) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8_t pb = pack8(sacc[kb], s2);
#pragma unroll
                    for (int db = 0; db < 2; ++db)
                        mfma16(tmp[db], tr_frag(st.v() + db * slab_stride(SPAD) + (kb * 32 + 16 * s2) * SLAB_BYTES, tro), pb);
                }
            }
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[db][r] = fmaf(tmp[db][r], norm, oacc[db][r]);
        });
        __syncthreads();
        cur ^= 1;
    }
    flush_tile_t(reinterpret_cast<float*>(smem + wave * OUT_STAGE_BYTES), oacc, O + h * HD, d.ldo, (long)qb * d.T + wave * 32, d.q_rows,
                 d.T - wave * 32, false, lane);
}

// ---------------------------------------------------------------------------------------------
// Round 6: the same forward at ONE wave per SIMD with 64 queries per wave (entities of <= 128 keys, not causal: the text memory's
// cross-attention and the encoder's self-attention).
//
// STATE (round 6): built, parity-green (58 attention tests through MMSUM_LIB), and SLOWER than the kernel above on the bench shape:
// 1,152 us against 927 us (B = 128, trailing pads, compact K / V; same box).  Counters per wave and entity (64 queries): 4,700 cycles
// issuing (VALU 3,260 -- 27 % more VALU instructions than two 32-query waves: the pre-scaled probabilities cost a multiply each, the
// 16-register accumulators a zero fill per key block), 2,340 waiting (one wave per SIMD: nobody covers an s_waitcnt or a barrier),
// MFMA pipe busy 1,460 of 7,960.  Two 32-query waves of the kernel above issue 4,380 cycles for the same work and overlap each
// other's waits: 5,570 elapsed.  Not dispatched (MMSUM_ATTN_W64 = 0); profiles/r06_attention_w64_fwd_pmc.txt has the counters.
//
// Why it was tried: the kernel above is issue-bound, not MFMA-bound (profiles/NOTES_r05.md section 7: per wave and entity ~1,580 issue cycles --
// 1,300 of them softmax VALU -- against 1,024 cycles of MFMA pipe), and its two waves per SIMD do not hide each other's dependency
// stalls: 6,300 cycles per entity and pair of waves, MFMA pipe 24 % busy.  Here a workgroup is TWO waves (128 queries = the query
// block), two workgroups share a CU (one wave per SIMD, 512 registers each), and a wave owns two independent 32-query groups whose
// instruction streams the scheduler overlaps inside ONE wave: the S^T MFMAs of group 1 run under the softmax arithmetic of group 0,
// the P V MFMAs of group 0 under the softmax of group 1.  Staging, masks, entity walk and the arithmetic per score are the kernel
// above's (same results bit for bit per query: a query's row never leaves its lane).
// ---------------------------------------------------------------------------------------------
constexpr int W64_THREADS = 128;

// Row maximum of one 32-query group's scores in the log2 domain (first half of scores_tr's arithmetic, scalar forms: beside MFMAs at
// one wave per SIMD the packed f32 instructions cost more than the two scalar ones they replace -- MI355X_MICROARCH.md, 'price of
// one filler').  Key blocks kb >= NFAST leave t = s * c2 + bias in place of s.  Returns the maximum (0 when every key is masked).
template <int NKB, int NACT, int NFAST>
__device__ __forceinline__ float w64_row_max(f32x16_t (&sacc)[NKB], const float* biasf, float c2, int lane) {
    const int h = lane >> 5;
    float mraw = -INFINITY, m = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NACT; ++kb) {
        if (kb < NFAST) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mraw = fmaxf(mraw, sacc[kb][r]);
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4_t bias = *reinterpret_cast<const f32x4_t*>(biasf + kb * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float t = fmaf(sacc[kb][4 * g + j], c2, bias[j]);
                    sacc[kb][4 * g + j] = t;
                    m = fmaxf(m, t);
                }
            }
        }
    }
    m = wave_half_max(fmaxf(m, mraw * c2));                   // scale > 0
    return (m == -INFINITY) ? 0.f : m;
}
// p = 2^(t - m) for registers r0 .. r0 + 3 of one key block, added into four running sums.
template <bool FAST>
__device__ __forceinline__ void w64_exp4(f32x16_t& s, int r0, float c2, float ms, float (&l)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float t = FAST ? fmaf(s[r0 + j], c2, -ms) : s[r0 + j] - ms;
        const float pv = __builtin_amdgcn_exp2f(t);
        s[r0 + j] = pv;
        l[j] += pv;
    }
}
// The B operand of one 16-key step of P V: registers 8 s2 .. 8 s2 + 7 of a key block's probabilities, normalised, as bf16.
__device__ __forceinline__ bf16x8_t w64_pack(const f32x16_t& p, int s2, float norm) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x8_t r;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {                         // one v_cvt_pk_bf16_f32 per pair
        const f32x2_t x = {p[8 * s2 + j] * norm, p[8 * s2 + j + 1] * norm};
        const bf16x2_t y = __builtin_convertvector(x, bf16x2_t);
        r[j] = y[0];
        r[j + 1] = y[1];
    }
    return r;
}

// O^T of the wave's two query groups lives in NAMED accumulation registers a[192:255] (block 2 g + db at a[192 + 16 (2 g + db)]):
// the P V MFMAs are issued from inline asm on those names, so the register allocator never sees a 64-register value that is
// live across the eight (key blocks, masked blocks) variants of the entity loop -- as compiler-managed values it copied them
// around every variant (~200 v_accvgpr moves per entity).  The compiler allocates its own AGPRs from a0 upwards and does not know these
// are taken: a build that dispatches this kernel must check that no compiler-generated instruction of it names a192 or above (done by
// hand on the device assembly for the measured build -- none did; not a Makefile check, since the kernel is not dispatched).
#define W64_AGPRS_0 "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207"
#define W64_AGPRS_1 "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223"
#define W64_AGPRS_2 "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239"
#define W64_AGPRS_3 "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
// (s_nop 1: an operand the compiler's VALU code wrote just before the statement -- the converted probabilities -- needs two wait states
// before an MFMA reads it, and the compiler does not know this statement is one)
template <int BLK>
__device__ __forceinline__ void w64_pv(const bf16x8_t a, const bf16x8_t b) {
    if constexpr (BLK == 0) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[192:207], %0, %1, a[192:207]" ::"v"(a), "v"(b));
    else if constexpr (BLK == 1) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[208:223], %0, %1, a[208:223]" ::"v"(a), "v"(b));
    else if constexpr (BLK == 2) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[224:239], %0, %1, a[224:239]" ::"v"(a), "v"(b));
    else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[240:255], %0, %1, a[240:255]" ::"v"(a), "v"(b));
}
__device__ __forceinline__ void w64_zero_acc() {
    const bf16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[192:207], %0, %0, 0" ::"v"(z) : W64_AGPRS_0);
    asm volatile("v_mfma_f32_32x32x16_bf16 a[208:223], %0, %0, 0" ::"v"(z) : W64_AGPRS_1);
    asm volatile("v_mfma_f32_32x32x16_bf16 a[224:239], %0, %0, 0" ::"v"(z) : W64_AGPRS_2);
    asm volatile("v_mfma_f32_32x32x16_bf16 a[240:255], %0, %0, 0" ::"v"(z) : W64_AGPRS_3);
}
#define W64_RD(dst, i, reg) asm volatile("v_accvgpr_read_b32 %0, " reg : "=v"(w64_t_)); dst[i] = w64_t_;
#define W64_RD16(dst, base)                                                                                                             \
    W64_RD(dst, 0, "a" #base "+0") W64_RD(dst, 1, "a" #base "+1")
// (register names cannot be computed in an asm string: the sixteen reads of a block are spelled out per block below)
template <int BLK>
__device__ __forceinline__ f32x16_t w64_read_acc() {
    f32x16_t o;
    float t;
#define W64_R(i, name) asm volatile("v_accvgpr_read_b32 %0, " name : "=v"(t)); o[i] = t;
    if constexpr (BLK == 0) {
        W64_R(0, "a192") W64_R(1, "a193") W64_R(2, "a194") W64_R(3, "a195") W64_R(4, "a196") W64_R(5, "a197") W64_R(6, "a198") W64_R(7, "a199")
        W64_R(8, "a200") W64_R(9, "a201") W64_R(10, "a202") W64_R(11, "a203") W64_R(12, "a204") W64_R(13, "a205") W64_R(14, "a206") W64_R(15, "a207")
    } else if constexpr (BLK == 1) {
        W64_R(0, "a208") W64_R(1, "a209") W64_R(2, "a210") W64_R(3, "a211") W64_R(4, "a212") W64_R(5, "a213") W64_R(6, "a214") W64_R(7, "a215")
        W64_R(8, "a216") W64_R(9, "a217") W64_R(10, "a218") W64_R(11, "a219") W64_R(12, "a220") W64_R(13, "a221") W64_R(14, "a222") W64_R(15, "a223")
    } else if constexpr (BLK == 2) {
        W64_R(0, "a224") W64_R(1, "a225") W64_R(2, "a226") W64_R(3, "a227") W64_R(4, "a228") W64_R(5, "a229") W64_R(6, "a230") W64_R(7, "a231")
        W64_R(8, "a232") W64_R(9, "a233") W64_R(10, "a234") W64_R(11, "a235") W64_R(12, "a236") W64_R(13, "a237") W64_R(14, "a238") W64_R(15, "a239")
    } else {
        W64_R(0, "a240") W64_R(1, "a241") W64_R(2, "a242") W64_R(3, "a243") W64_R(4, "a244") W64_R(5, "a245") W64_R(6, "a246") W64_R(7, "a247")
        W64_R(8, "a248") W64_R(9, "a249") W64_R(10, "a250") W64_R(11, "a251") W64_R(12, "a252") W64_R(13, "a253") W64_R(14, "a254") W64_R(15, "a255")
    }
#undef W64_R
    return o;
}
#undef W64_RD
#undef W64_RD16

template <int NKB, bool KVMAP>
__global__ __launch_bounds__(W64_THREADS, 1) void attn_w64_fwd_kernel(mmsum_attn_desc d) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SPAD = NKB * 32;
    typedef TrStage<NKB> Stage;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const FragOff fo = frag_off<T>(lane);
    const TrOff tro = tr_off(lane);
    const int h = blockIdx.x, qb = blockIdx.y;
    const int b = qb / d.qpb;
    const int excl = d.exclude_self ? (qb % d.qpb) : -1;
    uint32_t rem = valid_entities(d, b, excl);
    const int cnt = __popc(rem);
    const float inv_cnt = cnt > 0 ? 1.f / (float)cnt : 0.f;
    const float c2 = d.scale * LOG2E_F;

    const T* Q = static_cast<const T*>(d.q);
    const T* K = static_cast<const T*>(d.k);
    const T* V = static_cast<const T*>(d.v);
    T* O = static_cast<T*>(d.out);

    Frag qf[2][2];                                            // [query group][slab]
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int qpos = wave * 64 + g * 32 + (lane & 31);
        const long qr = phys_row(d.q_rows, (long)qb * d.T + qpos, qpos < d.T);
        const T* qrow = Q + (qr >= 0 ? qr : 0) * d.ldq + h * HD;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) qf[g][sl] = global_frag<T>(qrow + sl * 32, lane, qr >= 0);
    }
    pin_frags(qf[0]);
    pin_frags(qf[1]);
    w64_zero_acc();

    // K and V of the next entity are requested at the top of an iteration and committed to the other stage in its last phase: a whole
    // iteration (> 4,000 cycles) for the rows to arrive -- with one wave per SIMD nobody else covers a wait
    BufTile<SPAD, W64_THREADS> kreg, vreg;
    RowIdx<SPAD, W64_THREADS> kvidx;
    uint8_t mreg = 1;
    auto lookup = [&](int n) { kvidx.load(d.kv_rows, ((long)b * d.N + n) * d.S, d.S, tid); };
    auto request = [&](int n) {
        if constexpr (KVMAP) {
            kreg.load_mapped(K + h * HD, d.ldk, kvidx, tid);
            vreg.load_mapped(V + h * HD, d.ldv, kvidx, tid);
        } else {
            kreg.load(K + h * HD, d.ldk, ((long)b * d.N + n) * d.S, d.S, tid);
            vreg.load(V + h * HD, d.ldv, ((long)b * d.N + n) * d.S, d.S, tid);
        }
        mreg = (tid >= d.S) ? 1 : (d.pad ? d.pad[((long)b * d.N + n) * d.S + tid] : 0);
    };
    int cur = 0;
    if (rem) {
        const int n0 = __builtin_ctz(rem);
        if constexpr (KVMAP) lookup(n0);
        request(n0);
        if constexpr (KVMAP) if (rem & (rem - 1)) lookup(__builtin_ctz(rem & (rem - 1)));
        kreg.commit(Stage{smem}.k(), tid);
        vreg.commit(Stage{smem}.v(), tid);
        publish_key_mask(Stage{smem}.bias(), Stage{smem}.slots(), mreg, d.S, SPAD, tid);
    }
    __syncthreads();
    // a wave whose 64 queries lie past T (T <= 64: wave 1) still stages and meets the barriers, and skips the arithmetic
    const bool wave_live = wave * 64 < d.T;
    while (rem) {
        rem &= rem - 1;
        const Stage st{smem + cur * Stage::BYTES}, nx{smem + (cur ^ 1) * Stage::BYTES};
        const int nn = rem ? __builtin_ctz(rem) : 0;          // the entity staged during this iteration (none left: entity 0 again, nobody reads that stage)
        int slen, fmask;
        read_key_mask<W64_THREADS / 64>(st.slots(), slen, fmask);
        request(nn);
        if constexpr (KVMAP) lookup((rem & (rem - 1)) ? __builtin_ctz(rem & (rem - 1)) : nn);      // one entity ahead of the loads that use it
        bool done = false;
        if (wave_live) {
            dispatch_blocks<NKB>((slen + 31) >> 5, fmask >> 5, [&](auto nact, auto nfast) {
                constexpr int NACT = decltype(nact)::value, NFAST = decltype(nfast)::value;
                constexpr int NGAP = 4 * NACT;                 // MFMAs of one product of one group (S^T or P V): the gaps the other group's VALU work is dealt into
                constexpr int J0 = (NACT + 1) / 2;             // gaps that take the row maximum (two key blocks each); the exponentials follow
                f32x16_t s0[NKB], s1[NKB];
                const char* kt = st.k();
                const char* vt = st.v();
                // operand fragment f (two 16-deep steps) of the K tile: key block f >> 1, slab f & 1
                auto kfrag = [&](int f) { return lds_frag_o(kt + (f & 1) * slab_stride(SPAD) + (f >> 1) * 32 * SLAB_BYTES, fo); };
                // A operand of P V step `stp` (16 keys) for output columns db * 32 ..
                auto vfrag = [&](int stp, int db) { return tr_frag(vt + db * slab_stride(SPAD) + ((stp >> 1) * 32 + 16 * (stp & 1)) * SLAB_BYTES, tro); };
                // the VALU work of one group's softmax, dealt into gaps: gap j < J0 folds key blocks 2 j, 2 j + 1 into the running maximum
                // (the last of them finishes it across the wave's halves), gap j >= J0 takes its share of the 4 NACT exp units
                auto softmax_gap = [&](int j, f32x16_t (&sg)[NKB], float& mraw, float& m, float& ms, float (&l)[4]) {
                    if (j < J0) {
                        const int h2 = lane >> 5;
#pragma unroll
                        for (int kb = 2 * j; kb < 2 * j + 2 && kb < NACT; ++kb) {
                            if (kb < NFAST) {
#pragma unroll
                                for (int r = 0; r < 16; ++r) mraw = fmaxf(mraw, sg[kb][r]);
                            } else {
#pragma unroll
                                for (int g4 = 0; g4 < 4; ++g4) {
                                    const f32x4_t bias = *reinterpret_cast<const f32x4_t*>(st.bias() + kb * 32 + 8 * g4 + 4 * h2);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        const float t = fmaf(sg[kb][4 * g4 + e], c2, bias[e]);
                                        sg[kb][4 * g4 + e] = t;
                                        m = fmaxf(m, t);
                                    }
                                }
                            }
                        }
                        if (j == J0 - 1) {
                            m = wave_half_max(fmaxf(m, mraw * c2));        // scale > 0
                            ms = (m == -INFINITY) ? 0.f : m;
                        }
                    } else {
                        const int u0 = (j - J0) * (4 * NACT) / (NGAP - J0), u1 = (j - J0 + 1) * (4 * NACT) / (NGAP - J0);
#pragma unroll
                        for (int u = u0; u < u1; ++u) {
                            if ((u >> 2) < NFAST) w64_exp4<true>(sg[u >> 2], 4 * (u & 3), c2, ms, l);
                            else w64_exp4<false>(sg[u >> 2], 4 * (u & 3), c2, ms, l);
                        }
                    }
                };
                // ---- phase 1: S^T of group 0 (nothing to put beside it: the compiler requests all fragments up front)
#pragma unroll
                for (int kb = 0; kb < NACT; ++kb) {
                    s0[kb] = zero_acc();
#pragma unroll
                    for (int sl = 0; sl < 2; ++sl) mma_slab<bf16_t>(s0[kb], kfrag(2 * kb + sl), qf[0][sl]);
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- phase 2: S^T of group 1 on the MFMA pipe, gap by gap, under the softmax arithmetic of group 0
                float mr0 = -INFINITY, mm0 = -INFINITY, m0 = 0.f, l0[4] = {0.f, 0.f, 0.f, 0.f};
                {
                    Frag fa = kfrag(0), fb = kfrag(1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < NGAP; ++j) {
                        const int f = j >> 1, kb = j >> 2, sl = (j >> 1) & 1;
                        const Frag& cur = (f & 1) ? fb : fa;
                        if ((j & 3) == 0) s1[kb] = zero_acc();
                        s1[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, cur.c[j & 1]), __builtin_bit_cast(bf16x8_t, qf[1][sl].c[j & 1]), s1[kb], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        softmax_gap(j, s0, mr0, mm0, m0, l0);
                        if ((j & 1) == 1 && f + 2 < 2 * NACT) {           // the fragment two ahead replaces the one just used up
                            if (f & 1) fb = kfrag(f + 2); else fa = kfrag(f + 2);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                const float ls0 = wave_half_sum((l0[0] + l0[1]) + (l0[2] + l0[3]));
                const float norm0 = (ls0 > 0.f) ? inv_cnt * __builtin_amdgcn_rcpf(ls0) : 0.f;
                // ---- phase 3: P V of group 0 (asm MFMAs on the named accumulators), gap by gap, under the softmax of group 1; K of the next
                // (K / V of the next entity are committed after phase 4)
                float mr1 = -INFINITY, mm1 = -INFINITY, m1 = 0.f, l1[4] = {0.f, 0.f, 0.f, 0.f};
                {
                    bf16x8_t pb = w64_pack(s0[0], 0, norm0);
                    bf16x8_t va = vfrag(0, 0), vb = vfrag(0, 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < NGAP; ++j) {
                        const int stp = j >> 1;
                        if (j & 1) w64_pv<1>(vb, pb); else w64_pv<0>(va, pb);
                        __builtin_amdgcn_sched_barrier(0);
                        softmax_gap(j, s1, mr1, mm1, m1, l1);
                        if ((j & 1) == 0) {
                            if (stp + 1 < 2 * NACT) va = vfrag(stp + 1, 0);
                        } else if (stp + 1 < 2 * NACT) {
                            vb = vfrag(stp + 1, 1);
                            pb = w64_pack(s0[(stp + 1) >> 1], (stp + 1) & 1, norm0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                // ---- phase 4: P V of group 1 (its operands converted one step ahead)
                const float ls1 = wave_half_sum((l1[0] + l1[1]) + (l1[2] + l1[3]));
                const float norm1 = (ls1 > 0.f) ? inv_cnt * __builtin_amdgcn_rcpf(ls1) : 0.f;
                {
                    bf16x8_t pb = w64_pack(s1[0], 0, norm1);
                    bf16x8_t va = vfrag(0, 0), vb = vfrag(0, 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < NGAP; ++j) {
                        const int stp = j >> 1;
                        if (j & 1) w64_pv<3>(vb, pb); else w64_pv<2>(va, pb);
                        __builtin_amdgcn_sched_barrier(0);
                        if ((j & 1) == 0) {
                            if (stp + 1 < 2 * NACT) va = vfrag(stp + 1, 0);
                        } else if (stp + 1 < 2 * NACT) {
                            vb = vfrag(stp + 1, 1);
                            pb = w64_pack(s1[(stp + 1) >> 1], (stp + 1) & 1, norm1);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                kreg.commit(nx.k(), tid);
                vreg.commit(nx.v(), tid);
                done = true;
            });
        }
        if (!done) {                                          // (an entity without an unmasked key, or an idle wave: the staging goes on)
            kreg.commit(nx.k(), tid);
            vreg.commit(nx.v(), tid);
        }
        publish_key_mask(nx.bias(), nx.slots(), mreg, d.S, SPAD, tid);
        __syncthreads();
        cur ^= 1;
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");         // the last MFMA's result -> v_accvgpr_read (the loop's barrier lies between already)
    {
        float* stg = reinterpret_cast<float*>(smem + wave * OUT_STAGE_BYTES);
        f32x16_t o[2];
        o[0] = w64_read_acc<0>();
        o[1] = w64_read_acc<1>();
        flush_tile_t(stg, o, O + h * HD, d.ldo, (long)qb * d.T + wave * 64, d.q_rows, d.T - wave * 64, false, lane);
        o[0] = w64_read_acc<2>();
        o[1] = w64_read_acc<3>();
        flush_tile_t(stg, o, O + h * HD, d.ldo, (long)qb * d.T + wave * 64 + 32, d.q_rows, d.T - wave * 64 - 32, false, lane);
    }
}

// Entities of more than 128 keys (the image memory: 196 keys per image) walked in CHUNKS of up to 128 keys with a running softmax
// per entity, so that the two LDS stages are four key blocks each (as for the text memory) and TWO workgroups share a CU: the
// seven-block stages of the kernel above fill 117 KB and leave its four waves alone on the CU, one per SIMD, with nobody to issue
// MFMAs while a wave works through its exponentials (measured, B = 128: 800 us per layer for half the executed FLOPs of the text
// memory's 540 us).  Per (entity, chunk): scores against the running maximum m (scores_tr's m_floor), p = 2^(t - m); the entity's
// un-normalised output E is rescaled by 2^(m_old - m) when the maximum moved and P V accumulates INTO it (the MFMA's C operand), so
// the register budget is the text kernel's; after the entity's last chunk O += E / (count * l).  The iteration unit of the staging
// scheme (request at the top, commit in the middle, one barrier) is the (entity, chunk) pair.
struct ChunkUnit { uint32_t r; int n, c; bool ok; };
__device__ __forceinline__ ChunkUnit chunk_first(uint32_t rem) {
    ChunkUnit u;
    u.r = rem; u.ok = rem != 0; u.n = u.ok ? __builtin_ctz(rem) : 0; u.c = 0;
    return u;
}
__device__ __forceinline__ ChunkUnit chunk_next(ChunkUnit u, int nch) {
    if (!u.ok) return u;
    if (u.c + 1 < nch) { ++u.c; return u; }
    u.r &= u.r - 1;
    u.ok = u.r != 0;
    u.n = u.ok ? __builtin_ctz(u.r) : 0;
    u.c = 0;
    return u;
}

template <bool KVMAP>
__global__ __launch_bounds__(ATT_THREADS, 2) void attn_tr_fwd_chunk_kernel(mmsum_attn_desc d) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NKB = 4, SPAD = NKB * 32;
    typedef TrStage<NKB> Stage;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const FragOff fo = frag_off<T>(lane);
    const TrOff tro = tr_off(lane);
    const int h = blockIdx.x, qb = blockIdx.y;
    const int b = qb / d.qpb;
    const int excl = d.exclude_self ? (qb % d.qpb) : -1;
    const uint32_t rem0 = valid_entities(d, b, excl);
    const int cnt = __popc(rem0);
    const float inv_cnt = cnt > 0 ? 1.f / (float)cnt : 0.f;
    const float c2 = d.scale * LOG2E_F;
    const int nch = (d.S + SPAD - 1) / SPAD;

    const T* Q = static_cast<const T*>(d.q);
    const T* K = static_cast<const T*>(d.k);
    const T* V = static_cast<const T*>(d.v);
    T* O = static_cast<T*>(d.out);

    const int qpos = wave * 32 + (lane & 31);
    Frag qf[2];
    {
        const long qr = phys_row(d.q_rows, (long)qb * d.T + qpos, qpos < d.T);
        const T* qrow = Q + (qr >= 0 ? qr : 0) * d.ldq + h * HD;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) qf[sl] = global_frag<T>(qrow + sl * 32, lane, qr >= 0);
    }
    pin_frags(qf);
    f32x16_t oacc[2] = {zero_acc(), zero_acc()}, eacc[2] = {zero_acc(), zero_acc()};
    float m_run = 0.f, l_run = 0.f;                           // this lane's query: running maximum (log2 domain) and sum of the entity

    BufTile<SPAD> kreg, vreg;
    RowIdx<SPAD> kvidx;
    uint8_t mreg = 1;
    auto lookup = [&](const ChunkUnit& u) {
        const int k0 = u.c * SPAD;
        kvidx.load(d.kv_rows, ((long)b * d.N + u.n) * d.S + k0, min(SPAD, d.S - k0), tid);
    };
    auto prefetch = [&](const ChunkUnit& u) {
        const long ent = (long)b * d.N + u.n;
        const int k0 = u.c * SPAD, len = min(SPAD, d.S - k0);
        if constexpr (KVMAP) {
            kreg.load_mapped(K + h * HD, d.ldk, kvidx, tid);
            vreg.load_mapped(V + h * HD, d.ldv, kvidx, tid);
        } else {
            kreg.load(K + h * HD, d.ldk, ent * d.S + k0, len, tid);
            vreg.load(V + h * HD, d.ldv, ent * d.S + k0, len, tid);
        }
        mreg = (tid >= len) ? 1 : (d.pad ? d.pad[ent * d.S + k0 + tid] : 0);
    };
    auto commit = [&](const Stage& st, const ChunkUnit& u) {
        kreg.commit(st.k(), tid);
        vreg.commit(st.v(), tid);
        publish_key_mask(st.bias(), st.slots(), mreg, min(SPAD, d.S - u.c * SPAD), SPAD, tid);
    };
    ChunkUnit cu = chunk_first(rem0);
    int cur = 0;
    if (cu.ok) {
        if constexpr (KVMAP) lookup(cu);
        prefetch(cu);
        const ChunkUnit n1 = chunk_next(cu, nch);
        if constexpr (KVMAP) if (n1.ok) lookup(n1);
        commit(Stage{smem}, cu);
    }
    __syncthreads();
    while (cu.ok) {
        const ChunkUnit nx1 = chunk_next(cu, nch), nx2 = chunk_next(nx1, nch);
        const Stage st{smem + cur * Stage::BYTES}, nx{smem + (cur ^ 1) * Stage::BYTES};
        int slen, fmask;
        read_key_mask(st.slots(), slen, fmask);
        if (nx1.ok) {
            prefetch(nx1);
            if constexpr (KVMAP) if (nx2.ok) lookup(nx2);     // one unit ahead of the loads that use it
        }
        if (cu.c == 0) {                                      // a new entity (wave-uniform)
            l_run = 0.f;
            m_run = 0.f;
            eacc[0] = zero_acc();
            eacc[1] = zero_acc();
        }
        bool committed = false;
        dispatch_blocks<NKB>(active_blocks<false>(slen, wave), fmask >> 5, [&](auto nact, auto nfast) {
            constexpr int NACT = decltype(nact)::value, NFAST = decltype(nfast)::value;
            f32x16_t sacc[NKB];
            float m, l;
            scores_tr<NKB, NACT, NFAST, false>(sacc, st.k(), qf, st.bias(), c2, qpos, lane, fo, m, l, l_run > 0.f ? m_run : -INFINITY);
            if (nx1.ok) commit(nx, nx1);
            committed = true;
            const float alpha = l_run > 0.f ? __builtin_amdgcn_exp2f(m_run - m) : 0.f;      // m >= m_run: the maximum only grows
            l_run = fmaf(l_run, alpha, l);
            m_run = m;
            if (cu.c > 0) {
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) eacc[db][r] *= alpha;
            }
#pragma unroll
            for (int kb = 0; kb < NACT; ++kb) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8_t pb = pack8(sacc[kb], s2);
#pragma unroll
                    for (int db = 0; db < 2; ++db)
                        mfma16(eacc[db], tr_frag(st.v() + db * slab_stride(SPAD) + (kb * 32 + 16 * s2) * SLAB_BYTES, tro), pb);
                }
            }
        });
        if (!committed && nx1.ok) commit(nx, nx1);            // a chunk without an unmasked key: nothing to add, the staging goes on
        if (cu.c == nch - 1) {                                 // the entity is complete: O += E / (count * l)
            const float norm = (l_run > 0.f) ? inv_cnt * __builtin_amdgcn_rcpf(l_run) : 0.f;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[db][r] = fmaf(eacc[db][r], norm, oacc[db][r]);
        }
        __syncthreads();
        cur ^= 1;
        cu = nx1;
    }
    flush_tile_t(reinterpret_cast<float*>(smem + wave * OUT_STAGE_BYTES), oacc, O + h * HD, d.ldo, (long)qb * d.T + wave * 32, d.q_rows,
                 d.T - wave * 32, false, lane);
}

// dQ (+ per-entity statistics for the dK/dV kernel: log-sum-exp in the log2 domain and delta' = scale * sum_k P dP, i.e.
// delta * count * scale).  Per entity: scores and probabilities for all key blocks stay in registers; dP^T = V dO^T is formed
// twice (once for delta, once for dS) instead of being kept, which is what lets two workgroups share a CU.
template <int NKB, bool CAUSAL, bool KVMAP>
__global__ __launch_bounds__(ATT_THREADS, NKB <= 4 ? 2 : 1) void attn_tr_bwd_dq_kernel(mmsum_attn_desc d, const bf16_t* __restrict__ dO, long lddo,
                                                                                      bf16_t* __restrict__ dQ, long lddq, int accumulate_dq,
                                                                                      float* __restrict__ stats) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SPAD = NKB * 32;
    typedef TrStage<NKB> Stage;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const FragOff fo = frag_off<T>(lane);
    const TrOff tro = tr_off(lane);
    const int h = blockIdx.x, qb = blockIdx.y;
    const int b = qb / d.qpb;
    const int excl = d.exclude_self ? (qb % d.qpb) : -1;
    uint32_t rem = valid_entities(d, b, excl);
    const int cnt = __popc(rem);
    const float inv_cnt = cnt > 0 ? 1.f / (float)cnt : 0.f;
    const float c2 = d.scale * LOG2E_F;

    const T* Q = static_cast<const T*>(d.q);
    const T* K = static_cast<const T*>(d.k);
    const T* V = static_cast<const T*>(d.v);

    const int qpos = wave * 32 + (lane & 31);
    const bool qvalid = qpos < d.T;
    Frag qf[2], dof[2];
    {
        const long qr = phys_row(d.q_rows, (long)qb * d.T + qpos, qvalid);
        const T* qrow = Q + (qr >= 0 ? qr : 0) * d.ldq + h * HD;
        const T* drow = dO + (qr >= 0 ? qr : 0) * lddo + h * HD;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            qf[sl] = global_frag<T>(qrow + sl * 32, lane, qr >= 0);
            dof[sl] = global_frag<T>(drow + sl * 32, lane, qr >= 0);
        }
    }
    pin_frags(qf);
    pin_frags(dof);
    f32x16_t dqacc[2] = {zero_acc(), zero_acc()};

    BufTile<SPAD> kreg, vreg;
    RowIdx<SPAD> kvidx;
    uint8_t mreg = 1;
    int cur_n = 0, next_n = 0;
    auto lookup = [&](int n) { kvidx.load(d.kv_rows, ((long)b * d.N + n) * d.S, d.S, tid); };
    auto prefetch = [&](int n) {
        const long ent = (long)b * d.N + n;
        if constexpr (KVMAP) {
            kreg.load_mapped(K + h * HD, d.ldk, kvidx, tid);
            vreg.load_mapped(V + h * HD, d.ldv, kvidx, tid);
        } else {
            kreg.load(K + h * HD, d.ldk, ent * d.S, d.S, tid);
            vreg.load(V + h * HD, d.ldv, ent * d.S, d.S, tid);
        }
        mreg = (tid >= d.S) ? 1 : (d.pad ? d.pad[ent * d.S + tid] : 0);
        next_n = n;
    };
    auto commit = [&](const Stage& st) {
        kreg.commit(st.k(), tid);
        vreg.commit(st.v(), tid);
        publish_key_mask(st.bias(), st.slots(), mreg, d.S, SPAD, tid);
    };
    int cur = 0;
    if (rem) {
        if constexpr (KVMAP) lookup(__builtin_ctz(rem));
        prefetch(__builtin_ctz(rem));
        if constexpr (KVMAP) if (rem & (rem - 1)) lookup(__builtin_ctz(rem & (rem - 1)));
        commit(Stage{smem});
    }
    __syncthreads();
    while (rem) {
        rem &= rem - 1;
        const Stage st{smem + cur * Stage::BYTES}, nx{smem + (cur ^ 1) * Stage::BYTES};
        int slen, fmask;
        read_key_mask(st.slots(), slen, fmask);
        cur_n = next_n;
        if (rem) {
            prefetch(__builtin_ctz(rem));
            if constexpr (KVMAP) if (rem & (rem - 1)) lookup(__builtin_ctz(rem & (rem - 1)));
        }
        dispatch_blocks<NKB>(active_blocks<CAUSAL>(slen, wave + (d.causal_q0 >> 5)), fmask >> 5, [&](auto nact, auto nfast) {
            constexpr int NACT = decltype(nact)::value, NFAST = decltype(nfast)::value;
            f32x16_t p[NKB];
            float m, l;
            scores_tr<NKB, NACT, NFAST, CAUSAL>(p, st.k(), qf, st.bias(), c2, qpos + d.causal_q0, lane, fo, m, l);
            if (rem) commit(nx);
            const float invl = (l > 0.f) ? __builtin_amdgcn_rcpf(l) : 0.f;
            typedef float pair_t __attribute__((ext_vector_type(2)));
            pair_t raw2 = {0.f, 0.f};                             // sum_k p * dP, un-normalised, on register pairs (packed FMAs)
#pragma unroll
            for (int kb = 0; kb < NACT; ++kb) {
                f32x16_t dpk = zero_acc();
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    const Frag a = lds_frag_o(st.v() + sl * slab_stride(SPAD) + kb * 32 * SLAB_BYTES, fo);
                    mma_slab<T>(dpk, a, dof[sl]);
                }
#pragma unroll
                for (int r = 0; r < 16; r += 2) raw2 += pair_t{p[kb][r], p[kb][r + 1]} * pair_t{dpk[r], dpk[r + 1]};
            }
            const float dprime = wave_half_sum(raw2.x + raw2.y) * invl * d.scale;          // delta * count * scale
            if (lane < 32 && qvalid) {
                float* sp = stats + ((((long)qb * d.N + cur_n) * d.H + h) * d.T + qpos) * 2;
                sp[0] = m + __log2f(l);
                sp[1] = dprime;
            }
            // dS^T = P (dP / count - delta) scale = p * (dP * ca - cb)
            const float ca = invl * inv_cnt * d.scale, cb = invl * inv_cnt * dprime;
#pragma unroll
            for (int kb = 0; kb < NACT; ++kb) {
                f32x16_t dpk = zero_acc();
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    const Frag a = lds_frag_o(st.v() + sl * slab_stride(SPAD) + kb * 32 * SLAB_BYTES, fo);
                    mma_slab<T>(dpk, a, dof[sl]);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) dpk[r] = p[kb][r] * fmaf(dpk[r], ca, -cb);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8_t sb = pack8(dpk, s2);
#pragma unroll
                    for (int db = 0; db < 2; ++db)
                        mfma16(dqacc[db], tr_frag(st.k() + db * slab_stride(SPAD) + (kb * 32 + 16 * s2) * SLAB_BYTES, tro), sb);
                }
            }
        });
        __syncthreads();
        cur ^= 1;
    }
    flush_tile_t(reinterpret_cast<float*>(smem + wave * OUT_STAGE_BYTES), dqacc, dQ + h * HD, lddq, (long)qb * d.T + wave * 32, d.q_rows,
                 d.T - wave * 32, accumulate_dq != 0, lane);
}

// ---------------------------------------------------------------------------------------------
// One entity per business, attended by all qpb query blocks of the business (table and image memory: N == 1, no
// leave-one-out): a workgroup stages the entity's K and V ONCE and walks its share of the query blocks, the next block's
// Q (and dO) rows requested while the current one computes.  Nothing in the walk needs a workgroup barrier: the tiles are
// read-only and every wave stages its own output rows.  grid = (H, businesses, splits of the qpb query blocks).
// ---------------------------------------------------------------------------------------------
template <int NKB>
__global__ __launch_bounds__(ATT_THREADS, NKB <= 4 ? 2 : 1) void attn_tr_fwd_shared_kernel(mmsum_attn_desc d) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SPAD = NKB * 32;
    typedef TrStage<NKB> Stage;
    const Stage st{smem};

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const FragOff fo = frag_off<T>(lane);
    const TrOff tro = tr_off(lane);
    const int h = blockIdx.x, b = blockIdx.y;
    const int per = (d.qpb + gridDim.z - 1) / gridDim.z;
    const int i0 = blockIdx.z * per, i1 = min(d.qpb, i0 + per);
    const bool live = (valid_entities(d, b, -1) & 1u) != 0;
    const float c2 = d.scale * LOG2E_F;
    const T* Q = static_cast<const T*>(d.q);
    T* O = static_cast<T*>(d.out);
    const int qpos = wave * 32 + (lane & 31);
    const bool qvalid = qpos < d.T;

    int slen = 0, fmask = 0;
    if (live) {
        BufTile<SPAD> kreg, vreg;
        const long row0 = (long)b * d.S;
        if (d.kv_rows) {
            RowIdx<SPAD> kvidx;
            kvidx.load(d.kv_rows, row0, d.S, tid);
            kreg.load_mapped(static_cast<const T*>(d.k) + h * HD, d.ldk, kvidx, tid);
            vreg.load_mapped(static_cast<const T*>(d.v) + h * HD, d.ldv, kvidx, tid);
        } else {
            kreg.load(static_cast<const T*>(d.k) + h * HD, d.ldk, row0, d.S, tid);
            vreg.load(static_cast<const T*>(d.v) + h * HD, d.ldv, row0, d.S, tid);
        }
        const uint8_t mreg = (tid >= d.S) ? 1 : (d.pad ? d.pad[row0 + tid] : 0);
        kreg.commit(st.k(), tid);
        vreg.commit(st.v(), tid);
        publish_key_mask(st.bias(), st.slots(), mreg, d.S, SPAD, tid);
    }
    __syncthreads();
    if (live) read_key_mask(st.slots(), slen, fmask);
    float* stg = reinterpret_cast<float*>(smem + Stage::BYTES + wave * OUT_STAGE_BYTES);

    auto load_q = [&](Frag (&f)[2], int i) {
        const long qr = phys_row(d.q_rows, ((long)b * d.qpb + i) * d.T + qpos, qvalid && i < i1);
        const T* qrow = Q + (qr >= 0 ? qr : 0) * d.ldq + h * HD;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) f[sl] = global_frag<T>(qrow + sl * 32, lane, qr >= 0);
    };
    Frag qf[2], qn[2];
    load_q(qf, i0);
    for (int i = i0; i < i1; ++i) {
        load_q(qn, i + 1);
        f32x16_t oacc[2] = {zero_acc(), zero_acc()};
        if (live) {
            dispatch_blocks<NKB>(active_blocks<false>(slen, wave), fmask >> 5, [&](auto nact, auto nfast) {
                constexpr int NACT = decltype(nact)::value, NFAST = decltype(nfast)::value;
                f32x16_t sacc[NKB];
                float m, l;
                scores_tr<NKB, NACT, NFAST, false>(sacc, st.k(), qf, st.bias(), c2, qpos, lane, fo, m, l);
                const float norm = (l > 0.f) ? __builtin_amdgcn_rcpf(l) : 0.f;
#pragma unroll
                for (int kb = 0; kb < NACT; ++kb) {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const bf16x8_t pb = pack8(sacc[kb], s2);
#pragma unroll
                        for (int db = 0; db < 2; ++db)
                            mfma16(oacc[db], tr_frag(st.v() + db * slab_stride(SPAD) + (kb * 32 + 16 * s2) * SLAB_BYTES, tro), pb);
                    }
                }
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[db][r] *= norm;
            });
        }
        flush_tile_t(stg, oacc, O + h * HD, d.ldo, ((long)b * d.qpb + i) * d.T + wave * 32, d.q_rows, d.T - wave * 32, false, lane);
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) qf[sl] = qn[sl];
    }
}

template <int NKB>
__global__ __launch_bounds__(ATT_THREADS, NKB <= 4 ? 2 : 1) void attn_tr_bwd_dq_shared_kernel(mmsum_attn_desc d, const bf16_t* __restrict__ dO, long lddo,
                                                                                             bf16_t* __restrict__ dQ, long lddq, int accumulate_dq,
                                                                                             float* __restrict__ stats) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SPAD = NKB * 32;
    typedef TrStage<NKB> Stage;
    const Stage st{smem};

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const FragOff fo = frag_off<T>(lane);
    const TrOff tro = tr_off(lane);
    const int h = blockIdx.x, b = blockIdx.y;
    const int per = (d.qpb + gridDim.z - 1) / gridDim.z;
    const int i0 = blockIdx.z * per, i1 = min(d.qpb, i0 + per);
    const bool live = (valid_entities(d, b, -1) & 1u) != 0;
    const float c2 = d.scale * LOG2E_F;
    const T* Q = static_cast<const T*>(d.q);
    const int qpos = wave * 32 + (lane & 31);
    const bool qvalid = qpos < d.T;

    int slen = 0, fmask = 0;
    if (live) {
        BufTile<SPAD> kreg, vreg;
        const long row0 = (long)b * d.S;
        if (d.kv_rows) {
            RowIdx<SPAD> kvidx;
            kvidx.load(d.kv_rows, row0, d.S, tid);
            kreg.load_mapped(static_cast<const T*>(d.k) + h * HD, d.ldk, kvidx, tid);
            vreg.load_mapped(static_cast<const T*>(d.v) + h * HD, d.ldv, kvidx, tid);
        } else {
            kreg.load(static_cast<const T*>(d.k) + h * HD, d.ldk, row0, d.S, tid);
            vreg.load(static_cast<const T*>(d.v) + h * HD, d.ldv, row0, d.S, tid);
        }
        const uint8_t mreg = (tid >= d.S) ? 1 : (d.pad ? d.pad[row0 + tid] : 0);
        kreg.commit(st.k(), tid);
        vreg.commit(st.v(), tid);
        publish_key_mask(st.bias(), st.slots(), mreg, d.S, SPAD, tid);
    }
    __syncthreads();
    if (live) read_key_mask(st.slots(), slen, fmask);
    float* stg = reinterpret_cast<float*>(smem + Stage::BYTES + wave * OUT_STAGE_BYTES);

    auto load_q = [&](Frag (&fq)[2], Frag (&fd)[2], int i) {
        const long qr = phys_row(d.q_rows, ((long)b * d.qpb + i) * d.T + qpos, qvalid && i < i1);
        const T* qrow = Q + (qr >= 0 ? qr : 0) * d.ldq + h * HD;
        const T* drow = dO + (qr >= 0 ? qr : 0) * lddo + h * HD;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            fq[sl] = global_frag<T>(qrow + sl * 32, lane, qr >= 0);
            fd[sl] = global_frag<T>(drow + sl * 32, lane, qr >= 0);
        }
    };
    Frag qf[2], dof[2], qn[2], don[2];
    load_q(qf, dof, i0);
    for (int i = i0; i < i1; ++i) {
        load_q(qn, don, i + 1);
        const int qb = b * d.qpb + i;
        f32x16_t dqacc[2] = {zero_acc(), zero_acc()};
        if (live) {
            dispatch_blocks<NKB>(active_blocks<false>(slen, wave), fmask >> 5, [&](auto nact, auto nfast) {
                constexpr int NACT = decltype(nact)::value, NFAST = decltype(nfast)::value;
                f32x16_t p[NKB];
                float m, l;
                scores_tr<NKB, NACT, NFAST, false>(p, st.k(), qf, st.bias(), c2, qpos, lane, fo, m, l);
                const float invl = (l > 0.f) ? __builtin_amdgcn_rcpf(l) : 0.f;
                typedef float pair_t __attribute__((ext_vector_type(2)));
                pair_t raw2 = {0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < NACT; ++kb) {
                    f32x16_t dpk = zero_acc();
#pragma unroll
                    for (int sl = 0; sl < 2; ++sl) {
                        const Frag a = lds_frag_o(st.v() + sl * slab_stride(SPAD) + kb * 32 * SLAB_BYTES, fo);
                        mma_slab<T>(dpk, a, dof[sl]);
                    }
#pragma unroll
                    for (int r = 0; r < 16; r += 2) raw2 += pair_t{p[kb][r], p[kb][r + 1]} * pair_t{dpk[r], dpk[r + 1]};
                }
                const float dprime = wave_half_sum(raw2.x + raw2.y) * invl * d.scale;          // delta * count * scale, count = 1
                if (lane < 32 && qvalid) {
                    float* sp = stats + (((long)qb * d.H + h) * d.T + qpos) * 2;   // N == 1
                    sp[0] = m + __log2f(l);
                    sp[1] = dprime;
                }
                const float ca = invl * d.scale, cb = invl * dprime;
#pragma unroll
                for (int kb = 0; kb < NACT; ++kb) {
                    f32x16_t dpk = zero_acc();
#pragma unroll
                    for (int sl = 0; sl < 2; ++sl) {
                        const Frag a = lds_frag_o(st.v() + sl * slab_stride(SPAD) + kb * 32 * SLAB_BYTES, fo);
                        mma_slab<T>(dpk, a, dof[sl]);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) dpk[r] = p[kb][r] * fmaf(dpk[r], ca, -cb);
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const bf16x8_t sb = pack8(dpk, s2);
#pragma unroll
                        for (int db = 0; db < 2; ++db)
                            mfma16(dqacc[db], tr_frag(st.k() + db * slab_stride(SPAD) + (kb * 32 + 16 * s2) * SLAB_BYTES, tro), sb);
                    }
                }
            });
        }
        flush_tile_t(stg, dqacc, dQ + h * HD, lddq, (long)qb * d.T + wave * 32, d.q_rows, d.T - wave * 32, accumulate_dq != 0, lane);
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) { qf[sl] = qn[sl]; dof[sl] = don[sl]; }
    }
}

// dK / dV: one workgroup = (entity, head, round of four key blocks); a wave owns one 32-key block (its K / V fragments stay in
// registers as B operands) and sweeps the query chunks that attend to the entity; chunks are double-buffered like the entities above.
// Per score: p = 2^(s c2 - lse), P' = p / count (0 on this lane's key if it is masked), dS = P' (dP scale - delta').
// Query rows past T are zero rows of the staged Q and dO, so they add nothing whatever P' is; a wave whose key block
// holds only masked keys skips the arithmetic altogether.
template <int NKB, bool CAUSAL, bool QMAP>
__global__ __launch_bounds__(ATT_THREADS, 2) void attn_tr_bwd_dkv_kernel(mmsum_attn_desc d, const bf16_t* __restrict__ dO, long lddo,
                                                                                       bf16_t* __restrict__ dK, long lddk, bf16_t* __restrict__ dV, long lddv,
                                                                                       const float* __restrict__ stats) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TQ = 128;                                  // query rows per staged chunk: a whole query block per iteration
    constexpr int NOWN = 1;                                  // one key block per wave; entities of more than 128 keys take gridDim.z rounds of four blocks
    const int kb0 = blockIdx.z * 4;
    constexpr int QT_TILE = tr_tile_bytes(TQ);
    constexpr int STAGE = 2 * QT_TILE + 2 * TQ * 4;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const FragOff fo = frag_off<T>(lane);
    const TrOff tro = tr_off(lane);
    const float c2 = d.scale * LOG2E_F;
    const int hh = lane >> 5;
    const int h = blockIdx.x;
    const long ent = blockIdx.y;
    const int b = (int)(ent / d.N), n = (int)(ent % d.N);
    const long row0 = ent * d.S;
    const T* Q = static_cast<const T*>(d.q);
    const T* K = static_cast<const T*>(d.k);
    const T* V = static_cast<const T*>(d.v);
    const bool is_null = d.null_entity && d.null_entity[ent];

    Frag kf[NOWN][2], vf[NOWN][2];
    float keep[NOWN];                                         // 1 / 0: this lane's key takes part
    bool alive[NOWN];                                         // wave-uniform: the block has an unmasked key
    f32x16_t dkacc[NOWN][2], dvacc[NOWN][2];
#pragma unroll
    for (int o = 0; o < NOWN; ++o) {
        const int key = (kb0 + wave + 4 * o) * 32 + (lane & 31);
        const long kr = phys_row(d.kv_rows, row0 + key, key < d.S);
        const bool kvalid = kr >= 0;
        const bool masked = !kvalid || (d.pad && d.pad[ent * d.S + key]);
        keep[o] = masked ? 0.f : 1.f;
        alive[o] = __ballot(!masked) != 0;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            kf[o][sl] = global_frag<T>(K + (kvalid ? kr : 0) * d.ldk + h * HD + sl * 32, lane, kvalid);
            vf[o][sl] = global_frag<T>(V + (kvalid ? kr : 0) * d.ldv + h * HD + sl * 32, lane, kvalid);
        }
        dkacc[o][0] = dkacc[o][1] = dvacc[o][0] = dvacc[o][1] = zero_acc();
    }
#pragma unroll
    for (int o = 0; o < NOWN; ++o) { pin_frags(kf[o]); pin_frags(vf[o]); }
    const uint32_t live = valid_entities(d, b, -1);
    const int nchunks = (d.T + TQ - 1) / TQ;
    const int nqb = is_null ? 0 : (d.qpb - ((d.exclude_self && n < d.qpb) ? 1 : 0));
    const int n_it = nqb * nchunks;
    BufTile<TQ> qreg, doreg;
    RowIdx<TQ> qidx;                                          // physical rows of the chunk to request next (q_rows only)
    float streg = 0.f;
    auto coords = [&](int it, int& qb, int& qc, int& i) {
        const int idx = it / nchunks;
        i = idx + ((d.exclude_self && idx >= n) ? 1 : 0);
        qb = b * d.qpb + i;
        qc = (it % nchunks) * TQ;
    };
    auto lookup = [&](int it) {
        int qb, qc, i;
        coords(it, qb, qc, i);
        qidx.load(d.q_rows, (long)qb * d.T + qc, d.T - qc, tid);
    };
    auto prefetch = [&](int it) {
        int qb, qc, i;
        coords(it, qb, qc, i);
        if constexpr (QMAP) {
            qreg.load_mapped(Q + h * HD, d.ldq, qidx, tid);
            doreg.load_mapped(dO + h * HD, lddo, qidx, tid);
        } else {
            qreg.load(Q + h * HD, d.ldq, (long)qb * d.T + qc, d.T - qc, tid);
            doreg.load(dO + h * HD, lddo, (long)qb * d.T + qc, d.T - qc, tid);
        }
        const float* sbase = stats + (((long)qb * d.N + n) * d.H + h) * d.T * 2;
        streg = (tid < TQ * 2 && qc + (tid >> 1) < d.T) ? sbase[(long)qc * 2 + tid] : 0.f;      // tid = 2*query + {0: lse, 1: delta'}
    };
    auto commit = [&](char* stage) {
        qreg.commit(stage, tid);
        doreg.commit(stage + QT_TILE, tid);
        if (tid < TQ * 2) reinterpret_cast<float*>(stage + 2 * QT_TILE)[(tid & 1) * TQ + (tid >> 1)] = streg;
    };
    int cur = 0;
    if (n_it > 0) {
        if constexpr (QMAP) lookup(0);
        prefetch(0);
        if constexpr (QMAP) if (n_it > 1) lookup(1);
        commit(smem);
    }
    __syncthreads();
    for (int it = 0; it < n_it; ++it) {
        int qb, qc, i;
        coords(it, qb, qc, i);
        const int cnt = __popc(d.exclude_self ? (live & ~(1u << i)) : live);
        const float inv_cnt = cnt > 0 ? 1.f / (float)cnt : 0.f;
        const char* qn = smem + cur * STAGE;
        const char* don = qn + QT_TILE;
        const float* st = reinterpret_cast<const float*>(qn + 2 * QT_TILE);
        const bool more = it + 1 < n_it;
        if (more) {
            prefetch(it + 1);
            if constexpr (QMAP) if (it + 2 < n_it) lookup(it + 2);       // one chunk ahead of the loads that use it
        }
#pragma unroll
        for (int o = 0; o < NOWN; ++o) {
            const int kb = kb0 + wave + 4 * o;
            if (kb >= NKB || !alive[o]) continue;
            const int key = kb * 32 + (lane & 31);
            const float icl = keep[o] * inv_cnt;
#pragma unroll
            for (int qq = 0; qq < TQ / 32; ++qq) {
                if (qc + qq * 32 >= d.T) continue;
                if (CAUSAL && kb * 32 > d.causal_q0 + qc + qq * 32 + 31) continue;          // every key of the block lies above every query
                f32x16_t s = zero_acc(), dp = zero_acc();
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    const Frag aq = lds_frag_o(qn + sl * slab_stride(TQ) + qq * 32 * SLAB_BYTES, fo);
                    mma_slab<T>(s, aq, kf[o][sl]);
                    const Frag ad = lds_frag_o(don + sl * slab_stride(TQ) + qq * 32 * SLAB_BYTES, fo);
                    mma_slab<T>(dp, ad, vf[o][sl]);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ql0 = qq * 32 + 8 * g + 4 * hh;
                    const f32x4_t lse4 = *reinterpret_cast<const f32x4_t*>(st + ql0);
                    const f32x4_t del4 = *reinterpret_cast<const f32x4_t*>(st + TQ + ql0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = 4 * g + j;
                        float pr = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -lse4[j])) * icl;   // P / count
                        if (CAUSAL && key > d.causal_q0 + qc + ql0 + j) pr = 0.f;
                        s[r] = pr;
                        dp[r] = pr * fmaf(dp[r], d.scale, -del4[j]);                         // dS
                    }
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8_t pb = pack8(s, s2), sb = pack8(dp, s2);
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        mfma16(dvacc[o][db], tr_frag(don + db * slab_stride(TQ) + (qq * 32 + 16 * s2) * SLAB_BYTES, tro), pb);
                        mfma16(dkacc[o][db], tr_frag(qn + db * slab_stride(TQ) + (qq * 32 + 16 * s2) * SLAB_BYTES, tro), sb);
                    }
                }
            }
        }
        if (more) commit(smem + (cur ^ 1) * STAGE);
        __syncthreads();
        cur ^= 1;
    }
    float* stg = reinterpret_cast<float*>(smem + wave * OUT_STAGE_BYTES);
#pragma unroll
    for (int o = 0; o < NOWN; ++o) {
        const int kb = kb0 + wave + 4 * o;
        if (kb >= NKB) continue;
        flush_tile_t(stg, dkacc[o], dK + h * HD, lddk, row0 + kb * 32, d.kv_rows, d.S - kb * 32, false, lane);
        flush_tile_t(stg, dvacc[o], dV + h * HD, lddv, row0 + kb * 32, d.kv_rows, d.S - kb * 32, false, lane);
    }
}

// ---------------------------------------------------------------------------------------------
// SELF-attention backward in ONE kernel (N == 1 entity, qpb == 1 query block per entity, T and S <= 128: the encoder's and the
// decoder's self-attention).  Nothing leaves the workgroup here -- every query of the (sequence, head) is in it and so is every key --
// so S, the exponentials and dP are computed ONCE: 5 MFMA units (S, dP twice, dQ + dK + dV counted as 3... see below) instead of the
// 8 of the dQ kernel followed by the dK/dV kernel, and half the v_exp.
//   query phase  (wave w = queries 32 w ..): exactly the dQ kernel's body on the staged K / V tiles: scores, p, delta, dS, dQ += dS K.
//                P' = p / l goes to LDS as bf16, transposed into a [key][query] image X (row = key, 264 bytes: the 32 lanes of a
//                read then cover all 64 banks); dS stays packed in registers (32 VGPRs).
//   key phase    (wave w = keys 32 w ..): after a barrier the K / V stage is overwritten by the Q and dO tiles (requested into
//                registers at kernel start), and dV^T += dO^T P' with P' read from X as the MFMA's B operand (8 queries per lane in
//                the permuted order the transposing A read uses); then X is rewritten with dS and dK^T += Q^T dS the same way.
// LDS: stage 33 KB + X / output staging 34 KB = 67 KB: two workgroups per CU, as the two kernels it replaces.
// Causal: wave w's P' exists for key blocks <= w and key block w reads query blocks >= w only.  grid = (H, sequences).
// ---------------------------------------------------------------------------------------------
template <int NKB, bool CAUSAL, bool MAPPED>
__global__ __launch_bounds__(ATT_THREADS, 2) void attn_tr_bwd_self_kernel(mmsum_attn_desc d, const bf16_t* __restrict__ dO, long lddo,
                                                                                        bf16_t* __restrict__ dQ, long lddq, int accumulate_dq,
                                                                                        bf16_t* __restrict__ dK, long lddk, bf16_t* __restrict__ dV, long lddv) {
    typedef bf16_t T;
    static_assert(NKB <= 4, "one key block per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SPAD = NKB * 32, TQ = 128, QT_TILE = tr_tile_bytes(TQ), XROW = 264;
    typedef TrStage<NKB> Stage;
    constexpr int XOFF = ((Stage::BYTES > 2 * QT_TILE ? Stage::BYTES : 2 * QT_TILE) + 15) & ~15;
    const Stage st{smem};
    char* X = smem + XOFF;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const FragOff fo = frag_off<T>(lane);
    const TrOff tro = tr_off(lane);
    const int h = blockIdx.x, qb = blockIdx.y;                  // qpb == 1: the sequence
    const bool live_ent = valid_entities(d, qb, -1) != 0;
    const float c2 = d.scale * LOG2E_F;
    const T* Q = static_cast<const T*>(d.q);
    const T* K = static_cast<const T*>(d.k);
    const T* V = static_cast<const T*>(d.v);

    const int qpos = wave * 32 + (lane & 31);
    const bool qvalid = qpos < d.T;
    Frag qf[2], dof[2];
    {
        const long qr = phys_row(d.q_rows, (long)qb * d.T + qpos, qvalid);
        const T* qrow = Q + (qr >= 0 ? qr : 0) * d.ldq + h * HD;
        const T* drow = dO + (qr >= 0 ? qr : 0) * lddo + h * HD;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            qf[sl] = global_frag<T>(qrow + sl * 32, lane, qr >= 0);
            dof[sl] = global_frag<T>(drow + sl * 32, lane, qr >= 0);
        }
    }
    pin_frags(qf);
    pin_frags(dof);
    f32x16_t dqacc[2] = {zero_acc(), zero_acc()};

    BufTile<TQ> qreg, doreg;                                    // the key phase's A operands: requested now, committed after the query phase
    {
        BufTile<SPAD> kreg, vreg;
        if (MAPPED && d.kv_rows != nullptr) {
            RowIdx<SPAD> kvidx;
            kvidx.load(d.kv_rows, (long)qb * d.S, d.S, tid);
            kreg.load_mapped(K + h * HD, d.ldk, kvidx, tid);
            vreg.load_mapped(V + h * HD, d.ldv, kvidx, tid);
        } else {
            kreg.load(K + h * HD, d.ldk, (long)qb * d.S, d.S, tid);
            vreg.load(V + h * HD, d.ldv, (long)qb * d.S, d.S, tid);
        }
        const uint8_t mreg = (tid >= d.S) ? 1 : (d.pad ? d.pad[(long)qb * d.S + tid] : 0);
        if (MAPPED && d.q_rows != nullptr) {
            RowIdx<TQ> qidx;
            qidx.load(d.q_rows, (long)qb * d.T, d.T, tid);
            qreg.load_mapped(Q + h * HD, d.ldq, qidx, tid);
            doreg.load_mapped(dO + h * HD, lddo, qidx, tid);
        } else {
            qreg.load(Q + h * HD, d.ldq, (long)qb * d.T, d.T, tid);
            doreg.load(dO + h * HD, lddo, (long)qb * d.T, d.T, tid);
        }
        kreg.commit(st.k(), tid);
        vreg.commit(st.v(), tid);
        publish_key_mask(st.bias(), st.slots(), mreg, d.S, SPAD, tid);
    }
    __syncthreads();
    int slen, fmask;
    read_key_mask(st.slots(), slen, fmask);

    // ---- query phase ----
    bf16x8_t dspk[NKB][2];                                      // dS of this wave's 32 queries, packed as the dQ product's operand
    int nact_w = 0;
    if (live_ent) {
        dispatch_blocks<NKB>(active_blocks<CAUSAL>(slen, wave), fmask >> 5, [&](auto nact, auto nfast) {
            constexpr int NACT = decltype(nact)::value, NFAST = decltype(nfast)::value;
            nact_w = NACT;
            f32x16_t p[NKB];
            float m, l;
            scores_tr<NKB, NACT, NFAST, CAUSAL>(p, st.k(), qf, st.bias(), c2, qpos, lane, fo, m, l);
            const float invl = (l > 0.f) ? __builtin_amdgcn_rcpf(l) : 0.f;
            char* xq = X + qpos * 2;
#pragma unroll
            for (int kb = 0; kb < NACT; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    *reinterpret_cast<bf16_t*>(xq + (kb * 32 + acc_row(r, lane)) * XROW) = (bf16_t)(p[kb][r] * invl);
            typedef float pair_t __attribute__((ext_vector_type(2)));
            pair_t raw2 = {0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < NACT; ++kb) {
                f32x16_t dpk = zero_acc();
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    const Frag a = lds_frag_o(st.v() + sl * slab_stride(SPAD) + kb * 32 * SLAB_BYTES, fo);
                    mma_slab<T>(dpk, a, dof[sl]);
                }
#pragma unroll
                for (int r = 0; r < 16; r += 2) raw2 += pair_t{p[kb][r], p[kb][r + 1]} * pair_t{dpk[r], dpk[r + 1]};
            }
            const float dprime = wave_half_sum(raw2.x + raw2.y) * invl * d.scale;          // delta * scale
            const float ca = invl * d.scale, cb = invl * dprime;
#pragma unroll
            for (int kb = 0; kb < NACT; ++kb) {
                f32x16_t dpk = zero_acc();
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    const Frag a = lds_frag_o(st.v() + sl * slab_stride(SPAD) + kb * 32 * SLAB_BYTES, fo);
                    mma_slab<T>(dpk, a, dof[sl]);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) dpk[r] = p[kb][r] * fmaf(dpk[r], ca, -cb);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    dspk[kb][s2] = pack8(dpk, s2);
#pragma unroll
                    for (int db = 0; db < 2; ++db)
                        mfma16(dqacc[db], tr_frag(st.k() + db * slab_stride(SPAD) + (kb * 32 + 16 * s2) * SLAB_BYTES, tro), dspk[kb][s2]);
                }
            }
        });
    }
    __syncthreads();                                            // the K / V tiles are dead; X holds P'
    qreg.commit(smem, tid);
    doreg.commit(smem + QT_TILE, tid);
    __syncthreads();

    // ---- key phase ----
    const int key = wave * 32 + (lane & 31);
    bool masked = true;
    if (wave < NKB && key < d.S) {
        const long kr = phys_row(d.kv_rows, (long)qb * d.S + key, true);
        masked = kr < 0 || (d.pad && d.pad[(long)qb * d.S + key]);
    }
    const bool alive = live_ent && wave < NKB && __ballot(!masked) != 0;          // wave-uniform
    const int nqb = (d.T + 31) >> 5;
    f32x16_t dvacc[2] = {zero_acc(), zero_acc()}, dkacc[2] = {zero_acc(), zero_acc()};
    const char* xk = X + key * XROW + 8 * (lane >> 5);
    auto sweep = [&](f32x16_t (&acc)[2], const char* tile) {
        for (int qq = CAUSAL ? wave : 0; qq < nqb; ++qq) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const u32x2_t lo = *reinterpret_cast<const u32x2_t*>(xk + (qq * 32 + 16 * s2) * 2);
                const u32x2_t hi = *reinterpret_cast<const u32x2_t*>(xk + (qq * 32 + 16 * s2) * 2 + 16);
                const u32x4_t w = {lo[0], lo[1], hi[0], hi[1]};
                const bf16x8_t b = __builtin_bit_cast(bf16x8_t, w);
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    mfma16(acc[db], tr_frag(tile + db * slab_stride(TQ) + (qq * 32 + 16 * s2) * SLAB_BYTES, tro), b);
            }
        }
    };
    if (alive) sweep(dvacc, smem + QT_TILE);
    __syncthreads();                                            // every wave has read P': X takes dS
    {
        char* xq = X + qpos * 2;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
            if (kb < nact_w) {
#pragma unroll
                for (int r = 0; r < 16; ++r) *reinterpret_cast<bf16_t*>(xq + (kb * 32 + acc_row(r, lane)) * XROW) = dspk[kb][r >> 3][r & 7];
            }
    }
    __syncthreads();
    if (alive) sweep(dkacc, smem);
    __syncthreads();                                            // X is read: the output staging takes its place
    float* stg = reinterpret_cast<float*>(smem + XOFF + wave * OUT_STAGE_BYTES);
    flush_tile_t(stg, dqacc, dQ + h * HD, lddq, (long)qb * d.T + wave * 32, d.q_rows, d.T - wave * 32, accumulate_dq != 0, lane);
    if (wave < NKB) {
        flush_tile_t(stg, dkacc, dK + h * HD, lddk, (long)qb * d.S + wave * 32, d.kv_rows, d.S - wave * 32, false, lane);
        flush_tile_t(stg, dvacc, dV + h * HD, lddv, (long)qb * d.S + wave * 32, d.kv_rows, d.S - wave * 32, false, lane);
    }
}

__global__ void entity_null_kernel(const uint8_t* __restrict__ pad, uint8_t* __restrict__ null_entity, int S) {
    __shared__ int any_live;
    if (threadIdx.x == 0) any_live = 0;
    __syncthreads();
    int live = 0;
    for (int s = threadIdx.x; s < S; s += blockDim.x) live |= (pad[(long)blockIdx.x * S + s] == 0);
    if (live) any_live = 1;
    __syncthreads();
    if (threadIdx.x == 0) null_entity[blockIdx.x] = any_live ? 0 : 1;
}

template <typename T> size_t fwd_lds(int nkb) { return (size_t)nkb * 32 * HD * sizeof(T) + 4 * ImageTraits<T>::kBytes + nkb * 32 + 16; }
template <typename T> size_t dkv_lds() { return 4 * (size_t)64 * HD * sizeof(T) + 8 * ImageTraits<T>::kBytes + 64 * 2 * sizeof(float); }

// Dynamic LDS above 64 KiB must be opted into per kernel: once per launch site, in a function-local static (initialised
// exactly once and thread-safe by the language rules; read-only afterwards).  The bound is the device's 160 KiB, so the
// attribute does not depend on the first caller's sizes.
template <typename KernelT>
inline bool allow_lds(KernelT kernel) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
}
#define LAUNCH_LDS(kern, grid, block, lds, s, ...)            \
    do {                                                      \
        static const bool allowed = allow_lds(kern);          \
        (void)allowed;                                        \
        kern<<<grid, block, lds, s>>>(__VA_ARGS__);           \
    } while (0)

inline int nkb_for(int S) { return S <= 64 ? 2 : (S <= 128 ? 4 : 7); }

constexpr size_t LDS_MAX = 160 * 1024;
template <typename T> size_t pipe_lds(int nkb, int ntiles) {
    const size_t need = (size_t)ntiles * nkb * 32 * HD * sizeof(T) + 4 * ImageTraits<T>::kBytes + nkb * 32 * sizeof(float) + 16;
    return need > 4 * (size_t)OUT_STAGE_BYTES ? need : 4 * (size_t)OUT_STAGE_BYTES;          // the output staging reuses it
}

// One entity per business shared by its qpb > 1 query blocks (table / image memory of the decoder's cross-attention).
// Self-attention: one entity per sequence, attended by that sequence's one query block (the merged backward kernel)
inline bool self_attention(const mmsum_attn_desc& d) { return d.N == 1 && d.qpb == 1 && !d.exclude_self && d.S <= 128 && d.T <= 128 && d.causal_q0 == 0; }
inline size_t self_lds(int nkb) {
    const size_t stage = (size_t)2 * tr_tile_bytes(nkb * 32) + nkb * 32 * sizeof(float) + 32, tiles = 2 * (size_t)tr_tile_bytes(128);
    const size_t xoff = ((stage > tiles ? stage : tiles) + 15) & ~(size_t)15, x = (size_t)nkb * 32 * 264;
    return xoff + (x > 4 * (size_t)OUT_STAGE_BYTES ? x : 4 * (size_t)OUT_STAGE_BYTES);
}
inline bool shared_entity(const mmsum_attn_desc& d) { return d.N == 1 && d.qpb > 1 && !d.exclude_self && !d.causal; }
inline int shared_splits(const mmsum_attn_desc& d) { return d.qpb % 3 == 0 ? 3 : 1; }
template <typename T> size_t shared_lds(int nkb) { return (size_t)2 * tr_tile_bytes(nkb * 32) + nkb * 32 * sizeof(float) + 32 + 4 * (size_t)OUT_STAGE_BYTES; }
template <typename T> size_t tr_lds(int nkb) {
    const size_t need = 2 * ((size_t)2 * tr_tile_bytes(nkb * 32) + nkb * 32 * sizeof(float) + 32);      // two TrStage
    return need > 4 * (size_t)OUT_STAGE_BYTES ? need : 4 * (size_t)OUT_STAGE_BYTES;          // the output staging reuses it
}
#define LAUNCH_TR(kern, nkb, causal, mapped, grid, block, lds, s, ...)                                                \
    do {                                                                                                              \
        if (mapped) LAUNCH_TR1(kern, nkb, causal, true, grid, block, lds, s, __VA_ARGS__);                            \
        else LAUNCH_TR1(kern, nkb, causal, false, grid, block, lds, s, __VA_ARGS__);                                  \
    } while (0)
#define LAUNCH_TR1(kern, nkb, causal, MP, grid, block, lds, s, ...)                                                   \
    do {                                                                                                              \
        if (nkb == 2) { if (causal) LAUNCH_LDS((kern<2, true, MP>), grid, block, lds, s, __VA_ARGS__); else LAUNCH_LDS((kern<2, false, MP>), grid, block, lds, s, __VA_ARGS__); } \
        else if (nkb == 4) { if (causal) LAUNCH_LDS((kern<4, true, MP>), grid, block, lds, s, __VA_ARGS__); else LAUNCH_LDS((kern<4, false, MP>), grid, block, lds, s, __VA_ARGS__); } \
        else { if (causal) LAUNCH_LDS((kern<7, true, MP>), grid, block, lds, s, __VA_ARGS__); else LAUNCH_LDS((kern<7, false, MP>), grid, block, lds, s, __VA_ARGS__); } \
    } while (0)

template <typename T>
int attn_fwd_t(const mmsum_attn_desc& d, hipStream_t s) {
    const dim3 grid(d.H, d.n_qblocks), block(ATT_THREADS);
    const int nkb = nkb_for(d.S);
    if constexpr (sizeof(T) == 2) {
        if (shared_entity(d)) {
            const dim3 sgrid(d.H, d.n_qblocks / d.qpb, shared_splits(d));
            const size_t lds = shared_lds<T>(nkb);
            if (nkb == 2) LAUNCH_LDS((attn_tr_fwd_shared_kernel<2>), sgrid, block, lds, s, d);
            else if (nkb == 4) LAUNCH_LDS((attn_tr_fwd_shared_kernel<4>), sgrid, block, lds, s, d);
            else LAUNCH_LDS((attn_tr_fwd_shared_kernel<7>), sgrid, block, lds, s, d);
            return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
        }
        if (nkb > 4 && !d.causal) {          // more than 128 keys per entity (the image memory): chunks of four key blocks, two workgroups per CU
            const size_t lds = tr_lds<T>(4);
            if (d.kv_rows != nullptr) LAUNCH_LDS((attn_tr_fwd_chunk_kernel<true>), grid, block, lds, s, d);
            else LAUNCH_LDS((attn_tr_fwd_chunk_kernel<false>), grid, block, lds, s, d);
            return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
        }
        const size_t lds = tr_lds<T>(nkb);
#if MMSUM_ATTN_W64
        if (!d.causal) {                  // entities of <= 128 keys, not causal: 64 queries per wave, one wave per SIMD (two 2-wave workgroups per CU)
            const dim3 wblock(W64_THREADS);
            const bool mp = d.kv_rows != nullptr;
            if (nkb == 2) { if (mp) LAUNCH_LDS((attn_w64_fwd_kernel<2, true>), grid, wblock, lds, s, d); else LAUNCH_LDS((attn_w64_fwd_kernel<2, false>), grid, wblock, lds, s, d); }
            else { if (mp) LAUNCH_LDS((attn_w64_fwd_kernel<4, true>), grid, wblock, lds, s, d); else LAUNCH_LDS((attn_w64_fwd_kernel<4, false>), grid, wblock, lds, s, d); }
            return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
        }
#endif
        LAUNCH_TR(attn_tr_fwd_kernel, nkb, d.causal, d.kv_rows != nullptr, grid, block, lds, s, d);
        return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
    } else if (pipe_lds<T>(nkb, 2) <= LDS_MAX) {            // f32 (parity mode): the register-prefetch kernels
        const size_t lds = pipe_lds<T>(nkb, 2);
        if (nkb == 2) if (d.causal) LAUNCH_LDS((attn_fwd_pipe_kernel<T, 2, true>), grid, block, lds, s, d); else LAUNCH_LDS((attn_fwd_pipe_kernel<T, 2, false>), grid, block, lds, s, d);
        else if (nkb == 4) if (d.causal) LAUNCH_LDS((attn_fwd_pipe_kernel<T, 4, true>), grid, block, lds, s, d); else LAUNCH_LDS((attn_fwd_pipe_kernel<T, 4, false>), grid, block, lds, s, d);
        else if (d.causal) LAUNCH_LDS((attn_fwd_pipe_kernel<T, 7, true>), grid, block, lds, s, d); else LAUNCH_LDS((attn_fwd_pipe_kernel<T, 7, false>), grid, block, lds, s, d);
        return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
    } else {
        const size_t lds = fwd_lds<T>(nkb);
        if (nkb == 2) LAUNCH_LDS((attn_fwd_kernel<T, 2>), grid, block, lds, s, d);
        else if (nkb == 4) LAUNCH_LDS((attn_fwd_kernel<T, 4>), grid, block, lds, s, d);
        else LAUNCH_LDS((attn_fwd_kernel<T, 7>), grid, block, lds, s, d);
        return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
    }
}

template <typename T>
int attn_bwd_t(const mmsum_attn_desc& d, const void* dout, long lddo, void* dq, long lddq, int accumulate_dq, void* dk, long lddk,
               void* dv, long lddv, void* stats, hipStream_t s) {
    const int nkb = nkb_for(d.S);
    if constexpr (sizeof(T) == 2) {
        if (self_attention(d)) {          // one entity, one query block per entity: one kernel for dQ, dK and dV
            const dim3 grid(d.H, d.n_qblocks), block(ATT_THREADS);
            const size_t lds = self_lds(nkb);
            const bool mapped = d.q_rows != nullptr || d.kv_rows != nullptr;
#define SELF_CASE(N, C, M) LAUNCH_LDS((attn_tr_bwd_self_kernel<N, C, M>), grid, block, lds, s, d, (const T*)dout, lddo, (T*)dq, lddq, accumulate_dq, (T*)dk, lddk, (T*)dv, lddv)
#define SELF_NKB(N) do { if (d.causal) { if (mapped) SELF_CASE(N, true, true); else SELF_CASE(N, true, false); } \
                         else { if (mapped) SELF_CASE(N, false, true); else SELF_CASE(N, false, false); } } while (0)
            if (nkb == 2) SELF_NKB(2); else SELF_NKB(4);
#undef SELF_NKB
#undef SELF_CASE
            return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
        }
        if (shared_entity(d)) {
            const dim3 sgrid(d.H, d.n_qblocks / d.qpb, shared_splits(d)), block(ATT_THREADS);
            const size_t lds = shared_lds<T>(nkb);
            if (nkb == 2) LAUNCH_LDS((attn_tr_bwd_dq_shared_kernel<2>), sgrid, block, lds, s, d, (const T*)dout, lddo, (T*)dq, lddq, accumulate_dq, (float*)stats);
            else if (nkb == 4) LAUNCH_LDS((attn_tr_bwd_dq_shared_kernel<4>), sgrid, block, lds, s, d, (const T*)dout, lddo, (T*)dq, lddq, accumulate_dq, (float*)stats);
            else LAUNCH_LDS((attn_tr_bwd_dq_shared_kernel<7>), sgrid, block, lds, s, d, (const T*)dout, lddo, (T*)dq, lddq, accumulate_dq, (float*)stats);
        } else {
            const dim3 grid(d.H, d.n_qblocks), block(ATT_THREADS);
            // (entities of more than 128 keys keep the seven-block stages here: the chunked form of dQ needs the entity's log-sum-exp
            // and delta before its first key can be finished, i.e. two walks over the chunks = five MFMA products per key block against
            // four and every chunk staged twice -- measured 8 % SLOWER than this kernel at one workgroup per CU, profiles/NOTES_r04.md)
            const size_t lds = tr_lds<T>(nkb);
            LAUNCH_TR(attn_tr_bwd_dq_kernel, nkb, d.causal, d.kv_rows != nullptr, grid, block, lds, s, d, (const T*)dout, lddo, (T*)dq, lddq, accumulate_dq, (float*)stats);
        }
        {
            const int n_ent = (d.n_qblocks / d.qpb) * d.N;
            const dim3 grid(d.H, n_ent, (nkb + 3) / 4), block(ATT_THREADS);
            const size_t lds = 2 * (2 * (size_t)tr_tile_bytes(128) + 2 * 128 * sizeof(float));      // two stages of Q + dO tiles + statistics (> the output staging)
            LAUNCH_TR(attn_tr_bwd_dkv_kernel, nkb, d.causal, d.q_rows != nullptr, grid, block, lds, s, d, (const T*)dout, lddo, (T*)dk, lddk, (T*)dv, lddv, (const float*)stats);
        }
        return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
    } else {            // f32 (parity mode): the register-prefetch kernels
    if (pipe_lds<T>(nkb, 3) <= LDS_MAX) {
        const dim3 grid(d.H, d.n_qblocks), block(ATT_THREADS);
        const size_t lds = pipe_lds<T>(nkb, 3);
        if (nkb == 2) if (d.causal) LAUNCH_LDS((attn_bwd_dq_pipe_kernel<T, 2, true>), grid, block, lds, s, d, (const T*)dout, lddo, (T*)dq, lddq, accumulate_dq, (float*)stats); else LAUNCH_LDS((attn_bwd_dq_pipe_kernel<T, 2, false>), grid, block, lds, s, d, (const T*)dout, lddo, (T*)dq, lddq, accumulate_dq, (float*)stats);
        else if (nkb == 4) if (d.causal) LAUNCH_LDS((attn_bwd_dq_pipe_kernel<T, 4, true>), grid, block, lds, s, d, (const T*)dout, lddo, (T*)dq, lddq, accumulate_dq, (float*)stats); else LAUNCH_LDS((attn_bwd_dq_pipe_kernel<T, 4, false>), grid, block, lds, s, d, (const T*)dout, lddo, (T*)dq, lddq, accumulate_dq, (float*)stats);
        else if (d.causal) LAUNCH_LDS((attn_bwd_dq_pipe_kernel<T, 7, true>), grid, block, lds, s, d, (const T*)dout, lddo, (T*)dq, lddq, accumulate_dq, (float*)stats); else LAUNCH_LDS((attn_bwd_dq_pipe_kernel<T, 7, false>), grid, block, lds, s, d, (const T*)dout, lddo, (T*)dq, lddq, accumulate_dq, (float*)stats);
    } else {
        const dim3 grid(d.H, d.n_qblocks), block(ATT_THREADS);
        const size_t lds = fwd_lds<T>(nkb);
        if (nkb == 2) LAUNCH_LDS((attn_bwd_dq_kernel<T, 2>), grid, block, lds, s, d, (const T*)dout, lddo, (T*)dq, lddq, accumulate_dq, (float*)stats);
        else if (nkb == 4) LAUNCH_LDS((attn_bwd_dq_kernel<T, 4>), grid, block, lds, s, d, (const T*)dout, lddo, (T*)dq, lddq, accumulate_dq, (float*)stats);
        else LAUNCH_LDS((attn_bwd_dq_kernel<T, 7>), grid, block, lds, s, d, (const T*)dout, lddo, (T*)dq, lddq, accumulate_dq, (float*)stats);
    }
    {
        const int n_ent = (d.n_qblocks / d.qpb) * d.N;
        const dim3 grid(d.H, n_ent), block(ATT_THREADS);
        const size_t lds = dkv_lds<T>();
        if (nkb == 2) if (d.causal) LAUNCH_LDS((attn_bwd_dkv_pipe_kernel<T, 2, true>), grid, block, lds, s, d, (const T*)dout, lddo, (T*)dk, lddk, (T*)dv, lddv, (const float*)stats); else LAUNCH_LDS((attn_bwd_dkv_pipe_kernel<T, 2, false>), grid, block, lds, s, d, (const T*)dout, lddo, (T*)dk, lddk, (T*)dv, lddv, (const float*)stats);
        else if (nkb == 4) if (d.causal) LAUNCH_LDS((attn_bwd_dkv_pipe_kernel<T, 4, true>), grid, block, lds, s, d, (const T*)dout, lddo, (T*)dk, lddk, (T*)dv, lddv, (const float*)stats); else LAUNCH_LDS((attn_bwd_dkv_pipe_kernel<T, 4, false>), grid, block, lds, s, d, (const T*)dout, lddo, (T*)dk, lddk, (T*)dv, lddv, (const float*)stats);
        else if (d.causal) LAUNCH_LDS((attn_bwd_dkv_pipe_kernel<T, 7, true>), grid, block, lds, s, d, (const T*)dout, lddo, (T*)dk, lddk, (T*)dv, lddv, (const float*)stats); else LAUNCH_LDS((attn_bwd_dkv_pipe_kernel<T, 7, false>), grid, block, lds, s, d, (const T*)dout, lddo, (T*)dk, lddk, (T*)dv, lddv, (const float*)stats);
    }
    return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
    }
}

int check_desc(const mmsum_attn_desc* d, int dtype) {
    if (!d || d->T <= 0 || d->T > 128 || d->S <= 0 || d->S > 224 || d->N <= 0 || d->N > 32 || d->H <= 0 || d->qpb <= 0 || d->n_qblocks <= 0 ||      // N <= 32: entity sets are 32-bit masks
        d->n_qblocks % d->qpb)
        return MMSUM_ERR_BAD_SHAPE;
    if (dtype != MMSUM_F32 && dtype != MMSUM_BF16) return MMSUM_ERR_BAD_DTYPE;
    if ((d->q_rows || d->kv_rows) && dtype != MMSUM_BF16) return MMSUM_ERR_BAD_DTYPE;      // row maps: the bf16 kernels only
    if (d->causal_q0 != 0 && (!d->causal || d->causal_q0 < 0 || (d->causal_q0 & 31) || d->causal_q0 + d->T > d->S)) return MMSUM_ERR_BAD_SHAPE;
    const long es = dtype == MMSUM_BF16 ? 2 : 4;
    if (((uintptr_t)d->q | (uintptr_t)d->k | (uintptr_t)d->v) & 15) return MMSUM_ERR_BAD_ALIGN;
    if (((d->ldq | d->ldk | d->ldv) * es) & 15) return MMSUM_ERR_BAD_ALIGN;
    return MMSUM_OK;
}

}  // namespace

extern "C" int mmsum_entity_null(const uint8_t* pad, uint8_t* null_entity, int n_entities, int S, void* stream) {
    if (n_entities <= 0 || S <= 0) return MMSUM_ERR_BAD_SHAPE;
    entity_null_kernel<<<dim3(n_entities), dim3(64), 0, (hipStream_t)stream>>>(pad, null_entity, S);
    return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
}

extern "C" int mmsum_attn_fwd(int dtype, const mmsum_attn_desc* d, void* stream) {
    const int rc = check_desc(d, dtype);
    if (rc != MMSUM_OK) return rc;
    return dtype == MMSUM_BF16 ? attn_fwd_t<bf16_t>(*d, (hipStream_t)stream) : attn_fwd_t<float>(*d, (hipStream_t)stream);
}

extern "C" long mmsum_attn_bwd_workspace(const mmsum_attn_desc* d) {
    if (!d) return 0;
    return (long)d->n_qblocks * d->N * d->H * d->T * 2 * (long)sizeof(float);
}

extern "C" int mmsum_attn_bwd(int dtype, const mmsum_attn_desc* d, const void* dout, long lddo, void* dq, long lddq,
                              int accumulate_dq, void* dk, long lddk, void* dv, long lddv, void* stats, void* stream) {
    const int rc = check_desc(d, dtype);
    if (rc != MMSUM_OK) return rc;
    const long es = dtype == MMSUM_BF16 ? 2 : 4;
    if (((uintptr_t)dout & 15) || ((lddo * es) & 15)) return MMSUM_ERR_BAD_ALIGN;
    if (!stats) return MMSUM_ERR_WORKSPACE;
    return dtype == MMSUM_BF16 ? attn_bwd_t<bf16_t>(*d, dout, lddo, dq, lddq, accumulate_dq, dk, lddk, dv, lddv, stats, (hipStream_t)stream)
                               : attn_bwd_t<float>(*d, dout, lddo, dq, lddq, accumulate_dq, dk, lddk, dv, lddv, stats, (hipStream_t)stream);
}
