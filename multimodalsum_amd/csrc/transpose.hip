// bf16 matrix transposes that let every GEMM of the step run as the K-contiguous "NT" product:
//   * weights:     W [N,K] -> W^T [K,N] once per optimiser step (batched over all 2-D weights), so
//                  dgrad dx = dy W becomes dy (W^T)^T;
//   * activations: dy [M,N] -> dy^T [N,Mp], x [M,K] -> x^T [K,Mp] (Mp = M rounded up to 64, zero
//                  filled), so wgrad dW = dy^T x becomes (dy^T)(x^T)^T.
// 64x64 tiles through LDS; 16-byte global accesses on both sides.
#include "mmsum_device.h"
#include "mmsum_kernels.h"

namespace {

// dst[c*ld_dst + r] = src[r*ld_src + c] for r < rows, c < cols; dst[c][rows .. rows_pad) = 0.
// 64x64 tile: 16-byte global loads -> 16-byte LDS row writes (pitch 72 elements) -> each thread reads a
// 2-column x 8-row block as eight 32-bit LDS words, splits it into the two transposed 16-byte rows and
// stores them.  Optionally accumulates the column sums of the tile (bias gradients) with one f32 atomic per
// column per tile.
constexpr int TP = 72;     // LDS row pitch in elements (144 B: 16-byte aligned rows, odd multiple of 16 B -> conflict-light)
__device__ __forceinline__ void transpose_tile(const uint16_t* __restrict__ src, long ld_src, uint16_t* __restrict__ dst, long ld_dst,
                                               int rows, int cols, int rows_pad, int tr, int tc, uint16_t* tile, float* colsum,
                                               float* red) {
    const int tid = threadIdx.x;
    const int r0 = tr * 64, c0 = tc * 64;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int id = tid + it * 256;
        const int r = id >> 3, ch = id & 7;
        const int gr = r0 + r, gc = c0 + ch * 8;
        u32x4_t w = u32x4_t{0, 0, 0, 0};
        if (gr < rows && gc + 8 <= cols) {
            w = *reinterpret_cast<const u32x4_t*>(src + (long)gr * ld_src + gc);
        } else if (gr < rows && gc < cols) {
            uint16_t v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (gc + j < cols) ? src[(long)gr * ld_src + gc + j] : (uint16_t)0;
            __builtin_memcpy(&w, v, 16);
        }
        *reinterpret_cast<u32x4_t*>(tile + r * TP + ch * 8) = w;
    }
    __syncthreads();
    {
        const int cp = tid & 31, rg = tid >> 5;          // column pair (2 cols), row group (8 rows)
        uint32_t w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = *reinterpret_cast<const uint32_t*>(tile + (rg * 8 + j) * TP + cp * 2);
        u32x4_t lo, hi;                                   // lo: column 2cp, hi: column 2cp+1; 8 consecutive source rows each
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lo[j] = (w[2 * j] & 0xFFFFu) | (w[2 * j + 1] << 16);
            hi[j] = (w[2 * j] >> 16) | (w[2 * j + 1] & 0xFFFF0000u);
        }
        const int gr = r0 + rg * 8;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int gc = c0 + cp * 2 + k;
            const u32x4_t v = k ? hi : lo;
            if (gc < cols && gr < rows_pad) {
                if (gr + 8 <= rows_pad) {
                    *reinterpret_cast<u32x4_t*>(dst + (long)gc * ld_dst + gr) = v;
                } else {
                    uint16_t e[8];
                    __builtin_memcpy(e, &v, 16);
                    for (int j = 0; j < 8 && gr + j < rows_pad; ++j) dst[(long)gc * ld_dst + gr + j] = e[j];
                }
            }
        }
        if (colsum != nullptr) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s0 += (float)__builtin_bit_cast(bf16_t, (uint16_t)(w[j] & 0xFFFFu));
                s1 += (float)__builtin_bit_cast(bf16_t, (uint16_t)(w[j] >> 16));
            }
            red[rg * 64 + cp * 2] = s0;
            red[rg * 64 + cp * 2 + 1] = s1;
        }
    }
    __syncthreads();
    if (colsum != nullptr && tid < 64 && c0 + tid < cols) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k * 64 + tid];
        atomicAdd(colsum + c0 + tid, t);
    }
}

__global__ __launch_bounds__(256) void transpose_kernel(const uint16_t* __restrict__ src, long ld_src, uint16_t* __restrict__ dst, long ld_dst,
                                                        int rows, int cols, int rows_pad, float* __restrict__ colsum) {
    __shared__ __attribute__((aligned(16))) uint16_t tile[64 * TP];
    __shared__ float red[8 * 64];
    const int tiles_r = (rows_pad + 63) / 64, tiles_c = (cols + 63) / 64;
    for (int t = blockIdx.x; t < tiles_r * tiles_c; t += gridDim.x)
        transpose_tile(src, ld_src, dst, ld_dst, rows, cols, rows_pad, t / tiles_c, t % tiles_c, tile, colsum, red);
}

// desc[i] = {src_off, dst_off, rows, cols, ld_src, ld_dst} (elements), one matrix per blockIdx.y
__global__ __launch_bounds__(256) void transpose_batched_kernel(const uint16_t* __restrict__ src_base, uint16_t* __restrict__ dst_base,
                                                                const long* __restrict__ desc) {
    __shared__ __attribute__((aligned(16))) uint16_t tile[64 * TP];
    const long* d = desc + (long)blockIdx.y * 6;
    const int rows = (int)d[2], cols = (int)d[3];
    const int tiles_r = (rows + 63) / 64, tiles_c = (cols + 63) / 64;
    for (int t = blockIdx.x; t < tiles_r * tiles_c; t += gridDim.x)
        transpose_tile(src_base + d[0], d[4], dst_base + d[1], d[5], rows, cols, rows, t / tiles_c, t % tiles_c, tile, nullptr, nullptr);
}

}  // namespace

extern "C" int mmsum_transpose_bf16(const void* src, long ld_src, void* dst, long ld_dst, int rows, int cols, int rows_pad, float* colsum,
                                    void* stream) {
    if (rows <= 0 || cols <= 0 || rows_pad < rows || ld_dst < rows_pad) return MMSUM_ERR_BAD_SHAPE;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) || ((ld_src * 2) & 15) || ((ld_dst * 2) & 15)) return MMSUM_ERR_BAD_ALIGN;
    const long tiles = (long)((rows_pad + 63) / 64) * ((cols + 63) / 64);
    const int grid = (int)(tiles > 4096 ? 4096 : tiles);
    transpose_kernel<<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>((const uint16_t*)src, ld_src, (uint16_t*)dst, ld_dst, rows, cols, rows_pad, colsum);
    return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
}

extern "C" int mmsum_transpose_bf16_batched(const void* src_base, void* dst_base, const long* desc, int n, int max_tiles, void* stream) {
    if (n <= 0 || max_tiles <= 0) return MMSUM_ERR_BAD_SHAPE;
    const int gx = max_tiles > 256 ? 256 : max_tiles;
    transpose_batched_kernel<<<dim3(gx, n), dim3(256), 0, (hipStream_t)stream>>>((const uint16_t*)src_base, (uint16_t*)dst_base, desc);
    return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
}
