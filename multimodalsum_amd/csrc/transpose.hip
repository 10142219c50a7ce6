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
__device__ __forceinline__ void transpose_tile(const uint16_t* __restrict__ src, long ld_src, uint16_t* __restrict__ dst, long ld_dst,
                                               int rows, int cols, int rows_pad, int tr, int tc, uint16_t (*tile)[66]) {
    const int tid = threadIdx.x;
    const int r0 = tr * 64, c0 = tc * 64;
    // load 64 rows x 64 cols (8 chunks of 8 elements per row)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int id = tid + it * 256;
        const int r = id >> 3, ch = id & 7;
        const int gr = r0 + r, gc = c0 + ch * 8;
        uint16_t v[8];
        if (gr < rows && gc + 8 <= cols) {
            const u32x4_t w = *reinterpret_cast<const u32x4_t*>(src + (long)gr * ld_src + gc);
            __builtin_memcpy(v, &w, 16);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (gr < rows && gc + j < cols) ? src[(long)gr * ld_src + gc + j] : (uint16_t)0;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) tile[r][ch * 8 + j] = v[j];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int id = tid + it * 256;
        const int c = id >> 3, ch = id & 7;      // output row = source column c, 8 consecutive source rows
        const int gc = c0 + c, gr = r0 + ch * 8;
        if (gc < cols && gr < rows_pad) {
            uint16_t v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = tile[ch * 8 + j][c];
            if (gr + 8 <= rows_pad) {
                u32x4_t w;
                __builtin_memcpy(&w, v, 16);
                *reinterpret_cast<u32x4_t*>(dst + (long)gc * ld_dst + gr) = w;
            } else {
                for (int j = 0; j < 8 && gr + j < rows_pad; ++j) dst[(long)gc * ld_dst + gr + j] = v[j];
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void transpose_kernel(const uint16_t* __restrict__ src, long ld_src, uint16_t* __restrict__ dst, long ld_dst,
                                                        int rows, int cols, int rows_pad) {
    __shared__ uint16_t tile[64][66];
    const int tiles_r = (rows_pad + 63) / 64, tiles_c = (cols + 63) / 64;
    for (int t = blockIdx.x; t < tiles_r * tiles_c; t += gridDim.x)
        transpose_tile(src, ld_src, dst, ld_dst, rows, cols, rows_pad, t / tiles_c, t % tiles_c, tile);
}

// desc[i] = {src_off, dst_off, rows, cols, ld_src, ld_dst} (elements), one matrix per blockIdx.y
__global__ __launch_bounds__(256) void transpose_batched_kernel(const uint16_t* __restrict__ src_base, uint16_t* __restrict__ dst_base,
                                                                const long* __restrict__ desc) {
    __shared__ uint16_t tile[64][66];
    const long* d = desc + (long)blockIdx.y * 6;
    const int rows = (int)d[2], cols = (int)d[3];
    const int tiles_r = (rows + 63) / 64, tiles_c = (cols + 63) / 64;
    for (int t = blockIdx.x; t < tiles_r * tiles_c; t += gridDim.x)
        transpose_tile(src_base + d[0], d[4], dst_base + d[1], d[5], rows, cols, rows, t / tiles_c, t % tiles_c, tile);
}

}  // namespace

extern "C" int mmsum_transpose_bf16(const void* src, long ld_src, void* dst, long ld_dst, int rows, int cols, int rows_pad, void* stream) {
    if (rows <= 0 || cols <= 0 || rows_pad < rows || ld_dst < rows_pad) return MMSUM_ERR_BAD_SHAPE;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) || ((ld_src * 2) & 15) || ((ld_dst * 2) & 15)) return MMSUM_ERR_BAD_ALIGN;
    const long tiles = (long)((rows_pad + 63) / 64) * ((cols + 63) / 64);
    const int grid = (int)(tiles > 4096 ? 4096 : tiles);
    transpose_kernel<<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>((const uint16_t*)src, ld_src, (uint16_t*)dst, ld_dst, rows, cols, rows_pad);
    return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
}

extern "C" int mmsum_transpose_bf16_batched(const void* src_base, void* dst_base, const long* desc, int n, int max_tiles, void* stream) {
    if (n <= 0 || max_tiles <= 0) return MMSUM_ERR_BAD_SHAPE;
    const int gx = max_tiles > 256 ? 256 : max_tiles;
    transpose_batched_kernel<<<dim3(gx, n), dim3(256), 0, (hipStream_t)stream>>>((const uint16_t*)src_base, (uint16_t*)dst_base, desc);
    return hipGetLastError() == hipSuccess ? MMSUM_OK : MMSUM_ERR_HIP;
}
