// Store-bandwidth microbenchmark for the GEMM epilogue's access shape (one 512-thread workgroup per 256x256 bf16 tile).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/store_bw tools/micro/store_bw.hip && /tmp/store_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void tile_store(uint16_t* C, int M, int N, int tiles_n, int BN, int BM) {
    const int t = blockIdx.x, tm = t / tiles_n, tn = t % tiles_n;
    const int tid = threadIdx.x;
    const int cpr = BN / 8;                       // 16-byte chunks per tile row
    const int rows_per_pass = 512 / cpr;
    u32x4 v = {(uint32_t)tid, 1u, 2u, 3u};
    for (int r0 = 0; r0 < BM; r0 += rows_per_pass) {
        const int r = r0 + tid / cpr, c = (tid % cpr) * 8;
        const long row = (long)tm * BM + r;
        if (row < M) {
            u32x4* p = reinterpret_cast<u32x4*>(C + row * N + (long)tn * BN + c);
            if (MODE == 0) *p = v;
            else __builtin_nontemporal_store(v, p);
        }
    }
}

__global__ __launch_bounds__(512) void stream_store(u32x4* C, long n) {
    u32x4 v = {1u, 1u, 2u, 3u};
    for (long i = blockIdx.x * 512L + threadIdx.x; i < n; i += (long)gridDim.x * 512) C[i] = v;
}

int main() {
    const int M = 16128;
    uint16_t* C;
    hipMalloc(&C, (size_t)M * 4096 * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int N : {1024, 4096}) {
        for (int cfg = 0; cfg < 3; ++cfg) {
            const int BN = cfg == 2 ? 512 : 256, BM = cfg == 2 ? 128 : 256;
            const int tiles_n = N / BN, tiles = ((M + BM - 1) / BM) * tiles_n;
            for (int mode = 0; mode < 2; ++mode) {
                if (cfg == 1 && mode == 1) continue;
                float best = 1e9;
                for (int it = 0; it < 5; ++it) {
                    hipEventRecord(e0);
                    if (mode == 0) tile_store<0><<<tiles, 512>>>(C, M, N, tiles_n, BN, BM);
                    else tile_store<1><<<tiles, 512>>>(C, M, N, tiles_n, BN, BM);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                if (cfg == 1) continue;
                printf("N=%4d tile %dx%d %s: %7.1f us  %5.2f TB/s\n", N, BM, BN, mode ? "nontemporal" : "plain      ", best * 1e3,
                       (double)M * N * 2 / best / 1e9);
            }
        }
        float best = 1e9;
        for (int it = 0; it < 5; ++it) {
            hipEventRecord(e0);
            stream_store<<<2048, 512>>>(reinterpret_cast<u32x4*>(C), (long)M * N * 2 / 16);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("N=%4d linear stream        : %7.1f us  %5.2f TB/s\n", N, best * 1e3, (double)M * N * 2 / best / 1e9);
    }
    return 0;
}
