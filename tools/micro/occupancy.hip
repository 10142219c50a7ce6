// How many workgroups does a CU hold at a given dynamic-LDS size / register budget?  Every workgroup spins for a fixed
// time; the launch time of 12 workgroups per CU divided by one spin gives the number of rounds, i.e. 12 / resident.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/occupancy tools/micro/occupancy.hip && /tmp/occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int THREADS>
__global__ __launch_bounds__(THREADS) void spin(float* out, long long cycles) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    float a = lds[(threadIdx.x + 1) % THREADS];
    while (__builtin_readcyclecounter() - t0 < cycles) a = a * 1.0001f + 0.5f;
    if (a == 12345.f) out[0] = a;
}

template <int THREADS>
void sweep(float* d) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(spin<THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int per_cu = 12;
    for (int kb : {4, 16, 32, 40, 42, 48, 50, 52, 54, 64, 72, 80, 96, 128, 160}) {
        float best = 1e9f, one = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            spin<THREADS><<<256 * per_cu, THREADS, (size_t)kb * 1024>>>(d, 2000000);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
            hipEventRecord(e0);
            spin<THREADS><<<256, THREADS, (size_t)kb * 1024>>>(d, 2000000);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            one = ms < one ? ms : one;
        }
        printf("threads %3d  lds %3d KB: %6.2f ms for %d per CU, %5.2f ms for one  -> %.1f rounds -> %.1f resident per CU  (%s)\n", THREADS, kb, best,
               per_cu, one, best / one, per_cu / (best / one), hipGetErrorString(hipGetLastError()));
    }
}

int main() {
    float* d;
    hipMalloc(&d, 4);
    sweep<256>(d);
    sweep<512>(d);
    return 0;
}
