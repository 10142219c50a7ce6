// Probe of v_permlane32_swap_b32 on gfx950 and of the wave_half_max / wave_half_sum helpers built on it.
#include "../../multimodalsum_amd/csrc/mmsum_device.h"
#include <stdio.h>
__global__ void k(float* out) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[threadIdx.x] = r[0];
    out[64 + threadIdx.x] = r[1];
    float v = (float)((threadIdx.x * 37) % 64);
    out[128 + threadIdx.x] = wave_half_max(v);
    out[192 + threadIdx.x] = wave_half_sum(v);
    out[256 + threadIdx.x] = v;
}
int main() {
    float* d; (void)hipMalloc(&d, 320 * 4);
    k<<<1, 64>>>(d);
    float h[320]; (void)hipMemcpy(h, d, 320 * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < 320; ++i) printf("%g%c", h[i], (i % 32 == 31) ? '\n' : ' ');
    return 0;
}
