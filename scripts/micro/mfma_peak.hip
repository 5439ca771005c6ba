// Micro-benchmark: what does a bare v_mfma_f32_32x32x2_f32 stream sustain on this device?
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    __shared__ float sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = seed * (i & 7);
    __syncthreads();
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = seed + threadIdx.x, b = seed * 0.5f;
    const float* p = sm + (threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (LDS) { a = p[(u * 64) & 4095]; b = p[(u * 64 + 32) & 4095]; }
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool LDS>
void run(const char* name, int blocks_per_cu) {
    float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    const int iters = 2000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, LDS><<<grid, 256>>>(out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC, LDS><<<grid, 256>>>(out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)grid * 4 * iters * 16 * NACC * 4096.0;
    printf("%-40s blocks/CU=%d  %.3f ms  %.1f TFLOP/s\n", name, blocks_per_cu, ms, flop / ms / 1e9);
    hipFree(out);
}

int main() {
    run<4, false>("bare MFMA, 4 acc", 1);
    run<4, false>("bare MFMA, 4 acc", 2);
    run<4, false>("bare MFMA, 4 acc", 4);
    run<8, false>("bare MFMA, 8 acc", 1);
    run<4, true>("MFMA + 2 ds_read per 4 MFMA", 1);
    run<4, true>("MFMA + 2 ds_read per 4 MFMA", 2);
    run<1, true>("MFMA + 2 ds_read per 1 MFMA", 2);
    run<2, true>("MFMA + 2 ds_read per 2 MFMA", 2);
    return 0;
}
