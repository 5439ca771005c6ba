// Micro-benchmark: what does a bare v_mfma_f32_32x32x16_bf16 stream sustain on this device, and at which clock?
// Operands are random bf16 values in registers (zeros would draw less power and hold a higher clock than a real kernel).
// s_memtime (core clock) against s_memrealtime (100 MHz) over the kernel gives the clock the chip held.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_peak_bf16.hip -o /tmp/mfma_peak_bf16 && /tmp/mfma_peak_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256) void k(const uint4* in, float* out, unsigned long long* clk, int iters) {
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a[4], b[NACC];
    for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(bf16x8, in[threadIdx.x * 4 + i]);
    for (int i = 0; i < NACC; ++i) b[i] = __builtin_bit_cast(bf16x8, in[1024 + threadIdx.x * NACC + i]);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[i], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int NACC>
void run(const char* name, int blocks_per_cu, const uint4* in) {
    float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    unsigned long long* clk; hipMalloc(&clk, 256 * 8 * 2 * sizeof(unsigned long long));
    const int iters = 20000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<grid, 256>>>(in, out, clk, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<grid, 256>>>(in, out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double flop = (double)grid * 4 * iters * 4 * NACC * 32768.0;
    printf("%-28s blocks/CU=%d  %.3f ms  %.1f TFLOP/s   clock %.3f GHz (s_memtime / s_memrealtime x 100 MHz)\n", name, blocks_per_cu, ms, flop / ms / 1e9,
           (double)h[0] / (double)h[1] * 0.1);
    hipFree(out); hipFree(clk);
}

int main(int argc, char** argv) {
    const bool zeros = argc > 1;
    uint4* in; hipMalloc(&in, 8192 * sizeof(uint4));
    unsigned short* hbuf = (unsigned short*)malloc(8192 * 16);
    for (int i = 0; i < 8192 * 8; ++i) hbuf[i] = zeros ? 0 : (unsigned short)((rand() & 0x807F) | ((120 + rand() % 10) << 7));   // random sign/mantissa, exponent near 1
    hipMemcpy(in, hbuf, 8192 * 16, hipMemcpyHostToDevice);
    printf("%s operands\n", zeros ? "zero" : "random");
    run<4>("bare bf16 MFMA, 4 acc", 1, in);
    run<8>("bare bf16 MFMA, 8 acc", 1, in);
    run<4>("bare bf16 MFMA, 4 acc", 2, in);
    return 0;
}
