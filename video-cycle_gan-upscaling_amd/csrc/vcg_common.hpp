// Shared device/host helpers for the gfx950 hot-path kernels (wave64, MFMA f32 32x32x2).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "vcg.h"

#define VCG_CHECK_PTR(p) do { if ((p) == nullptr) return VCG_E_NULL; } while (0)
#define VCG_LAUNCH_CHECK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// v_mfma_f32_32x32x2_f32: A lane l holds A[i=l&31][k=l>>5]; B lane l holds B[k=l>>5][j=l&31];
// D reg r of lane l is D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31]   (cdna_hip_programming.md section 3)
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ float apply_act(float v, int act, float alpha) {
    // alpha = LeakyReLU slope or the channel's PReLU slope
    if (act == VCG_ACT_LRELU) return v >= 0.f ? v : v * alpha;
    if (act == VCG_ACT_PRELU) return fmaxf(v, 0.f) + alpha * fminf(v, 0.f);
    if (act == VCG_ACT_TANH) return tanhf(v);
    return v;
}

// A readable zero for "optional pointer is NULL": lets epilogues load bias / slope unconditionally
// (pointer and index are selected with scalar ops) instead of branching around every load.
__device__ __attribute__((aligned(16))) const float vcg_zero_word[4] = {0.f, 0.f, 0.f, 0.f};

// ---- buffer loads with hardware range checking (cdna_hip_programming.md T8) --------------------------------------
// Staging loads of the convolution kernels read through a buffer descriptor: an element outside the tensor (zero
// padding, channels past the end, ragged tile edges) is given the byte offset VCG_OOB and the hardware returns 0 for
// it.  What matters is WHERE the mask sits: a select on the ADDRESS in front of the load, not on the DATA behind it --
// a post-load `ok ? v : 0` makes the compiler wait for the whole prefetch (vmcnt(0)) before the MFMA loop it was
// meant to fly under.  A descriptor covers < 4 GiB: callers base it on the image (n) they are working on.
typedef __amdgpu_buffer_rsrc_t vcg_rsrc;
constexpr unsigned VCG_OOB = 0xFFFFFFF0u;
__device__ __forceinline__ vcg_rsrc make_rsrc(const void* base, size_t bytes) {
    // the descriptor must be PROVABLY wave-uniform or hipcc wraps every load in a waterfall loop (guide T20): pass its
    // inputs through readfirstlane as 32-bit halves and keep the saturation in 32-bit scalar arithmetic
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)base);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)base >> 32));
    const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)bytes);
    const unsigned bhi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)bytes >> 32));
    const unsigned nrec = bhi ? 0xFFFFFFE0u : (blo > 0xFFFFFFE0u ? 0xFFFFFFE0u : blo);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0, nrec, 0x00020000);
}
__device__ __forceinline__ float buf_load(vcg_rsrc r, unsigned byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}

// wave64 sum via DPP-free shuffles
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// reduce-scatter over the 32 lanes of a half-wave: every lane holds 32 values; on return lane r (= lane & 31) holds the sum of
// value r over those 32 lanes.  Five butterfly steps, the live set halves at each: 31 shuffles instead of 160; the order of the
// additions is fixed, so the result is deterministic.  (Both half-waves run it at once on their own 32 values.)
template <int CNT>
__device__ __forceinline__ void vcg_rs_step(float (&v)[32], bool up, int m) {
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        const float send = up ? v[i] : v[i + CNT], keep = up ? v[i + CNT] : v[i];
        v[i] = keep + __shfl_xor(send, m, 64);
    }
}
__device__ __forceinline__ float half_wave_reduce_scatter32(float (&v)[32], int r) {
    vcg_rs_step<16>(v, (r & 16) != 0, 16);
    vcg_rs_step<8>(v, (r & 8) != 0, 8);
    vcg_rs_step<4>(v, (r & 4) != 0, 4);
    vcg_rs_step<2>(v, (r & 2) != 0, 2);
    vcg_rs_step<1>(v, (r & 1) != 0, 1);
    return v[0];
}

// block-wide sum of up to 3 values, blockDim.x == 256; result valid in thread 0
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* smem /* >= 4*NV floats */) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) smem[wid * NV + i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = blockDim.x >> 6;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float s = 0.f;
            for (int w = 0; w < nw; ++w) s += smem[w * NV + i];
            v[i] = s;
        }
    }
    __syncthreads();
}
