// bf16-storage normalisation kernels (activations bf16 NHWC; statistics, affine and activation arithmetic fp32):
// BatchNormalization / instance norm + PReLU / LeakyReLU + Add of residual_block (upscaling/upscaler/model.py:20-25)
// on the layout of bf16_conv.hip.  HBM-bound: a thread owns 8 consecutive channels (16 bytes) of a pixel; the
// per-channel reductions run over pixels, so a wavefront's 8 lanes of one pixel never have to exchange anything and
// the cross-pixel sums go through LDS once per block.  vcg_norm_finalize (fp32, norm.hip) turns the statistics
// into scale / shift and maintains the moving averages exactly as on the fp32 path.
#include "vcg_common.hpp"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int SPB = 2048;        // pixels per block of the partial pass

// partial sums of (x - K) and (x - K)^2 per channel over a slab of pixels; K = the group's first pixel (a shift
// that keeps E[d^2] - E[d]^2 from cancelling when |mean| >> std).  group = image (instance) or the whole batch.
__global__ __launch_bounds__(256) void stats_partial_bf16_kernel(const bf16x8* __restrict__ x, int c8, long group_pixels,
                                                                  int slabs_per_group, float* __restrict__ part) {
    extern __shared__ float red[];                                   // [256 / c8 lanes... ] see below
    const int grp = blockIdx.x / slabs_per_group, slab = blockIdx.x - grp * slabs_per_group;
    const int ch = threadIdx.x % c8, pl = threadIdx.x / c8, npl = 256 / c8;
    const bf16x8* xg = x + (long)grp * group_pixels * c8;
    const bf16x8 k8 = xg[ch];
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
    const long p0 = (long)slab * SPB, p1 = p0 + SPB < group_pixels ? p0 + SPB : group_pixels;
    if (pl < npl)
        for (long p = p0 + pl; p < p1; p += npl) {
            const bf16x8 v = xg[p * c8 + ch];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = (float)v[j] - (float)k8[j];
                s[j] += d;
                q[j] += d * d;
            }
        }
    // fixed-order reduction over the pixel lanes
    float* rs = red;                       // [npl][c8*8] sums, then squares
    float* rq = red + 256 * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        rs[threadIdx.x * 8 + j] = s[j];
        rq[threadIdx.x * 8 + j] = q[j];
    }
    __syncthreads();
    if (threadIdx.x < c8 * 8) {
        const int cc = threadIdx.x;        // channel = chunk*8 + j  <->  thread (pl, ch=cc/8) element j=cc%8
        float ts = 0.f, tq = 0.f;
        for (int l = 0; l < npl; ++l) {
            ts += rs[(l * c8 + (cc >> 3)) * 8 + (cc & 7)];
            tq += rq[(l * c8 + (cc >> 3)) * 8 + (cc & 7)];
        }
        part[((long)blockIdx.x * c8 * 8 + cc) * 2] = ts;
        part[((long)blockIdx.x * c8 * 8 + cc) * 2 + 1] = tq;
    }
}

__global__ void stats_final_bf16_kernel(const __bf16* __restrict__ x, const float* __restrict__ part, int c, long group_pixels,
                                        int slabs_per_group, int groups, float* __restrict__ mean, float* __restrict__ var) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // (group, channel)
    if (i >= groups * c) return;
    const int grp = i / c, ch = i - grp * c;
    double s = 0.0, q = 0.0;
    for (int b = 0; b < slabs_per_group; ++b) {
        const float* pp = part + (((long)grp * slabs_per_group + b) * c + ch) * 2;
        s += pp[0];
        q += pp[1];
    }
    const double k = (double)(float)x[(long)grp * group_pixels * c + ch];
    const double m = s / (double)group_pixels;
    mean[i] = (float)(k + m);
    const double v = q / (double)group_pixels - m * m;
    var[i] = (float)(v > 0.0 ? v : 0.0);
}

__global__ __launch_bounds__(256) void norm_act_fwd_bf16_kernel(const bf16x8* __restrict__ x, long total8, int c8, long pix_per_img,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 int per_sample, int act, float alpha, const float* __restrict__ prelu,
                                                                 const bf16x8* __restrict__ res, bf16x8* __restrict__ y) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total8) return;
    const int ch = (int)(i % c8) * 8;
    const long img = per_sample ? i / (pix_per_img * c8) : 0;
    const float* sc = scale + img * c8 * 8 + ch;
    const float* sh = shift + img * c8 * 8 + ch;
    const bf16x8 v = x[i];
    bf16x8 r8;
    if (res) r8 = res[i];
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float u = (float)v[j] * sc[j] + sh[j];
        if (act == VCG_ACT_PRELU) u = u > 0.f ? u : u * prelu[ch + j];
        else if (act == VCG_ACT_LRELU) u = u > 0.f ? u : u * alpha;
        if (res) u += (float)r8[j];
        o[j] = (__bf16)u;
    }
    y[i] = o;
}

}  // namespace

extern "C" {

size_t vcg_norm_stats_bf16_workspace_bytes(int n, int c, int hw, int mode) {
    const long gp = mode == VCG_NORM_INSTANCE ? hw : (long)n * hw;
    const long groups = mode == VCG_NORM_INSTANCE ? n : 1;
    return (size_t)(groups * ((gp + SPB - 1) / SPB) * c * 2 * sizeof(float));
}

int vcg_norm_stats_bf16(const void* x, int n, int c, int hw, int mode, float* mean, float* var, void* ws, size_t ws_bytes,
                        hipStream_t stream) {
    VCG_CHECK_PTR(x); VCG_CHECK_PTR(mean); VCG_CHECK_PTR(var); VCG_CHECK_PTR(ws);
    if (n <= 0 || c <= 0 || hw <= 0) return VCG_E_SHAPE;
    if (c % 8 != 0 || c > 256) return VCG_E_UNSUPPORTED;
    if (ws_bytes < vcg_norm_stats_bf16_workspace_bytes(n, c, hw, mode)) return VCG_E_WORKSPACE;
    const long gp = mode == VCG_NORM_INSTANCE ? hw : (long)n * hw;
    const int groups = mode == VCG_NORM_INSTANCE ? n : 1;
    const int slabs = (int)((gp + SPB - 1) / SPB);
    stats_partial_bf16_kernel<<<groups * slabs, 256, 2 * 256 * 8 * sizeof(float), stream>>>((const bf16x8*)x, c / 8, gp, slabs, (float*)ws);
    VCG_LAUNCH_CHECK();
    stats_final_bf16_kernel<<<ceil_div(groups * c, 128), 128, 0, stream>>>((const __bf16*)x, (const float*)ws, c, gp, slabs, groups, mean, var);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_norm_act_fwd_bf16(const void* x, int n, int c, int hw, const float* scale, const float* shift, int per_sample, int act,
                          float alpha, const float* prelu_alpha, const void* residual, void* y, hipStream_t stream) {
    VCG_CHECK_PTR(x); VCG_CHECK_PTR(scale); VCG_CHECK_PTR(shift); VCG_CHECK_PTR(y);
    if (n <= 0 || c <= 0 || hw <= 0) return VCG_E_SHAPE;
    if (c % 8 != 0) return VCG_E_UNSUPPORTED;
    if (act == VCG_ACT_PRELU && !prelu_alpha) return VCG_E_NULL;
    if (act == VCG_ACT_TANH) return VCG_E_UNSUPPORTED;
    const long total8 = (long)n * hw * (c / 8);
    norm_act_fwd_bf16_kernel<<<(unsigned)((total8 + 255) / 256), 256, 0, stream>>>((const bf16x8*)x, total8, c / 8, hw, scale, shift,
                                                                                  per_sample, act, alpha, prelu_alpha, (const bf16x8*)residual,
                                                                                  (bf16x8*)y);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

}  // extern "C"
