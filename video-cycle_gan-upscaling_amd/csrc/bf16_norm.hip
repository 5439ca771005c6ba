// bf16-storage normalisation kernels (activations bf16 NHWC; statistics, affine and activation arithmetic fp32):
// BatchNormalization / instance norm + PReLU / LeakyReLU + Add of residual_block (upscaling/upscaler/model.py:20-25)
// on the layout of bf16_conv.hip.  HBM-bound: a thread owns 8 consecutive channels (16 bytes) of a pixel; the
// per-channel reductions run over pixels, so a wavefront's 8 lanes of one pixel never have to exchange anything and
// the cross-pixel sums go through LDS once per block.  vcg_norm_finalize (fp32, norm.hip) turns the statistics
// into scale / shift and maintains the moving averages exactly as on the fp32 path.
#include "vcg_common.hpp"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int SPB = 512;         // pixels per block of the partial passes (4 blocks per CU at the C2 trunk shape: 2048 ran one block per CU at 1.4 TB/s)

// partial sums of (x - K) and (x - K)^2 per channel over a slab of pixels; K = the group's first pixel (a shift
// that keeps E[d^2] - E[d]^2 from cancelling when |mean| >> std).  group = image (instance) or the whole batch.
// A block walks the slabs b, b + bpg, ... of its group (bpg blocks per group, at most NBLK in all): one LDS reduction and one partial
// record per BLOCK, so the final pass sums <= NBLK records instead of one per 512 pixels (1024 at the C2 trunk shape: 15 us in a
// single workgroup).
__global__ __launch_bounds__(256) void stats_partial_bf16_kernel(const bf16x8* __restrict__ x, int c8, long group_pixels,
                                                                  int slabs_per_group, int bpg, float* __restrict__ part) {
    extern __shared__ float red[];                                   // [256 / c8 lanes... ] see below
    const int grp = blockIdx.x / bpg, blk = blockIdx.x - grp * bpg;
    const int ch = threadIdx.x % c8, pl = threadIdx.x / c8, npl = 256 / c8;
    const bf16x8* xg = x + (long)grp * group_pixels * c8;
    const bf16x8 k8 = xg[ch];
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
    if (pl < npl)
        for (int slab = blk; slab < slabs_per_group; slab += bpg) {
            const long p0 = (long)slab * SPB, p1 = p0 + SPB < group_pixels ? p0 + SPB : group_pixels;
            // four pixels per trip, their loads issued together (one load in flight per thread left the pass latency-bound: 2.3 TB/s)
            long p = p0 + pl;
            for (; p + 3 * npl < p1; p += 4 * npl) {
                bf16x8 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = xg[(p + u * npl) * c8 + ch];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float d = (float)v[u][j] - (float)k8[j];
                        s[j] += d;
                        q[j] += d * d;
                    }
            }
            for (; p < p1; p += npl) {
                const bf16x8 v = xg[p * c8 + ch];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = (float)v[j] - (float)k8[j];
                    s[j] += d;
                    q[j] += d * d;
                }
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
    for (int cc = threadIdx.x; cc < c8 * 8; cc += 256) {      // channel = chunk*8 + j  <->  thread (pl, ch=cc/8) element j=cc%8
        float ts = 0.f, tq = 0.f;
        for (int l = 0; l < npl; ++l) {
            ts += rs[(l * c8 + (cc >> 3)) * 8 + (cc & 7)];
            tq += rq[(l * c8 + (cc >> 3)) * 8 + (cc & 7)];
        }
        part[((long)blockIdx.x * c8 * 8 + cc) * 2] = ts;
        part[((long)blockIdx.x * c8 * 8 + cc) * 2 + 1] = tq;
    }
}

// block = (group, 64 channels): 16 row groups sum the slabs in a fixed interleaved order, then one fixed-order combine
__global__ __launch_bounds__(1024) void stats_final_bf16_kernel(const __bf16* __restrict__ x, const float* __restrict__ part, int c, long group_pixels,
                                                                 int slabs_per_group, int groups, float* __restrict__ mean, float* __restrict__ var) {
    __shared__ double rs[16][64], rq[16][64];
    const int cblocks = (c + 63) / 64, grp = blockIdx.x / cblocks, ch = (blockIdx.x - grp * cblocks) * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    double s = 0.0, q = 0.0;
    if (ch < c) {
        // four records in flight per thread (one at a time left this single-workgroup pass latency-bound: 15 us for 1024 records)
        int b = g;
        for (; b + 48 < slabs_per_group; b += 64) {
            float2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const float2*)(part + (((long)grp * slabs_per_group + b + 16 * u) * c + ch) * 2);
#pragma unroll
            for (int u = 0; u < 4; ++u) { s += v[u].x; q += v[u].y; }
        }
        for (; b < slabs_per_group; b += 16) {
            const float* pp = part + (((long)grp * slabs_per_group + b) * c + ch) * 2;
            s += pp[0];
            q += pp[1];
        }
    }
    rs[g][threadIdx.x & 63] = s;
    rq[g][threadIdx.x & 63] = q;
    __syncthreads();
    if (g != 0 || ch >= c) return;
#pragma unroll
    for (int i = 1; i < 16; ++i) { s += rs[i][threadIdx.x]; q += rq[i][threadIdx.x]; }
    const double k = (double)(float)x[(long)grp * group_pixels * c + ch];
    const double m = s / (double)group_pixels;
    mean[grp * c + ch] = (float)(k + m);
    const double v = q / (double)group_pixels - m * m;
    var[grp * c + ch] = (float)(v > 0.0 ? v : 0.0);
}

constexpr int APB = 512;        // pixels per block of the apply passes

// y = act(x*scale + shift) + residual.  block = (image, APB pixels); a thread keeps ITS 8 channels' parameters in registers
// and walks the pixels (the first version re-read 64-128 bytes of parameters for every 16 bytes of data: 1.7 TB/s)
__global__ __launch_bounds__(256) void norm_act_fwd_bf16_kernel(const bf16x8* __restrict__ x, int c8, long pix_per_img, int blocks_per_img,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 int per_sample, int act, float alpha, const float* __restrict__ prelu,
                                                                 const bf16x8* __restrict__ res, bf16x8* __restrict__ y) {
    const int img = blockIdx.x / blocks_per_img, pb = blockIdx.x - img * blocks_per_img;
    const int ch = threadIdx.x % c8, pl = threadIdx.x / c8, npl = 256 / c8;
    if (pl >= npl) return;
    const int pidx = (per_sample ? img * c8 * 8 : 0) + ch * 8;
    float sc[8], sh[8], sl[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = scale[pidx + j];
        sh[j] = shift[pidx + j];
        sl[j] = act == VCG_ACT_PRELU ? prelu[ch * 8 + j] : (act == VCG_ACT_LRELU ? alpha : 1.f);
    }
    const long p0 = (long)pb * APB, p1 = p0 + APB < pix_per_img ? p0 + APB : pix_per_img;
    const long base = (long)img * pix_per_img * c8;
    auto one = [&](const bf16x8& v, const bf16x8& r8) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float u = (float)v[j] * sc[j] + sh[j];
            u = u > 0.f ? u : u * sl[j];
            if (res) u += (float)r8[j];
            o[j] = (__bf16)u;
        }
        return o;
    };
    // four pixels per trip: all their loads first, then the arithmetic and the stores
    long p = p0 + pl;
    for (; p + 3 * npl < p1; p += 4 * npl) {
        bf16x8 v[4], r8[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = base + (p + u * npl) * c8 + ch;
            v[u] = x[i];
            if (res) r8[u] = res[i];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) y[base + (p + u * npl) * c8 + ch] = one(v[u], r8[u]);
    }
    for (; p < p1; p += npl) {
        const long i = base + p * c8 + ch;
        const bf16x8 v = x[i];
        bf16x8 r8;
        if (res) r8 = res[i];
        y[i] = one(v, r8);
    }
}

// ---- backward -------------------------------------------------------------------------------------------------
// y = act(u), u = gamma*xhat + beta, xhat = (x - mean)*invstd.   pass 1: per channel (per image in instance mode)
// sum dz, sum dz*xhat, sum dy*min(u,0) over a slab of pixels -> partials;  pass 2: the partials are summed in a fixed
// order by every block that needs them (tiny: <= 128 slabs) and dx = gamma*invstd*(dz - S1/M - xhat*S2/M) is applied.
__device__ __forceinline__ float act_grad_bf16(float u, int act, float slope) {
    if (act == VCG_ACT_PRELU || act == VCG_ACT_LRELU) return u > 0.f ? 1.f : slope;
    return 1.f;
}

__global__ __launch_bounds__(256) void norm_bwd_partial_bf16_kernel(const bf16x8* __restrict__ x, const bf16x8* __restrict__ dy, int c8,
                                                                     long group_pixels, int slabs_per_group, int per_sample,
                                                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                     int act, float alpha, const float* __restrict__ prelu, int bpg,
                                                                     float* __restrict__ part) {
    extern __shared__ float red[];                                   // [3][256*8]
    const int grp = blockIdx.x / bpg, blk = blockIdx.x - grp * bpg;
    const int ch = threadIdx.x % c8, pl = threadIdx.x / c8, npl = 256 / c8;
    const long gbase = (long)grp * group_pixels * c8;
    const int sidx = (per_sample ? grp * c8 * 8 : 0) + ch * 8;
    float mu[8], is[8], ga[8], be[8], sl[8], s1[8], s2[8], s3[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        mu[j] = mean[sidx + j];
        is[j] = invstd[sidx + j];
        ga[j] = gamma ? gamma[ch * 8 + j] : 1.f;
        be[j] = beta ? beta[ch * 8 + j] : 0.f;
        sl[j] = act == VCG_ACT_PRELU ? prelu[ch * 8 + j] : alpha;
        s1[j] = s2[j] = s3[j] = 0.f;
    }
    if (pl < npl)
        for (int slab = blk; slab < slabs_per_group; slab += bpg) {
            const long p0 = (long)slab * SPB, p1 = p0 + SPB < group_pixels ? p0 + SPB : group_pixels;
            auto one = [&](const bf16x8& xv, const bf16x8& dv) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = ((float)xv[j] - mu[j]) * is[j], u = xh * ga[j] + be[j], d = (float)dv[j];
                    const float dz = d * act_grad_bf16(u, act, sl[j]);
                    s1[j] += dz;
                    s2[j] += dz * xh;
                    s3[j] += d * fminf(u, 0.f);
                }
            };
            // four pixels per trip, their eight loads issued together (one pair in flight per thread: 2.4 TB/s)
            long p = p0 + pl;
            for (; p + 3 * npl < p1; p += 4 * npl) {
                bf16x8 xv[4], dv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    xv[u] = x[gbase + (p + u * npl) * c8 + ch];
                    dv[u] = dy[gbase + (p + u * npl) * c8 + ch];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) one(xv[u], dv[u]);
            }
            for (; p < p1; p += npl) one(x[gbase + p * c8 + ch], dy[gbase + p * c8 + ch]);
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[threadIdx.x * 8 + j] = s1[j];
        red[2048 + threadIdx.x * 8 + j] = s2[j];
        red[4096 + threadIdx.x * 8 + j] = s3[j];
    }
    __syncthreads();
    for (int cc = threadIdx.x; cc < c8 * 8; cc += 256) {
        float t1 = 0.f, t2 = 0.f, t3 = 0.f;
        for (int l = 0; l < npl; ++l) {
            const int o = (l * c8 + (cc >> 3)) * 8 + (cc & 7);
            t1 += red[o];
            t2 += red[2048 + o];
            t3 += red[4096 + o];
        }
        float* pp = part + ((long)blockIdx.x * c8 * 8 + cc) * 3;
        pp[0] = t1;
        pp[1] = t2;
        pp[2] = t3;
    }
}

// group sums [groups*c][2]; block = (group, 64 channels), 16 row groups sum the slabs, fixed-order combine
__global__ __launch_bounds__(1024) void norm_bwd_sums_bf16_kernel(const float* __restrict__ part, int c, int slabs_per_group, int groups,
                                                                   float* __restrict__ sums, float* __restrict__ gsum, float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta, float* __restrict__ dalpha) {
    __shared__ double r1[16][64], r2[16][64], r3[16][64];
    const int cblocks = (c + 63) / 64, grp = blockIdx.x / cblocks, ch = (blockIdx.x - grp * cblocks) * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (ch < c) {
        int b = g;
        for (; b + 48 < slabs_per_group; b += 64) {           // four records in flight per thread, summed in record order
            float v[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* pp = part + (((long)grp * slabs_per_group + b + 16 * u) * c + ch) * 3;
                v[u][0] = pp[0]; v[u][1] = pp[1]; v[u][2] = pp[2];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { s1 += v[u][0]; s2 += v[u][1]; s3 += v[u][2]; }
        }
        for (; b < slabs_per_group; b += 16) {
            const float* pp = part + (((long)grp * slabs_per_group + b) * c + ch) * 3;
            s1 += pp[0];
            s2 += pp[1];
            s3 += pp[2];
        }
    }
    r1[g][threadIdx.x & 63] = s1;
    r2[g][threadIdx.x & 63] = s2;
    r3[g][threadIdx.x & 63] = s3;
    __syncthreads();
    if (g != 0 || ch >= c) return;
#pragma unroll
    for (int i = 1; i < 16; ++i) { s1 += r1[i][threadIdx.x]; s2 += r2[i][threadIdx.x]; s3 += r3[i][threadIdx.x]; }
    sums[((long)grp * c + ch) * 2] = (float)s1;
    sums[((long)grp * c + ch) * 2 + 1] = (float)s2;
    if (groups == 1) {                                    // batch statistics: the group totals ARE the parameter gradients (no second kernel)
        if (dbeta) dbeta[ch] = (float)s1;
        if (dgamma) dgamma[ch] = (float)s2;
        if (dalpha) dalpha[ch] = (float)s3;
        return;
    }
    float* gs = gsum + ((long)grp * c + ch) * 3;          // per-group totals for the parameter gradients
    gs[0] = (float)s1;
    gs[1] = (float)s2;
    gs[2] = (float)s3;
}

// dbeta / dgamma / dalpha [c]: the per-group totals summed over the groups (the images, in instance mode) in order
__global__ void norm_bwd_params_bf16_kernel(const float* __restrict__ gsum, int c, int groups, float* __restrict__ dgamma,
                                            float* __restrict__ dbeta, float* __restrict__ dalpha) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    double g1 = 0.0, g2 = 0.0, g3 = 0.0;
    for (int grp = 0; grp < groups; ++grp) {
        const float* gs = gsum + ((long)grp * c + ch) * 3;
        g1 += gs[0];
        g2 += gs[1];
        g3 += gs[2];
    }
    if (dbeta) dbeta[ch] = (float)g1;
    if (dgamma) dgamma[ch] = (float)g2;
    if (dalpha) dalpha[ch] = (float)g3;
}

__global__ __launch_bounds__(256) void norm_bwd_apply_bf16_kernel(const bf16x8* __restrict__ x, const bf16x8* __restrict__ dy, int c8,
                                                                   long pix_per_img, int blocks_per_img, long group_pixels, int per_sample,
                                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                                                   float alpha, const float* __restrict__ prelu, const float* __restrict__ sums,
                                                                   int use_batch_stats, bf16x8* __restrict__ dx) {
    const int img = blockIdx.x / blocks_per_img, pb = blockIdx.x - img * blocks_per_img;
    const int ch = threadIdx.x % c8, pl = threadIdx.x / c8, npl = 256 / c8;
    if (pl >= npl) return;
    const int pidx = (per_sample ? img * c8 * 8 : 0) + ch * 8;
    const float inv_m = 1.f / (float)group_pixels;
    float mu[8], is[8], ga[8], be[8], sl[8], m1[8], m2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        mu[j] = mean[pidx + j];
        is[j] = invstd[pidx + j];
        ga[j] = gamma ? gamma[ch * 8 + j] : 1.f;
        be[j] = beta ? beta[ch * 8 + j] : 0.f;
        sl[j] = act == VCG_ACT_PRELU ? prelu[ch * 8 + j] : (act == VCG_ACT_LRELU ? alpha : 1.f);
        m1[j] = use_batch_stats ? sums[(pidx + j) * 2] * inv_m : 0.f;
        m2[j] = use_batch_stats ? sums[(pidx + j) * 2 + 1] * inv_m : 0.f;
    }
    const long p0 = (long)pb * APB, p1 = p0 + APB < pix_per_img ? p0 + APB : pix_per_img;
    const long base = (long)img * pix_per_img * c8;
    auto one = [&](const bf16x8& xv, const bf16x8& dv) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = ((float)xv[j] - mu[j]) * is[j], u = xh * ga[j] + be[j];
            const float dz = (float)dv[j] * (u > 0.f ? 1.f : sl[j]) - m1[j] - xh * m2[j];
            o[j] = (__bf16)(ga[j] * is[j] * dz);
        }
        return o;
    };
    long p = p0 + pl;
    for (; p + 3 * npl < p1; p += 4 * npl) {          // four pixels per trip: eight loads in flight per thread
        bf16x8 xv[4], dv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = base + (p + u * npl) * c8 + ch;
            xv[u] = x[i];
            dv[u] = dy[i];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) dx[base + (p + u * npl) * c8 + ch] = one(xv[u], dv[u]);
    }
    for (; p < p1; p += npl) {
        const long i = base + p * c8 + ch;
        dx[i] = one(x[i], dy[i]);
    }
}

// blocks per group of the partial passes: one per slab up to NBLK blocks in all (2 per CU, each walking several slabs)
constexpr int NBLK = 512;
inline int blocks_per_group(int slabs, int groups) {
    int b = NBLK / groups;
    b = b < 1 ? 1 : b;
    return slabs < b ? slabs : b;
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
    if (c % 8 != 0 || c > 2048) return VCG_E_UNSUPPORTED;
    if (ws_bytes < vcg_norm_stats_bf16_workspace_bytes(n, c, hw, mode)) return VCG_E_WORKSPACE;
    const long gp = mode == VCG_NORM_INSTANCE ? hw : (long)n * hw;
    const int groups = mode == VCG_NORM_INSTANCE ? n : 1;
    const int slabs = (int)((gp + SPB - 1) / SPB), bpg = blocks_per_group(slabs, groups);
    stats_partial_bf16_kernel<<<groups * bpg, 256, 2 * 256 * 8 * sizeof(float), stream>>>((const bf16x8*)x, c / 8, gp, slabs, bpg, (float*)ws);
    VCG_LAUNCH_CHECK();
    stats_final_bf16_kernel<<<groups * ceil_div(c, 64), 1024, 0, stream>>>((const __bf16*)x, (const float*)ws, c, gp, bpg, groups, mean, var);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_norm_act_fwd_bf16(const void* x, int n, int c, int hw, const float* scale, const float* shift, int per_sample, int act,
                          float alpha, const float* prelu_alpha, const void* residual, void* y, hipStream_t stream) {
    VCG_CHECK_PTR(x); VCG_CHECK_PTR(scale); VCG_CHECK_PTR(shift); VCG_CHECK_PTR(y);
    if (n <= 0 || c <= 0 || hw <= 0) return VCG_E_SHAPE;
    if (c % 8 != 0 || c > 2048) return VCG_E_UNSUPPORTED;
    if (act == VCG_ACT_PRELU && !prelu_alpha) return VCG_E_NULL;
    if (act == VCG_ACT_TANH) return VCG_E_UNSUPPORTED;
    const int bpi = ceil_div(hw, APB);
    norm_act_fwd_bf16_kernel<<<n * bpi, 256, 0, stream>>>((const bf16x8*)x, c / 8, hw, bpi, scale, shift, per_sample, act, alpha, prelu_alpha,
                                                          (const bf16x8*)residual, (bf16x8*)y);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

size_t vcg_norm_act_bwd_bf16_workspace_bytes(int n, int c, int hw, int mode) {
    const long gp = mode == VCG_NORM_INSTANCE ? hw : (long)n * hw;
    const long groups = mode == VCG_NORM_INSTANCE ? n : 1;
    return (size_t)(groups * ((gp + SPB - 1) / SPB) * c * 3 + groups * c * 5) * sizeof(float);
}

int vcg_norm_act_bwd_bf16(const void* x, const void* dy, int n, int c, int hw, int mode, const float* mean, const float* invstd,
                          const float* gamma, const float* beta, int act, float act_alpha, const float* prelu_alpha, int use_batch_stats,
                          void* dx, float* dgamma, float* dbeta, float* dprelu_alpha, void* ws, size_t ws_bytes, hipStream_t stream) {
    VCG_CHECK_PTR(x); VCG_CHECK_PTR(dy); VCG_CHECK_PTR(mean); VCG_CHECK_PTR(invstd); VCG_CHECK_PTR(dx); VCG_CHECK_PTR(ws);
    if (n <= 0 || c <= 0 || hw <= 0) return VCG_E_SHAPE;
    if (c % 8 != 0 || c > 2048 || act == VCG_ACT_TANH) return VCG_E_UNSUPPORTED;
    if (act == VCG_ACT_PRELU && !prelu_alpha) return VCG_E_NULL;
    if (ws_bytes < vcg_norm_act_bwd_bf16_workspace_bytes(n, c, hw, mode)) return VCG_E_WORKSPACE;
    const int inst = mode == VCG_NORM_INSTANCE;
    const long gp = inst ? hw : (long)n * hw;
    const int groups = inst ? n : 1;
    const int slabs = (int)((gp + SPB - 1) / SPB), bpg = blocks_per_group(slabs, groups);
    float* part = (float*)ws;
    float* sums = part + (size_t)groups * slabs * c * 3;
    norm_bwd_partial_bf16_kernel<<<groups * bpg, 256, 3 * 2048 * sizeof(float), stream>>>((const bf16x8*)x, (const bf16x8*)dy, c / 8, gp, slabs, inst,
                                                                                           mean, invstd, gamma, beta, act, act_alpha, prelu_alpha, bpg, part);
    VCG_LAUNCH_CHECK();
    float* gsum = sums + (size_t)groups * c * 2;
    norm_bwd_sums_bf16_kernel<<<groups * ceil_div(c, 64), 1024, 0, stream>>>(part, c, bpg, groups, sums, gsum, dgamma, dbeta, dprelu_alpha);
    VCG_LAUNCH_CHECK();
    if (groups > 1 && (dgamma || dbeta || dprelu_alpha)) {
        norm_bwd_params_bf16_kernel<<<ceil_div(c, 64), 64, 0, stream>>>(gsum, c, groups, dgamma, dbeta, dprelu_alpha);
        VCG_LAUNCH_CHECK();
    }
    const int bpi = ceil_div(hw, APB);
    norm_bwd_apply_bf16_kernel<<<n * bpi, 256, 0, stream>>>((const bf16x8*)x, (const bf16x8*)dy, c / 8, hw, bpi, gp, inst, mean, invstd, gamma, beta,
                                                            act, act_alpha, prelu_alpha, sums, use_batch_stats, (bf16x8*)dx);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

}  // extern "C"
