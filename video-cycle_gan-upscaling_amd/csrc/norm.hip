// BatchNormalization / instance-norm statistics, fused normalise+activation(+residual) and their
// backward (reference call sites: model.py:20-25,284-285,840-841,877-882).  All HBM-bound passes:
// float4 loads where the row length allows, wave-shuffle reductions, two-stage deterministic sums.
//
// A "group" is what shares statistics: channel c over (n,hw) in batch mode (G = c groups,
// M = n*hw elements) or (n,c) over hw in instance mode (G = n*c, M = hw).
#include "vcg_common.hpp"

namespace {

constexpr int kMaxSplit = 64;

// number of slices each group is cut into so that the grid has >= ~1024 blocks
inline int pick_split(int groups, size_t elems_per_group) {
    int s = 1;
    while (s < kMaxSplit && (size_t)groups * s < 1024 && elems_per_group / (s * 2) >= 2048) s *= 2;
    return s;
}

// element (group g, linear i in [0,M)) -> offset in x[n][c][hw]
struct GroupMap {
    int n, c, hw, mode;
    __device__ __forceinline__ size_t offset(int g, size_t i) const {
        if (mode == VCG_NORM_INSTANCE) return (size_t)g * hw + i;        // g = n*c + ch
        const size_t nn = i / hw, r = i - nn * hw;                        // g = ch
        return (nn * c + g) * hw + r;
    }
};

// visit the offsets of elements [beg,end) of group g, contiguous runs per (n) plane, 256 threads
template <typename F>
__device__ __forceinline__ void for_each_in_slice(const GroupMap& gm, int g, size_t beg, size_t end, F f) {
    if (beg >= end) return;
    if (gm.mode == VCG_NORM_INSTANCE) {
        const size_t base = (size_t)g * gm.hw;
        for (size_t i = beg + threadIdx.x; i < end; i += 256) f(base + i);
        return;
    }
    const size_t hw = (size_t)gm.hw;
    const size_t n0 = beg / hw, n1 = (end - 1) / hw;
    for (size_t nn = n0; nn <= n1; ++nn) {
        const size_t lo = (beg > nn * hw ? beg : nn * hw) - nn * hw;
        const size_t hi = (end < (nn + 1) * hw ? end : (nn + 1) * hw) - nn * hw;
        const size_t base = (nn * gm.c + g) * hw;
        for (size_t r = lo + threadIdx.x; r < hi; r += 256) f(base + r);
    }
}

// float4 variant: requires hw % 4 == 0 and slice bounds that are multiples of 4 (the host rounds `per` up to 4);
// f(offset of 4 consecutive elements)
template <typename F>
__device__ __forceinline__ void for_each_in_slice4(const GroupMap& gm, int g, size_t beg, size_t end, F f) {
    if (beg >= end) return;
    if (gm.mode == VCG_NORM_INSTANCE) {
        const size_t base = (size_t)g * gm.hw;
        for (size_t i = beg + 4 * threadIdx.x; i < end; i += 1024) f(base + i);
        return;
    }
    const size_t hw = (size_t)gm.hw;
    const size_t n0 = beg / hw, n1 = (end - 1) / hw;
    for (size_t nn = n0; nn <= n1; ++nn) {
        const size_t lo = (beg > nn * hw ? beg : nn * hw) - nn * hw;
        const size_t hi = (end < (nn + 1) * hw ? end : (nn + 1) * hw) - nn * hw;
        const size_t base = (nn * gm.c + g) * hw;
        for (size_t r = lo + 4 * threadIdx.x; r < hi; r += 1024) f(base + r);
    }
}

// slice length: ceil(M / split), rounded up to a multiple of 4 when the float4 path is taken
__device__ __forceinline__ size_t slice_len(size_t M, int split, bool vec) {
    size_t per = (M + split - 1) / split;
    if (vec) per = (per + 3) & ~(size_t)3;
    return per;
}

// ---- statistics: shifted sums (shift = first element of the group) to avoid cancellation ---------
template <bool VEC>
__global__ __launch_bounds__(256) void stats_partial_kernel(const float* x, GroupMap gm, size_t M, int split,
                                                            float* part /* [G][split][2] */) {
    __shared__ float red[8];
    const int g = blockIdx.x / split, s = blockIdx.x % split;
    const float shift = x[gm.offset(g, 0)];
    const size_t per = slice_len(M, split, VEC);
    const size_t beg = (size_t)s * per, end = beg + per < M ? beg + per : M;
    float v[2] = {0.f, 0.f};
    if (VEC) {
        float a0 = 0.f, a1 = 0.f, q0 = 0.f, q1 = 0.f;      // two chains per sum
        for_each_in_slice4(gm, g, beg, end, [&](size_t o) {
            const float4 t = *reinterpret_cast<const float4*>(x + o);
            const float d0 = t.x - shift, d1 = t.y - shift, d2 = t.z - shift, d3 = t.w - shift;
            a0 += d0 + d1; a1 += d2 + d3;
            q0 += d0 * d0 + d1 * d1; q1 += d2 * d2 + d3 * d3;
        });
        v[0] = a0 + a1; v[1] = q0 + q1;
    } else
    for_each_in_slice(gm, g, beg, end, [&](size_t o) {
        const float d = x[o] - shift;
        v[0] += d;
        v[1] += d * d;
    });
    block_sum<2>(v, red);
    if (threadIdx.x == 0) {
        part[((size_t)g * split + s) * 2 + 0] = v[0];
        part[((size_t)g * split + s) * 2 + 1] = v[1];
    }
}

__global__ void stats_final_kernel(const float* x, GroupMap gm, size_t M, int split, int groups,
                                   const float* part, float* mean, float* var) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= groups) return;
    const float shift = x[gm.offset(g, 0)];
    float s0 = 0.f, s1 = 0.f;
    for (int s = 0; s < split; ++s) {
        s0 += part[((size_t)g * split + s) * 2 + 0];
        s1 += part[((size_t)g * split + s) * 2 + 1];
    }
    const float inv = 1.f / (float)M;
    const float dm = s0 * inv;
    mean[g] = shift + dm;
    const float vv = s1 * inv - dm * dm;
    var[g] = vv > 0.f ? vv : 0.f;
}

__global__ void finalize_kernel(const float* mean, const float* var, const float* gamma, const float* beta,
                                int c, int rows, float eps, float* scale, float* shift, float* invstd,
                                float* mmean, float* mvar, float momentum, int unbiased_count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c * rows) return;
    const int ch = i % c;
    const float is = rsqrtf(var[i] + eps);
    const float g = gamma ? gamma[ch] : 1.f, bt = beta ? beta[ch] : 0.f;
    const float sc = g * is;
    scale[i] = sc;
    shift[i] = bt - mean[i] * sc;
    if (invstd) invstd[i] = is;
    if (mmean && rows == 1) {
        mmean[i] = mmean[i] * momentum + mean[i] * (1.f - momentum);
        float v = var[i];
        if (unbiased_count > 1) v *= (float)unbiased_count / (float)(unbiased_count - 1);
        mvar[i] = mvar[i] * momentum + v * (1.f - momentum);
    }
}

// out[ch] = scale * sum over records of part[rec][ch], in double and in a fixed order (block = 64 channels x 16 record lanes)
__global__ __launch_bounds__(1024) void sum_records_kernel(const float* __restrict__ part, int nrec, int c, float scale, float* __restrict__ out) {
    __shared__ double rs[16][64];
    const int ch = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    double s = 0.0;
    if (ch < c) {
        int b = g;
        for (; b + 16 * 7 < nrec; b += 16 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(b + 16 * u) * c + ch];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < nrec; b += 16) s += part[(size_t)b * c + ch];
    }
    rs[g][threadIdx.x & 63] = s;
    __syncthreads();
    if (g != 0 || ch >= c) return;
#pragma unroll
    for (int i = 1; i < 16; ++i) s += rs[i][threadIdx.x];
    out[ch] = (float)(s * (double)scale);
}

// inference-mode BatchNormalization behind a convolution as the convolution's epilogue: scale = gamma * rsqrt(moving_var + eps),
// shift = (bias - moving_mean) * scale + beta  (model.py:19-20 in learning phase 0)
__global__ void bn_fold_kernel(const float* bias, const float* mmean, const float* mvar, const float* gamma, const float* beta, int c, float eps,
                               float* scale, float* shift) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    const float sc = (gamma ? gamma[i] : 1.f) * rsqrtf(mvar[i] + eps);
    scale[i] = sc;
    shift[i] = ((bias ? bias[i] : 0.f) - mmean[i]) * sc + (beta ? beta[i] : 0.f);
}

// the same for every (convolution, BatchNormalization) pair of a model in one launch: blockIdx.y = pair, scale / shift [pair][c]
constexpr int FOLD_BATCH_MAX = 48;
struct FoldBatch {
    const float* bias[FOLD_BATCH_MAX];
    const float* mmean[FOLD_BATCH_MAX];
    const float* mvar[FOLD_BATCH_MAX];
    const float* gamma[FOLD_BATCH_MAX];
    const float* beta[FOLD_BATCH_MAX];
};
__global__ void bn_fold_batch_kernel(FoldBatch fb, int c, float eps, float* scale, float* shift) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, l = blockIdx.y;
    if (i >= c) return;
    const float sc = (fb.gamma[l] ? fb.gamma[l][i] : 1.f) * rsqrtf(fb.mvar[l][i] + eps);
    scale[l * c + i] = sc;
    shift[l * c + i] = ((fb.bias[l] ? fb.bias[l][i] : 0.f) - fb.mmean[l][i]) * sc + (fb.beta[l] ? fb.beta[l][i] : 0.f);
}

// mean / biased variance from the partial records a convolution's epilogue wrote (part[group][nrec][2][c]: sums, sums of squares of the
// stored values), then what finalize_kernel does -- one launch instead of statistics partial + final + finalize.
// block = (group, 64 channels) x 16 record lanes; a thread sums its records (eight in flight) in double, the 16 lanes combine in a
// fixed order.  E[x^2] - E[x]^2 is formed in double from fp32 partial sums of <= a few thousand values each: its relative error is
// ~1e-7 (1 + mean^2/var), fine for the layers behind a convolution (|mean| of the order of the standard deviation).
__global__ __launch_bounds__(1024) void finalize_partials_kernel(const float* __restrict__ part, int nrec, int c, double inv_count,
                                                                  const float* gamma, const float* beta, float eps, float* mean_out,
                                                                  float* scale, float* shift, float* invstd, float* mmean, float* mvar,
                                                                  float momentum, int unbiased_count, int groups, const float* kshift) {
    __shared__ double rs[16][64], rq[16][64];
    const int cblocks = (c + 63) / 64, grp = blockIdx.x / cblocks, ch = (blockIdx.x - grp * cblocks) * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    double s = 0.0, q = 0.0;
    if (ch < c) {
        const float* pg = part + (size_t)grp * nrec * 2 * c + ch;
        int b = g;
        for (; b + 16 * 7 < nrec; b += 16 * 8) {
            float vs[8], vq[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                vs[u] = pg[(size_t)(b + 16 * u) * 2 * c];
                vq[u] = pg[((size_t)(b + 16 * u) * 2 + 1) * c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s += vs[u]; q += vq[u]; }
        }
        for (; b < nrec; b += 16) {
            s += pg[(size_t)b * 2 * c];
            q += pg[((size_t)b * 2 + 1) * c];
        }
    }
    rs[g][threadIdx.x & 63] = s;
    rq[g][threadIdx.x & 63] = q;
    __syncthreads();
    if (g != 0 || ch >= c) return;
#pragma unroll
    for (int i = 1; i < 16; ++i) { s += rs[i][threadIdx.x]; q += rq[i][threadIdx.x]; }
    // kshift: the records hold sums of (x - kshift[ch]) and of its square (the fp32 convolution's epilogue sums its accumulators, i.e. the
    // output minus the bias): the variance is that of the shifted values, the mean gets the shift back
    const double ms = s * inv_count, vd = q * inv_count - ms * ms, m = ms + (kshift ? (double)kshift[ch] : 0.0);
    const int i = grp * c + ch;
    const float mean = (float)m, var = vd > 0.0 ? (float)vd : 0.f;
    const float is = rsqrtf(var + eps);
    const float ga = gamma ? gamma[ch] : 1.f, bt = beta ? beta[ch] : 0.f;
    const float sc = ga * is;
    mean_out[i] = mean;
    scale[i] = sc;
    shift[i] = bt - mean * sc;
    if (invstd) invstd[i] = is;
    if (mmean && groups == 1) {
        mmean[i] = mmean[i] * momentum + mean * (1.f - momentum);
        float v = var;
        if (unbiased_count > 1) v *= (float)unbiased_count / (float)(unbiased_count - 1);
        mvar[i] = mvar[i] * momentum + v * (1.f - momentum);
    }
}

// ---- y = act(x*scale + shift) + residual -----------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void norm_act_fwd_kernel(const float* x, int n, int c, int hw,
                                                           const float* scale, const float* shift, int per_sample,
                                                           int act, float act_alpha, const float* prelu,
                                                           const float* residual, float* y) {
    // grid.y = n*c planes, grid.x covers hw
    const int plane = blockIdx.y;
    const int ch = plane % c;
    const int si = per_sample ? plane : ch;
    const float sc = scale ? scale[si] : 1.f, sh = shift ? shift[si] : 0.f;
    const float al = (act == VCG_ACT_PRELU) ? prelu[ch] : act_alpha;
    const size_t base = (size_t)plane * hw;
    if (VEC) {
        const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
        if (i >= hw) return;
        float4 v = *reinterpret_cast<const float4*>(x + base + i);
        v.x = apply_act(v.x * sc + sh, act, al);
        v.y = apply_act(v.y * sc + sh, act, al);
        v.z = apply_act(v.z * sc + sh, act, al);
        v.w = apply_act(v.w * sc + sh, act, al);
        if (residual) {
            const float4 r = *reinterpret_cast<const float4*>(residual + base + i);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        *reinterpret_cast<float4*>(y + base + i) = v;
    } else {
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i >= hw) return;
        float v = apply_act(x[base + i] * sc + sh, act, al);
        if (residual) v += residual[base + i];
        y[base + i] = v;
    }
}

__global__ __launch_bounds__(256) void norm_act_fwd_flat_kernel(const float* x, size_t total, int c, int hw,
                                                                const float* scale, const float* shift, int per_sample,
                                                                int act, float act_alpha, const float* prelu,
                                                                const float* residual, float* y) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t plane = i / hw;
    const int ch = (int)(plane % c);
    const size_t si = per_sample ? plane : (size_t)ch;
    const float sc = scale ? scale[si] : 1.f, sh = shift ? shift[si] : 0.f;
    const float al = (act == VCG_ACT_PRELU) ? prelu[ch] : act_alpha;
    float v = apply_act(x[i] * sc + sh, act, al);
    if (residual) v += residual[i];
    y[i] = v;
}

// ---- backward -------------------------------------------------------------------------------------
__device__ __forceinline__ float act_grad(float z, int act, float al) {
    if (act == VCG_ACT_LRELU) return z > 0.f ? 1.f : al;
    if (act == VCG_ACT_PRELU) return z > 0.f ? 1.f : al;
    return 1.f;
}

// pass 1: per group slice: sum dz, sum dz*xhat, sum dy*min(z,0)
template <bool VEC>
__global__ __launch_bounds__(256) void norm_bwd_partial_kernel(const float* x, const float* dy, GroupMap gm, size_t M,
                                                               int split, const float* mean, const float* invstd,
                                                               const float* gamma, const float* beta, int act,
                                                               float act_alpha, const float* prelu,
                                                               float* part /* [G][split][3] */) {
    __shared__ float red[12];
    const int g = blockIdx.x / split, s = blockIdx.x % split;
    const int ch = (gm.mode == VCG_NORM_INSTANCE) ? g % gm.c : g;
    const float mu = mean[g], is = invstd[g];
    const float ga = gamma ? gamma[ch] : 1.f, be = beta ? beta[ch] : 0.f;
    const float al = (act == VCG_ACT_PRELU) ? prelu[ch] : act_alpha;
    const size_t per = slice_len(M, split, VEC);
    const size_t beg = (size_t)s * per, end = beg + per < M ? beg + per : M;
    float v[3] = {0.f, 0.f, 0.f};
    auto one = [&](float xv, float d) {
        const float xh = (xv - mu) * is;
        const float z = ga * xh + be;
        const float dz = d * act_grad(z, act, al);
        v[0] += dz;
        v[1] += dz * xh;
        v[2] += d * fminf(z, 0.f);
    };
    if (VEC)
        for_each_in_slice4(gm, g, beg, end, [&](size_t o) {
            const float4 a = *reinterpret_cast<const float4*>(x + o);
            const float4 d = *reinterpret_cast<const float4*>(dy + o);
            one(a.x, d.x); one(a.y, d.y); one(a.z, d.z); one(a.w, d.w);
        });
    else
        for_each_in_slice(gm, g, beg, end, [&](size_t o) { one(x[o], dy[o]); });
    block_sum<3>(v, red);
    if (threadIdx.x == 0) {
        float* o = part + ((size_t)g * split + s) * 3;
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
    }
}

// combine slices -> sums[G][3]; channel-level dgamma/dbeta/dalpha summed over n for instance mode
__global__ void norm_bwd_final_kernel(const float* part, int groups, int split, float* sums /* [G][3] */) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= groups) return;
    float a = 0.f, b = 0.f, c = 0.f;
    for (int s = 0; s < split; ++s) {
        const float* o = part + ((size_t)g * split + s) * 3;
        a += o[0]; b += o[1]; c += o[2];
    }
    sums[g * 3 + 0] = a; sums[g * 3 + 1] = b; sums[g * 3 + 2] = c;
}

__global__ void norm_bwd_params_kernel(const float* sums, int n, int c, int mode, float* dgamma, float* dbeta,
                                       float* dalpha) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    float a = 0.f, b = 0.f, al = 0.f;
    if (mode == VCG_NORM_INSTANCE) {
        for (int nn = 0; nn < n; ++nn) {
            const float* o = sums + ((size_t)nn * c + ch) * 3;
            a += o[0]; b += o[1]; al += o[2];
        }
    } else {
        a = sums[ch * 3 + 0]; b = sums[ch * 3 + 1]; al = sums[ch * 3 + 2];
    }
    if (dbeta) dbeta[ch] = a;
    if (dgamma) dgamma[ch] = b;
    if (dalpha) dalpha[ch] = al;
}

// pass 2: dx = gamma*invstd*(dz - sum_dz/M - xhat*sum_dz_xhat/M)
template <bool VEC>
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const float* x, const float* dy, int n, int c, int hw,
                                                             int mode, const float* mean, const float* invstd,
                                                             const float* gamma, const float* beta, int act,
                                                             float act_alpha, const float* prelu, const float* sums,
                                                             float invM, int use_batch_stats, float* dx) {
    // gridDim.y is capped at 65535: the block walks the remaining planes (Dense BN: c = 1024, n >= 64)
    for (int plane = blockIdx.y; plane < n * c; plane += gridDim.y) {
        const int ch = plane % c;
        const int g = (mode == VCG_NORM_INSTANCE) ? plane : ch;
        const float mu = mean[g], is = invstd[g];
        const float ga = gamma ? gamma[ch] : 1.f, be = beta ? beta[ch] : 0.f;
        const float al = (act == VCG_ACT_PRELU) ? prelu[ch] : act_alpha;
        const float s0 = use_batch_stats ? sums[g * 3 + 0] * invM : 0.f;
        const float s1 = use_batch_stats ? sums[g * 3 + 1] * invM : 0.f;
        const size_t base = (size_t)plane * hw;
        const float gi = ga * is;
        auto one = [&](float xv, float d) {
            const float xh = (xv - mu) * is;
            const float z = ga * xh + be;
            const float dz = d * act_grad(z, act, al);
            return gi * (dz - s0 - xh * s1);
        };
        if (VEC) {
            for (int i = (blockIdx.x * 256 + threadIdx.x) * 4; i < hw; i += gridDim.x * 1024) {
                const float4 a = *reinterpret_cast<const float4*>(x + base + i);
                const float4 d = *reinterpret_cast<const float4*>(dy + base + i);
                float4 o;
                o.x = one(a.x, d.x); o.y = one(a.y, d.y); o.z = one(a.z, d.z); o.w = one(a.w, d.w);
                *reinterpret_cast<float4*>(dx + base + i) = o;
            }
        } else {
            for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) dx[base + i] = one(x[base + i], dy[base + i]);
        }
    }
}

}  // namespace

extern "C" {

size_t vcg_norm_stats_workspace_bytes(int n, int c, int hw, int mode) {
    const int groups = (mode == VCG_NORM_INSTANCE) ? n * c : c;
    return (size_t)groups * kMaxSplit * 2 * sizeof(float);
}

int vcg_norm_stats(const float* x, int n, int c, int hw, int mode, float* mean, float* var, void* ws,
                   size_t ws_bytes, vcg_stream_t stream) {
    VCG_CHECK_PTR(x); VCG_CHECK_PTR(mean); VCG_CHECK_PTR(var); VCG_CHECK_PTR(ws);
    if (n <= 0 || c <= 0 || hw <= 0) return VCG_E_SHAPE;
    if (ws_bytes < vcg_norm_stats_workspace_bytes(n, c, hw, mode)) return VCG_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int groups = (mode == VCG_NORM_INSTANCE) ? n * c : c;
    const size_t M = (mode == VCG_NORM_INSTANCE) ? (size_t)hw : (size_t)n * hw;
    const int split = pick_split(groups, M);
    GroupMap gm{n, c, hw, mode};
    if (hw % 4 == 0)
        hipLaunchKernelGGL(stats_partial_kernel<true>, dim3(groups * split), dim3(256), 0, st, x, gm, M, split, (float*)ws);
    else
        hipLaunchKernelGGL(stats_partial_kernel<false>, dim3(groups * split), dim3(256), 0, st, x, gm, M, split, (float*)ws);
    VCG_LAUNCH_CHECK();
    hipLaunchKernelGGL(stats_final_kernel, dim3(ceil_div(groups, 256)), dim3(256), 0, st, x, gm, M, split, groups,
                       (const float*)ws, mean, var);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_norm_finalize(const float* mean, const float* var, const float* gamma, const float* beta, int c, int rows,
                      float eps, float* scale, float* shift, float* invstd, float* moving_mean, float* moving_var,
                      float momentum, int unbiased_count, vcg_stream_t stream) {
    VCG_CHECK_PTR(mean); VCG_CHECK_PTR(var); VCG_CHECK_PTR(scale); VCG_CHECK_PTR(shift);
    if (c <= 0 || rows <= 0) return VCG_E_SHAPE;
    if ((moving_mean == nullptr) != (moving_var == nullptr)) return VCG_E_NULL;
    hipLaunchKernelGGL(finalize_kernel, dim3(ceil_div(c * rows, 256)), dim3(256), 0, (hipStream_t)stream, mean, var,
                       gamma, beta, c, rows, eps, scale, shift, invstd, moving_mean, moving_var, momentum,
                       unbiased_count);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_sum_records(const float* part, int nrec, int c, float scale, float* out, vcg_stream_t stream) {
    VCG_CHECK_PTR(part); VCG_CHECK_PTR(out);
    if (nrec <= 0 || c <= 0) return VCG_E_SHAPE;
    hipLaunchKernelGGL(sum_records_kernel, dim3(ceil_div(c, 64)), dim3(1024), 0, (hipStream_t)stream, part, nrec, c, scale, out);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_bn_fold(const float* bias, const float* moving_mean, const float* moving_var, const float* gamma, const float* beta, int c, float eps,
                float* scale, float* shift, vcg_stream_t stream) {
    VCG_CHECK_PTR(moving_mean); VCG_CHECK_PTR(moving_var); VCG_CHECK_PTR(scale); VCG_CHECK_PTR(shift);
    if (c <= 0) return VCG_E_SHAPE;
    hipLaunchKernelGGL(bn_fold_kernel, dim3(ceil_div(c, 256)), dim3(256), 0, (hipStream_t)stream, bias, moving_mean, moving_var, gamma, beta, c, eps,
                       scale, shift);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_bn_fold_batch(const float* const* bias, const float* const* moving_mean, const float* const* moving_var, const float* const* gamma,
                      const float* const* beta, int count, int c, float eps, float* scale, float* shift, vcg_stream_t stream) {
    VCG_CHECK_PTR(bias); VCG_CHECK_PTR(moving_mean); VCG_CHECK_PTR(moving_var); VCG_CHECK_PTR(gamma); VCG_CHECK_PTR(beta);
    VCG_CHECK_PTR(scale); VCG_CHECK_PTR(shift);
    if (c <= 0 || count <= 0 || count > FOLD_BATCH_MAX) return VCG_E_SHAPE;
    FoldBatch fb;
    for (int i = 0; i < count; ++i) {
        VCG_CHECK_PTR(moving_mean[i]); VCG_CHECK_PTR(moving_var[i]);
        fb.bias[i] = bias[i]; fb.mmean[i] = moving_mean[i]; fb.mvar[i] = moving_var[i]; fb.gamma[i] = gamma[i]; fb.beta[i] = beta[i];
    }
    hipLaunchKernelGGL(bn_fold_batch_kernel, dim3(ceil_div(c, 256), count), dim3(256), 0, (hipStream_t)stream, fb, c, eps, scale, shift);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_norm_finalize_partials(const float* part, int nrec, int groups, int c, double count, const float* gamma, const float* beta,
                               float eps, float* mean, float* scale, float* shift, float* invstd, float* moving_mean,
                               float* moving_var, float momentum, int unbiased_count, vcg_stream_t stream) {
    VCG_CHECK_PTR(part); VCG_CHECK_PTR(mean); VCG_CHECK_PTR(scale); VCG_CHECK_PTR(shift);
    if (c <= 0 || groups <= 0 || nrec <= 0 || !(count > 0.0)) return VCG_E_SHAPE;
    if ((moving_mean == nullptr) != (moving_var == nullptr)) return VCG_E_NULL;
    hipLaunchKernelGGL(finalize_partials_kernel, dim3(groups * ceil_div(c, 64)), dim3(1024), 0, (hipStream_t)stream, part, nrec, c,
                       1.0 / count, gamma, beta, eps, mean, scale, shift, invstd, moving_mean, moving_var, momentum, unbiased_count, groups, nullptr);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_norm_finalize_partials_shifted(const float* part, int nrec, int groups, int c, double count, const float* kshift, const float* gamma,
                                       const float* beta, float eps, float* mean, float* scale, float* shift, float* invstd, float* moving_mean,
                                       float* moving_var, float momentum, int unbiased_count, vcg_stream_t stream) {
    VCG_CHECK_PTR(part); VCG_CHECK_PTR(mean); VCG_CHECK_PTR(scale); VCG_CHECK_PTR(shift);
    if (c <= 0 || groups <= 0 || nrec <= 0 || !(count > 0.0)) return VCG_E_SHAPE;
    if ((moving_mean == nullptr) != (moving_var == nullptr)) return VCG_E_NULL;
    hipLaunchKernelGGL(finalize_partials_kernel, dim3(groups * ceil_div(c, 64)), dim3(1024), 0, (hipStream_t)stream, part, nrec, c,
                       1.0 / count, gamma, beta, eps, mean, scale, shift, invstd, moving_mean, moving_var, momentum, unbiased_count, groups, kshift);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_norm_act_fwd(const float* x, int n, int c, int hw, const float* scale, const float* shift, int per_sample,
                     int act, float act_alpha, const float* prelu_alpha, const float* residual, float* y,
                     vcg_stream_t stream) {
    VCG_CHECK_PTR(x); VCG_CHECK_PTR(y);
    if (n <= 0 || c <= 0 || hw <= 0) return VCG_E_SHAPE;
    if (act == VCG_ACT_PRELU && prelu_alpha == nullptr) return VCG_E_NULL;
    if ((scale == nullptr) != (shift == nullptr)) return VCG_E_NULL;
    hipStream_t st = (hipStream_t)stream;
    const long planes = (long)n * c;
    if (hw < 64 || planes > 65535) {
        // many tiny planes (Dense BN: hw == 1): one thread per element
        const size_t total = (size_t)planes * hw;
        hipLaunchKernelGGL(norm_act_fwd_flat_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, total,
                           c, hw, scale, shift, per_sample, act, act_alpha, prelu_alpha, residual, y);
        VCG_LAUNCH_CHECK();
        return VCG_OK;
    }
    const bool vec = (hw % 4) == 0;
    if (vec) {
        hipLaunchKernelGGL(norm_act_fwd_kernel<true>, dim3(ceil_div(hw / 4, 256), (unsigned)planes), dim3(256), 0, st, x,
                           n, c, hw, scale, shift, per_sample, act, act_alpha, prelu_alpha, residual, y);
    } else {
        hipLaunchKernelGGL(norm_act_fwd_kernel<false>, dim3(ceil_div(hw, 256), (unsigned)planes), dim3(256), 0, st, x,
                           n, c, hw, scale, shift, per_sample, act, act_alpha, prelu_alpha, residual, y);
    }
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

size_t vcg_norm_act_bwd_workspace_bytes(int n, int c, int hw, int mode) {
    const int groups = (mode == VCG_NORM_INSTANCE) ? n * c : c;
    return (size_t)groups * (kMaxSplit + 1) * 3 * sizeof(float);
}

int vcg_norm_act_bwd(const float* x, const float* dy, int n, int c, int hw, int mode, const float* mean,
                     const float* invstd, const float* gamma, const float* beta, int act, float act_alpha,
                     const float* prelu_alpha, int use_batch_stats, float* dx, float* dgamma, float* dbeta,
                     float* dprelu_alpha, void* ws, size_t ws_bytes, vcg_stream_t stream) {
    VCG_CHECK_PTR(x); VCG_CHECK_PTR(dy); VCG_CHECK_PTR(mean); VCG_CHECK_PTR(invstd); VCG_CHECK_PTR(dx);
    VCG_CHECK_PTR(ws);
    if (n <= 0 || c <= 0 || hw <= 0) return VCG_E_SHAPE;
    if (act == VCG_ACT_PRELU && prelu_alpha == nullptr) return VCG_E_NULL;
    if (ws_bytes < vcg_norm_act_bwd_workspace_bytes(n, c, hw, mode)) return VCG_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int groups = (mode == VCG_NORM_INSTANCE) ? n * c : c;
    const size_t M = (mode == VCG_NORM_INSTANCE) ? (size_t)hw : (size_t)n * hw;
    const int split = pick_split(groups, M);
    GroupMap gm{n, c, hw, mode};
    float* part = (float*)ws;
    float* sums = part + (size_t)groups * kMaxSplit * 3;
    const bool vec = (hw % 4) == 0;
    if (vec)
        hipLaunchKernelGGL(norm_bwd_partial_kernel<true>, dim3(groups * split), dim3(256), 0, st, x, dy, gm, M, split, mean,
                           invstd, gamma, beta, act, act_alpha, prelu_alpha, part);
    else
        hipLaunchKernelGGL(norm_bwd_partial_kernel<false>, dim3(groups * split), dim3(256), 0, st, x, dy, gm, M, split, mean,
                           invstd, gamma, beta, act, act_alpha, prelu_alpha, part);
    VCG_LAUNCH_CHECK();
    hipLaunchKernelGGL(norm_bwd_final_kernel, dim3(ceil_div(groups, 256)), dim3(256), 0, st, (const float*)part, groups,
                       split, sums);
    VCG_LAUNCH_CHECK();
    if (dgamma || dbeta || dprelu_alpha) {
        hipLaunchKernelGGL(norm_bwd_params_kernel, dim3(ceil_div(c, 256)), dim3(256), 0, st, (const float*)sums, n, c,
                           mode, dgamma, dbeta, dprelu_alpha);
        VCG_LAUNCH_CHECK();
    }
    int gx = ceil_div(hw, vec ? 1024 : 256);
    if (gx > 64) gx = 64;
    const dim3 agrid(gx, (unsigned)((long)n * c > 65535 ? 65535 : n * c));
    if (vec)
        hipLaunchKernelGGL(norm_bwd_apply_kernel<true>, agrid, dim3(256), 0, st, x, dy, n, c, hw, mode, mean, invstd, gamma, beta, act,
                           act_alpha, prelu_alpha, (const float*)sums, 1.f / (float)M, use_batch_stats, dx);
    else
        hipLaunchKernelGGL(norm_bwd_apply_kernel<false>, agrid, dim3(256), 0, st, x, dy, n, c, hw, mode, mean, invstd, gamma, beta, act,
                           act_alpha, prelu_alpha, (const float*)sums, 1.f / (float)M, use_batch_stats, dx);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

}  // extern "C"
