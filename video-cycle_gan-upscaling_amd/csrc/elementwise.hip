// HBM-bound helper kernels of the train step: activation backward, per-channel sums (bias
// gradients), loss reductions, Keras-form Adam, frame-edge layout/value conversion, per-tap kernel
// transposes.  Reference call sites are quoted in include/vcg.h next to each entry point.
#include "vcg_common.hpp"

namespace {

constexpr int kSumSplitMax = 64;

inline int pick_split_cs(int c, size_t per_channel) {
    int s = 1;
    while (s < kSumSplitMax && (size_t)c * s < 1024 && per_channel / (s * 2) >= 2048) s *= 2;
    return s;
}

// ---- activation backward (+ optional PReLU slope gradient and per-channel sum(dx) partials) -----------
__device__ __forceinline__ float act_bwd_one(float s, float d, int act, float al, float& dal) {
    float g;
    if (act == VCG_ACT_TANH) g = 1.f - s * s;                 // saved = output
    else if (act == VCG_ACT_LRELU) g = s > 0.f ? 1.f : al;    // saved = output; '>' as TF's LeakyReluGrad: exact for slope 0 (ReLU) too
    else if (act == VCG_ACT_PRELU) { g = s > 0.f ? 1.f : al; dal += d * fminf(s, 0.f); }  // saved = input
    else g = 1.f;
    return d * g;
}

template <bool VEC>
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* saved, const float* dy, int n, int c, int hw,
                                                      int act, float act_alpha, const float* prelu, float* dx,
                                                      float* part_alpha, float* part_sum /* [n*c][gridDim.x] or null */) {
    __shared__ float red[8];
    // planes beyond the 65535 limit of gridDim.y are walked by the same block (Dense BatchNormalization with
    // 1024 channels and >= 64 frames: n*c > 65535)
    for (int plane = blockIdx.y; plane < n * c; plane += gridDim.y) {
    const int ch = plane % c;
    const float al = (act == VCG_ACT_PRELU) ? prelu[ch] : act_alpha;
    const size_t base = (size_t)plane * hw;
    float v[2] = {0.f, 0.f};   // dalpha partial, sum(dx) partial
    if (VEC) {
        for (int i = (blockIdx.x * 256 + threadIdx.x) * 4; i < hw; i += gridDim.x * 1024) {
            const float4 s = *reinterpret_cast<const float4*>(saved + base + i);
            const float4 d = *reinterpret_cast<const float4*>(dy + base + i);
            float4 o;
            o.x = act_bwd_one(s.x, d.x, act, al, v[0]);
            o.y = act_bwd_one(s.y, d.y, act, al, v[0]);
            o.z = act_bwd_one(s.z, d.z, act, al, v[0]);
            o.w = act_bwd_one(s.w, d.w, act, al, v[0]);
            v[1] += (o.x + o.y) + (o.z + o.w);
            *reinterpret_cast<float4*>(dx + base + i) = o;
        }
    } else {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) {
            const float o = act_bwd_one(saved[base + i], dy[base + i], act, al, v[0]);
            v[1] += o;
            dx[base + i] = o;
        }
    }
    if (part_alpha || part_sum) {
        block_sum<2>(v, red);
        if (threadIdx.x == 0) {
            if (part_alpha) part_alpha[(size_t)plane * gridDim.x + blockIdx.x] = v[0];
            if (part_sum) part_sum[(size_t)plane * gridDim.x + blockIdx.x] = v[1];
        }
    }
    }
}

// out[ch] = sum_{n, k} part[(n*c+ch)*per + k]
__global__ void plane_partial_final_kernel(const float* part, int n, int c, int per, float* out) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    float s = 0.f;
    for (int nn = 0; nn < n; ++nn)
        for (int k = 0; k < per; ++k) s += part[((size_t)nn * c + ch) * per + k];
    out[ch] = s;
}

// ---- per-channel sum over (n, hw) --------------------------------------------------------------------
__global__ __launch_bounds__(256) void channel_sum_partial_kernel(const float* x, int n, int c, int hw, int split,
                                                                  float* part /* [c][split] */) {
    __shared__ float red[4];
    const int ch = blockIdx.x / split, s = blockIdx.x % split;
    const size_t M = (size_t)n * hw;
    const size_t per = (M + split - 1) / split;
    const size_t beg = (size_t)s * per, end = beg + per < M ? beg + per : M;
    float v[1] = {0.f};
    if (beg < end) {
        const size_t hws = (size_t)hw;
        const size_t n0 = beg / hws, n1 = (end - 1) / hws;
        for (size_t nn = n0; nn <= n1; ++nn) {
            const size_t lo = (beg > nn * hws ? beg : nn * hws) - nn * hws;
            const size_t hi = (end < (nn + 1) * hws ? end : (nn + 1) * hws) - nn * hws;
            const float* xp = x + (nn * c + ch) * hws;
            for (size_t r = lo + threadIdx.x; r < hi; r += 256) v[0] += xp[r];
        }
    }
    block_sum<1>(v, red);
    if (threadIdx.x == 0) part[(size_t)ch * split + s] = v[0];
}

__global__ void channel_sum_final_kernel(const float* part, int c, int split, float* out) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    float s = 0.f;
    for (int k = 0; k < split; ++k) s += part[(size_t)ch * split + k];
    out[ch] = s;
}

// ---- flat reductions ----------------------------------------------------------------------------------
constexpr int kRedBlocks = 1024;

__global__ __launch_bounds__(256) void sum_partial_kernel(const float* x, size_t count, float* part) {
    __shared__ float red[4];
    float v[1] = {0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) v[0] += x[i];
    block_sum<1>(v, red);
    if (threadIdx.x == 0) part[blockIdx.x] = v[0];
}

__global__ __launch_bounds__(256) void sum_final_kernel(const float* part, int nparts, float scale, float* out) {
    __shared__ float red[4];
    float v[1] = {0.f};
    for (int i = threadIdx.x; i < nparts; i += 256) v[0] += part[i];
    block_sum<1>(v, red);
    if (threadIdx.x == 0) out[0] = v[0] * scale;
}

__global__ __launch_bounds__(256) void pixel_loss_kernel(const float* pred, const float* target, size_t count, int kind,
                                                         float gscale, float* part, float* dpred) {
    __shared__ float red[4];
    float v[1] = {0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        const float d = pred[i] - target[i];
        if (kind == VCG_LOSS_MSE) {
            v[0] += d * d;
            if (dpred) dpred[i] = 2.f * d * gscale;
        } else {
            v[0] += fabsf(d);
            if (dpred) dpred[i] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * gscale;
        }
    }
    block_sum<1>(v, red);
    if (threadIdx.x == 0) part[blockIdx.x] = v[0];
}

__global__ void fill_kernel(float* y, size_t count, float value) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) y[i] = value;
}

__global__ void axpby_kernel(const float* x, float* y, size_t count, float a, float b) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) y[i] = a * x[i] + (b == 0.f ? 0.f : b * y[i]);
}

// ---- Keras-form Adam over a flat buffer ---------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, size_t count,
                                                   float lr_t, float b1, float b2, float eps, float gscale) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
}

// same update with the step count t kept in device memory, so that a captured hipGraph replays correctly as t
// advances: one thread evaluates lr_t = lr*sqrt(1-b2^t)/(1-b1^t) in double (1-b2^t cancels to ~1e-3: in fp32 it
// would carry 6e-5 of relative error, and the replayed step would drift from the eagerly launched one, whose lr_t the
// host computes in double), leaves it beside the counter and advances the counter
__global__ void adam_lr_kernel(int* t_dev, float lr, float b1, float b2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double t = (double)(t_dev[0] + 1);
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t));
    ((float*)t_dev)[1] = (float)lr_t;
    t_dev[0] += 1;
}

__global__ __launch_bounds__(256) void adam_dev_kernel(float* p, const float* g, float* m, float* v, size_t count, float b1,
                                                       float b2, float eps, float gscale, const int* t_dev) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const float lr_t = ((const float*)t_dev)[1];
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
}

// ---- layout / value conversion at the frame edge ------------------------------------------------------
__global__ void u8_to_nchw_kernel(const uint8_t* src, float* dst, int n, int h, int w, int c) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // index in dst (NCHW)
    const size_t total = (size_t)n * c * h * w;
    if (i >= total) return;
    const int x = (int)(i % w);
    const int y = (int)((i / w) % h);
    const int ch = (int)((i / ((size_t)w * h)) % c);
    const size_t nn = i / ((size_t)w * h * c);
    const uint8_t u = src[((nn * h + y) * w + x) * c + ch];
    dst[i] = (float)((double)u / 127.5 - 1.0);               // data.py:266-270 computes in float64
}

__global__ void nchw_to_u8_kernel(const float* src, uint8_t* dst, int n, int h, int w, int c) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // index in dst (NHWC)
    const size_t total = (size_t)n * c * h * w;
    if (i >= total) return;
    const int ch = (int)(i % c);
    const int x = (int)((i / c) % w);
    const int y = (int)((i / ((size_t)c * w)) % h);
    const size_t nn = i / ((size_t)c * w * h);
    const float a = src[((nn * c + ch) * h + y) * w + x];
    float r = rintf((a + 1.f) * 127.5f);                     // np.around = round-half-even (data.py:254)
    r = fminf(fmaxf(r, 0.f), 255.f);
    dst[i] = (uint8_t)r;
}

__global__ void nhwc_to_nchw_kernel(const float* src, float* dst, int n, int h, int w, int c) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // dst index
    const size_t total = (size_t)n * c * h * w;
    if (i >= total) return;
    const int x = (int)(i % w);
    const int y = (int)((i / w) % h);
    const int ch = (int)((i / ((size_t)w * h)) % c);
    const size_t nn = i / ((size_t)w * h * c);
    dst[i] = src[((nn * h + y) * w + x) * c + ch];
}

__global__ void nchw_to_nhwc_kernel(const float* src, float* dst, int n, int h, int w, int c) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // dst index
    const size_t total = (size_t)n * c * h * w;
    if (i >= total) return;
    const int ch = (int)(i % c);
    const int x = (int)((i / c) % w);
    const int y = (int)((i / ((size_t)c * w)) % h);
    const size_t nn = i / ((size_t)c * w * h);
    dst[i] = src[((nn * c + ch) * h + y) * w + x];
}

// MaxPooling2D((2,2), strides (2,2), 'valid') of keras.applications.VGG19 (the perceptual losses, model.py:101-157)
__global__ void maxpool2x2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t planes, int h, int w) {
    const int oh = h >> 1, ow = w >> 1;
    const size_t total = planes * oh * ow;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ox = (int)(i % ow), oy = (int)((i / ow) % oh);
    const size_t pl = i / ((size_t)ow * oh);
    const float* s = x + (pl * h + 2 * oy) * w + 2 * ox;
    y[i] = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[w], s[w + 1]));
}

// gradient to the FIRST maximal element of each window in row-major order (TF / torch convention); rows / columns
// beyond 2*floor(h/2) take no part in any window and get zero
__global__ void maxpool2x2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                      size_t planes, int h, int w) {
    const int oh = h >> 1, ow = w >> 1;
    const size_t total = planes * h * w;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ix = (int)(i % w), iy = (int)((i / w) % h);
    const size_t pl = i / ((size_t)w * h);
    const int oy = iy >> 1, ox = ix >> 1;
    float g = 0.f;
    if (oy < oh && ox < ow) {
        const float* s = x + (pl * h + 2 * oy) * w + 2 * ox;
        const float v0 = s[0], v1 = s[1], v2 = s[w], v3 = s[w + 1];
        int arg = 0;
        float m = v0;
        if (v1 > m) { m = v1; arg = 1; }
        if (v2 > m) { m = v2; arg = 2; }
        if (v3 > m) { m = v3; arg = 3; }
        if (arg == (iy & 1) * 2 + (ix & 1)) g = dy[(pl * oh + oy) * ow + ox];
    }
    dx[i] = g;
}

// (taps, a, b) -> (taps, b, a), LDS-tiled 32x32 transpose
__global__ __launch_bounds__(256) void kernel_transpose_kernel(const float* src, float* dst, int a, int b) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int a0 = blockIdx.y * 32, b0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const float* s = src + (size_t)t * a * b;
    float* d = dst + (size_t)t * a * b;
    for (int r = ty; r < 32; r += 8)
        if (a0 + r < a && b0 + tx < b) tile[r][tx] = s[(size_t)(a0 + r) * b + b0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (b0 + r < b && a0 + tx < a) d[(size_t)(b0 + r) * a + a0 + tx] = tile[tx][r];
}


// ---- discriminator output squashing (model.py:885-892) and GanLosses.loss_activation (model.py:172-181) -----------
// sigmoid / log(sigmoid) / tanh / bi-log: x/(1+|x|) * log(|x|+2).  log(sigmoid(x)) is evaluated as
// min(x,0) - log1p(exp(-|x|)) (finite where Keras' K.log(K.sigmoid(x)) underflows to -inf); derivatives in closed form
// (d|x|/dx = sign(x), 0 at 0, as TF's AbsGrad).
template <typename T>
__device__ __forceinline__ T head_value(T x, int kind) {
    const T a = x < T(0) ? -x : x;
    switch (kind) {
        case VCG_HEAD_SIGMOID: return T(1) / (T(1) + exp(-x));
        case VCG_HEAD_LOGSIGM: return (x < T(0) ? x : T(0)) - log1p(exp(-a));
        case VCG_HEAD_TANH: return tanh(x);
        case VCG_HEAD_BILOG: return (x / (T(1) + a)) * log(a + T(2));
        default: return x;
    }
}
template <typename T>
__device__ __forceinline__ T head_deriv(T x, int kind) {
    const T a = x < T(0) ? -x : x;
    switch (kind) {
        case VCG_HEAD_SIGMOID: return T(1) / ((T(1) + exp(-x)) * (T(1) + exp(x)));   // s(x) s(-x): no 1 - s cancellation
        case VCG_HEAD_LOGSIGM: return T(1) / (T(1) + exp(x));                  // 1 - sigmoid(x)
        case VCG_HEAD_TANH: { const T u = exp(T(-2) * a); return T(4) * u / ((T(1) + u) * (T(1) + u)); }   // sech^2, no 1 - t^2 cancellation
        case VCG_HEAD_BILOG: return log(a + T(2)) / ((T(1) + a) * (T(1) + a)) + a / ((T(1) + a) * (a + T(2)));
        default: return T(1);
    }
}

__global__ void head_act_fwd_kernel(const float* z, float* y, size_t count, int kind) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) y[i] = head_value<float>(z[i], kind);
}

__global__ void head_act_bwd_kernel(const float* z, const float* dy, float* dz, size_t count, int kind) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) dz[i] = dy[i] * head_deriv<float>(z[i], kind);
}

// loss = act((mean_a - mean_b) * mean_scale), evaluated in double by every thread from the two device scalars (no host
// read: the step stays hipGraph-capturable); da[i] = act'(.) * ga, db[i] = act'(.) * gb  -- the broadcast gradients
// of the two means.  One launch serves the relativistic D / G losses and (kind = none) the Wasserstein ones.
__global__ void gan_loss_kernel(const float* mean_a, const float* mean_b, float mean_scale, int kind, float* loss_out,
                                float* da, size_t na, float ga, float* db, size_t nb, float gb) {
    const double delta = ((double)mean_a[0] - (mean_b ? (double)mean_b[0] : 0.0)) * (double)mean_scale;
    const double g = head_deriv<double>(delta, kind);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i == 0 && loss_out) loss_out[0] = (float)head_value<double>(delta, kind);
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t k = i; k < na; k += stride) da[k] = (float)(g * (double)ga);
    for (size_t k = i; k < nb; k += stride) db[k] = (float)(g * (double)gb);
}

// ---- attention gates of make_upscaler_attention (model.py:34-36, 86-89): y = sigmoid(a) * m -------------------------------
__global__ void sigmoid_gate_fwd_kernel(const float* a, const float* m, float* y, size_t count) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) y[i] = m[i] / (1.f + expf(-a[i]));
}

// da = dy * m * s(1-s)  (the gradient in front of the sigmoid), dm = dy * s
__global__ void sigmoid_gate_bwd_kernel(const float* a, const float* m, const float* dy, float* da, float* dm, size_t count) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const float av = a[i], d = dy[i];
    const float s = 1.f / (1.f + expf(-av)), sd = 1.f / ((1.f + expf(-av)) * (1.f + expf(av)));
    da[i] = d * m[i] * sd;
    dm[i] = d * s;
}

// Lambda(lambda x: tf.math.atanh(0.99999 * x)) on the network input (model.py:94): y = atanh(scale * x), evaluated in double
__global__ void atanh_scale_kernel(const float* x, float* y, size_t count, float scale) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) y[i] = (float)atanh((double)scale * (double)x[i]);
}

// zero insertion: dst[plane][y*s][x*s] = src[plane][y][x], zeros elsewhere (data gradient of a stride-3 convolution as a
// stride-1 correlation over the dilated gradient)
__global__ void dilate2d_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t planes, int h, int w, int s) {
    const int dh = (h - 1) * s + 1, dw = (w - 1) * s + 1;
    const size_t total = planes * dh * dw;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % dw), y = (int)((i / dw) % dh);
    const size_t pl = i / ((size_t)dw * dh);
    dst[i] = (x % s == 0 && y % s == 0) ? src[(pl * h + y / s) * w + x / s] : 0.f;
}

// K.resize_images on TF 1.14: source coordinate = destination / f (align_corners=False, no half-pixel centres)
__global__ void resize2d_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t planes, int h, int w, int f, int bilinear) {
    const int oh = h * f, ow = w * f;
    const size_t total = planes * oh * ow;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % ow), y = (int)((i / ow) % oh);
    const float* s = src + (i / ((size_t)ow * oh)) * h * w;
    const int y0 = y / f, x0 = x / f;
    if (!bilinear) {
        dst[i] = s[y0 * w + x0];
        return;
    }
    // the fractions j/f are formed as TF forms them: in_y = y * (h / (float)oh), lerp weight = in_y - floor(in_y)
    const float sc = 1.0f / (float)f;
    const float fy = (float)y * sc - (float)y0, fx = (float)x * sc - (float)x0;
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float tl = s[y0 * w + x0], tr = s[y0 * w + x1], bl = s[y1 * w + x0], br = s[y1 * w + x1];
    const float top = tl + (tr - tl) * fx, bot = bl + (br - bl) * fx;
    dst[i] = top + (bot - top) * fy;
}

// dst (oh x ow) = window of src (h x w) at (top, left); PAD: the reverse (src placed into a zeroed dst)
template <bool PAD>
__global__ void crop_pad2d_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t planes, int h, int w, int top, int left,
                                  int oh, int ow) {
    const size_t total = planes * oh * ow;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % ow), y = (int)((i / ow) % oh);
    const size_t pl = i / ((size_t)ow * oh);
    if (PAD) {
        const int sy = y - top, sx = x - left;
        dst[i] = ((unsigned)sy < (unsigned)h && (unsigned)sx < (unsigned)w) ? src[(pl * h + sy) * w + sx] : 0.f;
    } else {
        dst[i] = src[(pl * h + y + top) * w + x + left];
    }
}

__global__ void copy_channels_kernel(const float* __restrict__ src, float* __restrict__ dst, int c_src, int c_src_off, int c_dst, int c_dst_off,
                                     int c_count, size_t hw, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t p = i % hw, c = (i / hw) % c_count, n = i / (hw * c_count);
    dst[(n * c_dst + c_dst_off + c) * hw + p] = src[(n * c_src + c_src_off + c) * hw + p];
}

// counter-based uniform in [0, 1): two rounds of the splitmix64 finaliser over (seed, step, index)
__device__ __forceinline__ float hash_uniform(unsigned long long seed, unsigned long long step, unsigned long long idx) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (step + 1) + 0xD1B54A32D192ED03ull * idx;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(unsigned)(z >> 40) * (1.0f / 16777216.0f);
}

__global__ void dropout_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ mask, size_t count, float rate,
                                   unsigned long long seed, const unsigned long long* __restrict__ step) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const bool keep = hash_uniform(seed, step ? *step : 0ull, i) >= rate;
    mask[i] = keep ? 1 : 0;
    y[i] = keep ? x[i] / (1.0f - rate) : 0.f;
}

__global__ void dropout_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ mask, float* __restrict__ dx, size_t count, float rate) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    dx[i] = mask[i] ? dy[i] / (1.0f - rate) : 0.f;
}

__global__ void counter_inc_kernel(unsigned long long* c) { *c += 1; }

inline unsigned blocks_for(size_t count) { return (unsigned)((count + 255) / 256); }

}  // namespace

extern "C" {

size_t vcg_act_bwd_workspace_bytes(int n, int c, int hw) { return (size_t)2 * n * c * 64 * sizeof(float); }

int vcg_act_bwd(const float* saved, const float* dy, int n, int c, int hw, int act, float act_alpha,
                const float* prelu_alpha, float* dx, float* dprelu_alpha, float* dsum, void* ws, size_t ws_bytes,
                vcg_stream_t stream) {
    VCG_CHECK_PTR(saved); VCG_CHECK_PTR(dy); VCG_CHECK_PTR(dx);
    if (n <= 0 || c <= 0 || hw <= 0 || (long)n * c > 0x7fffffffL) return VCG_E_SHAPE;
    if (act == VCG_ACT_PRELU && prelu_alpha == nullptr) return VCG_E_NULL;
    hipStream_t st = (hipStream_t)stream;
    const unsigned gy = (unsigned)((long)n * c > 65535 ? 65535 : n * c);
    int gx = ceil_div(hw, 256 * 4);
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    float* part_alpha = nullptr;
    float* part_sum = nullptr;
    if (dprelu_alpha || dsum) {
        if (ws == nullptr || ws_bytes < vcg_act_bwd_workspace_bytes(n, c, hw)) return VCG_E_WORKSPACE;
        if (dprelu_alpha) part_alpha = (float*)ws;
        if (dsum) part_sum = (float*)ws + (size_t)n * c * 64;
    }
    if (hw % 4 == 0) {
        hipLaunchKernelGGL(act_bwd_kernel<true>, dim3(gx, gy), dim3(256), 0, st, saved, dy, n, c, hw, act,
                           act_alpha, prelu_alpha, dx, part_alpha, part_sum);
    } else {
        hipLaunchKernelGGL(act_bwd_kernel<false>, dim3(gx, gy), dim3(256), 0, st, saved, dy, n, c, hw, act,
                           act_alpha, prelu_alpha, dx, part_alpha, part_sum);
    }
    VCG_LAUNCH_CHECK();
    if (dprelu_alpha) {
        hipLaunchKernelGGL(plane_partial_final_kernel, dim3(ceil_div(c, 256)), dim3(256), 0, st, (const float*)part_alpha,
                           n, c, gx, dprelu_alpha);
        VCG_LAUNCH_CHECK();
    }
    if (dsum) {
        hipLaunchKernelGGL(plane_partial_final_kernel, dim3(ceil_div(c, 256)), dim3(256), 0, st, (const float*)part_sum, n,
                           c, gx, dsum);
        VCG_LAUNCH_CHECK();
    }
    return VCG_OK;
}

size_t vcg_channel_sum_workspace_bytes(int n, int c, int hw) { return (size_t)c * kSumSplitMax * sizeof(float); }

int vcg_channel_sum(const float* x, int n, int c, int hw, float* out, void* ws, size_t ws_bytes,
                    vcg_stream_t stream) {
    VCG_CHECK_PTR(x); VCG_CHECK_PTR(out); VCG_CHECK_PTR(ws);
    if (n <= 0 || c <= 0 || hw <= 0) return VCG_E_SHAPE;
    if (ws_bytes < vcg_channel_sum_workspace_bytes(n, c, hw)) return VCG_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int split = pick_split_cs(c, (size_t)n * hw);
    hipLaunchKernelGGL(channel_sum_partial_kernel, dim3(c * split), dim3(256), 0, st, x, n, c, hw, split, (float*)ws);
    VCG_LAUNCH_CHECK();
    hipLaunchKernelGGL(channel_sum_final_kernel, dim3(ceil_div(c, 256)), dim3(256), 0, st, (const float*)ws, c, split,
                       out);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

size_t vcg_mean_reduce_workspace_bytes(size_t count) { return kRedBlocks * sizeof(float); }

int vcg_mean_reduce(const float* x, size_t count, float* out, void* ws, size_t ws_bytes, vcg_stream_t stream) {
    VCG_CHECK_PTR(x); VCG_CHECK_PTR(out); VCG_CHECK_PTR(ws);
    if (count == 0) return VCG_E_SHAPE;
    if (ws_bytes < kRedBlocks * sizeof(float)) return VCG_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    unsigned nb = blocks_for(count);
    if (nb > kRedBlocks) nb = kRedBlocks;
    hipLaunchKernelGGL(sum_partial_kernel, dim3(nb), dim3(256), 0, st, x, count, (float*)ws);
    VCG_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, (int)nb, 1.f / (float)count, out);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_pixel_loss(const float* pred, const float* target, size_t count, int kind, float grad_scale, float* out,
                   float* dpred, void* ws, size_t ws_bytes, vcg_stream_t stream) {
    VCG_CHECK_PTR(pred); VCG_CHECK_PTR(target); VCG_CHECK_PTR(out); VCG_CHECK_PTR(ws);
    if (count == 0) return VCG_E_SHAPE;
    if (kind != VCG_LOSS_MSE && kind != VCG_LOSS_MAE) return VCG_E_UNSUPPORTED;
    if (ws_bytes < kRedBlocks * sizeof(float)) return VCG_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    unsigned nb = blocks_for(count);
    if (nb > kRedBlocks) nb = kRedBlocks;
    hipLaunchKernelGGL(pixel_loss_kernel, dim3(nb), dim3(256), 0, st, pred, target, count, kind,
                       grad_scale / (float)count, (float*)ws, dpred);
    VCG_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, (int)nb, 1.f / (float)count, out);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_head_act_fwd(const float* z, float* y, size_t count, int kind, vcg_stream_t stream) {
    VCG_CHECK_PTR(z); VCG_CHECK_PTR(y);
    if (kind < VCG_HEAD_NONE || kind > VCG_HEAD_BILOG) return VCG_E_UNSUPPORTED;
    if (count == 0) return VCG_OK;
    hipLaunchKernelGGL(head_act_fwd_kernel, dim3(blocks_for(count)), dim3(256), 0, (hipStream_t)stream, z, y, count, kind);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_head_act_bwd(const float* z, const float* dy, float* dz, size_t count, int kind, vcg_stream_t stream) {
    VCG_CHECK_PTR(z); VCG_CHECK_PTR(dy); VCG_CHECK_PTR(dz);
    if (kind < VCG_HEAD_NONE || kind > VCG_HEAD_BILOG) return VCG_E_UNSUPPORTED;
    if (count == 0) return VCG_OK;
    hipLaunchKernelGGL(head_act_bwd_kernel, dim3(blocks_for(count)), dim3(256), 0, (hipStream_t)stream, z, dy, dz, count, kind);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_gan_loss(const float* mean_a, const float* mean_b, float mean_scale, int kind, float* loss_out, float* da, size_t na,
                 float ga, float* db, size_t nb, float gb, vcg_stream_t stream) {
    VCG_CHECK_PTR(mean_a);
    if (kind < VCG_HEAD_NONE || kind > VCG_HEAD_BILOG) return VCG_E_UNSUPPORTED;
    if ((na && da == nullptr) || (nb && db == nullptr)) return VCG_E_NULL;
    const size_t most = na > nb ? na : nb;
    unsigned nbk = blocks_for(most ? most : 1);
    if (nbk > 1024) nbk = 1024;
    hipLaunchKernelGGL(gan_loss_kernel, dim3(nbk), dim3(256), 0, (hipStream_t)stream, mean_a, mean_b, mean_scale, kind, loss_out,
                       da, na, ga, db, nb, gb);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_sigmoid_gate_fwd(const float* a, const float* m, float* y, size_t count, vcg_stream_t stream) {
    VCG_CHECK_PTR(a); VCG_CHECK_PTR(m); VCG_CHECK_PTR(y);
    if (count == 0) return VCG_OK;
    hipLaunchKernelGGL(sigmoid_gate_fwd_kernel, dim3(blocks_for(count)), dim3(256), 0, (hipStream_t)stream, a, m, y, count);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_sigmoid_gate_bwd(const float* a, const float* m, const float* dy, float* da, float* dm, size_t count, vcg_stream_t stream) {
    VCG_CHECK_PTR(a); VCG_CHECK_PTR(m); VCG_CHECK_PTR(dy); VCG_CHECK_PTR(da); VCG_CHECK_PTR(dm);
    if (count == 0) return VCG_OK;
    hipLaunchKernelGGL(sigmoid_gate_bwd_kernel, dim3(blocks_for(count)), dim3(256), 0, (hipStream_t)stream, a, m, dy, da, dm, count);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_atanh_scale(const float* x, float* y, size_t count, float scale, vcg_stream_t stream) {
    VCG_CHECK_PTR(x); VCG_CHECK_PTR(y);
    if (count == 0) return VCG_OK;
    hipLaunchKernelGGL(atanh_scale_kernel, dim3(blocks_for(count)), dim3(256), 0, (hipStream_t)stream, x, y, count, scale);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_dilate2d(const float* src, float* dst, size_t planes, int h, int w, int stride, vcg_stream_t stream) {
    VCG_CHECK_PTR(src); VCG_CHECK_PTR(dst);
    if (planes == 0 || h <= 0 || w <= 0 || stride < 1) return VCG_E_SHAPE;
    const size_t total = planes * ((size_t)(h - 1) * stride + 1) * ((size_t)(w - 1) * stride + 1);
    hipLaunchKernelGGL(dilate2d_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, planes, h, w, stride);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_resize2d(const float* src, float* dst, size_t planes, int h, int w, int factor, int bilinear, vcg_stream_t stream) {
    VCG_CHECK_PTR(src); VCG_CHECK_PTR(dst);
    if (planes == 0 || h <= 0 || w <= 0 || factor < 1) return VCG_E_SHAPE;
    const size_t total = planes * (size_t)h * factor * w * factor;
    hipLaunchKernelGGL(resize2d_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, planes, h, w, factor, bilinear);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_crop2d(const float* src, float* dst, size_t planes, int h, int w, int top, int left, int oh, int ow, vcg_stream_t stream) {
    VCG_CHECK_PTR(src); VCG_CHECK_PTR(dst);
    if (planes == 0 || h <= 0 || w <= 0 || oh <= 0 || ow <= 0 || top < 0 || left < 0 || top + oh > h || left + ow > w) return VCG_E_SHAPE;
    hipLaunchKernelGGL(crop_pad2d_kernel<false>, dim3(blocks_for(planes * oh * ow)), dim3(256), 0, (hipStream_t)stream, src, dst, planes, h, w, top,
                       left, oh, ow);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_pad2d(const float* src, float* dst, size_t planes, int h, int w, int top, int left, int oh, int ow, vcg_stream_t stream) {
    VCG_CHECK_PTR(src); VCG_CHECK_PTR(dst);
    if (planes == 0 || h <= 0 || w <= 0 || oh <= 0 || ow <= 0 || top < 0 || left < 0 || top + h > oh || left + w > ow) return VCG_E_SHAPE;
    hipLaunchKernelGGL(crop_pad2d_kernel<true>, dim3(blocks_for(planes * oh * ow)), dim3(256), 0, (hipStream_t)stream, src, dst, planes, h, w, top,
                       left, oh, ow);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_copy_channels(const float* src, float* dst, int n, int c_src, int c_src_off, int c_dst, int c_dst_off, int c_count, size_t hw,
                      vcg_stream_t stream) {
    VCG_CHECK_PTR(src); VCG_CHECK_PTR(dst);
    if (n <= 0 || c_count <= 0 || hw == 0 || c_src_off < 0 || c_dst_off < 0 || c_src_off + c_count > c_src || c_dst_off + c_count > c_dst)
        return VCG_E_SHAPE;
    const size_t total = (size_t)n * c_count * hw;
    hipLaunchKernelGGL(copy_channels_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, c_src, c_src_off, c_dst, c_dst_off,
                       c_count, hw, total);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_dropout_fwd(const float* x, float* y, unsigned char* mask, size_t count, float rate, unsigned long long seed,
                    const unsigned long long* step, vcg_stream_t stream) {
    VCG_CHECK_PTR(x); VCG_CHECK_PTR(y); VCG_CHECK_PTR(mask);
    if (!(rate >= 0.f && rate < 1.f)) return VCG_E_SHAPE;
    if (count == 0) return VCG_OK;
    hipLaunchKernelGGL(dropout_fwd_kernel, dim3(blocks_for(count)), dim3(256), 0, (hipStream_t)stream, x, y, mask, count, rate, seed, step);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_dropout_bwd(const float* dy, const unsigned char* mask, float* dx, size_t count, float rate, vcg_stream_t stream) {
    VCG_CHECK_PTR(dy); VCG_CHECK_PTR(mask); VCG_CHECK_PTR(dx);
    if (!(rate >= 0.f && rate < 1.f)) return VCG_E_SHAPE;
    if (count == 0) return VCG_OK;
    hipLaunchKernelGGL(dropout_bwd_kernel, dim3(blocks_for(count)), dim3(256), 0, (hipStream_t)stream, dy, mask, dx, count, rate);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_counter_inc(unsigned long long* counter, vcg_stream_t stream) {
    VCG_CHECK_PTR(counter);
    hipLaunchKernelGGL(counter_inc_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_fill(float* y, size_t count, float value, vcg_stream_t stream) {
    VCG_CHECK_PTR(y);
    if (count == 0) return VCG_OK;
    hipLaunchKernelGGL(fill_kernel, dim3(blocks_for(count)), dim3(256), 0, (hipStream_t)stream, y, count, value);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_axpby(const float* x, float* y, size_t count, float a, float b, vcg_stream_t stream) {
    VCG_CHECK_PTR(x); VCG_CHECK_PTR(y);
    if (count == 0) return VCG_OK;
    hipLaunchKernelGGL(axpby_kernel, dim3(blocks_for(count)), dim3(256), 0, (hipStream_t)stream, x, y, count, a, b);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_adam_keras_multi(float* p, const float* g, float* m, float* v, size_t count, float lr_t, float beta_1,
                         float beta_2, float eps, float grad_scale, vcg_stream_t stream) {
    VCG_CHECK_PTR(p); VCG_CHECK_PTR(g); VCG_CHECK_PTR(m); VCG_CHECK_PTR(v);
    if (count == 0) return VCG_OK;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(count)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, count, lr_t,
                       beta_1, beta_2, eps, grad_scale);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_adam_keras_multi_dev(float* p, const float* g, float* m, float* v, size_t count, float lr, float beta_1,
                             float beta_2, float eps, float grad_scale, int* t_dev, vcg_stream_t stream) {
    VCG_CHECK_PTR(p); VCG_CHECK_PTR(g); VCG_CHECK_PTR(m); VCG_CHECK_PTR(v); VCG_CHECK_PTR(t_dev);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_lr_kernel, dim3(1), dim3(64), 0, st, (int*)t_dev, lr, beta_1, beta_2);
    VCG_LAUNCH_CHECK();
    if (count) {
        hipLaunchKernelGGL(adam_dev_kernel, dim3(blocks_for(count)), dim3(256), 0, st, p, g, m, v, count, beta_1, beta_2, eps,
                           grad_scale, (const int*)t_dev);
        VCG_LAUNCH_CHECK();
    }
    return VCG_OK;
}

int vcg_frames_u8_to_nchw(const uint8_t* src, float* dst, int n, int h, int w, int c, vcg_stream_t stream) {
    VCG_CHECK_PTR(src); VCG_CHECK_PTR(dst);
    if (n <= 0 || h <= 0 || w <= 0 || c <= 0) return VCG_E_SHAPE;
    const size_t total = (size_t)n * h * w * c;
    hipLaunchKernelGGL(u8_to_nchw_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, n, h, w, c);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_nchw_to_frames_u8(const float* src, uint8_t* dst, int n, int h, int w, int c, vcg_stream_t stream) {
    VCG_CHECK_PTR(src); VCG_CHECK_PTR(dst);
    if (n <= 0 || h <= 0 || w <= 0 || c <= 0) return VCG_E_SHAPE;
    const size_t total = (size_t)n * h * w * c;
    hipLaunchKernelGGL(nchw_to_u8_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, n, h, w, c);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_nhwc_to_nchw(const float* src, float* dst, int n, int h, int w, int c, vcg_stream_t stream) {
    VCG_CHECK_PTR(src); VCG_CHECK_PTR(dst);
    if (n <= 0 || h <= 0 || w <= 0 || c <= 0) return VCG_E_SHAPE;
    const size_t total = (size_t)n * h * w * c;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, n, h, w, c);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_nchw_to_nhwc(const float* src, float* dst, int n, int h, int w, int c, vcg_stream_t stream) {
    VCG_CHECK_PTR(src); VCG_CHECK_PTR(dst);
    if (n <= 0 || h <= 0 || w <= 0 || c <= 0) return VCG_E_SHAPE;
    const size_t total = (size_t)n * h * w * c;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, n, h, w, c);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_kernel_transpose(const float* src, float* dst, int taps, int a, int b, vcg_stream_t stream) {
    VCG_CHECK_PTR(src); VCG_CHECK_PTR(dst);
    if (taps <= 0 || a <= 0 || b <= 0 || taps > 65535) return VCG_E_SHAPE;
    hipLaunchKernelGGL(kernel_transpose_kernel, dim3(ceil_div(b, 32), ceil_div(a, 32), taps), dim3(256), 0,
                       (hipStream_t)stream, src, dst, a, b);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_maxpool2x2_fwd(const float* x, float* y, int n, int c, int h, int w, vcg_stream_t stream) {
    VCG_CHECK_PTR(x); VCG_CHECK_PTR(y);
    if (n <= 0 || c <= 0 || h < 2 || w < 2) return VCG_E_SHAPE;
    const size_t total = (size_t)n * c * (h >> 1) * (w >> 1);
    hipLaunchKernelGGL(maxpool2x2_fwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, (size_t)n * c, h, w);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_maxpool2x2_bwd(const float* x, const float* dy, float* dx, int n, int c, int h, int w, vcg_stream_t stream) {
    VCG_CHECK_PTR(x); VCG_CHECK_PTR(dy); VCG_CHECK_PTR(dx);
    if (n <= 0 || c <= 0 || h < 2 || w < 2) return VCG_E_SHAPE;
    const size_t total = (size_t)n * c * h * w;
    hipLaunchKernelGGL(maxpool2x2_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, (size_t)n * c, h, w);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

}  // extern "C"
