// One-output-channel convolution on bf16 NHWC activations: the 70x70 PatchGAN's last layer, Conv2D(1, 4, stride 1, zero padding 1) on 512
// channels (the reference's critics end in Dense layers, model.py:836-896), in the bf16 configs (BASELINE.json
// C3 / C4).  Forward, data gradient and weight gradient read / write the [n][h][w][cin] bf16 tensor directly -- the fp32 NCHW copies the
// fp32 kernels needed (two layout conversions of a 512-channel tensor per application) are gone.
//
// All three are plain FMA reductions, as conv_cout1_kernel is: with one output channel an MFMA tile would be 31/32 padding.  With channels
// innermost a pixel's cin <= 512 channels are ONE wave-wide 16-byte load (lane l = channels 8l .. 8l+7), the K x K weights of those 8
// channels live in 8 K^2 registers per lane for the whole launch, and a wave walks a run of SEG consecutive outputs of one row with a
// sliding K x K window of pixels in registers (each input pixel is loaded once per output row, not K times).  fp32 weights and accumulation;
// activations are the bf16 values as stored.
#include "vcg_common.hpp"
#include <utility>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int K> constexpr int seg_len() { return 16 / K * K; }         // outputs per wave job: a multiple of K (the window's slot rotation)

struct HeadParams {
    const unsigned char* x;         // bf16 NHWC [n][h][w][cin]          (fwd, wgrad: input;  dgrad: output)
    const float* w;                 // Keras (kh,kw,cin,1) = [tap][cin]
    const float* bias;              // [1] or null
    float* y;                       // fp32 [n][1][oh][ow]               (fwd: output;  dgrad, wgrad: the gradient dy)
    float* ws;                      // wgrad partial records [blocks][K*K*cin + 1]
    int n, cin, h, w_, oh, ow, pad_top, pad_left, segs, jobs;
};

template <int K>
__device__ __forceinline__ void load_weights(const HeadParams& p, int lane, float (&wr)[K * K][8]) {
    // range-checked 16-byte loads (channels past cin read 0): no branch per element
    const vcg_rsrc rw = make_rsrc(p.w, (size_t)K * K * p.cin * sizeof(float));
    const int c0 = lane * 8;
#pragma unroll
    for (int t = 0; t < K * K; ++t) {
        const unsigned off = c0 < p.cin ? (unsigned)(t * p.cin + c0) * 4u : VCG_OOB;
        const auto a = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)off, 0, 0);
        const auto b = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)(off == VCG_OOB ? VCG_OOB : off + 16u), 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // (a bit_cast straight from the vector-element lvalue a[j] reads element 0 for every j with this compiler: copy first)
            const unsigned ua = a[j], ub = b[j];
            wr[t][j] = __builtin_bit_cast(float, ua);
            wr[t][4 + j] = __builtin_bit_cast(float, ub);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// forward: y[n][oy][ox] = bias + sum_{ky,kx,c} x[n][oy+ky-pt][ox+kx-pl][c] w[ky][kx][c]
// ---------------------------------------------------------------------------------------------------------------
// sliding K x K window of pixels in registers: each input pixel is loaded once per output row.  (Measured against an input-stationary form
// with K running output sums per lane: 35 us against 63 at the PatchGAN head's size -- the window form keeps 4 K loads in flight per step.)
template <int K>
__global__ __launch_bounds__(256) void cout1_fwd_bf16_kernel(const HeadParams p) {
    constexpr int H_SEG = seg_len<K>();
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float wr[K * K][8];
    load_weights<K>(p, lane, wr);
    const float b = p.bias ? p.bias[0] : 0.f;
    const long img_bytes = (long)p.h * p.w_ * p.cin * 2;
    const unsigned lane_off = lane * 8 < p.cin ? (unsigned)lane * 16u : VCG_OOB;
    for (int job = blockIdx.x * 4 + wv; job < p.jobs; job += gridDim.x * 4) {
        const int seg = job % p.segs, t2 = job / p.segs, oy = t2 % p.oh, n = t2 / p.oh;
        const int x0 = seg * H_SEG;
        const vcg_rsrc rx = make_rsrc(p.x + n * img_bytes, (unsigned long)img_bytes);
        bf16x8 win[K][K];                                       // [column slot][ky]
        auto load_col = [&](int slot, int gx) {
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const int gy = oy + ky - p.pad_top;
                const bool ok = (unsigned)gy < (unsigned)p.h && (unsigned)gx < (unsigned)p.w_ && lane_off != VCG_OOB;
                const unsigned off = ok ? (unsigned)(gy * p.w_ + gx) * (unsigned)(p.cin * 2) + lane_off : VCG_OOB;
                win[slot][ky] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)off, 0, 0));
            }
        };
#pragma unroll
        for (int s = 0; s < K - 1; ++s) load_col(s, x0 - p.pad_left + s);
#pragma unroll 1
        for (int j0 = 0; j0 < H_SEG; j0 += K) {
#pragma unroll
            for (int jj = 0; jj < K; ++jj) {
                const int ox = x0 + j0 + jj;
                load_col((jj + K - 1) % K, ox - p.pad_left + K - 1);
                float acc[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
                for (int kx = 0; kx < K; ++kx)
#pragma unroll
                    for (int ky = 0; ky < K; ++ky) {
                        const bf16x8 v = win[(jj + kx) % K][ky];
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[j] = fmaf((float)v[j], wr[ky * K + kx][j], acc[j]);
                    }
                const float s = wave_sum(((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7])));
                if (lane == 0 && ox < p.ow) p.y[((long)n * p.oh + oy) * p.ow + ox] = s + b;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// data gradient: dx[n][iy][ix][c] = sum_{ky,kx} dy[n][iy+pt-ky][ix+pl-kx] w[ky][kx][c]        (bf16 NHWC out)
// ---------------------------------------------------------------------------------------------------------------
// A wave holds the K rows of dy it needs as lane vectors (lane j = column ix0 + pl - (K-1) + j) and broadcasts each scalar with a
// compile-time v_readlane; every lane then does 8 K^2 FMAs and stores its 16 bytes of the pixel.
template <int K>
__global__ __launch_bounds__(256) void cout1_dgrad_bf16_kernel(const HeadParams p) {
    constexpr int H_SEG = seg_len<K>();
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float wr[K * K][8];
    load_weights<K>(p, lane, wr);
    const long plane = (long)p.oh * p.ow;
    for (int job = blockIdx.x * 4 + wv; job < p.jobs; job += gridDim.x * 4) {
        const int seg = job % p.segs, t2 = job / p.segs, iy = t2 % p.h, n = t2 / p.h;
        const int x0 = seg * H_SEG;
        float dyr[K];                                           // row ky: dy[n][iy + pt - ky][x0 + pl - (K-1) + lane]
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int gy = iy + p.pad_top - ky, gx = x0 + p.pad_left - (K - 1) + lane;
            const bool ok = (unsigned)gy < (unsigned)p.oh && (unsigned)gx < (unsigned)p.ow && lane < H_SEG + K - 1;
            dyr[ky] = ok ? p.y[n * plane + (long)gy * p.ow + gx] : 0.f;
        }
#pragma unroll
        for (int jj = 0; jj < H_SEG; ++jj) {
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    // output column ix = x0 + jj reads dy column ix + pl - kx = lane index jj + (K-1) - kx
                    const float g = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dyr[ky]), jj + K - 1 - kx));
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = fmaf(g, wr[ky * K + kx][j], acc[j]);
                }
            const int ix = x0 + jj;
            if (ix < p.w_ && lane * 8 < p.cin) {
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (__bf16)acc[j];
                *(bf16x8*)(const_cast<unsigned char*>(p.x) + (((long)n * p.h + iy) * p.w_ + ix) * (long)(p.cin * 2) + lane * 16) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient: dW[ky][kx][c] = sum_{n,oy,ox} x[n][oy+ky-pt][ox+kx-pl][c] dy[n][oy][ox];   db = sum dy
// ---------------------------------------------------------------------------------------------------------------
// Input-stationary: a job is a run of SEG pixels of one INPUT row; a pixel is loaded and converted once and multiplied into all K x K taps
// with the K x K gradients dy[gy+pt-ky][gx+pl-kx] it meets (K lane vectors of dy, broadcast by compile-time v_readlane).  A wave keeps its
// 8 K^2 sums in registers over all its jobs; the four waves of a workgroup add up in LDS in wave order and the workgroup writes one record;
// a second kernel sums the records in a fixed order (deterministic).
template <int K>
__global__ __launch_bounds__(256, 2) void cout1_wgrad_bf16_kernel(const HeadParams p) {
    constexpr int H_SEG = seg_len<K>();
    extern __shared__ float red[];                              // [K*K*8][64] + 4
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float acc[K * K][8];
#pragma unroll
    for (int t = 0; t < K * K; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
    const long img_bytes = (long)p.h * p.w_ * p.cin * 2;
    const long plane = (long)p.oh * p.ow;
    const unsigned lane_off = lane * 8 < p.cin ? (unsigned)lane * 16u : VCG_OOB;
    for (int job = blockIdx.x * 4 + wv; job < p.jobs; job += gridDim.x * 4) {
        const int seg = job % p.segs, t2 = job / p.segs, gy = t2 % p.h, n = t2 / p.h;
        const int x0 = seg * H_SEG;
        const vcg_rsrc rx = make_rsrc(p.x + n * img_bytes, (unsigned long)img_bytes);
        float dyr[K];                                           // row ky: dy[n][gy + pt - ky][x0 + pl - (K-1) + lane]
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int oy = gy + p.pad_top - ky, ox = x0 + p.pad_left - (K - 1) + lane;
            const bool ok = (unsigned)oy < (unsigned)p.oh && (unsigned)ox < (unsigned)p.ow && lane < H_SEG + K - 1;
            dyr[ky] = ok ? p.y[n * plane + (long)oy * p.ow + ox] : 0.f;
        }
        const unsigned rowoff = lane_off != VCG_OOB ? (unsigned)(gy * p.w_) * (unsigned)(p.cin * 2) + lane_off : VCG_OOB;
#pragma unroll 2
        for (int m = 0; m < H_SEG; ++m) {
            const int gx = x0 + m;
            const unsigned off = gx < p.w_ && rowoff != VCG_OOB ? rowoff + (unsigned)gx * (unsigned)(p.cin * 2) : VCG_OOB;
            const bf16x8 v = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)off, 0, 0));
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = (float)v[j];
#pragma unroll
            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const float g = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dyr[ky]), m + K - 1 - kx));
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[ky * K + kx][j] = fmaf(f[j], g, acc[ky * K + kx][j]);
                }
        }
    }
    // bias gradient: every workgroup sums a slice of dy
    float dsum = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < p.n * plane; i += (long)gridDim.x * 256) dsum += p.y[i];
    dsum = wave_sum(dsum);
    if (lane == 0) red[K * K * 512 + wv] = dsum;
    for (int w = 0; w < 4; ++w) {
        if (wv == w) {
#pragma unroll
            for (int t = 0; t < K * K; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float* q = red + (t * 8 + j) * 64 + lane;
                    *q = w == 0 ? acc[t][j] : *q + acc[t][j];
                }
        }
        __syncthreads();
    }
    float* out = p.ws + (long)blockIdx.x * (K * K * p.cin + 1);
    for (int i = threadIdx.x; i < K * K * p.cin; i += 256) {
        const int t = i / p.cin, c = i - t * p.cin;
        out[i] = red[(t * 8 + (c & 7)) * 64 + (c >> 3)];
    }
    if (threadIdx.x == 0) out[K * K * p.cin] = (red[K * K * 512] + red[K * K * 512 + 1]) + (red[K * K * 512 + 2] + red[K * K * 512 + 3]);
}

// dw[i] = sum over the workgroup records (fixed order); i == count: the bias gradient.  32 outputs x 8 record slices per workgroup
__global__ __launch_bounds__(256) void cout1_wgrad_reduce_kernel(const float* __restrict__ ws, int blocks, int count, float* __restrict__ dw,
                                                                 float* __restrict__ db) {
    __shared__ float part[8][32];
    const int o = threadIdx.x & 31, ks = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + o;
    const long stride = count + 1;
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    if (i <= count) {
        const float* src = ws + i;
        int k = ks;
        for (; k + 24 < blocks; k += 32)
#pragma unroll
            for (int u = 0; u < 4; ++u) s4[u] += src[(k + 8 * u) * stride];
        for (; k < blocks; k += 8) s4[0] += src[k * stride];
    }
    part[ks][o] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    __syncthreads();
    if (ks == 0 && i <= count) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) s += part[q][o];
        if (i < count) dw[i] = s;
        else if (db) db[0] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// data gradient of a critic's FIRST layer (3 input channels) from the bf16 NHWC gradient dz in front of its activation
// ---------------------------------------------------------------------------------------------------------------
// dx has 3 channels: as an implicit GEMM of its own it would be 29/32 padding.  Instead the layer's gradient is written as a 3x3 stride-1
// convolution over dz with 64 "virtual" output channels and run on the LDS-tiled generic kernel (vcg_conv2d_nhwc_bf16_fwd, bf16_gconv.hip):
//   4x4 stride 2, pad 1 (PatchGAN block 1):  dx[c][2q+py][2p+px] = sum_{a,b,m} dz[m][q+a-1][p+b-1] Wv[a][b][m][(2py+px)*3 + c]
//       with Wv[a][b][m][..] = W[ky][kx][c][m] where ky = 1 - 2(a-1) + py (py = 0: a = 1 -> ky 1, a = 0 -> ky 3; py = 1: a = 2 -> ky 0, a = 1 -> ky 2)
//       and 0 where that ky / kx falls outside 0..3: the four sub-pixel phases are 12 of the 64 virtual channels;
//   3x3 stride 1, pad 1 (simple_512 / thin_512 block 1, model.py:839):  Wv[a][b][m][c] = W[2-a][2-b][c][m], 3 of the 64 virtual channels.
// unshuffle_phases_kernel then scatters the 12 (3) channels of every virtual pixel to the fp32 NCHW gradient of the frames.
__global__ void build_first_dgrad_kernel(const float* __restrict__ w, int k, int stride, int m_count, float* __restrict__ wv) {
    const int idx = blockIdx.x * 256 + threadIdx.x;              // wv[a][b][m][v], v = virtual output channel
    if (idx >= 9 * 64 * 64) return;
    const int v = idx & 63, m = (idx >> 6) & 63, ab = idx >> 12, a = ab / 3, b = ab - 3 * a;
    float r = 0.f;
    if (m < m_count) {
        if (stride == 2) {
            if (v < 12) {
                const int ph = v / 3, c = v - 3 * ph, py = ph >> 1, px = ph & 1;
                const int ky = 1 - 2 * (a - 1) + py, kx = 1 - 2 * (b - 1) + px;
                if (ky >= 0 && ky < k && kx >= 0 && kx < k) r = w[((ky * k + kx) * 3 + c) * m_count + m];
            }
        } else if (v < 3) {
            r = w[(((2 - a) * k + (2 - b)) * 3 + v) * m_count + m];
        }
    }
    wv[idx] = r;
}

__global__ __launch_bounds__(256) void unshuffle_phases_kernel(const __bf16* __restrict__ t, int qh, int qw, int h, int w_, int stride,
                                                               float* __restrict__ dx) {
    // thread = one virtual pixel (q, p) of image blockIdx.y; reads its first 16 channels (two 16-byte loads), writes stride^2 x 3 floats
    const int n = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)qh * qw) return;
    const int q = (int)(i / qw), p = (int)(i - (long)q * qw);
    const __bf16* src = t + ((long)n * qh * qw + i) * 64;
    const bf16x8 lo = *(const bf16x8*)src, hi = *(const bf16x8*)(src + 8);
    float v[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = (float)lo[j]; v[8 + j] = (float)hi[j]; }
    const long plane = (long)h * w_;
    float* dst = dx + (long)n * 3 * plane;
    if (stride == 2) {
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const int iy = 2 * q + (ph >> 1), ix = 2 * p + (ph & 1);
            if (iy < h && ix < w_) {
#pragma unroll
                for (int c = 0; c < 3; ++c) dst[c * plane + (long)iy * w_ + ix] = v[ph * 3 + c];
            }
        }
    } else if (q < h && p < w_) {
#pragma unroll
        for (int c = 0; c < 3; ++c) dst[c * plane + (long)q * w_ + p] = v[c];
    }
}

bool head_supported(const vcg_conv_desc* d) {
    return d->cout == 1 && d->stride == 1 && d->kh == d->kw && (d->kh == 3 || d->kh == 4) && d->cin % 8 == 0 && d->cin <= 512 &&
           (long)d->h * d->w * d->cin * 2 <= 0xFFFFFFE0l;
}

int head_check(const vcg_conv_desc* d) {
    if (d->n <= 0 || d->cin <= 0 || d->h <= 0 || d->w <= 0 || d->oh <= 0 || d->ow <= 0) return VCG_E_SHAPE;
    if (!head_supported(d)) return VCG_E_UNSUPPORTED;
    if (d->pad_top < 0 || d->pad_left < 0 || d->pad_top >= d->kh || d->pad_left >= d->kw) return VCG_E_SHAPE;
    if (d->oh > d->h + d->pad_top || d->ow > d->w + d->pad_left) return VCG_E_SHAPE;
    return VCG_OK;
}

HeadParams head_params(const vcg_conv_desc* d, int rows, int cols) {
    HeadParams p{};
    p.n = d->n; p.cin = d->cin; p.h = d->h; p.w_ = d->w; p.oh = d->oh; p.ow = d->ow; p.pad_top = d->pad_top; p.pad_left = d->pad_left;
    p.segs = ceil_div(cols, d->kh == 4 ? seg_len<4>() : seg_len<3>());
    p.jobs = d->n * rows * p.segs;
    return p;
}

constexpr int HEAD_WGRAD_BLOCKS = 256;

}  // namespace

extern "C" {

int vcg_conv2d_cout1_nhwc_bf16_fwd(const vcg_conv_desc* d, const void* x, const float* w_hwio, const float* bias, float* y, vcg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VCG_CHECK_PTR(d); VCG_CHECK_PTR(x); VCG_CHECK_PTR(w_hwio); VCG_CHECK_PTR(y);
    if (int e = head_check(d)) return e;
    HeadParams p = head_params(d, d->oh, d->ow);
    p.x = (const unsigned char*)x; p.w = w_hwio; p.bias = bias; p.y = y;
    const int grid = ceil_div(p.jobs, 4);
    if (d->kh == 4) cout1_fwd_bf16_kernel<4><<<grid, 256, 0, stream>>>(p);
    else cout1_fwd_bf16_kernel<3><<<grid, 256, 0, stream>>>(p);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_conv2d_cout1_nhwc_bf16_dgrad(const vcg_conv_desc* d, const float* dy, const float* w_hwio, void* dx, vcg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VCG_CHECK_PTR(d); VCG_CHECK_PTR(dy); VCG_CHECK_PTR(w_hwio); VCG_CHECK_PTR(dx);
    if (int e = head_check(d)) return e;
    HeadParams p = head_params(d, d->h, d->w);
    p.x = (const unsigned char*)dx; p.w = w_hwio; p.y = const_cast<float*>(dy);
    const int grid = ceil_div(p.jobs, 4);
    if (d->kh == 4) cout1_dgrad_bf16_kernel<4><<<grid, 256, 0, stream>>>(p);
    else cout1_dgrad_bf16_kernel<3><<<grid, 256, 0, stream>>>(p);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

size_t vcg_conv2d_cout1_nhwc_bf16_wgrad_workspace_bytes(const vcg_conv_desc* d) {
    if (d == nullptr || d->cin <= 0 || d->kh <= 0 || d->kw <= 0) return 0;
    return (size_t)HEAD_WGRAD_BLOCKS * ((size_t)d->kh * d->kw * d->cin + 1) * sizeof(float);
}

int vcg_conv2d_cout1_nhwc_bf16_wgrad(const vcg_conv_desc* d, const void* x, const float* dy, float* dw_hwio, float* dbias, void* ws, size_t ws_bytes,
                                     vcg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VCG_CHECK_PTR(d); VCG_CHECK_PTR(x); VCG_CHECK_PTR(dy); VCG_CHECK_PTR(dw_hwio); VCG_CHECK_PTR(ws);
    if (int e = head_check(d)) return e;
    if (ws_bytes < vcg_conv2d_cout1_nhwc_bf16_wgrad_workspace_bytes(d)) return VCG_E_WORKSPACE;
    HeadParams p = head_params(d, d->h, d->w);
    p.x = (const unsigned char*)x; p.y = const_cast<float*>(dy); p.ws = (float*)ws;
    const int grid = ceil_div(p.jobs, 4) < HEAD_WGRAD_BLOCKS ? ceil_div(p.jobs, 4) : HEAD_WGRAD_BLOCKS;
    const int kk = d->kh * d->kw;
    const size_t lds = ((size_t)kk * 512 + 4) * sizeof(float);
    if (d->kh == 4) cout1_wgrad_bf16_kernel<4><<<grid, 256, lds, stream>>>(p);
    else cout1_wgrad_bf16_kernel<3><<<grid, 256, lds, stream>>>(p);
    VCG_LAUNCH_CHECK();
    const int count = kk * d->cin;
    cout1_wgrad_reduce_kernel<<<ceil_div(count + 1, 32), 256, 0, stream>>>((const float*)ws, grid, count, dw_hwio, dbias);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

static bool first_dgrad_supported(const vcg_conv_desc* d) {
    return d->cin == 3 && d->cout == 64 && d->pad_top == 1 && d->pad_left == 1 &&
           ((d->kh == 4 && d->kw == 4 && d->stride == 2) || (d->kh == 3 && d->kw == 3 && d->stride == 1));
}

static void first_dgrad_virtual(const vcg_conv_desc* d, vcg_conv_desc* v) {
    // the virtual convolution: dz [n][oh][ow][64] -> t [n][qh][qw][64], 3x3 stride 1, pad 1 before
    v->n = d->n; v->cin = 64; v->h = d->oh; v->w = d->ow; v->cout = 64;
    v->oh = d->stride == 2 ? (d->h + 1) / 2 : d->h;
    v->ow = d->stride == 2 ? (d->w + 1) / 2 : d->w;
    v->kh = v->kw = 3; v->stride = 1; v->pad_top = v->pad_left = 1;
}

size_t vcg_conv3ch_bf16_dgrad_workspace_bytes(const vcg_conv_desc* d) {
    if (d == nullptr || d->n <= 0 || d->h <= 0 || d->w <= 0) return 0;
    vcg_conv_desc v;
    first_dgrad_virtual(d, &v);
    return align_up((size_t)9 * 64 * 64 * sizeof(float), 256) + align_up(vcg_conv_frag_bf16_bytes(9, 64, 64), 256) +
           (size_t)v.n * v.oh * v.ow * 64 * 2 + 256;
}

int vcg_conv3ch_bf16_dgrad(const vcg_conv_desc* d, const void* dz, const float* w_hwio, float* dx, void* ws, size_t ws_bytes, vcg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VCG_CHECK_PTR(d); VCG_CHECK_PTR(dz); VCG_CHECK_PTR(w_hwio); VCG_CHECK_PTR(dx); VCG_CHECK_PTR(ws);
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->oh <= 0 || d->ow <= 0 || d->n > 65535) return VCG_E_SHAPE;
    if (!first_dgrad_supported(d)) return VCG_E_UNSUPPORTED;
    if (ws_bytes < vcg_conv3ch_bf16_dgrad_workspace_bytes(d)) return VCG_E_WORKSPACE;
    vcg_conv_desc v;
    first_dgrad_virtual(d, &v);
    unsigned char* base = (unsigned char*)ws;
    base += (256 - ((size_t)base & 255)) & 255;
    float* wv = (float*)base;
    unsigned char* wfrag = base + align_up((size_t)9 * 64 * 64 * sizeof(float), 256);
    unsigned char* t = wfrag + align_up(vcg_conv_frag_bf16_bytes(9, 64, 64), 256);
    build_first_dgrad_kernel<<<9 * 64 * 64 / 256, 256, 0, stream>>>(w_hwio, d->kh, d->stride, d->cout, wv);
    VCG_LAUNCH_CHECK();
    if (int e = vcg_pack_conv_frag_bf16(wv, 9, 64, 64, 0, wfrag, stream)) return e;
    if (int e = vcg_conv2d_nhwc_bf16_fwd(&v, dz, wfrag, nullptr, VCG_ACT_NONE, 0.f, t, stream)) return e;
    const long px = (long)v.oh * v.ow;
    unshuffle_phases_kernel<<<dim3((unsigned)((px + 255) / 256), d->n), 256, 0, stream>>>((const __bf16*)t, v.oh, v.ow, d->h, d->w, d->stride, dx);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

}  // extern "C"
