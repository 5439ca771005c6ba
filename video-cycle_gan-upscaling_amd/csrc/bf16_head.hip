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
    const int c0 = lane * 8;
#pragma unroll
    for (int t = 0; t < K * K; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) wr[t][j] = c0 < p.cin ? p.w[t * p.cin + c0 + j] : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------
// forward: y[n][oy][ox] = bias + sum_{ky,kx,c} x[n][oy+ky-pt][ox+kx-pl][c] w[ky][kx][c]
// ---------------------------------------------------------------------------------------------------------------
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
// A wave keeps its 8 K^2 sums in registers over all its jobs; the four waves of a workgroup add up in LDS in wave order and the
// workgroup writes one record; a second kernel sums the records in a fixed order (deterministic).
template <int K>
__global__ __launch_bounds__(256) void cout1_wgrad_bf16_kernel(const HeadParams p) {
    constexpr int H_SEG = seg_len<K>();
    extern __shared__ float red[];                              // [K*K*8][64] + 1
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float acc[K * K][8];
#pragma unroll
    for (int t = 0; t < K * K; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
    float dsum = 0.f;
    const long img_bytes = (long)p.h * p.w_ * p.cin * 2;
    const long plane = (long)p.oh * p.ow;
    const unsigned lane_off = lane * 8 < p.cin ? (unsigned)lane * 16u : VCG_OOB;
    for (int job = blockIdx.x * 4 + wv; job < p.jobs; job += gridDim.x * 4) {
        const int seg = job % p.segs, t2 = job / p.segs, oy = t2 % p.oh, n = t2 / p.oh;
        const int x0 = seg * H_SEG;
        const vcg_rsrc rx = make_rsrc(p.x + n * img_bytes, (unsigned long)img_bytes);
        const int gxl = x0 + lane;
        const float dyv = lane < H_SEG && gxl < p.ow ? p.y[n * plane + (long)oy * p.ow + gxl] : 0.f;      // lane j = dy[n][oy][x0 + j]
        dsum += dyv;
        bf16x8 win[K][K];
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
#pragma unroll
        for (int j0 = 0; j0 < H_SEG; j0 += K) {
#pragma unroll
            for (int jj = 0; jj < K; ++jj) {
                load_col((jj + K - 1) % K, x0 + j0 + jj - p.pad_left + K - 1);
                const float g = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dyv), j0 + jj));
#pragma unroll
                for (int kx = 0; kx < K; ++kx)
#pragma unroll
                    for (int ky = 0; ky < K; ++ky) {
                        const bf16x8 v = win[(jj + kx) % K][ky];
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[ky * K + kx][j] = fmaf((float)v[j], g, acc[ky * K + kx][j]);
                    }
            }
        }
    }
    dsum = wave_sum(dsum);
    for (int w = 0; w < 4; ++w) {
        if (wv == w) {
#pragma unroll
            for (int t = 0; t < K * K; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float* q = red + (t * 8 + j) * 64 + lane;
                    *q = w == 0 ? acc[t][j] : *q + acc[t][j];
                }
            if (lane == 0) red[K * K * 512] = w == 0 ? dsum : red[K * K * 512] + dsum;
        }
        __syncthreads();
    }
    float* out = p.ws + (long)blockIdx.x * (K * K * p.cin + 1);
    for (int i = threadIdx.x; i < K * K * p.cin; i += 256) {
        const int t = i / p.cin, c = i - t * p.cin;
        out[i] = red[(t * 8 + (c & 7)) * 64 + (c >> 3)];
    }
    if (threadIdx.x == 0) out[K * K * p.cin] = red[K * K * 512];
}

__global__ __launch_bounds__(256) void cout1_wgrad_reduce_kernel(const float* __restrict__ ws, int blocks, int count, float* __restrict__ dw,
                                                                 float* __restrict__ db) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i > count) return;
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    const float* src = ws + i;
    const long stride = count + 1;
    int k = 0;
    for (; k + 4 <= blocks; k += 4)
#pragma unroll
        for (int u = 0; u < 4; ++u) s4[u] += src[(k + u) * stride];
    for (; k < blocks; ++k) s4[0] += src[k * stride];
    const float s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    if (i < count) dw[i] = s;
    else if (db) db[0] = s;
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
    HeadParams p = head_params(d, d->oh, d->ow);
    p.x = (const unsigned char*)x; p.y = const_cast<float*>(dy); p.ws = (float*)ws;
    const int grid = ceil_div(p.jobs, 4) < HEAD_WGRAD_BLOCKS ? ceil_div(p.jobs, 4) : HEAD_WGRAD_BLOCKS;
    const int kk = d->kh * d->kw;
    const size_t lds = ((size_t)kk * 512 + 1) * sizeof(float);
    if (d->kh == 4) cout1_wgrad_bf16_kernel<4><<<grid, 256, lds, stream>>>(p);
    else cout1_wgrad_bf16_kernel<3><<<grid, 256, lds, stream>>>(p);
    VCG_LAUNCH_CHECK();
    const int count = kk * d->cin;
    cout1_wgrad_reduce_kernel<<<ceil_div(count + 1, 256), 256, 0, stream>>>((const float*)ws, grid, count, dw_hwio, dbias);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

}  // extern "C"
