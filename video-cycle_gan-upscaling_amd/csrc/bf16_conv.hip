// bf16-storage kernels (configs C3-C5 of BASELINE.json): activations bf16 NHWC, weights bf16 packed
// [tap][out-channel][in-channel], fp32 accumulation on v_mfma_f32_32x32x16_bf16, fp32 epilogue.
//
// Why NHWC here while the fp32 path is NCHW: the bf16 MFMA takes 8 consecutive k per lane, and k of the
// implicit GEMM is the input channel -- with channels innermost one lane's operand fragment is ONE 16-byte LDS
// read, and a pixel's 64 channels are one 128-byte line in HBM.
//
// Reference ops served: Conv2D 3x3 'same' of residual_block / prefinal conv (upscaling/upscaler/model.py:19,22,283)
// with the inference-mode BatchNormalization folded into a per-channel scale/shift (model.py:20,23,284),
// PReLU (model.py:21) and the block's Add (model.py:25,285) fused into the epilogue.
#include "vcg_common.hpp"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    // lane l (r = l&31, h = l>>5): A[row r][k = 8h+j], B[k = 8h+j][col r], j = 0..7; D as the f32 form
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------
// layout / packing helpers
// ---------------------------------------------------------------------------------------------------------------
__global__ void pack_kernel_bf16(const float* __restrict__ w, __bf16* __restrict__ out, int taps, int a, int b,
                                 int transpose, int flip) {
    // out[tap'][i][j] (j contiguous) = transpose ? w[tap][j][i] : w[tap][i][j];  tap' = flip ? taps-1-tap : tap
    const long total = (long)taps * a * b;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int j = (int)(idx % b);
        const int i = (int)((idx / b) % a);
        const int t = (int)(idx / ((long)a * b));
        const int ts = flip ? taps - 1 - t : t;
        const float v = transpose ? w[((long)ts * b + j) * a + i] : w[((long)ts * a + i) * b + j];
        out[idx] = (__bf16)v;
    }
}

__global__ void f32_nchw_to_bf16_nhwc_kernel(const float* __restrict__ x, __bf16* __restrict__ y, int n, int c, int hw) {
    // one block per (n, 64-pixel segment): coalesced reads along pixels, coalesced writes along channels
    __shared__ float tile[64][65];
    const int p0 = blockIdx.x * 64, img = blockIdx.y;
    for (int c0 = 0; c0 < c; c0 += 64) {
        for (int e = threadIdx.x; e < 64 * 64; e += blockDim.x) {
            const int cc = e >> 6, pp = e & 63;
            float v = 0.f;
            if (c0 + cc < c && p0 + pp < hw) v = x[((long)img * c + c0 + cc) * hw + p0 + pp];
            tile[cc][pp] = v;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * 64; e += blockDim.x) {
            const int pp = e >> 6, cc = e & 63;
            if (c0 + cc < c && p0 + pp < hw) y[((long)img * hw + p0 + pp) * c + c0 + cc] = (__bf16)tile[cc][pp];
        }
        __syncthreads();
    }
}

__global__ void bf16_nhwc_to_f32_nchw_kernel(const __bf16* __restrict__ x, float* __restrict__ y, int n, int c, int hw) {
    __shared__ float tile[64][65];
    const int p0 = blockIdx.x * 64, img = blockIdx.y;
    for (int c0 = 0; c0 < c; c0 += 64) {
        for (int e = threadIdx.x; e < 64 * 64; e += blockDim.x) {
            const int pp = e >> 6, cc = e & 63;
            float v = 0.f;
            if (c0 + cc < c && p0 + pp < hw) v = (float)x[((long)img * hw + p0 + pp) * c + c0 + cc];
            tile[cc][pp] = v;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * 64; e += blockDim.x) {
            const int cc = e >> 6, pp = e & 63;
            if (c0 + cc < c && p0 + pp < hw) y[((long)img * c + c0 + cc) * hw + p0 + pp] = tile[cc][pp];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3 stride-1 'same' convolution, 64 -> 64 channels: the generator trunk
// ---------------------------------------------------------------------------------------------------------------
// Persistent workgroups of 8 waves.  LDS: all 9x64x64 weights (72 KiB, loaded once) + one 18x34-pixel halo tile
// (76.5 KiB).  Both images hold 128-byte rows (one pixel / one out-channel x 64 in-channels) whose eight 16-byte
// chunks are XOR-swizzled with (index>>1)&7: the ds_read_b128 of the 16-lane groups {0-3,12-15,20-27},
// {4-11,16-19,28-31} (MI355X_MICROARCH.md, LDS) then touches all 64 banks once, for every tap shift.
// Wave w owns output rows 2w, 2w+1 of the 16x32 tile: a 64-channel x 64-pixel accumulator (4 MFMA tiles, 64 VGPRs)
// fed by 2 weight + 2 pixel fragment reads per 4 MFMAs.  The next tile's halo is fetched into registers before the
// MFMA loop and written to LDS after it.
constexpr int TR = 16, TC = 32, HR = TR + 2, HC = TC + 2;
constexpr int ROWB = HC * 128;              // bytes per halo row
constexpr int XB = HR * ROWB;               // 78336
constexpr int WB = 9 * 64 * 128;            // 73728
constexpr int NCHUNK = HR * HC * 8;         // 16-byte chunks per halo tile (4896)
constexpr int NPRE = (NCHUNK + 511) / 512;  // per-thread prefetch registers (10)

struct C3Params {
    const uint4* x;
    const uint4* w;
    __bf16* y;
    const float* scale;
    const float* shift;
    const float* alpha;
    const __bf16* res;
    int n, h, w_, tiles_x, tiles_y, total;
    int act;
    float act_alpha;
};

__global__ __launch_bounds__(512, 1) void conv3x3_c64_bf16_kernel(C3Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wl = smem;
    unsigned char* xl = smem + WB;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, hh = lane >> 5;

    for (int c = tid; c < 9 * 64 * 8; c += 512) {
        const int chunk = c & 7, co = (c >> 3) & 63, tap = c >> 9;
        *(uint4*)(wl + tap * 8192 + co * 128 + ((chunk ^ ((co >> 1) & 7)) << 4)) = p.w[c];
    }

    int aoff[4], boff[3][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) aoff[s] = r * 128 + (((2 * s + hh) ^ ((r >> 1) & 7)) << 4);
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int pos = r + dx;
            boff[dx][s] = pos * 128 + (((2 * s + hh) ^ ((pos >> 1) & 7)) << 4);
        }

    // staging coordinates of this thread's chunks (tile-independent)
    int srow[NPRE], scol[NPRE];
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
        int c = tid + 512 * i;
        if (c >= NCHUNK) c = NCHUNK - 1;      // duplicates the last chunk (same data, same address): harmless
        const int pix = c >> 3;
        srow[i] = pix / HC;
        scol[i] = pix - srow[i] * HC;
    }

    uint4 pre[NPRE];
    auto fetch = [&](int tile) {
        const int txi = tile % p.tiles_x, t2 = tile / p.tiles_x, tyi = t2 % p.tiles_y, img = t2 / p.tiles_y;
        const int y0 = tyi * TR - 1, x0 = txi * TC - 1;
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int gy = y0 + srow[i], gx = x0 + scol[i];
            const bool ok = (unsigned)gy < (unsigned)p.h && (unsigned)gx < (unsigned)p.w_;
            const int cy = min(max(gy, 0), p.h - 1), cx = min(max(gx, 0), p.w_ - 1);
            const int chunk = min(tid + 512 * i, NCHUNK - 1) & 7;
            uint4 v = p.x[((long)(img * p.h + cy) * p.w_ + cx) * 8 + chunk];
            if (!ok) v = make_uint4(0, 0, 0, 0);
            pre[i] = v;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            int c = tid + 512 * i;
            if (c >= NCHUNK) c = NCHUNK - 1;
            const int chunk = c & 7;
            *(uint4*)(xl + (srow[i] * HC + scol[i]) * 128 + ((chunk ^ ((scol[i] >> 1) & 7)) << 4)) = pre[i];
        }
    };

    int tile = blockIdx.x;
    if (tile < p.total) fetch(tile);
    stash();
    __syncthreads();

    const unsigned char* xb = xl + (wv * 2) * ROWB;
    for (; tile < p.total; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        if (next < p.total) fetch(next);

        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const unsigned char* wa = wl + (dy * 3 + dx) * 8192 + aoff[s];
                    const bf16x8 a0 = *(const bf16x8*)(wa);
                    const bf16x8 a1 = *(const bf16x8*)(wa + 4096);
                    const bf16x8 b0 = *(const bf16x8*)(xb + dy * ROWB + boff[dx][s]);
                    const bf16x8 b1 = *(const bf16x8*)(xb + (dy + 1) * ROWB + boff[dx][s]);
                    acc[0][0] = mfma_bf16(a0, b0, acc[0][0]);
                    acc[0][1] = mfma_bf16(a0, b1, acc[0][1]);
                    acc[1][0] = mfma_bf16(a1, b0, acc[1][0]);
                    acc[1][1] = mfma_bf16(a1, b1, acc[1][1]);
                }

        // epilogue: y = act(acc * scale + shift) + residual, bf16, 4 consecutive channels (8 bytes) per store
        {
            const int txi = tile % p.tiles_x, t2 = tile / p.tiles_x, tyi = t2 % p.tiles_y, img = t2 / p.tiles_y;
            const int gx = txi * TC + r;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = mt * 32 + 8 * g + 4 * hh;
                    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f}, al = {p.act_alpha, p.act_alpha, p.act_alpha, p.act_alpha};
                    if (p.scale) sc = *(const f32x4*)(p.scale + co);
                    if (p.shift) sh = *(const f32x4*)(p.shift + co);
                    if (p.alpha) al = *(const f32x4*)(p.alpha + co);
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt) {
                        const int gy = tyi * TR + wv * 2 + pt;
                        if (gy < p.h && gx < p.w_) {
                            const long o = ((long)(img * p.h + gy) * p.w_ + gx) * 64 + co;
                            f32x4 v;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                float t = acc[mt][pt][4 * g + j] * sc[j] + sh[j];
                                if (p.act == VCG_ACT_LRELU || p.act == VCG_ACT_PRELU) t = fmaxf(t, 0.f) + al[j] * fminf(t, 0.f);
                                v[j] = t;
                            }
                            if (p.res) {
                                const bf16x4 rr = *(const bf16x4*)(p.res + o);
#pragma unroll
                                for (int j = 0; j < 4; ++j) v[j] += (float)rr[j];
                            }
                            *(bf16x4*)(p.y + o) = __builtin_convertvector(v, bf16x4);
                        }
                    }
                }
        }

        __syncthreads();
        if (next < p.total) stash();
        __syncthreads();
    }
}

}  // namespace

extern "C" {

int vcg_pack_conv_kernel_bf16(const void* w, int32_t taps, int32_t a, int32_t b, int32_t transpose, int32_t flip, void* out,
                              hipStream_t stream) {
    VCG_CHECK_PTR(w);
    VCG_CHECK_PTR(out);
    if (taps <= 0 || a <= 0 || b <= 0) return VCG_E_SHAPE;
    const long total = (long)taps * a * b;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    pack_kernel_bf16<<<blocks, 256, 0, stream>>>((const float*)w, (__bf16*)out, taps, a, b, transpose, flip);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_f32_nchw_to_bf16_nhwc(const void* x, void* y, int32_t n, int32_t c, int32_t h, int32_t w, hipStream_t stream) {
    VCG_CHECK_PTR(x);
    VCG_CHECK_PTR(y);
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0) return VCG_E_SHAPE;
    dim3 grid(ceil_div(h * w, 64), n);
    f32_nchw_to_bf16_nhwc_kernel<<<grid, 256, 0, stream>>>((const float*)x, (__bf16*)y, n, c, h * w);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_bf16_nhwc_to_f32_nchw(const void* x, void* y, int32_t n, int32_t c, int32_t h, int32_t w, hipStream_t stream) {
    VCG_CHECK_PTR(x);
    VCG_CHECK_PTR(y);
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0) return VCG_E_SHAPE;
    dim3 grid(ceil_div(h * w, 64), n);
    bf16_nhwc_to_f32_nchw_kernel<<<grid, 256, 0, stream>>>((const __bf16*)x, (float*)y, n, c, h * w);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_conv2d_bf16_fwd(const vcg_conv_desc* d, const void* x, const void* w_packed, void* y, const vcg_epilogue_bf16* ep,
                        hipStream_t stream) {
    VCG_CHECK_PTR(d);
    VCG_CHECK_PTR(x);
    VCG_CHECK_PTR(w_packed);
    VCG_CHECK_PTR(y);
    if (d->n <= 0 || d->h <= 0 || d->w <= 0) return VCG_E_SHAPE;
    if (d->oh != d->h || d->ow != d->w) return VCG_E_SHAPE;
    const int act = ep ? ep->act : VCG_ACT_NONE;
    if (act == VCG_ACT_TANH) return VCG_E_UNSUPPORTED;
    if (act == VCG_ACT_PRELU && (!ep || !ep->prelu_alpha)) return VCG_E_NULL;
    if (d->cin == 64 && d->cout == 64 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad_top == 1 && d->pad_left == 1) {
        C3Params p;
        p.x = (const uint4*)x;
        p.w = (const uint4*)w_packed;
        p.y = (__bf16*)y;
        p.scale = ep ? (const float*)ep->scale : nullptr;
        p.shift = ep ? (const float*)ep->shift : nullptr;
        p.alpha = (ep && act == VCG_ACT_PRELU) ? (const float*)ep->prelu_alpha : nullptr;
        p.res = ep ? (const __bf16*)ep->residual : nullptr;
        p.n = d->n;
        p.h = d->h;
        p.w_ = d->w;
        p.tiles_x = ceil_div(d->w, TC);
        p.tiles_y = ceil_div(d->h, TR);
        p.total = p.n * p.tiles_x * p.tiles_y;
        p.act = act;
        p.act_alpha = ep ? ep->act_alpha : 0.f;
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute((const void*)conv3x3_c64_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WB + XB);
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
        const int grid = p.total < 256 ? p.total : 256;
        conv3x3_c64_bf16_kernel<<<grid, 512, WB + XB, stream>>>(p);
        VCG_LAUNCH_CHECK();
        return VCG_OK;
    }
    return VCG_E_UNSUPPORTED;
}

}  // extern "C"
