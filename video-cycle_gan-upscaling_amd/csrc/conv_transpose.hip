// Stride-2 transposed convolution as sub-pixel phase convolutions on v_mfma_f32_32x32x2_f32, NCHW.
//
//   out[m][2q+p] = sum_{kc, k : phase(k)=p} W[k][kc][m] * in[kc][q - d(k)]      (per axis)
//   phase(k) = (k - cb) mod 2,  d(k) = (k - cb - phase(k)) / 2,  cb = crop-before of the full
//   transposed convolution (TF SAME: floor((K-2)/2); for a strided conv's dgrad: its pad-before).
//
// A wave owns one input row q_y and 32 consecutive q_x and accumulates all four output phases
// (2x2 output pixels per input position) for 64 output channels: 8 MFMA tiles = 128 accumulator
// VGPRs.  The two x-phases of a lane are interleaved in registers and stored as one float2, so the
// 2x up-sampled rows are written fully coalesced.
// Serves: Conv2DTranspose forward (model.py:72 via :288) and stride-2 Conv2D dgrad (D blocks 2-9,
// model.py:843-871; PatchGAN k4 s2).
#include "vcg_common.hpp"

namespace {

struct ConvTParams {
    const float* x;   // [n][cin][h][w]
    const float* w;   // [T][cin][cout]   (cout contiguous)
    float* y;         // [n][cout][oh][ow]
    const float* bias;
    const float* prelu;
    const float* residual;
    int n, cin, h, w_, cout, oh, ow;
    int tiles_x, tiles_y, co_blocks;
    int act;
    float alpha;
};

constexpr int phase_of(int k, int cb) { return (((k - cb) % 2) + 2) % 2; }
constexpr int d_of(int k, int cb) { return (k - cb - phase_of(k, cb)) / 2; }
constexpr int dmin_of(int K, int cb) { int m = 1000; for (int k = 0; k < K; ++k) m = d_of(k, cb) < m ? d_of(k, cb) : m; return m; }
constexpr int dmax_of(int K, int cb) { int m = -1000; for (int k = 0; k < K; ++k) m = d_of(k, cb) > m ? d_of(k, cb) : m; return m; }

template <int K, int CBY, int CBX, int CK>
struct ConvTCfg {
    static constexpr int QROWS = 4;  // one q-row per wave
    static constexpr int DMINY = dmin_of(K, CBY), DMAXY = dmax_of(K, CBY);
    static constexpr int DMINX = dmin_of(K, CBX), DMAXX = dmax_of(K, CBX);
    static constexpr int NDY = DMAXY - DMINY + 1, NDX = DMAXX - DMINX + 1;
    static constexpr int NR = QROWS + DMAXY - DMINY;
    static constexpr int NC = 32 + DMAXX - DMINX;
    static constexpr int PLANE = NR * NC;
    static constexpr int T = K * K;
    static constexpr int IN_ELEMS = CK * PLANE;
    static constexpr int W_ELEMS = CK * T * 64;
    static constexpr int IN_PT = (IN_ELEMS + 255) / 256;
    static constexpr int W_PT = (W_ELEMS + 255) / 256;
    static constexpr size_t LDS_BYTES = (size_t)(IN_ELEMS + W_ELEMS) * sizeof(float);
};

template <int K, int CBY, int CBX, int CK>
__global__ __launch_bounds__(256) void convt_kernel(const ConvTParams p) {
    using C = ConvTCfg<K, CBY, CBX, CK>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;               // [CK][NR][NC]
    float* s_w = smem + C::IN_ELEMS;  // [CK][T][64]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    int b = blockIdx.x;
    const int cb = b % p.co_blocks; b /= p.co_blocks;
    const int tx = b % p.tiles_x;   b /= p.tiles_x;
    const int ty = b % p.tiles_y;   b /= p.tiles_y;
    const int n = b;
    const int qx0 = tx * 32, qy0 = ty * C::QROWS, co0 = cb * 64;
    const int gy0 = qy0 - C::DMAXY, gx0 = qx0 - C::DMAXX;
    const float* xn = p.x + (size_t)n * p.cin * p.h * p.w_;

    float rin[C::IN_PT], rw[C::W_PT];
    auto load_chunk = [&](int ci0) {
#pragma unroll
        for (int i = 0; i < C::IN_PT; ++i) {
            const int e = tid + i * 256;
            float v = 0.f;
            if (e < C::IN_ELEMS) {
                const int ci = e / C::PLANE, rem = e % C::PLANE;
                const int r = rem / C::NC, c = rem % C::NC;
                const int gy = gy0 + r, gx = gx0 + c, ch = ci0 + ci;
                if (ch < p.cin && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w_)
                    v = xn[((size_t)ch * p.h + gy) * p.w_ + gx];
            }
            rin[i] = v;
        }
#pragma unroll
        for (int i = 0; i < C::W_PT; ++i) {
            const int e = tid + i * 256;
            float v = 0.f;
            if (e < C::W_ELEMS) {
                const int m = e & 63, q = e >> 6;
                const int t = q % C::T, ci = q / C::T;
                const int ch = ci0 + ci;
                if (ch < p.cin && co0 + m < p.cout) v = p.w[((size_t)t * p.cin + ch) * p.cout + co0 + m];
            }
            rw[i] = v;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < C::IN_PT; ++i) {
            const int e = tid + i * 256;
            if (e < C::IN_ELEMS) s_in[e] = rin[i];
        }
#pragma unroll
        for (int i = 0; i < C::W_PT; ++i) {
            const int e = tid + i * 256;
            if (e < C::W_ELEMS) s_w[e] = rw[i];
        }
    };

    f32x16 acc[2][2][2];  // [py][px][mt]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][c][m][r] = 0.f;

    // input row for tap-offset d: tile row = wv + DMAXY - d ; col = l31 + DMAXX - d
    const float* bbase = s_in + half * C::PLANE + wv * C::NC + l31;
    const float* abase = s_w + half * C::T * 64 + l31;

    const int nchunks = (p.cin + CK - 1) / CK;
    load_chunk(0);
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();
        store_chunk();
        __syncthreads();
        if (c + 1 < nchunks) load_chunk((c + 1) * CK);
#pragma unroll
        for (int cp = 0; cp < CK / 2; ++cp) {
            float bv[C::NDY][C::NDX];
#pragma unroll
            for (int iy = 0; iy < C::NDY; ++iy)
#pragma unroll
                for (int ix = 0; ix < C::NDX; ++ix)
                    bv[iy][ix] = bbase[cp * 2 * C::PLANE + (C::DMAXY - (C::DMINY + iy)) * C::NC +
                                       (C::DMAXX - (C::DMINX + ix))];
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    constexpr int dummy = 0; (void)dummy;
                    const int t = ky * K + kx;
                    const int py = phase_of(ky, CBY), px = phase_of(kx, CBX);
                    const int iy = d_of(ky, CBY) - C::DMINY, ix = d_of(kx, CBX) - C::DMINX;
                    const float a0 = abase[(cp * 2 * C::T + t) * 64];
                    const float a1 = abase[(cp * 2 * C::T + t) * 64 + 32];
                    acc[py][px][0] = mfma32(a0, bv[iy][ix], acc[py][px][0]);
                    acc[py][px][1] = mfma32(a1, bv[iy][ix], acc[py][px][1]);
                }
            }
        }
    }

    const int qy = qy0 + wv, qx = qx0 + l31;
    const bool vec_ok = (p.ow % 2) == 0;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + mt * 32 + mfma_row(r, lane);
            if (co >= p.cout) continue;
            const float bvv = p.bias ? p.bias[co] : 0.f;
            const float al = (p.act == VCG_ACT_PRELU) ? p.prelu[co] : p.alpha;
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const int oy = 2 * qy + py, ox = 2 * qx;
                if (oy >= p.oh || ox >= p.ow) continue;
                const size_t idx = (((size_t)n * p.cout + co) * p.oh + oy) * p.ow + ox;
                float v0 = apply_act(acc[py][0][mt][r] + bvv, p.act, al);
                float v1 = apply_act(acc[py][1][mt][r] + bvv, p.act, al);
                if (vec_ok) {
                    if (p.residual) {
                        const float2 rr = *reinterpret_cast<const float2*>(p.residual + idx);
                        v0 += rr.x; v1 += rr.y;
                    }
                    *reinterpret_cast<float2*>(p.y + idx) = make_float2(v0, v1);
                } else {
                    if (p.residual) v0 += p.residual[idx];
                    p.y[idx] = v0;
                    if (ox + 1 < p.ow) {
                        if (p.residual) v1 += p.residual[idx + 1];
                        p.y[idx + 1] = v1;
                    }
                }
            }
        }
    }
}

template <int K, int CBY, int CBX, int CK>
int launch_convt(ConvTParams p, hipStream_t st) {
    using C = ConvTCfg<K, CBY, CBX, CK>;
    p.tiles_x = ceil_div(ceil_div(p.ow, 2), 32);
    p.tiles_y = ceil_div(ceil_div(p.oh, 2), C::QROWS);
    p.co_blocks = ceil_div(p.cout, 64);
    const long grid = (long)p.tiles_x * p.tiles_y * p.co_blocks * p.n;
    if (grid <= 0 || grid > 0x7fffffffL) return VCG_E_SHAPE;
    auto kern = convt_kernel<K, CBY, CBX, CK>;
    if (C::LDS_BYTES > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3((int)grid), dim3(256), C::LDS_BYTES, st, p);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

}  // namespace

// w: [T][cin][cout] (cout contiguous); (cby, cbx) crop-before per axis.
int vcg_internal_convt(const float* x, const float* w, float* y, int n, int cin, int h, int wd, int cout,
                       int oh, int ow, int k, int cby, int cbx, const vcg_epilogue* ep, hipStream_t st) {
    ConvTParams p{};
    p.x = x; p.w = w; p.y = y;
    p.bias = ep ? ep->bias : nullptr;
    p.prelu = ep ? ep->prelu_alpha : nullptr;
    p.residual = ep ? ep->residual : nullptr;
    p.act = ep ? ep->act : VCG_ACT_NONE;
    p.alpha = ep ? ep->act_alpha : 0.f;
    if (p.act == VCG_ACT_PRELU && p.prelu == nullptr) return VCG_E_NULL;
    p.n = n; p.cin = cin; p.h = h; p.w_ = wd; p.cout = cout; p.oh = oh; p.ow = ow;
    if (n <= 0 || cin <= 0 || cout <= 0 || oh <= 0 || ow <= 0) return VCG_E_SHAPE;
#define VCG_CT(K_, A_, B_, CK_) if (k == K_ && cby == A_ && cbx == B_) return launch_convt<K_, A_, B_, CK_>(p, st)
    VCG_CT(3, 0, 0, 8); VCG_CT(3, 0, 1, 8); VCG_CT(3, 1, 0, 8); VCG_CT(3, 1, 1, 8);
    VCG_CT(4, 1, 1, 8);
    VCG_CT(5, 1, 1, 8); VCG_CT(5, 1, 2, 8); VCG_CT(5, 2, 1, 8); VCG_CT(5, 2, 2, 8);
#undef VCG_CT
    return VCG_E_UNSUPPORTED;
}
