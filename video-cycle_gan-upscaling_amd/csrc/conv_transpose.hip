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
#include <type_traits>

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

template <int K, int CBY, int CBX, int CK, int MT>
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
    static constexpr int MB = 32 * MT;   // output channels per workgroup
    static constexpr int W_ELEMS = CK * T * MB;
    static constexpr int IN_PT = (IN_ELEMS + 255) / 256;
    static constexpr int W_PT = (W_ELEMS + 255) / 256;
    static constexpr size_t LDS_BYTES = (size_t)(IN_ELEMS + W_ELEMS) * sizeof(float);
};

template <int K, int CBY, int CBX, int CK, int MT>
__global__ __launch_bounds__(256, (MT == 1 ? 2 : 1)) void convt_kernel(const ConvTParams p) {
    using C = ConvTCfg<K, CBY, CBX, CK, MT>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;               // [CK][NR][NC]
    float* s_w = smem + C::IN_ELEMS;  // [CK][T][MB]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    int b = blockIdx.x;
    const int cb = b % p.co_blocks; b /= p.co_blocks;
    const int tx = b % p.tiles_x;   b /= p.tiles_x;
    const int ty = b % p.tiles_y;   b /= p.tiles_y;
    const int n = b;
    const int qx0 = tx * 32, qy0 = ty * C::QROWS, co0 = cb * C::MB;
    const int gy0 = qy0 - C::DMAXY, gx0 = qx0 - C::DMAXX;
    const float* xn = p.x + (size_t)n * p.cin * p.h * p.w_;

    float rin[C::IN_PT], rw[C::W_PT];
    // byte offsets inside a chunk's [CK][h][w] block, computed once (VCG_OOB = zero padding); staging loads go through
    // range-checked buffer descriptors whose base moves with the chunk (vcg_common.hpp): no mask behind a load
    unsigned in_off[C::IN_PT];
#pragma unroll
    for (int i = 0; i < C::IN_PT; ++i) {
        const int e = tid + i * 256;
        unsigned off = VCG_OOB;
        if (e < C::IN_ELEMS) {
            const int ci = e / C::PLANE, rem = e % C::PLANE;
            const int r = rem / C::NC, c = rem % C::NC;
            const int gy = gy0 + r, gx = gx0 + c;
            if (gy >= 0 && gy < p.h && gx >= 0 && gx < p.w_) off = 4u * (unsigned)((ci * p.h + gy) * p.w_ + gx);
        }
        in_off[i] = off;
    }
    const int hw = p.h * p.w_;
    // weight element e -> row q = (ci, tap), column m; rows of channels past cin alias later rows (finite, meet zero
    // inputs) or fall off the end (-> 0); columns past cout are never stored
    unsigned w_off[C::W_PT];
#pragma unroll
    for (int i = 0; i < C::W_PT; ++i) {
        const int e = tid + i * 256;
        const int m = e % C::MB, q = e / C::MB;
        const int t = q % C::T, ci = q / C::T;
        const int col = co0 + m < p.cout ? co0 + m : p.cout - 1;
        w_off[i] = e < C::W_ELEMS ? 4u * (unsigned)((t * p.cin + ci) * p.cout + col) : VCG_OOB;
    }
    const size_t wbytes = (size_t)C::T * p.cin * p.cout * sizeof(float);
    auto load_chunk = [&](int ci0) {
        const vcg_rsrc rx = make_rsrc(xn + (size_t)ci0 * hw, (size_t)(p.cin - ci0) * hw * sizeof(float));
#pragma unroll
        for (int i = 0; i < C::IN_PT; ++i) rin[i] = buf_load(rx, in_off[i]);
        const vcg_rsrc rwt = make_rsrc(p.w + (size_t)ci0 * p.cout, wbytes - (size_t)ci0 * p.cout * sizeof(float));
#pragma unroll
        for (int i = 0; i < C::W_PT; ++i) rw[i] = buf_load(rwt, w_off[i]);
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

    f32x16 acc[2][2][MT];  // [py][px][mt]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][c][m][r] = 0.f;

    // input row for tap-offset d: tile row = wv + DMAXY - d ; col = l31 + DMAXX - d
    const float* bbase = s_in + half * C::PLANE + wv * C::NC + l31;
    const float* abase = s_w + half * C::T * C::MB + l31;

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
            // all A operands of this channel pair first (one batch of LDS reads), then the MFMAs: the
            // compiler overlaps the next pair's reads with this pair's 2*T MFMAs
            float av[C::T][MT];
#pragma unroll
            for (int t = 0; t < C::T; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) av[t][mt] = abase[(cp * 2 * C::T + t) * C::MB + mt * 32];
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const int t = ky * K + kx;
                    const int py = phase_of(ky, CBY), px = phase_of(kx, CBX);
                    const int iy = d_of(ky, CBY) - C::DMINY, ix = d_of(kx, CBX) - C::DMINX;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[py][px][mt] = mfma32(av[t][mt], bv[iy][ix], acc[py][px][mt]);
                }
            }
        }
    }

    // epilogue: the two x-phases of a lane form one float2.  Interior tiles of even-width outputs take a
    // guard-free path; none / LeakyReLU / PReLU share  v >= 0 ? v : v*slope  (slope 1 = identity).
    const int qy = qy0 + wv, qx = qx0 + l31;
    const bool vec_ok = (p.ow % 2) == 0;
    const bool full = vec_ok && (co0 + C::MB <= p.cout) && (2 * (qy0 + C::QROWS) <= p.oh) && (2 * (qx0 + 32) <= p.ow);
    const int plane = p.oh * p.ow;
    const size_t img = (size_t)n * p.cout * plane;
    float* yimg = p.y + img;
    const float* rimg = p.residual ? p.residual + img : nullptr;
    const int ob = (co0 + 4 * half) * plane + (2 * qy) * p.ow + 2 * qx;
    const float* bp = p.bias ? p.bias : vcg_zero_word;
    const int bmask = p.bias ? ~0 : 0;
    const bool is_prelu = p.act == VCG_ACT_PRELU;
    const float* ap = is_prelu ? p.prelu : vcg_zero_word;
    const int amask = is_prelu ? ~0 : 0;
    const float slope_u = (p.act == VCG_ACT_LRELU) ? p.alpha : 1.f;
    auto emit = [&](auto guard_tag, auto res_tag) {
        constexpr bool GUARD = decltype(guard_tag)::value, RES = decltype(res_tag)::value;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mt * 32 + mfma_row(r, lane);
                const int co = co0 + row;
                const bool co_ok = !GUARD || co < p.cout;
                const int cs = co_ok ? co : co0;
                const float bvv = bp[cs & bmask];
                const float pa = ap[cs & amask];
                const float al = is_prelu ? pa : slope_u;
#pragma unroll
                for (int py = 0; py < 2; ++py) {
                    const int o = ob + (row - 4 * half) * plane + py * p.ow;
                    float v0 = acc[py][0][mt][r] + bvv, v1 = acc[py][1][mt][r] + bvv;
                    v0 = v0 >= 0.f ? v0 : v0 * al;
                    v1 = v1 >= 0.f ? v1 : v1 * al;
                    if (!GUARD) {
                        if (RES) { const float2 rr = *reinterpret_cast<const float2*>(rimg + o); v0 += rr.x; v1 += rr.y; }
                        *reinterpret_cast<float2*>(yimg + o) = make_float2(v0, v1);
                    } else {
                        const int oy = 2 * qy + py, ox = 2 * qx;
                        if (co_ok && oy < p.oh && ox < p.ow) {
                            if (vec_ok) {
                                if (RES) { const float2 rr = *reinterpret_cast<const float2*>(rimg + o); v0 += rr.x; v1 += rr.y; }
                                *reinterpret_cast<float2*>(yimg + o) = make_float2(v0, v1);
                            } else {
                                if (RES) v0 += rimg[o];
                                yimg[o] = v0;
                                if (ox + 1 < p.ow) {
                                    if (RES) v1 += rimg[o + 1];
                                    yimg[o + 1] = v1;
                                }
                            }
                        }
                    }
                }
            }
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    if (full) {
        if (p.residual) emit(F_{}, T_{}); else emit(F_{}, F_{});
    } else {
        if (p.residual) emit(T_{}, T_{}); else emit(T_{}, F_{});
    }
}

template <int K, int CBY, int CBX, int CK, int MT>
int launch_convt(ConvTParams p, hipStream_t st) {
    using C = ConvTCfg<K, CBY, CBX, CK, MT>;
    p.tiles_x = ceil_div(ceil_div(p.ow, 2), 32);
    p.tiles_y = ceil_div(ceil_div(p.oh, 2), C::QROWS);
    p.co_blocks = ceil_div(p.cout, C::MB);
    const long grid = (long)p.tiles_x * p.tiles_y * p.co_blocks * p.n;
    if (grid <= 0 || grid > 0x7fffffffL) return VCG_E_SHAPE;
    auto kern = convt_kernel<K, CBY, CBX, CK, MT>;
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
    if (p.act == VCG_ACT_TANH) return VCG_E_UNSUPPORTED;
    p.n = n; p.cin = cin; p.h = h; p.w_ = wd; p.cout = cout; p.oh = oh; p.ow = ow;
    if (n <= 0 || cin <= 0 || cout <= 0 || oh <= 0 || ow <= 0) return VCG_E_SHAPE;
#define VCG_CT(K_, A_, B_, CK_, MT_) if (k == K_ && cby == A_ && cbx == B_) return launch_convt<K_, A_, B_, CK_, MT_>(p, st)
    // MT = 1: 32 output channels per workgroup, 64 accumulator registers -> 2+ workgroups per CU overlap each
    // other's staging and barriers (measured faster than 64 channels at one wave per SIMD)
    VCG_CT(3, 0, 0, 16, 1); VCG_CT(3, 0, 1, 16, 1); VCG_CT(3, 1, 0, 16, 1); VCG_CT(3, 1, 1, 16, 1);
    VCG_CT(4, 1, 1, 16, 1);
    VCG_CT(5, 1, 1, 8, 2); VCG_CT(5, 1, 2, 8, 2); VCG_CT(5, 2, 1, 8, 2); VCG_CT(5, 2, 2, 8, 2);
#undef VCG_CT
    return VCG_E_UNSUPPORTED;
}
