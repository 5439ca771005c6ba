// Implicit-GEMM direct convolution on v_mfma_f32_32x32x2_f32 (exact fp32), NCHW.
//
// GEMM view:  M = output channels, N = output pixels (32 consecutive x of one row per MFMA tile),
//             K = (input channel, ky, kx).  The two k of one MFMA are two consecutive input
//             channels at the same tap, so every operand read is "lane base + immediate".
// Workgroup:  256 threads = 4 waves; tile = 64 output channels x (8 rows x 32 cols);
//             wave w owns rows 2w, 2w+1  ->  2x2 MFMA tiles, 64 accumulator VGPRs.
// Staging:    one LDS buffer holding CK input channels' halo tile + their weights; the next
//             chunk's global loads are issued into registers BEFORE the MFMA loop of the current
//             chunk and written to LDS after it (issue-early / write-late).
// Serves:     Conv2D forward (model.py:19,22,275,283,839-871), stride-1 Conv2D dgrad (flipped
//             taps, per-tap transposed kernel), Conv2DTranspose dgrad (stride-2 conv).
//
// Second kernel (conv_smallm): for <=32/KW output channels (final 9x9 256->3 conv, model.py:290;
// PatchGAN's 1-channel head; dgrad into 3-channel images) the M dimension of the MFMA carries
// (channel, kx) pairs and the kx shift-sum is done in the epilogue through LDS, so the matrix core
// runs at 27/32 row utilisation instead of 3/32.
#include "vcg_common.hpp"

namespace {

struct ConvParams {
    const float* x;
    const float* w;        // [T][cin][cout]  (m contiguous)
    float* y;
    const float* bias;
    const float* prelu;
    const float* residual;
    int n, cin, h, w_, cout, oh, ow, pad_top, pad_left;
    int tiles_x, tiles_y, co_blocks;
    int act;
    float alpha;
    int flip;
    // small-M kernel only: weight element (mch, kc, tap) at w[tap*ws_t + mch*ws_m + kc*ws_k]
    int ws_t, ws_m, ws_k;
};

template <int KH, int KW, int S, int CK>
struct ConvCfg {
    static constexpr int ROWS = 8;
    static constexpr int IH = (ROWS - 1) * S + KH;
    static constexpr int IW = 31 * S + KW;
    static constexpr int PLANE = IH * IW;
    static constexpr int T = KH * KW;
    static constexpr int IN_ELEMS = CK * PLANE;
    static constexpr int W_ELEMS = CK * T * 64;
    static constexpr int IN_PT = (IN_ELEMS + 255) / 256;
    static constexpr int W_PT = (W_ELEMS + 255) / 256;
    static constexpr size_t LDS_BYTES = (size_t)(IN_ELEMS + W_ELEMS) * sizeof(float);
};

template <int KH, int KW, int S, int CK>
__global__ __launch_bounds__(256) void conv_fwd_kernel(const ConvParams p) {
    using C = ConvCfg<KH, KW, S, CK>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;               // [CK][IH][IW]
    float* s_w = smem + C::IN_ELEMS;  // [CK][T][64]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    int b = blockIdx.x;
    const int cb = b % p.co_blocks; b /= p.co_blocks;
    const int tx = b % p.tiles_x;   b /= p.tiles_x;
    const int ty = b % p.tiles_y;   b /= p.tiles_y;
    const int n = b;
    const int ox0 = tx * 32, oy0 = ty * C::ROWS, co0 = cb * 64;
    const int gy0 = oy0 * S - p.pad_top, gx0 = ox0 * S - p.pad_left;
    const float* xn = p.x + (size_t)n * p.cin * p.h * p.w_;

    float rin[C::IN_PT], rw[C::W_PT];

    auto load_chunk = [&](int ci0) {
#pragma unroll
        for (int i = 0; i < C::IN_PT; ++i) {
            const int e = tid + i * 256;
            float v = 0.f;
            if (e < C::IN_ELEMS) {
                const int ci = e / C::PLANE, rem = e % C::PLANE;
                const int r = rem / C::IW, c = rem % C::IW;
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
                const int ch = ci0 + ci, tap = p.flip ? (C::T - 1 - t) : t;
                if (ch < p.cin && co0 + m < p.cout) v = p.w[((size_t)tap * p.cin + ch) * p.cout + co0 + m];
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

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const float* bbase = s_in + half * C::PLANE + (wv * 2 * S) * C::IW + l31 * S;
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
#pragma unroll
            for (int ky = 0; ky < KH; ++ky) {
#pragma unroll
                for (int kx = 0; kx < KW; ++kx) {
                    const int t = ky * KW + kx;
                    const float a0 = abase[(cp * 2 * C::T + t) * 64];
                    const float a1 = abase[(cp * 2 * C::T + t) * 64 + 32];
                    const float b0 = bbase[cp * 2 * C::PLANE + ky * C::IW + kx];
                    const float b1 = bbase[cp * 2 * C::PLANE + (S + ky) * C::IW + kx];
                    acc[0][0] = mfma32(a0, b0, acc[0][0]);
                    acc[0][1] = mfma32(a0, b1, acc[0][1]);
                    acc[1][0] = mfma32(a1, b0, acc[1][0]);
                    acc[1][1] = mfma32(a1, b1, acc[1][1]);
                }
            }
        }
    }

    // epilogue: y = act(acc + bias) + residual
    const int ox = ox0 + l31;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + mt * 32 + mfma_row(r, lane);
            if (co >= p.cout) continue;
            const float bv = p.bias ? p.bias[co] : 0.f;
            const float al = (p.act == VCG_ACT_PRELU) ? p.prelu[co] : p.alpha;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int oy = oy0 + wv * 2 + rt;
                if (oy < p.oh && ox < p.ow) {
                    const size_t idx = (((size_t)n * p.cout + co) * p.oh + oy) * p.ow + ox;
                    float v = apply_act(acc[mt][rt][r] + bv, p.act, al);
                    if (p.residual) v += p.residual[idx];
                    p.y[idx] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// small-M variant (stride 1): MFMA rows = (out channel, kx); K = (in channel, ky)
// tile: 8 rows x 64 x' positions -> 64-(KW-1) valid output columns per row
// ------------------------------------------------------------------------------------------------
template <int KH, int KW, int CK>
struct SmallCfg {
    static constexpr int ROWS = 8;
    static constexpr int XW = 64;
    static constexpr int OUTW = XW - (KW - 1);
    static constexpr int IH = ROWS + KH - 1;
    static constexpr int PLANE = IH * XW;
    static constexpr int IN_ELEMS = CK * PLANE;
    static constexpr int W_ELEMS = CK * KH * 32;
    static constexpr int IN_PT = (IN_ELEMS + 255) / 256;
    static constexpr int W_PT = (W_ELEMS + 255) / 256;
    static constexpr int PST = XW + 1;                       // scratch row stride
    static constexpr int SCRATCH = 4 * 32 * PST;             // one 32 x 64 tile per wave
    static constexpr int LDS_ELEMS = (IN_ELEMS + W_ELEMS) > SCRATCH ? (IN_ELEMS + W_ELEMS) : SCRATCH;
    static constexpr size_t LDS_BYTES = (size_t)LDS_ELEMS * sizeof(float);
};

template <int KH, int KW, int CK>
__global__ __launch_bounds__(256) void conv_smallm_kernel(const ConvParams p) {
    using C = SmallCfg<KH, KW, CK>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;               // [CK][IH][64]
    float* s_w = smem + C::IN_ELEMS;  // [CK][KH][32]   (row mi = mch*KW + kx)

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    int b = blockIdx.x;
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y; b /= p.tiles_y;
    const int n = b;
    const int ox0 = tx * C::OUTW, oy0 = ty * C::ROWS;
    const int gy0 = oy0 - p.pad_top, gx0 = ox0 - p.pad_left;
    const float* xn = p.x + (size_t)n * p.cin * p.h * p.w_;
    const int mrows = p.cout * KW;

    float rin[C::IN_PT], rw[C::W_PT];
    auto load_chunk = [&](int ci0) {
#pragma unroll
        for (int i = 0; i < C::IN_PT; ++i) {
            const int e = tid + i * 256;
            float v = 0.f;
            if (e < C::IN_ELEMS) {
                const int ci = e / C::PLANE, rem = e % C::PLANE;
                const int r = rem / C::XW, c = rem % C::XW;
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
                const int mi = e & 31, q = e >> 5;
                const int ky = q % KH, ci = q / KH;
                const int ch = ci0 + ci;
                if (ch < p.cin && mi < mrows) {
                    const int mch = mi / KW, kx = mi % KW;
                    int tap = ky * KW + kx;
                    if (p.flip) tap = KH * KW - 1 - tap;
                    v = p.w[(size_t)tap * p.ws_t + (size_t)mch * p.ws_m + (size_t)ch * p.ws_k];
                }
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

    f32x16 acc[2][2];  // [row tile][x tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const float* bbase = s_in + half * C::PLANE + (wv * 2) * C::XW + l31;
    const float* abase = s_w + half * KH * 32 + l31;

    const int nchunks = (p.cin + CK - 1) / CK;
    load_chunk(0);
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();
        store_chunk();
        __syncthreads();
        if (c + 1 < nchunks) load_chunk((c + 1) * CK);
#pragma unroll
        for (int cp = 0; cp < CK / 2; ++cp) {
#pragma unroll
            for (int ky = 0; ky < KH; ++ky) {
                const float a = abase[(cp * 2 * KH + ky) * 32];
                const float b00 = bbase[cp * 2 * C::PLANE + ky * C::XW];
                const float b01 = bbase[cp * 2 * C::PLANE + ky * C::XW + 32];
                const float b10 = bbase[cp * 2 * C::PLANE + (ky + 1) * C::XW];
                const float b11 = bbase[cp * 2 * C::PLANE + (ky + 1) * C::XW + 32];
                acc[0][0] = mfma32(a, b00, acc[0][0]);
                acc[0][1] = mfma32(a, b01, acc[0][1]);
                acc[1][0] = mfma32(a, b10, acc[1][0]);
                acc[1][1] = mfma32(a, b11, acc[1][1]);
            }
        }
    }

    // epilogue: out[mch][x] = sum_kx P[mch*KW+kx][x+kx]; one output row at a time through LDS
    float* s_p = smem + wv * 32 * C::PST;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        __syncthreads();  // staging LDS (or previous scratch) no longer read by anyone
#pragma unroll
        for (int xt = 0; xt < 2; ++xt)
#pragma unroll
            for (int r = 0; r < 16; ++r) s_p[mfma_row(r, lane) * C::PST + xt * 32 + l31] = acc[rt][xt][r];
        __syncthreads();
        const int oy = oy0 + wv * 2 + rt;
        const int ox = ox0 + lane;
        if (lane < C::OUTW && oy < p.oh && ox < p.ow) {
            for (int mch = 0; mch < p.cout; ++mch) {
                float v = 0.f;
#pragma unroll
                for (int kx = 0; kx < KW; ++kx) v += s_p[(mch * KW + kx) * C::PST + lane + kx];
                const float bv = p.bias ? p.bias[mch] : 0.f;
                const float al = (p.act == VCG_ACT_PRELU) ? p.prelu[mch] : p.alpha;
                const size_t idx = (((size_t)n * p.cout + mch) * p.oh + oy) * p.ow + ox;
                v = apply_act(v + bv, p.act, al);
                if (p.residual) v += p.residual[idx];
                p.y[idx] = v;
            }
        }
    }
}

template <typename Kern>
int launch_with_lds(Kern kern, int grid, size_t lds, const ConvParams& p, hipStream_t st) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, p);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

template <int KH, int KW, int S, int CK>
int launch_conv(ConvParams p, hipStream_t st) {
    using C = ConvCfg<KH, KW, S, CK>;
    p.tiles_x = ceil_div(p.ow, 32);
    p.tiles_y = ceil_div(p.oh, C::ROWS);
    p.co_blocks = ceil_div(p.cout, 64);
    const long grid = (long)p.tiles_x * p.tiles_y * p.co_blocks * p.n;
    if (grid <= 0 || grid > 0x7fffffffL) return VCG_E_SHAPE;
    return launch_with_lds(conv_fwd_kernel<KH, KW, S, CK>, (int)grid, C::LDS_BYTES, p, st);
}

template <int KH, int KW, int CK>
int launch_smallm(ConvParams p, hipStream_t st) {
    using C = SmallCfg<KH, KW, CK>;
    p.tiles_x = ceil_div(p.ow, C::OUTW);
    p.tiles_y = ceil_div(p.oh, C::ROWS);
    p.co_blocks = 1;
    const long grid = (long)p.tiles_x * p.tiles_y * p.n;
    if (grid <= 0 || grid > 0x7fffffffL) return VCG_E_SHAPE;
    return launch_with_lds(conv_smallm_kernel<KH, KW, CK>, (int)grid, C::LDS_BYTES, p, st);
}

}  // namespace

// Generic entry used by the C ABI wrappers in api.hip.  Weight tensor w is [T][cin][cout] with cout
// contiguous (or addressed by ws_* strides for the small-M kernel).
int vcg_internal_conv(const float* x, const float* w, float* y, int n, int cin, int h, int wd, int cout,
                      int oh, int ow, int kh, int kw, int stride, int pad_top, int pad_left, int flip,
                      const vcg_epilogue* ep, int smallm, int ws_t, int ws_m, int ws_k, hipStream_t st) {
    ConvParams p{};
    p.x = x; p.w = w; p.y = y;
    p.bias = ep ? ep->bias : nullptr;
    p.prelu = ep ? ep->prelu_alpha : nullptr;
    p.residual = ep ? ep->residual : nullptr;
    p.act = ep ? ep->act : VCG_ACT_NONE;
    p.alpha = ep ? ep->act_alpha : 0.f;
    if (p.act == VCG_ACT_PRELU && p.prelu == nullptr) return VCG_E_NULL;
    p.n = n; p.cin = cin; p.h = h; p.w_ = wd; p.cout = cout; p.oh = oh; p.ow = ow;
    p.pad_top = pad_top; p.pad_left = pad_left; p.flip = flip;
    p.ws_t = ws_t; p.ws_m = ws_m; p.ws_k = ws_k;
    if (n <= 0 || cin <= 0 || cout <= 0 || oh <= 0 || ow <= 0) return VCG_E_SHAPE;
    if (smallm) {
        if (stride != 1 || cout * kw > 32) return VCG_E_UNSUPPORTED;
        if (kh == 9 && kw == 9) return launch_smallm<9, 9, 8>(p, st);
        if (kh == 4 && kw == 4) return launch_smallm<4, 4, 8>(p, st);
        if (kh == 3 && kw == 3) return launch_smallm<3, 3, 8>(p, st);
        if (kh == 5 && kw == 5) return launch_smallm<5, 5, 8>(p, st);
        return VCG_E_UNSUPPORTED;
    }
    if (kh == 3 && kw == 3 && stride == 1) return launch_conv<3, 3, 1, 8>(p, st);
    if (kh == 3 && kw == 3 && stride == 2) return launch_conv<3, 3, 2, 8>(p, st);
    if (kh == 4 && kw == 4 && stride == 1) return launch_conv<4, 4, 1, 8>(p, st);
    if (kh == 4 && kw == 4 && stride == 2) return launch_conv<4, 4, 2, 4>(p, st);
    if (kh == 5 && kw == 5 && stride == 1) return launch_conv<5, 5, 1, 8>(p, st);
    if (kh == 5 && kw == 5 && stride == 2) return launch_conv<5, 5, 2, 4>(p, st);
    if (kh == 9 && kw == 9 && stride == 1) return launch_conv<9, 9, 1, 4>(p, st);
    return VCG_E_UNSUPPORTED;
}
