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
#include <type_traits>

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
    float* stats;          // conv_fwd_kernel<..., STATS>: [n * tiles_y * tiles_x][2][cout] per-tile sums of the accumulators and of their squares
};

template <int KH, int KW, int S, int CK, int XT>
struct ConvCfg {
    static constexpr int ROWS = 8;
    static constexpr int IH = (ROWS - 1) * S + KH;
    static constexpr int IW = (32 * XT - 1) * S + KW;
    static constexpr int PLANE = IH * IW;
    static constexpr int T = KH * KW;
    static constexpr int IN_ELEMS = CK * PLANE;
    static constexpr int W_ELEMS = CK * T * 64;
    static constexpr int IN_PT = (IN_ELEMS + 255) / 256;
    static constexpr int W_PT = (W_ELEMS + 255) / 256;
    static constexpr size_t LDS_BYTES = (size_t)(IN_ELEMS + W_ELEMS) * sizeof(float);
};

// ---- shared epilogue: y = act(acc + bias) + residual for a 64-channel x (8 rows x 32*XT cols) tile ----------
template <int XT>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x16 (&acc)[2][2][XT], int n, int co0, int oy0, int ox0,
                                              int wv, int lane) {
    constexpr int ROWS_ = 8;
    const int half = lane >> 5, l31 = lane & 31;
    // epilogue: y = act(acc + bias) + residual.  Interior tiles take a guard-free path.
    const int ox = ox0 + l31;
    const bool full = (co0 + 64 <= p.cout) && (oy0 + ROWS_ <= p.oh) && (ox0 + 32 * XT <= p.ow);
    // 32-bit element offsets inside this image's [cout][oh][ow] block: one scalar base + a vector offset
    const int plane = p.oh * p.ow;
    const size_t img = (size_t)n * p.cout * plane;
    float* yb = p.y + img;
    const float* rb = p.residual ? p.residual + img : nullptr;
    const int ob = (co0 + 4 * half) * plane + (oy0 + wv * 2) * p.ow + ox;
    // none / LeakyReLU / PReLU share one straight-line form  v >= 0 ? v : v*slope  (slope 1 = identity);
    // tanh is only offered by the small-M kernel (the API rejects it here).
    const float* bp = p.bias ? p.bias : vcg_zero_word;
    const int bmask = p.bias ? ~0 : 0;
    const bool is_prelu = p.act == VCG_ACT_PRELU;
    const float* ap = is_prelu ? p.prelu : vcg_zero_word;
    const int amask = is_prelu ? ~0 : 0;
    const float slope_u = (p.act == VCG_ACT_LRELU) ? p.alpha : 1.f;
    auto emit = [&](auto guard_tag, auto res_tag) {
        constexpr bool GUARD = decltype(guard_tag)::value, RES = decltype(res_tag)::value;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mt * 32 + mfma_row(r, lane);
                const int co = co0 + row;
                const bool co_ok = !GUARD || co < p.cout;
                const int cs = co_ok ? co : co0;
                const float bv = bp[cs & bmask];
                const float pa = ap[cs & amask];
                const float al = is_prelu ? pa : slope_u;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
                    for (int xt = 0; xt < XT; ++xt) {
                        const int o = ob + (row - 4 * half) * plane + rt * p.ow + xt * 32;
                        if (!GUARD || (co_ok && oy0 + wv * 2 + rt < p.oh && ox + xt * 32 < p.ow)) {
                            float v = acc[mt][rt][xt][r] + bv;
                            v = v >= 0.f ? v : v * al;
                            if (RES) v += rb[o];
                            yb[o] = v;
                        }
                    }
                }
            }
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    if (full) {
        if (rb) emit(F_{}, T_{}); else emit(F_{}, F_{});
    } else {
        if (rb) emit(T_{}, T_{}); else emit(T_{}, F_{});
    }
}

// STATS epilogue: for the BatchNormalization / instance norm behind the convolution (model.py:19-25, 840) the workgroup also leaves, per output
// channel, the sum of its tile's accumulators and of their squares -- the output minus the bias: a shift that keeps E[d^2] - E[d]^2 from
// cancelling -- as one record per tile; vcg_norm_finalize_partials_shifted sums the records in a fixed order.  The statistics pass over the
// output (one more read of the tensor, two more launches per normalisation) is gone.  A lane's 32 (channel) x 2 partial sums are
// reduce-scattered over the 32 lanes of its half-wave, the four waves (row pairs) meet in LDS.
template <int XT>
__device__ __forceinline__ void conv_stats_epilogue(const ConvParams& p, f32x16 (&acc)[2][2][XT], float* smem, int rec, int co0, int oy0, int ox0,
                                                    int wv, int lane, int tid) {
    const int half = lane >> 5, l31 = lane & 31;
    float sv[32], sq[32];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int xt = 0; xt < XT; ++xt) {
                    const bool ok = oy0 + wv * 2 + rt < p.oh && ox0 + l31 + xt * 32 < p.ow;
                    const float v = ok ? acc[mt][rt][xt][r] : 0.f;
                    a += v;
                    b = fmaf(v, v, b);
                }
            sv[mt * 16 + r] = a;
            sq[mt * 16 + r] = b;
        }
    const float ts = half_wave_reduce_scatter32(sv, l31), tq = half_wave_reduce_scatter32(sq, l31);
    // lane (l31, half) now holds value index l31 = (mt, r) summed over the half-wave's 32 columns: channel mt*32 + mfma_row(r)
    const int ch = (l31 >> 4) * 32 + (l31 & 3) + 8 * ((l31 & 15) >> 2) + 4 * half;
    __syncthreads();                                   // every wave is done with the operand tiles: the LDS is free
    smem[(0 * 4 + wv) * 64 + ch] = ts;
    smem[(1 * 4 + wv) * 64 + ch] = tq;
    __syncthreads();
    if (tid < 128) {
        const int st = tid >> 6, c = tid & 63;
        const float t = (smem[(st * 4 + 0) * 64 + c] + smem[(st * 4 + 1) * 64 + c]) + (smem[(st * 4 + 2) * 64 + c] + smem[(st * 4 + 3) * 64 + c]);
        if (co0 + c < p.cout) p.stats[((size_t)rec * 2 + st) * p.cout + co0 + c] = t;
    }
}

template <int KH, int KW, int S, int CK, int XT, bool STATS = false>
__global__ __launch_bounds__(256, (KH >= 9 ? 1 : 2)) void conv_fwd_kernel(const ConvParams p) {
    using C = ConvCfg<KH, KW, S, CK, XT>;
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
    const int ox0 = tx * 32 * XT, oy0 = ty * C::ROWS, co0 = cb * 64;
    const int gy0 = oy0 * S - p.pad_top, gx0 = ox0 * S - p.pad_left;
    const float* xn = p.x + (size_t)n * p.cin * p.h * p.w_;

    float rin[C::IN_PT], rw[C::W_PT];

    // per-thread input BYTE offsets inside a chunk's [CK][h][w] block, computed once (VCG_OOB = zero padding); a chunk
    // only moves the descriptor's base (scalar), whose record count also cuts off the channels past cin: the
    // staging loop is one buffer_load per element, no per-element VALU and nothing that depends on the loaded data
    unsigned in_off[C::IN_PT];
#pragma unroll
    for (int i = 0; i < C::IN_PT; ++i) {
        const int e = tid + i * 256;
        unsigned off = VCG_OOB;
        if (e < C::IN_ELEMS) {
            const int ci = e / C::PLANE, rem = e % C::PLANE;
            const int r = rem / C::IW, c = rem % C::IW;
            const int gy = gy0 + r, gx = gx0 + c;
            if (gy >= 0 && gy < p.h && gx >= 0 && gx < p.w_) off = 4u * (unsigned)((ci * p.h + gy) * p.w_ + gx);
        }
        in_off[i] = off;
    }
    const int hw = p.h * p.w_;
    // weights: row q = (ci, tap) of the chunk is wave-uniform, the lane is the output channel.  Rows of channels past
    // cin alias the next tap's rows (finite values that meet zero inputs) or fall off the tensor's end (range check -> 0);
    // lanes past cout read the last valid column: their output rows are never stored
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    const unsigned wcol = 4u * (unsigned)(co0 + lane < p.cout ? co0 + lane : p.cout - 1);
    const size_t wbytes = (size_t)C::T * p.cin * p.cout * sizeof(float);

    auto load_chunk = [&](int ci0) {
        const vcg_rsrc rx = make_rsrc(xn + (size_t)ci0 * hw, (size_t)(p.cin - ci0) * hw * sizeof(float));
#pragma unroll
        for (int i = 0; i < C::IN_PT; ++i) rin[i] = buf_load(rx, in_off[i]);
        const vcg_rsrc rwt = make_rsrc(p.w + (size_t)ci0 * p.cout, wbytes - (size_t)ci0 * p.cout * sizeof(float));
#pragma unroll
        for (int i = 0; i < C::W_PT; ++i) {
            const int q = wvu + 4 * i;
            const int t = q % C::T, ci = q / C::T;
            const int tap = p.flip ? (C::T - 1 - t) : t;
            rw[i] = buf_load(rwt, 4u * (unsigned)((tap * p.cin + ci) * p.cout) + wcol);
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

    f32x16 acc[2][2][XT];   // [co tile][row][x tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int x = 0; x < XT; ++x)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][x][r] = 0.f;

    const float* bbase = s_in + half * C::PLANE + (wv * 2 * S) * C::IW + l31 * S;
    const float* abase = s_w + half * C::T * 64 + l31;

    const int nchunks = (p.cin + CK - 1) / CK;
    load_chunk(0);
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();
        store_chunk();
        __syncthreads();
        if (c + 1 < nchunks) load_chunk((c + 1) * CK);   // in flight during this chunk's MFMAs
        // operand fragments are read one tap AHEAD of the MFMAs that use them (register double buffer, statically renamed
        // by the full unroll); sched_group_barrier pins the order "reads of tap t+1, then the 4*XT MFMAs of tap t", so a
        // wave's own LDS latency hides under its own MFMAs instead of relying on the SIMD's other wave
        constexpr int NTAP = (CK / 2) * C::T;
        auto rd = [&](int q, float (&a)[2], float (&b)[2][XT]) {
            const int cp = q / C::T, t = q % C::T, ky = t / KW, kx = t % KW;
            a[0] = abase[(cp * 2 * C::T + t) * 64];
            a[1] = abase[(cp * 2 * C::T + t) * 64 + 32];
#pragma unroll
            for (int xt = 0; xt < XT; ++xt) {
                b[0][xt] = bbase[cp * 2 * C::PLANE + ky * C::IW + kx + xt * 32 * S];
                b[1][xt] = bbase[cp * 2 * C::PLANE + (S + ky) * C::IW + kx + xt * 32 * S];
            }
        };
        float a_[2][2], b_[2][2][XT];
        rd(0, a_[0], b_[0]);
#pragma unroll
        for (int q = 0; q < NTAP; ++q) {
            const int cur = q & 1, nxt = cur ^ 1;
            if (q + 1 < NTAP) rd(q + 1, a_[nxt], b_[nxt]);
#pragma unroll
            for (int xt = 0; xt < XT; ++xt) {
                acc[0][0][xt] = mfma32(a_[cur][0], b_[cur][0][xt], acc[0][0][xt]);
                acc[0][1][xt] = mfma32(a_[cur][0], b_[cur][1][xt], acc[0][1][xt]);
                acc[1][0][xt] = mfma32(a_[cur][1], b_[cur][0][xt], acc[1][0][xt]);
                acc[1][1][xt] = mfma32(a_[cur][1], b_[cur][1][xt], acc[1][1][xt]);
            }
            if (q + 1 < NTAP) __builtin_amdgcn_sched_group_barrier(0x100, 1 + 2 * XT, 0);   // DS reads of the next tap first
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * XT, 0);                          // then this tap's MFMAs
        }
    }

    conv_epilogue<XT>(p, acc, n, co0, oy0, ox0, wv, lane);
    if (STATS) conv_stats_epilogue<XT>(p, acc, smem, (n * p.tiles_y + ty) * p.tiles_x + tx, co0, oy0, ox0, wv, lane, tid);
}

// ------------------------------------------------------------------------------------------------
// <= 3 input channels (first 9x9 conv on RGB frames, model.py:275; data gradient of the final 9x9 conv, :290):
// the generic kernel pairs CHANNELS in the MFMA's k dimension and would pad 3 channels to 4 (25 % of the
// matrix work on zeros).  Here the whole (channel, ky, kx) range is one flat K axis staged once, and an MFMA
// takes two CONSECUTIVE k: the second half-wave reads one element further along the tap row, with the two
// row / channel wrap-arounds handled by two more per-lane base pointers chosen at compile time.
// ------------------------------------------------------------------------------------------------
template <int KH, int KW, int S>
struct C3Cfg {
    static constexpr int ROWS = 8;
    static constexpr int IH = (ROWS - 1) * S + KH;
    static constexpr int IW = 31 * S + KW;
    static constexpr int PLANE = IH * IW;
    static constexpr int T = KH * KW;
    static constexpr int KT = 3 * T;                       // flat K
    static constexpr int NP = (KT + 1) / 2;                // MFMA k-pairs
    static constexpr int IN_ELEMS = 3 * PLANE;
    static constexpr int W_ROWS = 2 * NP;                  // weight rows incl. one zero row when KT is odd
    static constexpr int W_ELEMS = W_ROWS * 64;
    static constexpr int IN_PT = (IN_ELEMS + 255) / 256;
    static constexpr int W_PT = W_ELEMS / 256;
    static constexpr size_t LDS_BYTES = (size_t)(IN_ELEMS + W_ELEMS) * sizeof(float);
    static constexpr int off(int k) { return (k / T) * PLANE + ((k % T) / KW) * IW + (k % T) % KW; }
    static constexpr int D_ROW = IW - (KW - 1);                              // kx wraps to the next tap row
    static constexpr int D_CH = PLANE - (KH - 1) * IW - (KW - 1);            // last tap -> first tap of next channel
};

template <int KH, int KW, int S>
__global__ __launch_bounds__(256, 2) void conv_c3_kernel(const ConvParams p) {
    using C = C3Cfg<KH, KW, S>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;               // [3][IH][IW]
    float* s_w = smem + C::IN_ELEMS;  // [W_ROWS][64]

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

    // ---- stage everything once: input halo tile of all channels + the whole weight slab of this block
#pragma unroll
    for (int i = 0; i < C::IN_PT; ++i) {
        const int e = tid + i * 256;
        if (e < C::IN_ELEMS) {
            const int ci = e / C::PLANE, rem = e % C::PLANE;
            const int r = rem / C::IW, c = rem % C::IW;
            const int gy = gy0 + r, gx = gx0 + c;
            const bool ok = ci < p.cin && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w_;
            const float v = xn[ok ? (ci * p.h + gy) * p.w_ + gx : 0];
            s_in[e] = ok ? v : 0.f;
        }
    }
#pragma unroll 4
    for (int i = 0; i < C::W_PT; ++i) {
        const int e = tid + i * 256;
        const int m = e & 63, k = e >> 6;              // k = ci*T + tap
        const int ci = k / C::T, t = k % C::T;
        const int tap = p.flip ? (C::T - 1 - t) : t;
        const bool ok = k < C::KT && ci < p.cin && co0 + m < p.cout;
        const float v = p.w[ok ? (tap * p.cin + ci) * p.cout + co0 + m : 0];
        s_w[e] = ok ? v : 0.f;
    }
    __syncthreads();

    f32x16 acc[2][2][1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][0][r] = 0.f;

    const float* b0p = s_in + (wv * 2 * S) * C::IW + l31 * S;     // lanes 0-31: k even
    const float* bb1 = b0p + half * 1;                             // lanes 32-63: next tap in the row
    const float* bb2 = b0p + half * C::D_ROW;                      //              first tap of the next row
    const float* bb3 = b0p + half * C::D_CH;                       //              first tap of the next channel
    const float* abase = s_w + half * 64 + l31;
#pragma unroll
    for (int pr = 0; pr < C::NP; ++pr) {
        constexpr int dummy = 0; (void)dummy;
        const int k0 = 2 * pr, k1 = 2 * pr + 1;
        const int o0 = C::off(k0);
        const int d = (k1 < C::KT) ? (C::off(k1) - o0) : 1;       // past the end: weights are zero, any valid read
        const float* bb = (d == 1) ? bb1 : (d == C::D_ROW ? bb2 : bb3);
        const float a0 = abase[k0 * 64];
        const float a1 = abase[k0 * 64 + 32];
        const float v0 = bb[o0];
        const float v1 = bb[o0 + S * C::IW];
        acc[0][0][0] = mfma32(a0, v0, acc[0][0][0]);
        acc[0][1][0] = mfma32(a0, v1, acc[0][1][0]);
        acc[1][0][0] = mfma32(a1, v0, acc[1][0][0]);
        acc[1][1][0] = mfma32(a1, v1, acc[1][1][0]);
    }
    conv_epilogue<1>(p, acc, n, co0, oy0, ox0, wv, lane);
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
    // the staged rows are exactly 64 floats wide: row (ci, r) is wave-uniform and the lane is the column,
    // so the input addressing is scalar + lane
    const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gx = gx0 + lane;
    const bool col_ok = gx >= 0 && gx < p.w_;
    const int gxc = min(max(gx, 0), p.w_ - 1);
    const int hw = p.h * p.w_;
    // range-checked buffer loads: padding / missing channels get the offset VCG_OOB and read as 0 (vcg_common.hpp)
    const vcg_rsrc rx = make_rsrc(xn, (size_t)p.cin * hw * sizeof(float));
    const vcg_rsrc rwt = make_rsrc(p.w, (size_t)KH * KW * p.cin * p.cout * sizeof(float));
    auto load_chunk = [&](int ci0) {
#pragma unroll
        for (int i = 0; i < C::IN_PT; ++i) {
            const int q = wvu + 4 * i;                  // row index inside the chunk tile
            const int r = q % C::IH, ci = q / C::IH;
            const int gy = gy0 + r, ch = ci0 + ci;
            const bool ok = col_ok && ch < p.cin && gy >= 0 && gy < p.h;
            rin[i] = buf_load(rx, ok ? 4u * (unsigned)(ch * hw + gy * p.w_ + gxc) : VCG_OOB);
        }
#pragma unroll
        for (int i = 0; i < C::W_PT; ++i) {
            const int e = tid + i * 256;
            const int mi = e & 31, q = e >> 5;
            const int ky = q % KH, ci = q / KH;
            const int ch = ci0 + ci;
            const bool ok = e < C::W_ELEMS && ch < p.cin && mi < mrows;
            const int mch = mi / KW, kx = mi % KW;
            int tap = ky * KW + kx;
            if (p.flip) tap = KH * KW - 1 - tap;
            rw[i] = buf_load(rwt, ok ? 4u * (unsigned)(tap * p.ws_t + mch * p.ws_m + ch * p.ws_k) : VCG_OOB);
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

// ------------------------------------------------------------------------------------------------
// one output channel, stride 1 (the PatchGAN head: Conv2D(1, 4) on 512 channels): 0.5 GFLOP over 65 MB of input -- memory
// work, not matrix work.  On the small-M MFMA kernel it occupies 4 of an MFMA's 32 rows and is bound by the wasted rows
// (0.32 ms); here it is a plain FMA reduction: block = (image, output row, 64 columns), wave = a quarter of the input
// channels, lane = column; the K shifted loads of a row overlap in L1; the four partial sums meet in LDS in a fixed order.
// ------------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void conv_cout1_kernel(const ConvParams p) {
    __shared__ float red[4][64];
    const int tid = threadIdx.x, xl = tid & 63;
    const int cg = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = blockIdx.x;
    const int seg = b % p.tiles_x; b /= p.tiles_x;
    const int oy = b % p.oh;
    const int n = b / p.oh;
    const int ox = seg * 64 + xl, gx0 = ox - p.pad_left, gy0 = oy - p.pad_top;
    const int hw = p.h * p.w_;
    const vcg_rsrc rx = make_rsrc(p.x + (size_t)n * p.cin * hw, (size_t)p.cin * hw * sizeof(float));
    unsigned coff[K];                                            // byte offset of (row 0, column gx0 + kx) or out of range
#pragma unroll
    for (int kx = 0; kx < K; ++kx) coff[kx] = (unsigned)(gx0 + kx) < (unsigned)p.w_ ? 4u * (unsigned)(gx0 + kx) : VCG_OOB;
    float acc = 0.f;
#pragma unroll 1                                                 // (unrolled x4 hipcc makes it 3x slower)
    for (int c = cg; c < p.cin; c += 4) {
        float wr[K * K];
#pragma unroll
        for (int t = 0; t < K * K; ++t) wr[t] = p.w[(p.flip ? K * K - 1 - t : t) * p.ws_t + c * p.ws_k];       // wave-uniform: scalar loads
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int gy = gy0 + ky;
            const bool rok = (unsigned)gy < (unsigned)p.h;
            const unsigned rbase = 4u * (unsigned)(c * hw + gy * p.w_);
#pragma unroll
            for (int kx = 0; kx < K; ++kx) acc += buf_load(rx, rok && coff[kx] != VCG_OOB ? rbase + coff[kx] : VCG_OOB) * wr[ky * K + kx];
        }
    }
    red[cg][xl] = acc;
    __syncthreads();
    if (cg == 0 && ox < p.ow) {
        float v = ((red[0][xl] + red[1][xl]) + (red[2][xl] + red[3][xl])) + (p.bias ? p.bias[0] : 0.f);
        if (p.act == VCG_ACT_LRELU) v = v >= 0.f ? v : v * p.alpha;
        else if (p.act == VCG_ACT_TANH) v = tanhf(v);
        const size_t o = ((size_t)n * p.oh + oy) * p.ow + ox;
        if (p.residual) v += p.residual[o];
        p.y[o] = v;
    }
}

template <int K>
int launch_cout1(ConvParams p, hipStream_t st) {
    p.tiles_x = ceil_div(p.ow, 64);
    const long grid = (long)p.tiles_x * p.oh * p.n;
    if (grid <= 0 || grid > 0x7fffffffL) return VCG_E_SHAPE;
    hipLaunchKernelGGL(conv_cout1_kernel<K>, dim3((unsigned)grid), dim3(256), 0, st, p);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
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

template <int KH, int KW, int S, int CK, int XT>
int launch_conv(ConvParams p, hipStream_t st) {
    using C = ConvCfg<KH, KW, S, CK, XT>;
    p.tiles_x = ceil_div(p.ow, 32 * XT);
    p.tiles_y = ceil_div(p.oh, C::ROWS);
    p.co_blocks = ceil_div(p.cout, 64);
    const long grid = (long)p.tiles_x * p.tiles_y * p.co_blocks * p.n;
    if (grid <= 0 || grid > 0x7fffffffL) return VCG_E_SHAPE;
    if (p.stats) {
        // instantiated for the layers a normalisation follows in the fp32 configs: the trunk's 3x3 and the critics' 3x3 / 4x4 stride 1 and 2
        if constexpr ((KH == 3 && KW == 3 && S <= 2) || (KH == 4 && KW == 4 && S <= 2))
            return launch_with_lds(conv_fwd_kernel<KH, KW, S, CK, XT, true>, (int)grid, C::LDS_BYTES, p, st);
        else
            return VCG_E_UNSUPPORTED;
    }
    return launch_with_lds(conv_fwd_kernel<KH, KW, S, CK, XT>, (int)grid, C::LDS_BYTES, p, st);
}

template <int KH, int KW, int S>
int launch_c3(ConvParams p, hipStream_t st) {
    using C = C3Cfg<KH, KW, S>;
    p.tiles_x = ceil_div(p.ow, 32);
    p.tiles_y = ceil_div(p.oh, C::ROWS);
    p.co_blocks = ceil_div(p.cout, 64);
    const long grid = (long)p.tiles_x * p.tiles_y * p.co_blocks * p.n;
    if (grid <= 0 || grid > 0x7fffffffL) return VCG_E_SHAPE;
    return launch_with_lds(conv_c3_kernel<KH, KW, S>, (int)grid, C::LDS_BYTES, p, st);
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

// conv_rowchain.hip: 9x9, 256 -> <=3 channels, width % 64 == 0 (the generator's final/conv); VCG_E_UNSUPPORTED otherwise
int vcg_internal_conv9_rowchain(const float* x, const float* w, float* y, int n, int cin, int h, int wd, int cout, const float* bias, int act,
                                int ws_t, int ws_m, int ws_k, hipStream_t st);

// Generic entry used by the C ABI wrappers in api.hip.  Weight tensor w is [T][cin][cout] with cout
// contiguous (or addressed by ws_* strides for the small-M kernel).
int vcg_internal_conv(const float* x, const float* w, float* y, int n, int cin, int h, int wd, int cout,
                      int oh, int ow, int kh, int kw, int stride, int pad_top, int pad_left, int flip,
                      const vcg_epilogue* ep, int smallm, int ws_t, int ws_m, int ws_k, hipStream_t st, float* stats) {
    ConvParams p{};
    p.x = x; p.w = w; p.y = y; p.stats = stats;
    if (stats && (smallm || cin <= 3)) return VCG_E_UNSUPPORTED;
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
        if (kh == 9 && kw == 9 && !flip && pad_top == 4 && pad_left == 4 && oh == h && ow == wd && !p.prelu && !p.residual) {
            const int rc = vcg_internal_conv9_rowchain(x, w, y, n, cin, h, wd, cout, p.bias, p.act, ws_t, ws_m, ws_k, st);
            if (rc != VCG_E_UNSUPPORTED) return rc;
        }
        // one output channel on many input channels: the FMA reduction (the MFMA rows of the small-M kernel would be 7/8 padding)
        if (cout == 1 && cin >= 64 && kh == kw && ws_m == 1 && !p.prelu && (size_t)cin * h * wd * 4 <= 0xFFFFFFE0u) {
            if (kh == 3) return launch_cout1<3>(p, st);
            if (kh == 4) return launch_cout1<4>(p, st);
            if (kh == 5) return launch_cout1<5>(p, st);
        }
        if (kh == 9 && kw == 9) return launch_smallm<9, 9, 8>(p, st);
        if (kh == 4 && kw == 4) return launch_smallm<4, 4, 8>(p, st);
        if (kh == 3 && kw == 3) return launch_smallm<3, 3, 8>(p, st);
        if (kh == 5 && kw == 5) return launch_smallm<5, 5, 8>(p, st);
        return VCG_E_UNSUPPORTED;
    }
    if (p.act == VCG_ACT_TANH) return VCG_E_UNSUPPORTED;   // tanh epilogue: small-M kernel only (<= 32/kw channels)
    if (cin <= 3) {   // flat-K kernel: no channel padding (first conv on RGB, data gradient of a 3-channel conv)
        if (kh == 9 && kw == 9 && stride == 1) return launch_c3<9, 9, 1>(p, st);
        if (kh == 3 && kw == 3 && stride == 1) return launch_c3<3, 3, 1>(p, st);
        if (kh == 4 && kw == 4 && stride == 2) return launch_c3<4, 4, 2>(p, st);
        if (kh == 5 && kw == 5 && stride == 1) return launch_c3<5, 5, 1>(p, st);     // sparse_512 block 1 (model.py:967)
    }
    // wide outputs use 64-column tiles (two MFMA x-tiles per wave: half the barriers and weight staging per
    // MFMA, and 1024 workgroups = 2 full rounds of 2 per CU at the C2 trunk shape instead of 2.67 rounds of 3)
    const bool wide = ow > 32;
    if (kh == 3 && kw == 3 && stride == 1) return wide ? launch_conv<3, 3, 1, 8, 2>(p, st) : launch_conv<3, 3, 1, 8, 1>(p, st);
    if (kh == 3 && kw == 3 && stride == 2) return launch_conv<3, 3, 2, 8, 1>(p, st);
    if (kh == 4 && kw == 4 && stride == 1) return launch_conv<4, 4, 1, 8, 1>(p, st);   // XT=2 spills here (measured slower)
    if (kh == 4 && kw == 4 && stride == 2) return launch_conv<4, 4, 2, 4, 1>(p, st);
    if (kh == 5 && kw == 5 && stride == 1) return launch_conv<5, 5, 1, 8, 1>(p, st);   // XT=2 spills (50 weight prefetch registers)
    if (kh == 5 && kw == 5 && stride == 2) return launch_conv<5, 5, 2, 4, 1>(p, st);
    if (kh == 5 && kw == 5 && stride == 3) return launch_conv<5, 5, 3, 4, 1>(p, st);   // sparse_512 blocks 2-6 (model.py:971-987)
    if (kh == 9 && kw == 9 && stride == 1) return launch_conv<9, 9, 1, 4, 1>(p, st);
    // make_upscaler_incep_resnet (model.py:372-440): 1x1, and the 1xk / kx1 pairs of the 2-path blocks (and their data gradients)
    if (stride == 1) {
        if (kh == 1 && kw == 1) return launch_conv<1, 1, 1, 8, 1>(p, st);
        if (kh == 1 && kw == 3) return launch_conv<1, 3, 1, 8, 1>(p, st);
        if (kh == 3 && kw == 1) return launch_conv<3, 1, 1, 8, 1>(p, st);
        if (kh == 1 && kw == 5) return launch_conv<1, 5, 1, 8, 1>(p, st);
        if (kh == 5 && kw == 1) return launch_conv<5, 1, 1, 8, 1>(p, st);
        if (kh == 1 && kw == 7) return launch_conv<1, 7, 1, 8, 1>(p, st);
        if (kh == 7 && kw == 1) return launch_conv<7, 1, 1, 8, 1>(p, st);
    }
    return VCG_E_UNSUPPORTED;
}
