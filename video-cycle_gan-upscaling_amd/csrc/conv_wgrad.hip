// Weight gradient as a pixel-contraction GEMM on v_mfma_f32_32x32x2_f32, NCHW.
//
//   R[m][j] = sum_{n, (ay,ax)}  A[n][m][ay][ax] * B[n][jc(j)][ay*S + ky(j) - PT][ax*S + kx(j) - PL]
//   j = jc*T + ky*K + kx  (taps flipped when FLIP)
//
// normal orientation  : A = dy (m = out channel), B = x  (jc = in channel)          -> any stride
// swapped orientation : A = x  (m = in channel),  B = dy (jc = out channel), FLIP=1 -> stride 1;
//                       used when the layer has <= 3 output channels (final 9x9 conv, PatchGAN head)
//                       so that the MFMA rows carry 64 real channels instead of 3.
// The MFMA's k dimension is the pixel index (2 consecutive ax per instruction); A is read from an
// LDS tile [m][pixels] (odd m-stride => conflict-free), B through a per-lane base offset that
// encodes (jc, ky, kx) into the halo tile, so the inner loop is tap-agnostic and every LDS read is
// "lane base + immediate".
// Each workgroup walks a contiguous range of pixel tiles and keeps its 64 x (2*NJ*32) block of R
// in registers (up to 208 accumulator VGPRs); the next tile's global loads are issued into
// registers BEFORE the MFMA loop of the current tile and written to LDS after it.  Partial blocks
// go to the workspace and a second kernel sums them in a fixed order (deterministic) and scatters
// into the Keras kernel layout.  When A = dy the per-channel sums of the staged A tiles give the
// bias gradient for free (db partials ride along in the same workspace).
#include "vcg_common.hpp"

namespace {

struct WgradParams {
    const float* A;
    const float* B;
    float* part;    // [slabs][m_pad][j_pad]
    float* dbpart;  // [slabs][m_pad] or nullptr
    int n, mtot, ah, aw;  // A dims
    int jctot, bh, bw;    // B dims
    int pt, pl, flip;     // effective pads, tap flip
    int jc;               // B channels per j-block (<= JCMAX)
    int m_blocks, j_blocks, slabs;
    int tiles_x, tiles_y, tiles_total, tiles_per_slab;
    int m_pad, j_pad;
};

template <int S, int NJ, int TH, int K>
struct WgCfg {
    static constexpr int T = K * K;
    static constexpr int JCMAX = (2 * NJ * 32) / T;
    static constexpr int AST = TH * 32 + 1;             // odd stride between m rows of the A tile
    static constexpr int BH = (TH - 1) * S + K;
    static constexpr int BW = 31 * S + K;
    // conflict-free gather: row stride = K and plane stride = K*K (mod 32) put element (jc,ky,kx) of a lane on
    // bank (T*jc + K*ky + kx) mod 32 = j mod 32, i.e. the 32 lanes of a j-tile hit 32 different banks
    static constexpr int round_to(int v, int r) { return v + ((r - v % 32) % 32 + 32) % 32; }
    static constexpr int BRS = round_to(BW, K % 32);
    static constexpr int BPS = round_to(BH * BRS, (K * K) % 32);
    static constexpr int A_ELEMS = 64 * TH * 32;
    static constexpr int B_ELEMS = JCMAX * BH * BW;
    static constexpr int A_PT = A_ELEMS / 256;
    static constexpr int B_PT = (B_ELEMS + 255) / 256;
    static constexpr size_t LDS_BYTES = ((size_t)64 * AST + (size_t)JCMAX * BPS + 64) * sizeof(float);
};

template <int S, int NJ, int TH, int K>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
    using C = WgCfg<S, NJ, TH, K>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_a = smem;                 // [64][AST]
    float* s_b = smem + 64 * C::AST;   // [JCMAX][BH][BRS] (plane stride BPS)

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    int b = blockIdx.x;
    const int jb = b % p.j_blocks; b /= p.j_blocks;
    const int mb = b % p.m_blocks; b /= p.m_blocks;
    const int slab = b;
    const int m0 = mb * 64, jc0 = jb * p.jc;
    const int jc_here = min(p.jc, p.jctot - jc0);
    const int jvalid = jc_here * C::T;

    // per-lane B base offsets for this wave's NJ j-tiles
    int boff[NJ];
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        const int j = ((wv >> 1) * NJ + i) * 32 + l31;
        int o = 0;
        if (j < jvalid) {
            const int jc = j / C::T, t = j % C::T;
            int ky = t / K, kx = t % K;
            if (p.flip) { ky = K - 1 - ky; kx = K - 1 - kx; }
            o = jc * C::BPS + ky * C::BRS + kx;
        }
        boff[i] = o + half * S;
    }
    const float* ap0 = s_a + ((wv & 1) * 32 + l31) * C::AST + half;

    f32x16 acc[NJ];
#pragma unroll
    for (int i = 0; i < NJ; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float dbacc = 0.f;

    const int t_begin = slab * p.tiles_per_slab;
    const int t_end = min(t_begin + p.tiles_per_slab, p.tiles_total);

    float ra[C::A_PT], rb[C::B_PT];
    auto load_tile = [&](int tile) {
        int q = tile;
        const int tx = q % p.tiles_x; q /= p.tiles_x;
        const int ty = q % p.tiles_y; q /= p.tiles_y;
        const int n = q;
        const int ax0 = tx * 32, ay0 = ty * TH;
        const float* An = p.A + (size_t)n * p.mtot * p.ah * p.aw;
#pragma unroll
        for (int i = 0; i < C::A_PT; ++i) {
            const int e = tid + i * 256;
            const int c = e & 31, r = (e >> 5) % TH, m = e / (32 * TH);
            const int ay = ay0 + r, ax = ax0 + c;
            const bool ok = m0 + m < p.mtot && ay < p.ah && ax < p.aw;
            const float v = An[ok ? ((m0 + m) * p.ah + ay) * p.aw + ax : 0];
            ra[i] = ok ? v : 0.f;
        }
        const float* Bn = p.B + ((size_t)n * p.jctot + jc0) * p.bh * p.bw;
        const int by0 = ay0 * S - p.pt, bx0 = ax0 * S - p.pl;
#pragma unroll
        for (int i = 0; i < C::B_PT; ++i) {
            const int e = tid + i * 256;
            const int jc = e / (C::BH * C::BW), rem = e % (C::BH * C::BW);
            const int r = rem / C::BW, c = rem % C::BW;
            const int by = by0 + r, bx = bx0 + c;
            const bool ok = e < C::B_ELEMS && jc < jc_here && by >= 0 && by < p.bh && bx >= 0 && bx < p.bw;
            const float v = Bn[ok ? (jc * p.bh + by) * p.bw + bx : 0];
            rb[i] = ok ? v : 0.f;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < C::A_PT; ++i) {
            const int e = tid + i * 256;
            const int c = e & 31, r = (e >> 5) % TH, m = e / (32 * TH);
            s_a[m * C::AST + r * 32 + c] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < C::B_PT; ++i) {
            const int e = tid + i * 256;
            if (e < C::B_ELEMS) {
                const int jc = e / (C::BH * C::BW), rem = e % (C::BH * C::BW);
                const int r = rem / C::BW, c = rem % C::BW;
                s_b[jc * C::BPS + r * C::BRS + c] = rb[i];
            }
        }
    };

    if (t_begin < t_end) load_tile(t_begin);
    for (int tile = t_begin; tile < t_end; ++tile) {
        __syncthreads();  // previous tile fully consumed
        store_tile();
        __syncthreads();
        if (tile + 1 < t_end) load_tile(tile + 1);  // in flight during the MFMA loop
        if (p.dbpart != nullptr && jb == 0) {
            // bias gradient: thread t sums a quarter of row (t>>2) of the staged A tile
            const float* row = s_a + (tid >> 2) * C::AST + (tid & 3) * (TH * 8);
#pragma unroll
            for (int i = 0; i < TH * 8; ++i) dbacc += row[i];
        }
#pragma unroll
        for (int r = 0; r < TH; ++r) {
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                const float a = ap0[r * 32 + 2 * st];
#pragma unroll
                for (int i = 0; i < NJ; ++i)
                    acc[i] = mfma32(a, s_b[boff[i] + (r * S) * C::BRS + 2 * st * S], acc[i]);
            }
        }
    }

    // ---- write the partial block: rows m, cols j
    float* out = p.part + ((size_t)slab * p.m_pad + m0 + (wv & 1) * 32) * p.j_pad + (size_t)jb * (2 * NJ * 32);
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        const int jcol = ((wv >> 1) * NJ + i) * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(size_t)mfma_row(r, lane) * p.j_pad + jcol] = acc[i][r];
    }
    if (p.dbpart != nullptr && jb == 0) {
        dbacc += __shfl_xor(dbacc, 1, 64);
        dbacc += __shfl_xor(dbacc, 2, 64);
        if ((tid & 3) == 0) p.dbpart[(size_t)slab * p.m_pad + m0 + (tid >> 2)] = dbacc;
    }
}

struct ReduceParams {
    const float* part;
    const float* dbpart;
    float* dw;
    float* db;
    int slabs, m_pad, j_pad, mtot, jctot, jc, jbw, T;
    int ts, sm, sj;  // dw[tap*ts + m*sm + jc*sj]
};

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const ReduceParams p) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)p.m_pad * p.j_pad;
    if (idx >= total) return;
    const int m = (int)(idx / p.j_pad), jg = (int)(idx % p.j_pad);
    if (m >= p.mtot) return;
    const int jb = jg / p.jbw, jl = jg % p.jbw;
    if (jl >= p.jc * p.T) return;
    const int jc = jb * p.jc + jl / p.T, t = jl % p.T;
    if (jc >= p.jctot) return;
    // fixed summation order (deterministic): 8 interleaved partial sums keep 8 loads in flight
    float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* src = p.part + idx;
    int k = 0;
    for (; k + 8 <= p.slabs; k += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) s8[u] += src[(size_t)(k + u) * total];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (k + u < p.slabs) s8[u] += src[(size_t)(k + u) * total];
    const float s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    p.dw[(size_t)t * p.ts + (size_t)m * p.sm + (size_t)jc * p.sj] = s;
}

// db[m] = sum over slabs of dbpart[slab][m]: 256 threads = 64 channels x 4 slab groups, fixed order
__global__ __launch_bounds__(256) void wgrad_db_reduce_kernel(const float* dbpart, float* db, int slabs, int m_pad, int mtot) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int m = blockIdx.x * 64 + c;
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    if (m < mtot) {
        int k = g;
        for (; k + 12 < slabs; k += 16) {
#pragma unroll
            for (int u = 0; u < 4; ++u) s4[u] += dbpart[(size_t)(k + 4 * u) * m_pad + m];
        }
        for (; k < slabs; k += 4) s4[0] += dbpart[(size_t)k * m_pad + m];
    }
    red[g][c] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    __syncthreads();
    if (g == 0 && m < mtot) db[m] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

struct Plan {
    int S, NJ, TH, K, jc, m_blocks, j_blocks, slabs, tiles_x, tiles_y, tiles_total, tiles_per_slab;
    int m_pad, j_pad;
    size_t ws_part_bytes, ws_bytes;
    bool ok;
};

// instantiated (K, NJ) pairs; every pair exists for S = 1 and (except K = 9) S = 2
inline int pick_nj(int K, int jtot) {
    const int c3[3] = {1, 4, 9}, c4[3] = {1, 4, 8}, c5[2] = {4, 13}, c9[1] = {4};
    const int* c; int n;
    switch (K) {
        case 3: c = c3; n = 3; break;
        case 4: c = c4; n = 3; break;
        case 5: c = c5; n = 2; break;
        case 9: c = c9; n = 1; break;
        default: return -1;
    }
    for (int i = 0; i < n; ++i)
        if (2 * c[i] * 32 >= jtot) return c[i];   // smallest block that holds the whole j range
    return c[n - 1];
}

Plan make_plan(int n, int mtot, int ah, int aw, int jctot, int kh, int kw, int S) {
    Plan pl{};
    pl.ok = false;
    if (kh != kw || (S != 1 && S != 2) || (kh == 9 && S != 1)) return pl;
    const int T = kh * kw;
    pl.S = S; pl.K = kh;
    pl.TH = (S == 1) ? 2 : 1;
    pl.NJ = pick_nj(kh, jctot * T);
    if (pl.NJ < 0) return pl;
    const int cap = 2 * pl.NJ * 32;
    pl.jc = cap / T;
    if (pl.jc < 1) return pl;
    if (pl.jc > jctot) pl.jc = jctot;
    pl.m_blocks = ceil_div(mtot, 64);
    pl.j_blocks = ceil_div(jctot, pl.jc);
    pl.tiles_x = ceil_div(aw, 32);
    pl.tiles_y = ceil_div(ah, pl.TH);
    pl.tiles_total = pl.tiles_x * pl.tiles_y * n;
    // one workgroup per CU is resident for the big accumulator blocks (NJ >= 8: >256 registers per lane),
    // so 256 workgroups fill the chip in one round and halve the partial-sum traffic
    int slabs = ceil_div(pl.NJ >= 8 ? 256 : 512, pl.m_blocks * pl.j_blocks);
    if (slabs > pl.tiles_total) slabs = pl.tiles_total;
    if (slabs < 1) slabs = 1;
    pl.tiles_per_slab = ceil_div(pl.tiles_total, slabs);
    pl.slabs = ceil_div(pl.tiles_total, pl.tiles_per_slab);
    pl.m_pad = pl.m_blocks * 64;
    pl.j_pad = pl.j_blocks * cap;
    pl.ws_part_bytes = align_up((size_t)pl.slabs * pl.m_pad * pl.j_pad * sizeof(float), 256);
    pl.ws_bytes = pl.ws_part_bytes + (size_t)pl.slabs * pl.m_pad * sizeof(float);
    pl.ok = true;
    return pl;
}

template <int S, int NJ, int TH, int K>
int launch_wgrad(const WgradParams& p, int grid, hipStream_t st) {
    using C = WgCfg<S, NJ, TH, K>;
    auto kern = wgrad_kernel<S, NJ, TH, K>;
    if (C::LDS_BYTES > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), C::LDS_BYTES, st, p);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

}  // namespace

size_t vcg_internal_wgrad_ws(int n, int mtot, int ah, int aw, int jctot, int kh, int kw, int S) {
    Plan pl = make_plan(n, mtot, ah, aw, jctot, kh, kw, S);
    return pl.ok ? pl.ws_bytes : 0;
}

// dw[tap*ts + m*sm + jc*sj] = R[m][(jc,tap)];  db[m] = sum of A over pixels (optional, normal orientation)
int vcg_internal_wgrad(const float* A, const float* B, float* dw, float* db, int n, int mtot, int ah, int aw,
                       int jctot, int bh, int bw, int kh, int kw, int S, int pt, int pl_, int flip, int ts, int sm,
                       int sj, void* ws, size_t ws_bytes, hipStream_t st) {
    Plan pl = make_plan(n, mtot, ah, aw, jctot, kh, kw, S);
    if (!pl.ok) return VCG_E_UNSUPPORTED;
    if (ws == nullptr || ws_bytes < pl.ws_bytes) return VCG_E_WORKSPACE;
    WgradParams p{};
    p.A = A; p.B = B; p.part = (float*)ws;
    p.dbpart = db ? (float*)((char*)ws + pl.ws_part_bytes) : nullptr;
    p.n = n; p.mtot = mtot; p.ah = ah; p.aw = aw; p.jctot = jctot; p.bh = bh; p.bw = bw;
    p.pt = pt; p.pl = pl_; p.flip = flip; p.jc = pl.jc;
    p.m_blocks = pl.m_blocks; p.j_blocks = pl.j_blocks; p.slabs = pl.slabs;
    p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.tiles_total = pl.tiles_total;
    p.tiles_per_slab = pl.tiles_per_slab;
    p.m_pad = pl.m_pad; p.j_pad = pl.j_pad;
    const int grid = pl.slabs * pl.m_blocks * pl.j_blocks;
    int rc = VCG_E_UNSUPPORTED;
#define VCG_WG(S_, NJ_, TH_, K_) \
    if (S == S_ && pl.NJ == NJ_ && pl.K == K_) rc = launch_wgrad<S_, NJ_, TH_, K_>(p, grid, st)
    VCG_WG(1, 1, 2, 3); VCG_WG(1, 4, 2, 3); VCG_WG(1, 9, 2, 3);
    VCG_WG(2, 1, 1, 3); VCG_WG(2, 4, 1, 3); VCG_WG(2, 9, 1, 3);
    VCG_WG(1, 1, 2, 4); VCG_WG(1, 4, 2, 4); VCG_WG(1, 8, 2, 4);
    VCG_WG(2, 1, 1, 4); VCG_WG(2, 4, 1, 4); VCG_WG(2, 8, 1, 4);
    VCG_WG(1, 4, 2, 5); VCG_WG(1, 13, 2, 5);
    VCG_WG(2, 4, 1, 5); VCG_WG(2, 13, 1, 5);
    VCG_WG(1, 4, 2, 9);
#undef VCG_WG
    if (rc != VCG_OK) return rc;
    ReduceParams r{};
    r.part = (const float*)ws; r.dbpart = p.dbpart; r.dw = dw; r.db = db;
    r.slabs = pl.slabs; r.m_pad = pl.m_pad; r.j_pad = pl.j_pad;
    r.mtot = mtot; r.jctot = jctot; r.jc = pl.jc; r.jbw = 2 * pl.NJ * 32; r.T = kh * kw;
    r.ts = ts; r.sm = sm; r.sj = sj;
    const size_t total = (size_t)pl.m_pad * pl.j_pad;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, r);
    VCG_LAUNCH_CHECK();
    if (db != nullptr) {
        hipLaunchKernelGGL(wgrad_db_reduce_kernel, dim3(ceil_div(mtot, 64)), dim3(256), 0, st, (const float*)p.dbpart, db,
                           pl.slabs, pl.m_pad, mtot);
        VCG_LAUNCH_CHECK();
    }
    return VCG_OK;
}
