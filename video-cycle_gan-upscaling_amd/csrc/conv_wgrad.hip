// Weight gradient as a pixel-contraction GEMM on v_mfma_f32_32x32x2_f32, NCHW.
//
//   R[m][j] = sum_{n, (ay,ax)}  A[n][m][ay][ax] * B[n][jc(j)][ay*S + ky(j) - PT][ax*S + kx(j) - PL]
//   j = jc*T + ky*KW + kx  (taps flipped when FLIP)
//
// normal orientation  : A = dy (m = out channel), B = x  (jc = in channel)          -> any stride
// swapped orientation : A = x  (m = in channel),  B = dy (jc = out channel), FLIP=1 -> stride 1;
//                       used when the layer has <= 3 output channels (final 9x9 conv, PatchGAN head)
//                       so that the MFMA rows carry 64 real channels instead of 3.
// The MFMA's k dimension is the pixel index (2 consecutive ax per instruction); A is read from an
// LDS tile [m][pixels] (odd m-stride => conflict-free), B through a per-lane base offset that
// encodes (jc, ky, kx) into the halo tile, so the inner loop is tap-agnostic.
// Each workgroup walks a contiguous range of pixel tiles and keeps its 64 x (2*NJ*32) block of R
// in registers; partial blocks go to the workspace and a second kernel sums them in a fixed order
// (deterministic) and scatters into the Keras kernel layout.
#include "vcg_common.hpp"

namespace {

struct WgradParams {
    const float* A;
    const float* B;
    float* part;  // [slabs][m_pad][j_pad]
    int n, mtot, ah, aw;        // A dims
    int jctot, bh, bw;          // B dims
    int kh, kw, pt, pl, flip;   // taps, effective pads
    int jc;                     // B channels per j-block
    int m_blocks, j_blocks, slabs;
    int tiles_x, tiles_y, tiles_total, tiles_per_slab;
    int bh_t, bw_t, brs, bps;   // B tile rows/cols, row stride, plane stride (floats)
    int m_pad, j_pad;
};

template <int S, int NJ, int TH>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
    constexpr int AST = TH * 32 + 1;  // odd stride between m rows of the A tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_a = smem;                // [64][AST]
    float* s_b = smem + 64 * AST;     // [jc][bh_t][brs] (plane stride bps)

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    int b = blockIdx.x;
    const int jb = b % p.j_blocks; b /= p.j_blocks;
    const int mb = b % p.m_blocks; b /= p.m_blocks;
    const int slab = b;
    const int m0 = mb * 64, jc0 = jb * p.jc;
    const int T = p.kh * p.kw;
    const int jvalid = min(p.jc, p.jctot - jc0) * T;

    // per-lane B base offsets for this wave's NJ j-tiles
    int boff[NJ];
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        const int j = ((wv >> 1) * NJ + i) * 32 + l31;
        int o = 0;
        if (j < jvalid) {
            const int jc = j / T, t = j % T;
            int ky = t / p.kw, kx = t % p.kw;
            if (p.flip) { ky = p.kh - 1 - ky; kx = p.kw - 1 - kx; }
            o = jc * p.bps + ky * p.brs + kx;
        }
        boff[i] = o + half * S;
    }
    const int aoff = ((wv & 1) * 32 + l31) * AST + half;

    f32x16 acc[NJ];
#pragma unroll
    for (int i = 0; i < NJ; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int t_begin = slab * p.tiles_per_slab;
    const int t_end = min(t_begin + p.tiles_per_slab, p.tiles_total);
    const int b_elems = p.bh_t * p.bw_t;
    const int jc_here = min(p.jc, p.jctot - jc0);

    for (int tile = t_begin; tile < t_end; ++tile) {
        int q = tile;
        const int tx = q % p.tiles_x; q /= p.tiles_x;
        const int ty = q % p.tiles_y; q /= p.tiles_y;
        const int n = q;
        const int ax0 = tx * 32, ay0 = ty * TH;
        __syncthreads();  // previous tile fully consumed
        // ---- stage A tile: 64 m x TH x 32
        const float* An = p.A + (size_t)n * p.mtot * p.ah * p.aw;
        for (int e = tid; e < 64 * TH * 32; e += 256) {
            const int c = e & 31, r = (e >> 5) % TH, m = e / (32 * TH);
            const int ay = ay0 + r, ax = ax0 + c;
            float v = 0.f;
            if (m0 + m < p.mtot && ay < p.ah && ax < p.aw) v = An[((size_t)(m0 + m) * p.ah + ay) * p.aw + ax];
            s_a[m * AST + r * 32 + c] = v;
        }
        // ---- stage B halo tile: jc_here channels x bh_t x bw_t
        const float* Bn = p.B + (size_t)n * p.jctot * p.bh * p.bw;
        const int by0 = ay0 * S - p.pt, bx0 = ax0 * S - p.pl;
        for (int jc = 0; jc < jc_here; ++jc) {
            const float* Bc = Bn + (size_t)(jc0 + jc) * p.bh * p.bw;
            float* sb = s_b + jc * p.bps;
            for (int e = tid; e < b_elems; e += 256) {
                const int r = e / p.bw_t, c = e - r * p.bw_t;
                const int by = by0 + r, bx = bx0 + c;
                float v = 0.f;
                if (by >= 0 && by < p.bh && bx >= 0 && bx < p.bw) v = Bc[(size_t)by * p.bw + bx];
                sb[r * p.brs + c] = v;
            }
        }
        __syncthreads();
        // ---- contraction over the tile's pixels
#pragma unroll
        for (int r = 0; r < TH; ++r) {
            const float* ap = s_a + aoff + r * 32;
            const float* bp = s_b + (r * S) * p.brs;
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                const float a = ap[2 * st];
#pragma unroll
                for (int i = 0; i < NJ; ++i) acc[i] = mfma32(a, bp[boff[i] + 2 * st * S], acc[i]);
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
}

struct ReduceParams {
    const float* part;
    float* dw;
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
    float s = 0.f;
    for (int k = 0; k < p.slabs; ++k) s += p.part[(size_t)k * total + idx];
    p.dw[(size_t)t * p.ts + (size_t)m * p.sm + (size_t)jc * p.sj] = s;
}

struct Plan {
    int S, NJ, TH, jc, m_blocks, j_blocks, slabs, tiles_x, tiles_y, tiles_total, tiles_per_slab;
    int bh_t, bw_t, brs, bps, m_pad, j_pad;
    size_t lds_bytes, ws_bytes;
    bool ok;
};

Plan make_plan(int n, int mtot, int ah, int aw, int jctot, int kh, int kw, int S) {
    Plan pl{};
    const int T = kh * kw;
    pl.S = S;
    pl.TH = (S == 1) ? 2 : 1;
    // pick NJ (j-tiles per wave; a block spans 2*NJ tiles of 32 j) among the instantiated sizes:
    // the smallest that holds the whole j range, else the exact-fit size for the tap count
    // (9 taps: 18 tiles = 64 channels; 16 taps: 16 tiles = 32 channels; 25 taps: 26 tiles = 33
    // channels; 81 taps: 8 tiles = 3 channels).
    const int cand[5] = {1, 4, 8, 9, 13};
    const int jtot = jctot * T;
    int best = -1;
    for (int i = 0; i < 5 && best < 0; ++i)
        if (2 * cand[i] * 32 >= jtot) best = cand[i];
    if (best < 0) {
        if (T <= 9) best = 9;
        else if (T <= 16) best = 8;
        else best = 13;
    }
    if (2 * best * 32 < T) { pl.ok = false; return pl; }
    pl.NJ = best;
    const int cap = 2 * best * 32;
    pl.jc = cap / T;
    if (pl.jc > jctot) pl.jc = jctot;
    if (pl.jc < 1) { pl.ok = false; return pl; }
    pl.m_blocks = ceil_div(mtot, 64);
    pl.j_blocks = ceil_div(jctot, pl.jc);
    pl.tiles_x = ceil_div(aw, 32);
    pl.tiles_y = ceil_div(ah, pl.TH);
    pl.tiles_total = pl.tiles_x * pl.tiles_y * n;
    int slabs = ceil_div(512, pl.m_blocks * pl.j_blocks);
    if (slabs > pl.tiles_total) slabs = pl.tiles_total;
    if (slabs < 1) slabs = 1;
    pl.tiles_per_slab = ceil_div(pl.tiles_total, slabs);
    pl.slabs = ceil_div(pl.tiles_total, pl.tiles_per_slab);
    pl.bh_t = (pl.TH - 1) * S + kh;
    pl.bw_t = 31 * S + kw;
    pl.brs = pl.bw_t | 1;                        // odd row stride
    pl.bps = pl.bh_t * pl.brs;
    if (T == 9 && S == 1) {                      // conflict-free gather for 3x3: row = 3, plane = 9 (mod 32)
        pl.brs = 35;
        pl.bps = ((pl.bh_t * pl.brs + 31) / 32) * 32 + 9;
    } else if ((pl.bps & 1) == 0) {
        pl.bps += 1;
    }
    pl.m_pad = pl.m_blocks * 64;
    pl.j_pad = pl.j_blocks * cap;
    pl.lds_bytes = ((size_t)64 * (pl.TH * 32 + 1) + (size_t)pl.jc * pl.bps + 64) * sizeof(float);
    pl.ws_bytes = (size_t)pl.slabs * pl.m_pad * pl.j_pad * sizeof(float);
    pl.ok = pl.lds_bytes <= 160 * 1024;
    return pl;
}

template <int S, int NJ, int TH>
int launch_wgrad(const WgradParams& p, size_t lds, int grid, hipStream_t st) {
    auto kern = wgrad_kernel<S, NJ, TH>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, p);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

}  // namespace

size_t vcg_internal_wgrad_ws(int n, int mtot, int ah, int aw, int jctot, int kh, int kw, int S) {
    Plan pl = make_plan(n, mtot, ah, aw, jctot, kh, kw, S);
    return pl.ok ? pl.ws_bytes : 0;
}

// dw[tap*ts + m*sm + jc*sj] = R[m][(jc,tap)]
int vcg_internal_wgrad(const float* A, const float* B, float* dw, int n, int mtot, int ah, int aw, int jctot,
                       int bh, int bw, int kh, int kw, int S, int pt, int pl_, int flip, int ts, int sm, int sj,
                       void* ws, size_t ws_bytes, hipStream_t st) {
    Plan pl = make_plan(n, mtot, ah, aw, jctot, kh, kw, S);
    if (!pl.ok) return VCG_E_UNSUPPORTED;
    if (ws == nullptr || ws_bytes < pl.ws_bytes) return VCG_E_WORKSPACE;
    WgradParams p{};
    p.A = A; p.B = B; p.part = (float*)ws;
    p.n = n; p.mtot = mtot; p.ah = ah; p.aw = aw; p.jctot = jctot; p.bh = bh; p.bw = bw;
    p.kh = kh; p.kw = kw; p.pt = pt; p.pl = pl_; p.flip = flip; p.jc = pl.jc;
    p.m_blocks = pl.m_blocks; p.j_blocks = pl.j_blocks; p.slabs = pl.slabs;
    p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.tiles_total = pl.tiles_total;
    p.tiles_per_slab = pl.tiles_per_slab;
    p.bh_t = pl.bh_t; p.bw_t = pl.bw_t; p.brs = pl.brs; p.bps = pl.bps;
    p.m_pad = pl.m_pad; p.j_pad = pl.j_pad;
    const int grid = pl.slabs * pl.m_blocks * pl.j_blocks;
    int rc = VCG_E_UNSUPPORTED;
#define VCG_WG(S_, NJ_, TH_) if (S == S_ && pl.NJ == NJ_) rc = launch_wgrad<S_, NJ_, TH_>(p, pl.lds_bytes, grid, st)
    VCG_WG(1, 1, 2); VCG_WG(1, 4, 2); VCG_WG(1, 8, 2); VCG_WG(1, 9, 2); VCG_WG(1, 13, 2);
    VCG_WG(2, 1, 1); VCG_WG(2, 4, 1); VCG_WG(2, 8, 1); VCG_WG(2, 9, 1); VCG_WG(2, 13, 1);
#undef VCG_WG
    if (rc != VCG_OK) return rc;
    ReduceParams r{};
    r.part = (const float*)ws; r.dw = dw; r.slabs = pl.slabs; r.m_pad = pl.m_pad; r.j_pad = pl.j_pad;
    r.mtot = mtot; r.jctot = jctot; r.jc = pl.jc; r.jbw = 2 * pl.NJ * 32; r.T = kh * kw;
    r.ts = ts; r.sm = sm; r.sj = sj;
    const size_t total = (size_t)pl.m_pad * pl.j_pad;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, r);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}
