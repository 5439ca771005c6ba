// Weight gradient as a pixel-contraction GEMM on v_mfma_f32_32x32x2_f32, NCHW.
//
//   R[m][j] = sum_{n, (ay,ax)}  A[n][m][ay][ax] * B[n][jc(j)][ay*S + ky(j) - PT][ax*S + kx(j) - PL]
//   j = jc*T + ky*KW + kx  (taps flipped when FLIP); kernels need not be square (1x7 / 7x1 / 1x1 of the inception-resnet generator)
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

// NW waves per workgroup, two layouts of the 64-channel x JTILES block over the waves:
//   MW = 2: every wave covers both 32-row m-tiles and NJ of the NW*NJ j-tiles (2*NJ accumulator tiles per wave;
//           small register footprint -> 2-3 workgroups per CU overlap each other's staging and barriers);
//   MW = 1: waves split the m-tiles too (wave = 1 m-tile x NJ j-tiles, JTILES = NW/2*NJ): the layout for the
//           64-channel 3x3 / 5x5 blocks whose 18 / 26 j-tiles need 144-208 accumulator registers per wave.
// Which layout serves which shape was measured (profiles/r01_kbench.txt).
template <int S, int NW, int NJ, int MW, int TH, int KH, int KW>
struct WgCfg {
    static constexpr int T = KH * KW;
    static constexpr int NT = 64 * NW;                   // threads per workgroup
    static constexpr int JTILES = (MW == 2 ? NW : NW / 2) * NJ;   // j-tiles per block
    static constexpr int JCMAX = (JTILES * 32) / T;      // B channels per block
    static constexpr int AST = TH * 32 + 1;              // odd stride between m rows of the A tile
    static constexpr int BH = (TH - 1) * S + KH;
    static constexpr int BW = 31 * S + KW;
    // conflict-free gather: row stride = KW and plane stride = KH*KW (mod 32) put element (jc,ky,kx) of a lane on
    // bank (T*jc + KW*ky + kx) mod 32 = j mod 32, i.e. the 32 lanes of a j-tile hit 32 different banks
    static constexpr int round_to(int v, int r) { return v + ((r - v % 32) % 32 + 32) % 32; }
    static constexpr int BRS = round_to(BW, KW % 32);
    static constexpr int BPS = round_to(BH * BRS, T % 32);
    static constexpr int A_ELEMS = 64 * TH * 32;
    static constexpr int B_ELEMS = JCMAX * BH * BW;
    static constexpr int A_PT = (A_ELEMS + NT - 1) / NT;
    // B staging: thread = (row group, column) over whole 32-column segments of the halo rows -- (channel, row)
    // come from one small division per ROW instead of three per element -- plus a flat pass for the BW % 32
    // remaining halo columns
    static constexpr int RG = NT / 32;
    static constexpr int ROWS = JCMAX * BH;
    static constexpr int QSEG = BW / 32, REM = BW % 32;
    static constexpr int RPT = (ROWS + RG - 1) / RG;
    static constexpr int HPT = (ROWS * REM + NT - 1) / NT;
    static constexpr int REMD = REM > 0 ? REM : 1;       // divisor of the remainder pass (which is empty when BW is a multiple of 32)
    static constexpr int B_PT = RPT * QSEG + HPT;
    static constexpr size_t LDS_BYTES = ((size_t)64 * AST + (size_t)JCMAX * BPS + 64) * sizeof(float);
};

template <int S, int NW, int NJ, int MW, int TH, int KH, int KW>
__global__ __launch_bounds__(64 * NW, (MW * NJ * 16 > 100 ? 1 : 2)) void wgrad_kernel(const WgradParams p) {
    using C = WgCfg<S, NW, NJ, MW, TH, KH, KW>;
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
        const int j = ((MW == 2 ? wv : (wv >> 1)) * NJ + i) * 32 + l31;
        int o = 0;
        if (j < jvalid) {
            const int jc = j / C::T, t = j % C::T;
            int ky = t / KW, kx = t % KW;
            if (p.flip) { ky = KH - 1 - ky; kx = KW - 1 - kx; }
            o = jc * C::BPS + ky * C::BRS + kx;
        }
        boff[i] = o + half * S;
    }
    const float* ap0 = s_a + ((MW == 2 ? 0 : (wv & 1)) * 32 + l31) * C::AST + half;

    f32x16 acc[MW][NJ];
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int i = 0; i < NJ; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][i][r] = 0.f;
    float dbacc = 0.f;
    const bool do_db = p.dbpart != nullptr && jb == 0 && wv == 0;

    const int t_begin = slab * p.tiles_per_slab;
    const int t_end = min(t_begin + p.tiles_per_slab, p.tiles_total);

    float ra[C::A_PT], rb[C::B_PT];
    auto load_tile = [&](int tile) {
        int q = tile;
        const int tx = q % p.tiles_x; q /= p.tiles_x;
        const int ty = q % p.tiles_y; q /= p.tiles_y;
        const int n = q;
        const int ax0 = tx * 32, ay0 = ty * TH;
        // range-checked buffer loads (vcg_common.hpp): the validity select sits on the ADDRESS, nothing depends on the
        // loaded data until store_tile -- the prefetch really flies under the MFMA loop
        const vcg_rsrc rA = make_rsrc(p.A + (size_t)n * p.mtot * p.ah * p.aw, (size_t)p.mtot * p.ah * p.aw * sizeof(float));
#pragma unroll
        for (int i = 0; i < C::A_PT; ++i) {
            const int e = tid + i * C::NT;
            const int c = e & 31, r = (e >> 5) % TH, m = e / (32 * TH);
            const int ay = ay0 + r, ax = ax0 + c;
            const bool ok = e < C::A_ELEMS && m0 + m < p.mtot && ay < p.ah && ax < p.aw;
            ra[i] = buf_load(rA, ok ? 4u * (unsigned)(((m0 + m) * p.ah + ay) * p.aw + ax) : VCG_OOB);
        }
        const vcg_rsrc rB = make_rsrc(p.B + ((size_t)n * p.jctot + jc0) * p.bh * p.bw, (size_t)(p.jctot - jc0) * p.bh * p.bw * sizeof(float));
        const int by0 = ay0 * S - p.pt, bx0 = ax0 * S - p.pl;
        const int c32 = tid & 31, rg = tid >> 5;
#pragma unroll
        for (int i = 0; i < C::RPT; ++i) {
            const int row = rg + i * C::RG;
            const int jc = row / C::BH, r = row % C::BH;
            const int by = by0 + r;
            const bool rok = row < C::ROWS && jc < jc_here && by >= 0 && by < p.bh;
            const int rbase = (jc * p.bh + by) * p.bw + bx0;
#pragma unroll
            for (int q = 0; q < C::QSEG; ++q) {
                const int bx = bx0 + q * 32 + c32;
                const bool ok = rok && bx >= 0 && bx < p.bw;
                rb[i * C::QSEG + q] = buf_load(rB, ok ? 4u * (unsigned)(rbase + q * 32 + c32) : VCG_OOB);
            }
        }
#pragma unroll
        for (int i = 0; i < C::HPT; ++i) {
            const int e = tid + i * C::NT;
            const int row = e / C::REMD, c = C::QSEG * 32 + e % C::REMD;
            const int jc = row / C::BH, r = row % C::BH;
            const int by = by0 + r, bx = bx0 + c;
            const bool ok = row < C::ROWS && jc < jc_here && by >= 0 && by < p.bh && bx >= 0 && bx < p.bw;
            rb[C::RPT * C::QSEG + i] = buf_load(rB, ok ? 4u * (unsigned)((jc * p.bh + by) * p.bw + bx) : VCG_OOB);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < C::A_PT; ++i) {
            const int e = tid + i * C::NT;
            if (e < C::A_ELEMS) {
                const int c = e & 31, r = (e >> 5) % TH, m = e / (32 * TH);
                s_a[m * C::AST + r * 32 + c] = ra[i];
            }
        }
        const int c32 = tid & 31, rg = tid >> 5;
#pragma unroll
        for (int i = 0; i < C::RPT; ++i) {
            const int row = rg + i * C::RG;
            if (row < C::ROWS) {
                const int jc = row / C::BH, r = row % C::BH;
#pragma unroll
                for (int q = 0; q < C::QSEG; ++q) s_b[jc * C::BPS + r * C::BRS + q * 32 + c32] = rb[i * C::QSEG + q];
            }
        }
#pragma unroll
        for (int i = 0; i < C::HPT; ++i) {
            const int e = tid + i * C::NT;
            const int row = e / C::REMD, c = C::QSEG * 32 + e % C::REMD;
            if (row < C::ROWS) {
                const int jc = row / C::BH, r = row % C::BH;
                s_b[jc * C::BPS + r * C::BRS + c] = rb[C::RPT * C::QSEG + i];
            }
        }
    };

    if (t_begin < t_end) load_tile(t_begin);
    for (int tile = t_begin; tile < t_end; ++tile) {
        __syncthreads();  // previous tile fully consumed
        store_tile();
        __syncthreads();
        if (tile + 1 < t_end) load_tile(tile + 1);  // in flight during the MFMA loop
        if (do_db) {
            // bias gradient: lane = channel row of the staged dy tile (odd row stride: conflict-free)
            const float* row = s_a + lane * C::AST;
#pragma unroll
            for (int i = 0; i < TH * 32; ++i) dbacc += row[i];
        }
#pragma unroll
        for (int r = 0; r < TH; ++r) {
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                float av[MW];
#pragma unroll
                for (int m = 0; m < MW; ++m) av[m] = ap0[m * 32 * C::AST + r * 32 + 2 * st];
#pragma unroll
                for (int i = 0; i < NJ; ++i) {
                    const float bv = s_b[boff[i] + (r * S) * C::BRS + 2 * st * S];
#pragma unroll
                    for (int m = 0; m < MW; ++m) acc[m][i] = mfma32(av[m], bv, acc[m][i]);
                }
            }
        }
    }

    // ---- write the partial block: rows m, cols j
    float* out = p.part + ((size_t)slab * p.m_pad + m0 + (MW == 2 ? 0 : (wv & 1)) * 32) * p.j_pad + (size_t)jb * (C::JTILES * 32);
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            const int jcol = ((MW == 2 ? wv : (wv >> 1)) * NJ + i) * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) out[(size_t)(m * 32 + mfma_row(r, lane)) * p.j_pad + jcol] = acc[m][i][r];
        }
    if (do_db) p.dbpart[(size_t)slab * p.m_pad + m0 + lane] = dbacc;
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
    int S, NW, NJ, MW, TH, KH, KW, jc, m_blocks, j_blocks, slabs, tiles_x, tiles_y, tiles_total, tiles_per_slab;
    int m_pad, j_pad;
    size_t ws_part_bytes, ws_bytes;
    bool ok;
};

// wave layout per kernel size / stride (measured choices): NW waves, NJ j-tiles per wave, MW m-tiles per wave
inline void pick_layout(int KH, int KW, int S, int jtot, int* nw, int* nj, int* mw) {
    *mw = 2;
    if (jtot <= 64) { *nw = 2; *nj = 1; return; }          // 3-channel first layers, 1-channel head
    if (KH != KW) { *nw = 4; *nj = 2; return; }            // 1xk / kx1 (inception-resnet paths): the small layout, 256 / T channels per block
    switch (KH) {
        case 1: *nw = 4; *nj = 2; return;                                                             // 1x1: 256 channels per block
        case 3: if (jtot <= 256) { *nw = 4; *nj = 2; } else { *nw = 4; *nj = 9; *mw = 1; } return;   // 18 tiles = 64 channels
        case 4: if (S == 2) { *nw = 4; *nj = 2; } else { *nw = 4; *nj = 8; *mw = 1; } return;        // 8 tiles = 16 ch / 16 tiles = 32 ch
        case 5: if (jtot <= 256 || S == 3) { *nw = 4; *nj = 2; } else { *nw = 4; *nj = 13; *mw = 1; } return;  // 26 tiles = 33 channels (stride 3: small layout)
        case 9: *nw = 4; *nj = 2; return;                                                             // 8 tiles = 3 channels
        default: *nw = 0; *nj = 0; return;
    }
}

Plan make_plan(int n, int mtot, int ah, int aw, int jctot, int kh, int kw, int S) {
    Plan pl{};
    pl.ok = false;
    if (S < 1 || S > 3 || (S == 3 && (kh != 5 || kw != 5)) || (kh == 9 && S != 1)) return pl;
    if (kh != kw && (S != 1 || (kh != 1 && kw != 1))) return pl;            // non-square: 1xk / kx1 at stride 1
    if (kh == 1 && kw == 1 && S != 1) return pl;
    const int T = kh * kw;
    pl.S = S; pl.KH = kh; pl.KW = kw;
    pl.TH = (S == 1) ? 2 : 1;
    pick_layout(kh, kw, S, jctot * T, &pl.NW, &pl.NJ, &pl.MW);
    if (pl.NW == 0) return pl;
    const int cap = (pl.MW == 2 ? pl.NW : pl.NW / 2) * pl.NJ * 32;
    pl.jc = cap / T;
    if (pl.jc < 1) return pl;
    if (pl.jc > jctot) pl.jc = jctot;
    pl.m_blocks = ceil_div(mtot, 64);
    pl.j_blocks = ceil_div(jctot, pl.jc);
    pl.tiles_x = ceil_div(aw, 32);
    pl.tiles_y = ceil_div(ah, pl.TH);
    pl.tiles_total = pl.tiles_x * pl.tiles_y * n;
    // big-accumulator layouts hold one workgroup per CU (256 fill the chip in one round); the small ones 2-3
    int slabs = ceil_div(pl.MW == 1 ? 256 : 768, pl.m_blocks * pl.j_blocks);
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

template <int S, int NW, int NJ, int MW, int TH, int KH, int KW>
int launch_wgrad(const WgradParams& p, int grid, hipStream_t st) {
    using C = WgCfg<S, NW, NJ, MW, TH, KH, KW>;
    static_assert(C::LDS_BYTES <= 160 * 1024, "wgrad tile does not fit the 160 KiB LDS");
    auto kern = wgrad_kernel<S, NW, NJ, MW, TH, KH, KW>;
    if (C::LDS_BYTES > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS_BYTES, st, p);
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
#define VCG_WG2(S_, NW_, NJ_, MW_, TH_, KH_, KW_)                                                                        \
    if (S == S_ && pl.NW == NW_ && pl.NJ == NJ_ && pl.MW == MW_ && pl.KH == KH_ && pl.KW == KW_ && pl.TH == TH_) \
    rc = launch_wgrad<S_, NW_, NJ_, MW_, TH_, KH_, KW_>(p, grid, st)
#define VCG_WG(S_, NW_, NJ_, MW_, TH_, K_) VCG_WG2(S_, NW_, NJ_, MW_, TH_, K_, K_)
    VCG_WG(1, 2, 1, 2, 2, 3); VCG_WG(1, 4, 2, 2, 2, 3); VCG_WG(1, 4, 9, 1, 2, 3);
    VCG_WG(2, 2, 1, 2, 1, 3); VCG_WG(2, 4, 2, 2, 1, 3); VCG_WG(2, 4, 9, 1, 1, 3);
    VCG_WG(1, 2, 1, 2, 2, 4); VCG_WG(1, 4, 8, 1, 2, 4);
    VCG_WG(2, 2, 1, 2, 1, 4); VCG_WG(2, 4, 2, 2, 1, 4);
    VCG_WG(1, 4, 2, 2, 2, 5); VCG_WG(1, 4, 13, 1, 2, 5);
    VCG_WG(2, 4, 2, 2, 1, 5); VCG_WG(2, 4, 13, 1, 1, 5);
    VCG_WG(3, 4, 2, 2, 1, 5);            // sparse_512 (model.py:971-987): 5x5 stride 3
    VCG_WG(1, 4, 2, 2, 2, 9);
    // make_upscaler_incep_resnet (model.py:372-440): 1x1, and the 1xk / kx1 pairs of the 2-path blocks
    VCG_WG(1, 2, 1, 2, 2, 1); VCG_WG(1, 4, 2, 2, 2, 1);
    VCG_WG2(1, 2, 1, 2, 2, 1, 3); VCG_WG2(1, 4, 2, 2, 2, 1, 3); VCG_WG2(1, 2, 1, 2, 2, 3, 1); VCG_WG2(1, 4, 2, 2, 2, 3, 1);
    VCG_WG2(1, 2, 1, 2, 2, 1, 5); VCG_WG2(1, 4, 2, 2, 2, 1, 5); VCG_WG2(1, 2, 1, 2, 2, 5, 1); VCG_WG2(1, 4, 2, 2, 2, 5, 1);
    VCG_WG2(1, 2, 1, 2, 2, 1, 7); VCG_WG2(1, 4, 2, 2, 2, 1, 7); VCG_WG2(1, 2, 1, 2, 2, 7, 1); VCG_WG2(1, 4, 2, 2, 2, 7, 1);
#undef VCG_WG2
#undef VCG_WG
    if (rc != VCG_OK) return rc;
    ReduceParams r{};
    r.part = (const float*)ws; r.dbpart = p.dbpart; r.dw = dw; r.db = db;
    r.slabs = pl.slabs; r.m_pad = pl.m_pad; r.j_pad = pl.j_pad;
    r.mtot = mtot; r.jctot = jctot; r.jc = pl.jc; r.jbw = (pl.MW == 2 ? pl.NW : pl.NW / 2) * pl.NJ * 32; r.T = kh * kw;
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
