// bf16-storage weight gradient of final/conv -- Conv2D(3, 9, 'same') on 256 channels (upscaling/upscaler/model.py:290) -- in the bf16
// configs (BASELINE.json C3/C4):
//     dW[ky][kx][ci][co] = sum over pixels p of  x[p][ci] * dz[p - (ky-4, kx-4)][co]            (fp32 result, Keras HWIO layout)
// As a GEMM: M = ci (256), N = (tap, co) = 81 x 3, K = pixels.  Both operands are needed pixel-major, i.e. transposed reads
// (ds_read_b64_tr_b16, as in bf16_wgrad.hip):
//   * x tile [4x32 pixels][128 channels of this workgroup's half] in LDS, 256-byte pixels whose four 64-byte blocks are XORed with
//     (pixel & 3): the four pixels of a transposed read fall into four different bank quarters.  x is NOT shifted (the tap shift is on
//     dz), so the tile needs no halo and x is read from HBM exactly once -- the kernel's bound (1.07 GB at batch 8);
//   * dz as bf16 [pixel][3 channels + 1 zero] = 8 bytes per pixel (vcg_pack_dz3_bf16 below; the fp32 path multiplies by fp32 dz: here
//     dz is rounded to bf16 like every other gradient operand of the bf16 path), a 12 x 40-pixel halo in LDS.  In a transposed read the 16
//     lanes of a group supply 16 block addresses: lane L addresses pixel (L >> 2) of the k-quad at the shift of tap slot (L & 3), so
//     one read hands 4 taps x 4 channels their 4 consecutive pixels and an MFMA column tile is 8 taps x (3 + 1) channels: 11 tiles
//     for the 81 taps;
//   * four waves = quarters of the 11 (+ 1 dummy) column tiles; a wave keeps its [4 row tiles of 32 channels] x [3 column tiles] block
//     in 192 accumulator registers for the whole launch (why this shape: see the k-loop); tiles stream by `buffer_load ... lds` through a ring of three stages, one barrier per tile; the
//     operand fragments of the next k-step are read behind the MFMAs of this one;
//   * per-wave partial blocks go to the workspace as raw register dumps and are summed in a fixed order (deterministic).
#include "vcg_common.hpp"
#include <utility>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

constexpr int W9_TR = 4, W9_TC = 32, W9_NT = 11;
constexpr int W9_NWV = 4, W9_NTH = W9_NWV * 64;                  // waves = quarters of the (11 + 1 dummy) column tiles; a wave holds ALL four row tiles
constexpr int W9_NTW = 3, W9_MT = 4;                             // column tiles / row tiles (32 channels) per wave
constexpr int W9_XB = W9_TR * W9_TC * 256;                       // 32768: x tile, 128 channels
constexpr int W9_DR = W9_TR + 8, W9_DC = W9_TC + 8;              // dz halo: 12 x 40 pixels of 8 bytes
constexpr int W9_DYB = W9_DR * W9_DC * 8;                        // 3840
constexpr int W9_CHUNKS = (W9_XB + W9_DYB) / 16;                 // 2288
constexpr int W9_NDMA = (W9_CHUNKS + W9_NTH - 1) / W9_NTH;       // 5
constexpr int W9_BUF = W9_NDMA * W9_NTH * 16;                    // 40960
constexpr int W9_NS = 3;                                         // ring of three stages: two tiles in flight while one is multiplied
constexpr int W9_KS = W9_TR * W9_TC / 16;                        // k-steps of 16 pixels per tile
static_assert(W9_NS * W9_BUF <= 160 * 1024 && W9_NDMA < 16 && W9_NDMA <= W9_NTW * W9_MT && W9_XB % (W9_NTH * 16) == 0, "wgrad9: stages / vmcnt / issue slots / regions");
constexpr int W9_WAIT_ONE_BEHIND = W9_NDMA | 0x0F70;             // s_waitcnt vmcnt(NDMA): all but the youngest stage's pieces have landed
constexpr int W9_GRID = 128;                                     // workgroups per channel half
constexpr int W9_WAVE_FLOATS = W9_NT * 16 * 64;

struct W9Params {
    const unsigned char* x;      // bf16 NHWC [n][h][w][256]
    const unsigned char* dz;     // bf16 [n][h][w][4] (channel 3 = 0)
    float* ws;                   // [2 halves][grid][4 waves][W9_WAVE_FLOATS]
    int n, h, w_, tiles_x, tiles_y, total, grid;
};

template <class F, int... Is>
__device__ __forceinline__ void w9_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void w9_static_for(F&& f) {
    w9_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ void w9_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ unsigned long long w9_tr_read(unsigned addr) {
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

__global__ __launch_bounds__(W9_NTH, 1) void wgrad9x9_c256to3_bf16_kernel(W9Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // channel half mh (input channels 128*mh ..) is the FASTEST index of the grid: the two workgroups that read the two 256-byte halves of
    // the same 512-byte pixels start together and stay in step, so a DRAM page is opened once for both (with the half as the slow index
    // the halves of a page were fetched at unrelated times: 3.0 TB/s whatever the tile pipeline looked like)
    const int mh = blockIdx.x & 1, bx = blockIdx.x >> 1, nbx = gridDim.x >> 1;
    const unsigned lds0 = (unsigned)(size_t)smem;
    const long ximg = (long)p.h * p.w_ * 512, dimg = (long)p.h * p.w_ * 8;

    // lane constants of the transposed reads.  A (x): pixel 8*h8 + q (+4t), channels 32*m + 16*((l>>4)&1) + 4*(l&3) .. of the half
    const int h8 = (lane >> 5) * 8, q = (lane & 15) >> 2;
    unsigned abase[W9_MT];
#pragma unroll
    for (int m = 0; m < W9_MT; ++m) abase[m] = (unsigned)((h8 + q) * 256 + ((m ^ q) << 6) + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);
    // B (dz): lane L of a 16-group addresses pixel (L >> 2) of the k-quad, shifted for tap 8*nt + 4*((lane>>4)&1) + (L & 3), nt = 3*wv + j
    unsigned bbase[W9_NTW];
#pragma unroll
    for (int j = 0; j < W9_NTW; ++j) {
        int tap = 8 * (wv * W9_NTW + j) + 4 * ((lane >> 4) & 1) + (lane & 3);
        tap = tap < 81 ? tap : 80;                                  // 15 dummy columns: any address inside the halo (their sums are discarded)
        const int ky = tap / 9, kx = tap - 9 * ky;
        bbase[j] = (unsigned)(W9_XB + ((8 - ky) * W9_DC + (8 - kx) + h8 + q) * 8);
    }

    f32x16 acc[W9_MT][W9_NTW];
#pragma unroll
    for (int m = 0; m < W9_MT; ++m)
#pragma unroll
        for (int j = 0; j < W9_NTW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][j][e] = 0.f;

    // one 1-KiB piece (DMA round k) of a tile's stage; the tile's origin and descriptors are decoded once per tile (scalars)
    struct Src { int y0, x0; vcg_rsrc rx, rd; };
    auto decode = [&](int tile) {
        const int txi = tile % p.tiles_x, t2 = tile / p.tiles_x, tyi = t2 % p.tiles_y, img = t2 / p.tiles_y;
        return Src{tyi * W9_TR, txi * W9_TC, make_rsrc(p.x + img * ximg, (unsigned long)ximg), make_rsrc(p.dz + img * dimg, (unsigned long)dimg)};
    };
    auto dma_piece = [&](const Src& sc, int buf, int k) {
        const int y0 = sc.y0, x0 = sc.x0;
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));                                 // keep the slot arithmetic inside the tile loop (no hoisted registers)
        const int s = k * W9_NTH + tid_o;
        constexpr int KX = W9_XB / (W9_NTH * 16);                       // rounds that carry x
        unsigned off;
        bool ok;
        if (k < KX) {                                                   // x: pixel P of the tile, 16-byte position pos of its 256 bytes
            const int P = s >> 4, pos = s & 15, row = P >> 5, col = P & 31;
            const int cs = (((pos >> 2) ^ (P & 3)) << 2) | (pos & 3);              // stored position pos holds source chunk cs
            const int gy = y0 + row, gx = x0 + col;
            ok = gy < p.h && gx < p.w_;
            off = (unsigned)(gy * p.w_ + gx) * 512u + (unsigned)(mh * 256 + cs * 16);
        } else {                                                        // dz halo: two pixels per 16 bytes (w is even: a pair never straddles a row)
            const int sd = s - W9_XB / 16, row = (2 * sd) / W9_DC, col = 2 * sd - row * W9_DC;
            const int gy = y0 - 4 + row, gx = x0 - 4 + col;
            ok = sd < W9_DYB / 16 && (unsigned)gy < (unsigned)p.h && (unsigned)gx < (unsigned)p.w_;
            off = (unsigned)(gy * p.w_ + gx) * 8u;
        }
        asm volatile("" : "+v"(off));
        off = ok ? off : VCG_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(k < KX ? sc.rx : sc.rd,
                                                 (void __attribute__((address_space(3)))*)(smem + buf * W9_BUF + (k * W9_NTH + wv * 64) * 16), 16, off, 0, 0, 0);
    };

    // What bounds this kernel is the LDS pipe: a ds_read_b64_tr_b16 occupies it for ~8 cycles, an MFMA its SIMD's matrix pipe for 32, and a
    // (rows x columns) block of MFMA tiles per wave needs 2*(rows + columns) reads per rows*columns MFMAs and k-step.  Measured at C3's shard
    // (1.07 GB of x, a 150-us MFMA stream): 1 x 11 tiles on four waves, read-then-multiply (round 2) 445 us; the same with the next
    // k-step's reads issued behind this one's MFMAs 358 us; 1 x 6 tiles on eight waves (two per SIMD) 352 us -- 96...112 reads per k-step
    // and CU either way.  Here a wave holds ALL four row tiles x 3 of the 12 column tiles (192 accumulator registers): 4 x 14 = 56
    // reads per k-step and CU for the same 48 MFMAs.  The fragments of k-step ks + 1 are read behind the MFMAs of k-step ks (a pair of
    // reads after each of the first seven), the 9 DMA pieces of the tile after next behind the first MFMAs of k-step 0.
    unsigned long long fa[2][W9_MT][2], fb[2][W9_NTW][2];
    auto read_a = [&](unsigned lb, int set, int ks, int m) {
        const int i = ks >> 1, cb = ks & 1;
#pragma unroll
        for (int t = 0; t < 2; ++t) fa[set][m][t] = w9_tr_read(lb + abase[m] + (unsigned)((i * W9_TC + cb * 16 + 4 * t) * 256));
    };
    auto read_b = [&](unsigned lb, int set, int ks, int j) {
        const int i = ks >> 1, cb = ks & 1;
#pragma unroll
        for (int t = 0; t < 2; ++t) fb[set][j][t] = w9_tr_read(lb + bbase[j] + (unsigned)((i * W9_DC + cb * 16 + 4 * t) * 8));
    };
#define W9_WAIT_SET(S_)                                                                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                                       \
                 : "+v"(fa[S_][0][0]), "+v"(fa[S_][0][1]), "+v"(fa[S_][1][0]), "+v"(fa[S_][1][1]), "+v"(fa[S_][2][0]), "+v"(fa[S_][2][1]),   \
                   "+v"(fa[S_][3][0]), "+v"(fa[S_][3][1]), "+v"(fb[S_][0][0]), "+v"(fb[S_][0][1]), "+v"(fb[S_][1][0]), "+v"(fb[S_][1][1]),   \
                   "+v"(fb[S_][2][0]), "+v"(fb[S_][2][1]))

    int tile = bx, buf = 0;
    if (tile < p.total) {
        const Src first = decode(tile);
#pragma unroll
        for (int k = 0; k < W9_NDMA; ++k) dma_piece(first, 0, k);
    }
    if (tile + nbx < p.total) {
        const Src second = decode(tile + nbx);
#pragma unroll
        for (int k = 0; k < W9_NDMA; ++k) dma_piece(second, 1, k);
    }
    for (; tile < p.total; tile += nbx, buf = buf + 1 == W9_NS ? 0 : buf + 1) {
        // a wave's loads retire in order: "at most NDMA outstanding" = this tile's stage is complete, the next one's may still be in flight
        if (tile + nbx < p.total) __builtin_amdgcn_s_waitcnt(W9_WAIT_ONE_BEHIND);
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        w9_barrier();                                // ... and everyone else's part; the stage multiplied last is free again
        const int next = tile + 2 * nbx;
        const bool has_next = next < p.total;
        const Src nsrc = decode(has_next ? next : tile);
        const int nbuf = buf == 0 ? W9_NS - 1 : buf - 1;
        const unsigned lb = lds0 + buf * W9_BUF;
#pragma unroll
        for (int m = 0; m < W9_MT; ++m) read_a(lb, 0, 0, m);
#pragma unroll
        for (int j = 0; j < W9_NTW; ++j) read_b(lb, 0, 0, j);
        W9_WAIT_SET(0);
        // k-steps of 16 pixels: tile row ks >> 1, column half ks & 1
        w9_static_for<W9_KS>([&](auto ic) {
            constexpr int ks = decltype(ic)::value, c = ks & 1, n = c ^ 1;
#pragma unroll
            for (int j = 0; j < W9_NTW; ++j) {
                const u64x2 bv = {fb[c][j][0], fb[c][j][1]};
                const bf16x8 b = __builtin_bit_cast(bf16x8, bv);
#pragma unroll
                for (int m = 0; m < W9_MT; ++m) {
                    const int u = j * W9_MT + m;                         // MFMA number inside the k-step
                    const u64x2 av = {fa[c][m][0], fa[c][m][1]};
                    acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), b, acc[m][j], 0, 0, 0);
                    if (ks + 1 < W9_KS) {
                        if (u < W9_MT) read_a(lb, n, ks + 1, u);
                        else if (u < W9_MT + W9_NTW) read_b(lb, n, ks + 1, u - W9_MT);
                    }
                    if (ks == 0 && u < W9_NDMA && has_next) dma_piece(nsrc, nbuf, u);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (ks + 1 < W9_KS) {
                if (n == 0) W9_WAIT_SET(0); else W9_WAIT_SET(1);
            }
        });
    }
#undef W9_WAIT_SET
    // raw register dumps into the row tiles' blocks (coalesced); decoded by the reduction
#pragma unroll
    for (int m = 0; m < W9_MT; ++m) {
        float* out = p.ws + (((long)mh * p.grid + bx) * 4 + m) * W9_WAVE_FLOATS;
#pragma unroll
        for (int j = 0; j < W9_NTW; ++j) {
            const int nt = wv * W9_NTW + j;
            if (nt < W9_NT) {
#pragma unroll
                for (int e = 0; e < 16; ++e) out[(nt * 16 + e) * 64 + lane] = acc[m][j][e];
            }
        }
    }
}

// dW[tap][ci][co] = sum over workgroups, in a fixed order, of the wave blocks.  block = 64 consecutive RAW dump elements x 16 record lanes
// (coalesced reads, eight records in flight per thread, fixed-order combine through LDS); the decoded (tap, ci, co) position is only used
// for the single store.  (One thread per OUTPUT element walking all records read scattered words one latency after the other: 47 us.)
__global__ __launch_bounds__(1024) void wgrad9_reduce_kernel(const float* __restrict__ ws, int grid, float* __restrict__ dw) {
    __shared__ float red[16][64];
    const int g = threadIdx.x >> 6, cl = threadIdx.x & 63;
    const int r = blockIdx.x * 64 + cl;                              // raw element: [channel half][row tile][W9_WAVE_FLOATS]
    const int mh = r / (4 * W9_WAVE_FLOATS), rem = r - mh * 4 * W9_WAVE_FLOATS, mt = rem / W9_WAVE_FLOATS, off = rem - mt * W9_WAVE_FLOATS;
    const float* src = ws + (((long)mh * grid) * 4 + mt) * W9_WAVE_FLOATS + off;
    const long stride = 4l * W9_WAVE_FLOATS;
    float s = 0.f;
    int b = g;
    for (; b + 16 * 7 < grid; b += 16 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(long)(b + 16 * u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < grid; b += 16) s += src[(long)b * stride];
    red[g][cl] = s;
    __syncthreads();
    if (g != 0) return;
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i][cl];
    // raw offset -> column tile nt, register e, lane l;  D[row][col]: row = (e&3) + 8*(e>>2) + 4*(l>>5), col = l&31 = (tap slot, co)
    const int l = off & 63, e = (off >> 6) & 15, nt = off >> 10;
    const int m = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5), n = l & 31, tap = 8 * nt + (n >> 2), co = n & 3, ci = mh * 128 + mt * 32 + m;
    if (tap < 81 && co < 3) dw[(tap * 256 + ci) * 3 + co] = t;
}

// dz fp32 NCHW [n][3][h][w] -> bf16 [n][h][w][4] (channel 3 = 0)
__global__ void pack_dz3_bf16_kernel(const float* __restrict__ dz, __bf16* __restrict__ out, int n, long hw) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * hw) return;
    const long img = i / hw, px = i - img * hw;
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 v;
    v[0] = (__bf16)dz[(img * 3 + 0) * hw + px];
    v[1] = (__bf16)dz[(img * 3 + 1) * hw + px];
    v[2] = (__bf16)dz[(img * 3 + 2) * hw + px];
    v[3] = (__bf16)0.f;
    *(bf16x4*)(out + i * 4) = v;
}

}  // namespace

extern "C" {

size_t vcg_conv9x9_to3_bf16_wgrad_workspace_bytes(const vcg_conv_desc* d) {
    if (d == nullptr || d->n <= 0 || d->h <= 0 || d->w <= 0) return 0;
    return (size_t)2 * W9_GRID * 4 * W9_WAVE_FLOATS * sizeof(float) + (size_t)d->n * d->h * d->w * 8 + 256;
}

int vcg_conv9x9_to3_bf16_wgrad(const vcg_conv_desc* d, const void* x, const float* dz, float* dw_hwio, void* ws, size_t ws_bytes, hipStream_t stream) {
    VCG_CHECK_PTR(d); VCG_CHECK_PTR(x); VCG_CHECK_PTR(dz); VCG_CHECK_PTR(dw_hwio); VCG_CHECK_PTR(ws);
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->oh != d->h || d->ow != d->w) return VCG_E_SHAPE;
    if (d->cin != 256 || d->cout != 3 || d->kh != 9 || d->kw != 9 || d->stride != 1 || d->pad_top != 4 || d->pad_left != 4) return VCG_E_UNSUPPORTED;
    if (d->w % 2 || (long)d->h * d->w * 512 > 0xFFFFFFE0l) return VCG_E_UNSUPPORTED;     // dz pairs must not straddle rows; one image per buffer descriptor
    if (ws_bytes < vcg_conv9x9_to3_bf16_wgrad_workspace_bytes(d)) return VCG_E_WORKSPACE;
    W9Params p;
    p.x = (const unsigned char*)x;
    p.ws = (float*)ws;
    unsigned char* dzb = (unsigned char*)ws + (size_t)2 * W9_GRID * 4 * W9_WAVE_FLOATS * sizeof(float);
    dzb += (256 - ((size_t)dzb & 255)) & 255;
    p.dz = dzb;
    p.n = d->n; p.h = d->h; p.w_ = d->w;
    p.tiles_x = ceil_div(d->w, W9_TC);
    p.tiles_y = ceil_div(d->h, W9_TR);
    p.total = p.n * p.tiles_x * p.tiles_y;
    p.grid = p.total < W9_GRID ? p.total : W9_GRID;
    const long px = (long)d->n * d->h * d->w;
    pack_dz3_bf16_kernel<<<(unsigned)((px + 255) / 256), 256, 0, stream>>>(dz, (__bf16*)dzb, d->n, (long)d->h * d->w);
    VCG_LAUNCH_CHECK();
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)wgrad9x9_c256to3_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, W9_NS * W9_BUF);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    wgrad9x9_c256to3_bf16_kernel<<<dim3(2 * p.grid), W9_NTH, W9_NS * W9_BUF, stream>>>(p);
    VCG_LAUNCH_CHECK();
    static_assert(W9_WAVE_FLOATS % 64 == 0, "wgrad9 reduce: 64 raw elements per block stay inside one wave block");
    wgrad9_reduce_kernel<<<2 * 4 * W9_WAVE_FLOATS / 64, 1024, 0, stream>>>((const float*)ws, p.grid, dw_hwio);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

}  // extern "C"
