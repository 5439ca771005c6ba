// bf16-storage weight gradient of final/conv -- Conv2D(3, 9, 'same') on 256 channels (upscaling/upscaler/model.py:290) -- in the bf16
// configs (BASELINE.json C3/C4):
//     dW[ky][kx][ci][co] = sum over pixels p of  x[p][ci] * dz[p - (ky-4, kx-4)][co]            (fp32 result, Keras HWIO layout)
// As a GEMM: M = ci (256), N = (tap, co) = 81 x 3, K = pixels.  Both operands are needed pixel-major, i.e. transposed reads
// (ds_read_b64_tr_b16, as in bf16_wgrad.hip):
//   * x tile [8x32 pixels][128 channels of this workgroup's half] in LDS, 256-byte pixels whose four 64-byte blocks are XORed with
//     (pixel & 3): the four pixels of a transposed read fall into four different bank quarters.  x is NOT shifted (the tap shift is on
//     dz), so the tile needs no halo and x is read from HBM exactly once -- the kernel's bound (1.07 GB at batch 8);
//   * dz as bf16 [pixel][3 channels + 1 zero] = 8 bytes per pixel (vcg_pack_dz3_bf16 below; the fp32 path multiplies by fp32 dz: here
//     dz is rounded to bf16 like every other gradient operand of the bf16 path), a 16 x 40-pixel halo in LDS.  In a transposed read the 16
//     lanes of a group supply 16 block addresses: lane L addresses pixel (L >> 2) of the k-quad at the shift of tap slot (L & 3), so
//     one read hands 4 taps x 4 channels their 4 consecutive pixels and an MFMA column tile is 8 taps x (3 + 1) channels: 11 tiles
//     for the 81 taps;
//   * four waves = four 32-channel row tiles of the workgroup's 128 input channels; a wave keeps its 32 x (11 x 32) block in 176
//     accumulator registers for the whole launch; tiles stream by `buffer_load ... lds`, double buffered, one barrier per tile;
//   * per-wave partial blocks go to the workspace as raw register dumps and are summed in a fixed order (deterministic).
#include "vcg_common.hpp"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

constexpr int W9_TR = 8, W9_TC = 32, W9_NT = 11;
constexpr int W9_XB = W9_TR * W9_TC * 256;                       // 65536: x tile, 128 channels
constexpr int W9_DR = W9_TR + 8, W9_DC = W9_TC + 8;              // dz halo: 16 x 40 pixels of 8 bytes
constexpr int W9_DYB = W9_DR * W9_DC * 8;                        // 5120
constexpr int W9_CHUNKS = (W9_XB + W9_DYB) / 16;                 // 4416
constexpr int W9_NDMA = (W9_CHUNKS + 255) / 256;                 // 18
constexpr int W9_BUF = W9_NDMA * 4096;                           // 73728
constexpr int W9_GRID = 128;                                     // workgroups per channel half
constexpr int W9_WAVE_FLOATS = W9_NT * 16 * 64;

struct W9Params {
    const unsigned char* x;      // bf16 NHWC [n][h][w][256]
    const unsigned char* dz;     // bf16 [n][h][w][4] (channel 3 = 0)
    float* ws;                   // [2 halves][grid][4 waves][W9_WAVE_FLOATS]
    int n, h, w_, tiles_x, tiles_y, total, grid;
};

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

__global__ __launch_bounds__(256, 1) void wgrad9x9_c256to3_bf16_kernel(W9Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mh = blockIdx.y;                                     // channel half: input channels 128*mh ..
    const unsigned lds0 = (unsigned)(size_t)smem;
    const long ximg = (long)p.h * p.w_ * 512, dimg = (long)p.h * p.w_ * 8;

    // lane constants of the transposed reads.  A (x): pixel 8*h8 + q (+4t), channels 32*wv + 16*((l>>4)&1) + 4*(l&3) .. of the half
    const int h8 = (lane >> 5) * 8, q = (lane & 15) >> 2;
    const unsigned abase = (unsigned)((h8 + q) * 256 + ((wv ^ q) << 6) + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);
    // B (dz): lane L of a 16-group addresses pixel (L >> 2) of the k-quad, shifted for tap 8*nt + 4*((lane>>4)&1) + (L & 3)
    unsigned bbase[W9_NT];
#pragma unroll
    for (int nt = 0; nt < W9_NT; ++nt) {
        int tap = 8 * nt + 4 * ((lane >> 4) & 1) + (lane & 3);
        tap = tap < 81 ? tap : 80;                                  // 7 dummy columns: any address inside the halo (their sums are discarded)
        const int ky = tap / 9, kx = tap - 9 * ky;
        bbase[nt] = (unsigned)(W9_XB + ((8 - ky) * W9_DC + (8 - kx) + h8 + q) * 8);
    }

    f32x16 acc[W9_NT];
#pragma unroll
    for (int nt = 0; nt < W9_NT; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;

    auto dma = [&](int tile, int buf) {
        const int txi = tile % p.tiles_x, t2 = tile / p.tiles_x, tyi = t2 % p.tiles_y, img = t2 / p.tiles_y;
        const int y0 = tyi * W9_TR, x0 = txi * W9_TC;
        const vcg_rsrc rx = make_rsrc(p.x + img * ximg, (unsigned long)ximg);
        const vcg_rsrc rd = make_rsrc(p.dz + img * dimg, (unsigned long)dimg);
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));                                 // keep the slot arithmetic inside the tile loop (no hoisted registers)
#pragma unroll
        for (int k = 0; k < W9_NDMA; ++k) {
            const int s = k * 256 + tid_o;
            unsigned off;
            bool ok;
            if (k < W9_XB / 4096) {                                     // x: pixel P of the tile, 16-byte position pos of its 256 bytes
                const int P = s >> 4, pos = s & 15, row = P >> 5, col = P & 31;
                const int cs = (((pos >> 2) ^ (P & 3)) << 2) | (pos & 3);          // stored position pos holds source chunk cs
                const int gy = y0 + row, gx = x0 + col;
                ok = gy < p.h && gx < p.w_;
                off = (unsigned)(gy * p.w_ + gx) * 512u + (unsigned)(mh * 256 + cs * 16);
            } else {                                                    // dz halo: two pixels per 16 bytes (w is even: a pair never straddles a row)
                const int sd = s - W9_XB / 16, row = (2 * sd) / W9_DC, col = 2 * sd - row * W9_DC;
                const int gy = y0 - 4 + row, gx = x0 - 4 + col;
                ok = sd < W9_DYB / 16 && (unsigned)gy < (unsigned)p.h && (unsigned)gx < (unsigned)p.w_;
                off = (unsigned)(gy * p.w_ + gx) * 8u;
            }
            asm volatile("" : "+v"(off));
            off = ok ? off : VCG_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(k < W9_XB / 4096 ? rx : rd,
                                                     (void __attribute__((address_space(3)))*)(smem + buf * W9_BUF + (k * 256 + wv * 64) * 16), 16, off, 0, 0, 0);
        }
    };

    int tile = blockIdx.x, buf = 0;
    if (tile < p.total) dma(tile, 0);
    for (; tile < p.total; tile += gridDim.x, buf ^= 1) {
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): this wave's part of the stage has landed
        w9_barrier();                                // ... and everyone else's; the other buffer is free again
        const int next = tile + gridDim.x;
        if (next < p.total) dma(next, buf ^ 1);
        const unsigned lb = lds0 + buf * W9_BUF;
        // 16 k-steps of 16 pixels: tile row i, column half cb
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int i = ks >> 1, cb = ks & 1;
            unsigned long long fa[2], fb[W9_NT][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) fa[t] = w9_tr_read(lb + abase + (unsigned)((i * W9_TC + cb * 16 + 4 * t) * 256));
#pragma unroll
            for (int nt = 0; nt < W9_NT; ++nt)
#pragma unroll
                for (int t = 0; t < 2; ++t) fb[nt][t] = w9_tr_read(lb + bbase[nt] + (unsigned)((i * W9_DC + cb * 16 + 4 * t) * 8));
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(fa[0]), "+v"(fa[1]), "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[1][0]), "+v"(fb[1][1]), "+v"(fb[2][0]), "+v"(fb[2][1]),
                           "+v"(fb[3][0]), "+v"(fb[3][1]), "+v"(fb[4][0]), "+v"(fb[4][1]), "+v"(fb[5][0]), "+v"(fb[5][1]), "+v"(fb[6][0]), "+v"(fb[6][1]),
                           "+v"(fb[7][0]), "+v"(fb[7][1]), "+v"(fb[8][0]), "+v"(fb[8][1]), "+v"(fb[9][0]), "+v"(fb[9][1]), "+v"(fb[10][0]), "+v"(fb[10][1]));
            const u64x2 av = {fa[0], fa[1]};
            const bf16x8 a = __builtin_bit_cast(bf16x8, av);
#pragma unroll
            for (int nt = 0; nt < W9_NT; ++nt) {
                const u64x2 bv = {fb[nt][0], fb[nt][1]};
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, bv), acc[nt], 0, 0, 0);
            }
        }
    }
    // raw register dump of this wave's block (coalesced); decoded by the reduction
    float* out = p.ws + (((long)mh * p.grid + blockIdx.x) * 4 + wv) * W9_WAVE_FLOATS;
#pragma unroll
    for (int nt = 0; nt < W9_NT; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) out[(nt * 16 + e) * 64 + lane] = acc[nt][e];
}

// dW[tap][ci][co] = sum over workgroups (fixed order) of the wave blocks; thread = one output element
__global__ __launch_bounds__(256) void wgrad9_reduce_kernel(const float* __restrict__ ws, int grid, float* __restrict__ dw) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 81 * 256 * 3) return;
    const int co = idx % 3, ci = (idx / 3) % 256, tap = idx / 768;
    const int mh = ci >> 7, wv = (ci >> 5) & 3, m = ci & 31, nt = tap >> 3, n = (tap & 7) * 4 + co;
    // MFMA 32x32 accumulator layout: lane (n, hh) register e holds row m = 8*(e>>2) + 4*hh + (e&3)
    const int hh = (m >> 2) & 1, e = ((m >> 3) << 2) | (m & 3), lane = hh * 32 + n;
    const float* src = ws + (((long)mh * grid) * 4 + wv) * W9_WAVE_FLOATS + (nt * 16 + e) * 64 + lane;
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 4 <= grid; k += 4)
#pragma unroll
        for (int u = 0; u < 4; ++u) s4[u] += src[(long)(k + u) * 4 * W9_WAVE_FLOATS];
    for (; k < grid; ++k) s4[0] += src[(long)k * 4 * W9_WAVE_FLOATS];
    dw[idx] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
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
        hipError_t e = hipFuncSetAttribute((const void*)wgrad9x9_c256to3_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W9_BUF);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    wgrad9x9_c256to3_bf16_kernel<<<dim3(p.grid, 2), 256, 2 * W9_BUF, stream>>>(p);
    VCG_LAUNCH_CHECK();
    wgrad9_reduce_kernel<<<(81 * 256 * 3 + 255) / 256, 256, 0, stream>>>((const float*)ws, p.grid, dw_hwio);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

}  // extern "C"
