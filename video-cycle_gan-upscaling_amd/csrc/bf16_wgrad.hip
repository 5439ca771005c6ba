// bf16-storage weight gradient of the 3x3 stride-1 'same' 64 -> 64 convolution (the generator trunk; gradient of
// upscaling/upscaler/model.py:19,22,283 in the bf16 configs C3/C4 of BASELINE.json):
//     dW[tap][ci][co] = sum over pixels of  dy[pixel][co] * x[pixel + tap][ci]          (fp32 result, Keras HWIO layout)
// The contraction runs over PIXELS while the NHWC activations keep CHANNELS contiguous, i.e. both MFMA operands are
// needed "transposed" (8 consecutive pixels of one channel per lane).  gfx950's ds_read_b64_tr_b16 delivers exactly
// that from the natural [pixel][64 channels] LDS image (semantics measured with scripts/micro/tr_read_probe.hip: in a
// 16-lane group, lane i receives halfword (i&3) of the 8 bytes addressed by lanes (i>>2)+4j, j = 0..3): a group fetches
// 4 pixels x 16 channels and hands every lane 4 pixels of ITS channel; two reads make one operand fragment.  No
// transposed staging pass, and the tap shift is plain address arithmetic on whole pixels (no alignment issue).
//   * LDS image: 128-byte pixel rows whose two 64-byte halves are swapped when bit 1 of the pixel index is set: the four
//     consecutive pixels of a transposed read then fall into four different 16-bank quarters (conflict-free);
//   * persistent workgroups of 6 compute waves = 3 tap rows x 2 pixel halves of an 8x32-pixel tile (+ 2 loader waves); a wave keeps the whole
//     [64 co] x [3 dx x 64 ci] block of its tap row in 192 accumulator VGPRs for the entire launch (12 MFMAs per
//     2 + 6 operand fragments);
//   * tiles stream HBM -> LDS by global_load_lds (double buffered, swizzle applied on the global side, out-of-image
//     pixels fetched from a zero page);
//   * per-wave partial blocks go to the workspace and are summed in a fixed order (deterministic) into Keras' layout.
#include "vcg_common.hpp"
#include <utility>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ unsigned long long tr_read(unsigned addr) {
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

constexpr int G_R = 8, G_C = 32;                       // output-pixel tile
constexpr int G_DYB = G_R * G_C * 128;                 // 32768
constexpr int G_XP = G_C + 2;                          // halo tile pitch (pixels)
constexpr int G_XB = (G_R + 2) * G_XP * 128;           // 43520
constexpr int G_NW = 6;                                // compute waves
constexpr int G_NL = 2, G_NWT = G_NW + G_NL;           // + loader waves (they only issue the tiles' DMA)
constexpr int G_CHUNKS = (G_DYB + G_XB) / 16;          // 4768 16-byte chunks per stage
constexpr int G_NDMA = (G_CHUNKS + G_NL * 64 - 1) / (G_NL * 64);      // 38 DMA instructions per loader lane and stage
constexpr int G_BUF = G_NDMA * G_NL * 64 * 16;         // one stage, padded to whole DMA instructions (77824 B): the lanes
                                                       // past the last chunk land in the padding
constexpr int G_WAVE_FLOATS = 12 * 16 * 64;            // a wave's partial block: 12 tiles x 16 registers x 64 lanes

struct WgParams {
    const unsigned char* x;      // bf16 NHWC [n][h][w][64]
    const unsigned char* dy;     // bf16 NHWC [n][h][w][64]
    float* ws;                   // [grid][3 tap rows][G_WAVE_FLOATS]
    float* wsb;                  // [grid][2][2][64] bias partials
    int n, h, w_, tiles_x, tiles_y, total;
};

// Diagnostic build only (-DVCG_WG_STAMPS, scripts/micro/wg_stamps.sh): per wave, s_memtime sums of [wait for the stage's DMA, barrier, DMA issue,
// k-steps, tiles, kernel clocks]
#ifdef VCG_WG_STAMPS
__device__ unsigned long long vcg_wg_stamp_sums[256 * G_NWT * 6];
#define WG_STAMP(t)                                                                   \
    do {                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");     \
        __builtin_amdgcn_sched_barrier(0);                                            \
    } while (0)
#else
#define WG_STAMP(t) do { } while (0)
#endif

__global__ __launch_bounds__(G_NWT * 64, 1) void wgrad3x3_c64_bf16_kernel(WgParams p) {
#ifdef VCG_WG_STAMPS
    unsigned long long ws0 = 0, ws1 = 0, ws2 = 0, ws3 = 0, wcnt = 0, wt0, wt1, wt2, wt3, wt4;
    const unsigned long long wk0 = __builtin_amdgcn_s_memtime();
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dyi = wv >> 1, ph = wv & 1;
    const unsigned lds0 = (unsigned)(size_t)smem;

    // lane constants of the transposed reads: pixel 8h + q (+dx), first channel 16*((l>>4)&1) + 4*(l&3) of the 32-block
    const int h8 = (lane >> 5) * 8, q = (lane & 15) >> 2;
    const int chb = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;             // byte offset inside a 64-byte half
    unsigned abase[2];                                                           // dy: [co half]
#pragma unroll
    for (int coh = 0; coh < 2; ++coh)
        abase[coh] = (unsigned)((h8 + q) * 128 + ((64 * coh + chb) ^ (64 * ((q >> 1) & 1))));
    unsigned bbase[3][2][2];                                                     // x: [dx][ci half][row parity of the k-step]
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int cih = 0; cih < 2; ++cih)
#pragma unroll
            for (int ip = 0; ip < 2; ++ip) {
                const int sb = ((ip + dyi) & 1) ^ (((q + dx) >> 1) & 1);
                bbase[dx][cih][ip] = (unsigned)(G_DYB + (dyi * G_XP + h8 + q + dx) * 128 + ((64 * cih + chb) ^ (64 * sb)));
            }

    float dbs[2] = {0.f, 0.f};              // bias gradient: this lane's channel (of each co half), its 8 pixels per k-step
    f32x16 acc[3][2][2];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][c][e] = 0.f;

    const long img_bytes = (long)p.h * p.w_ * 128;
    // Staging belongs to two LOADER waves (wv >= G_NW), 38 1-KiB pieces each per stage.  In-kernel stamps (scripts/micro/wg_stamps.*,
    // profiles/r03_wg_stamps.txt) of the form in which the six compute waves issued 13 pieces each: 2.6-3.7 k of a tile's 11.2 k ticks in
    // the issue block behind the barrier (every `buffer_load ... lds` waits for room in the CU's memory pipe; no wave multiplies meanwhile)
    // and next to nothing waiting for the data afterwards; with the pieces spread over the k-steps the same ticks moved into the k-steps
    // (an issue that stalls holds up the MFMAs behind it in program order).  A piece's first slot k*128 + lw*64 is wave-uniform, so dy
    // or x, and for dy the row, are scalar; out-of-image pixels get the out-of-range offset of the image's buffer descriptor (zero fill
    // by the range check: no zero page, no 64-bit lane pointers).
    if (wv >= G_NW) {
        // Loader waves.  Rounds 0..15 carry dy (2048 slots = 16 x 128), rounds 16..37 the x halo: which region a piece belongs to is a
        // compile-time property of k.  Everything that depends on (k, lane) only is computed ONCE (the loaders have the registers: no
        // accumulators): for dy the piece is 8 consecutive pixels of row k >> 1, the lane part has two variants (k & 1); for x a lane's
        // (row, column) of the 10 x 34 halo and its byte offset per round.  Per tile a piece is an add, its range check and a select --
        // the first loader version re-derived each slot per tile (~30 instructions per piece, 6.2-6.8 k ticks of issue per tile: the
        // loaders, not the multiplying waves, set the tile time).
        constexpr int KDY = G_DYB / 16 / (G_NL * 64), NX = G_NDMA - KDY;
        static_assert(G_DYB / 16 % (G_NL * 64) == 0 && G_C == 32, "wgrad3x3 loaders: dy rounds");
        const int lw = wv - G_NW, l3 = lane >> 3, l7 = lane & 7;
        int pdy[2];
        unsigned rdy[2];
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            pdy[par] = (2 * par + lw) * 8 + l3;
            rdy[par] = (unsigned)(pdy[par] * 128 + ((l7 ^ (4 * ((pdy[par] >> 1) & 1))) * 16));         // stored chunk l7 holds source chunk l7 ^ 4*((pixel >> 1) & 1)
        }
        int xr[NX], xc[NX];
        unsigned rxo[NX];
        static_for<NX>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int P = j * 16 + lw * 8 + l3, row = P / G_XP, col = P - row * G_XP;
            xr[j] = P < (G_R + 2) * G_XP ? row : 0x40000000;                                           // (slots past the halo: never in range)
            xc[j] = col;
            rxo[j] = (unsigned)(row * p.w_ + col) * 128u + (unsigned)((l7 ^ (4 * ((P >> 1) & 1))) * 16);
        });
        auto issue = [&](int tile, int buf) {
            const int txi = tile % p.tiles_x, t2 = tile / p.tiles_x, tyi = t2 % p.tiles_y, img = t2 / p.tiles_y;
            const int y0 = tyi * G_R, x0 = txi * G_C;
            const vcg_rsrc qdy = make_rsrc(p.dy + img * img_bytes, (unsigned long)img_bytes), qx = make_rsrc(p.x + img * img_bytes, (unsigned long)img_bytes);
            const unsigned dybase = (unsigned)(y0 * p.w_ + x0) * 128u, xbase = (unsigned)((y0 - 1) * p.w_ + x0 - 1) * 128u;   // (xbase: modulo 2^32)
            const int wlim = p.w_ - x0;
            unsigned char* stage = smem + buf * G_BUF + lw * 1024;
            static_for<KDY>([&](auto kc) {
                constexpr int k = decltype(kc)::value, row = k >> 1, par = k & 1;
                const int lim = y0 + row < p.h ? wlim : 0;                                            // scalar: a row below the image admits no column
                const unsigned off = pdy[par] < lim ? dybase + (unsigned)(row * p.w_) * 128u + rdy[par] : VCG_OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(qdy, (void __attribute__((address_space(3)))*)(stage + k * (G_NL * 1024)), 16, off, 0, 0, 0);
            });
            static_for<NX>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const bool ok = (unsigned)(xr[j] + y0 - 1) < (unsigned)p.h && (unsigned)(xc[j] + x0 - 1) < (unsigned)p.w_;
                const unsigned off = ok ? xbase + rxo[j] : VCG_OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(qx, (void __attribute__((address_space(3)))*)(stage + (KDY + j) * (G_NL * 1024)), 16, off, 0, 0, 0);
            });
        };
        // stage t + 1 is requested right behind barrier t (which frees it) and has landed before barrier t + 1
        int tile = blockIdx.x, buf = 0;
        if (tile < p.total) issue(tile, 0);
        for (; tile < p.total; tile += gridDim.x, buf ^= 1) {
            WG_STAMP(wt0);
            __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wave's half of the stage has landed
            WG_STAMP(wt1);
            lds_barrier();
            WG_STAMP(wt2);
            const int next = tile + gridDim.x;
            if (next < p.total) issue(next, buf ^ 1);
            WG_STAMP(wt3);
#ifdef VCG_WG_STAMPS
            ws0 += wt1 - wt0, ws1 += wt2 - wt1, ws2 += wt3 - wt2, ++wcnt;
#endif
        }
#ifdef VCG_WG_STAMPS
        if (lane == 0 && blockIdx.x < 256) {
            unsigned long long* o = vcg_wg_stamp_sums + (blockIdx.x * G_NWT + wv) * 6;
            o[0] = ws0, o[1] = ws1, o[2] = ws2, o[3] = 0, o[4] = wcnt, o[5] = __builtin_amdgcn_s_memtime() - wk0;
        }
#endif
        lds_barrier();                               // the two barriers of the compute waves' exchange below
        lds_barrier();
        return;
    }

    int tile = blockIdx.x, buf = 0;
    for (; tile < p.total; tile += gridDim.x, buf ^= 1) {
        WG_STAMP(wt0);
        WG_STAMP(wt1);
        lds_barrier();                               // the loaders' stage has landed; the other buffer is free again
        WG_STAMP(wt2);
        WG_STAMP(wt3);
        const unsigned lb = lds0 + buf * G_BUF;

        // 8 k-steps of 16 pixels: rows ph*4 .. ph*4+3, column halves 0/1
        static_for<8>([&](auto ic) {
            constexpr int ks = decltype(ic)::value, i = ks >> 1, cb = ks & 1;
            const int row = ph * 4 + i;
            unsigned long long fa[2][2], fb[3][2][2];
#pragma unroll
            for (int coh = 0; coh < 2; ++coh)
#pragma unroll
                for (int t = 0; t < 2; ++t) fa[coh][t] = tr_read(lb + abase[coh] + (unsigned)((row * G_C + cb * 16 + 4 * t) * 128));
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int cih = 0; cih < 2; ++cih)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        fb[dx][cih][t] = tr_read(lb + bbase[dx][cih][i & 1] + (unsigned)((row * G_XP + cb * 16 + 4 * t) * 128));
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fb[0][0][0]), "+v"(fb[0][0][1]),
                           "+v"(fb[0][1][0]), "+v"(fb[0][1][1]), "+v"(fb[1][0][0]), "+v"(fb[1][0][1]), "+v"(fb[1][1][0]), "+v"(fb[1][1][1]),
                           "+v"(fb[2][0][0]), "+v"(fb[2][0][1]), "+v"(fb[2][1][0]), "+v"(fb[2][1][1]));
            if (dyi == 0) {                         // wave-uniform: the two tap-row-0 waves cover every pixel once
#pragma unroll
                for (int coh = 0; coh < 2; ++coh)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                        const bf16x4 v4 = __builtin_bit_cast(bf16x4, fa[coh][t]);
                        dbs[coh] += ((float)v4[0] + (float)v4[1]) + ((float)v4[2] + (float)v4[3]);
                    }
            }
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int cih = 0; cih < 2; ++cih) {
                    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                    const u64x2 bv = {fb[dx][cih][0], fb[dx][cih][1]};
                    const bf16x8 b = __builtin_bit_cast(bf16x8, bv);
#pragma unroll
                    for (int coh = 0; coh < 2; ++coh) {
                        const u64x2 av = {fa[coh][0], fa[coh][1]};
                        acc[dx][cih][coh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), b, acc[dx][cih][coh], 0, 0, 0);
                    }
                }
        });
        WG_STAMP(wt4);
#ifdef VCG_WG_STAMPS
        ws0 += wt1 - wt0, ws1 += wt2 - wt1, ws2 += wt3 - wt2, ws3 += wt4 - wt3, ++wcnt;
#endif
    }
#ifdef VCG_WG_STAMPS
    if (lane == 0 && blockIdx.x < 256) {
        unsigned long long* o = vcg_wg_stamp_sums + (blockIdx.x * G_NWT + wv) * 6;
        o[0] = ws0, o[1] = ws1, o[2] = ws2, o[3] = ws3, o[4] = wcnt, o[5] = __builtin_amdgcn_s_memtime() - wk0;
    }
#endif

    // the two pixel-half waves of a tap row add their blocks through LDS (the stage buffers are free now): one raw
    // register dump per tap row and workgroup (coalesced), decoded by the reduction
    lds_barrier();
    float* xch = (float*)smem + dyi * G_WAVE_FLOATS;               // 3 x 48 KiB
    if (ph == 1) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int cih = 0; cih < 2; ++cih)
#pragma unroll
                for (int coh = 0; coh < 2; ++coh)
#pragma unroll
                    for (int e = 0; e < 16; ++e) xch[(((dx * 2 + cih) * 2 + coh) * 16 + e) * 64 + lane] = acc[dx][cih][coh][e];
    }
    lds_barrier();
    if (ph == 0) {
        float* out = p.ws + ((long)blockIdx.x * 3 + dyi) * G_WAVE_FLOATS;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int cih = 0; cih < 2; ++cih)
#pragma unroll
                for (int coh = 0; coh < 2; ++coh)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int o = (((dx * 2 + cih) * 2 + coh) * 16 + e) * 64 + lane;
                        out[o] = acc[dx][cih][coh][e] + xch[o];
                    }
    }
    if (dyi == 0) {                                 // bias partials: [workgroup][pixel half][lane half][64 channels]
        float* bo = p.wsb + (((long)blockIdx.x * 2 + ph) * 2 + (lane >> 5)) * 64;
        bo[lane & 31] = dbs[0];
        bo[32 + (lane & 31)] = dbs[1];
    }
}

// dW[tap = dyi*3+dx][ci][co] = sum over workgroups, in a fixed order; the decoded (tap, ci, co) position is only used for the single store.
// Both tails in ONE launch: block = 64 columns x 16 record lanes (eight records in flight per thread), blocks [0, 576) sum the partial
// weight blocks of their 64 raw columns, block 576 the 4 x grid partial bias rows (round 2 ran two kernels, 14 + 10 us per weight
// gradient on 144 workgroups and on one).
__global__ __launch_bounds__(1024) void wgrad3x3_c64_reduce2_kernel(const float* __restrict__ ws, const float* __restrict__ wsb, int nblocks,
                                                                     float* __restrict__ dw, float* __restrict__ db) {
    __shared__ float red[16][64];
    constexpr int NMAIN = 3 * G_WAVE_FLOATS / 64;
    const int g = threadIdx.x >> 6, cl = threadIdx.x & 63;
    const bool main_blk = (int)blockIdx.x < NMAIN;
    const int idx = blockIdx.x * 64 + cl, dyi = main_blk ? idx / G_WAVE_FLOATS : 0, off = idx - dyi * G_WAVE_FLOATS;
    const float* src = main_blk ? ws + (long)dyi * G_WAVE_FLOATS + off : wsb + cl;
    const long stride = main_blk ? 3l * G_WAVE_FLOATS : 64l;
    const int nrec = main_blk ? nblocks : nblocks * 4;
    if (!main_blk && db == nullptr) return;
    float s = 0.f;
    int b = g;
    for (; b + 16 * 7 < nrec; b += 16 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(long)(b + 16 * u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < nrec; b += 16) s += src[(long)b * stride];
    red[g][cl] = s;
    __syncthreads();
    if (g != 0) return;
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i][cl];
    if (!main_blk) {
        db[cl] = t;
        return;
    }
    // raw offset -> tile (dx, ci half, co half), register e, lane l;  D[row][col]: row = (e&3) + 8*(e>>2) + 4*(l>>5), col = l&31
    const int l = off & 63, e = (off >> 6) & 15, tile = off >> 10, coh = tile & 1, cih = (tile >> 1) & 1, dx = tile >> 2;
    const int co = 32 * coh + (e & 3) + 8 * (e >> 2) + 4 * (l >> 5), ci = 32 * cih + (l & 31);
    dw[((dyi * 3 + dx) * 64 + ci) * 64 + co] = t;
}

constexpr int G_GRID = 256;

}  // namespace

extern "C" {

size_t vcg_conv2d_bf16_wgrad_workspace_bytes(const vcg_conv_desc* d) {
    if (!d) return 0;
    return (size_t)G_GRID * 3 * G_WAVE_FLOATS * sizeof(float) + (size_t)G_GRID * 4 * 64 * sizeof(float);
}

int vcg_conv2d_bf16_wgrad(const vcg_conv_desc* d, const void* x, const void* dy, float* dw_hwio, float* dbias, void* ws, size_t ws_bytes,
                          hipStream_t stream) {
    VCG_CHECK_PTR(d); VCG_CHECK_PTR(x); VCG_CHECK_PTR(dy); VCG_CHECK_PTR(dw_hwio); VCG_CHECK_PTR(ws);
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->oh != d->h || d->ow != d->w) return VCG_E_SHAPE;
    if (d->cin != 64 || d->cout != 64 || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad_top != 1 || d->pad_left != 1) return VCG_E_UNSUPPORTED;
    if (ws_bytes < vcg_conv2d_bf16_wgrad_workspace_bytes(d)) return VCG_E_WORKSPACE;
    WgParams p;
    p.x = (const unsigned char*)x;
    p.dy = (const unsigned char*)dy;
    p.ws = (float*)ws;
    p.wsb = (float*)ws + (size_t)G_GRID * 3 * G_WAVE_FLOATS;
    if ((long)d->h * d->w * 128 > 0xFFFFFFE0l) return VCG_E_UNSUPPORTED;        // an image behind one buffer descriptor
    p.n = d->n; p.h = d->h; p.w_ = d->w;
    p.tiles_x = ceil_div(d->w, G_C);
    p.tiles_y = ceil_div(d->h, G_R);
    p.total = p.n * p.tiles_x * p.tiles_y;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)wgrad3x3_c64_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * G_BUF);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int grid = p.total < G_GRID ? p.total : G_GRID;
    wgrad3x3_c64_bf16_kernel<<<grid, G_NWT * 64, 2 * G_BUF, stream>>>(p);
    VCG_LAUNCH_CHECK();
    static_assert(G_WAVE_FLOATS % 64 == 0, "wgrad3x3 reduce: 64 raw columns per block stay inside one tap row");
    wgrad3x3_c64_reduce2_kernel<<<3 * G_WAVE_FLOATS / 64 + 1, 1024, 0, stream>>>((const float*)ws, p.wsb, grid, dw_hwio, dbias);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

#ifdef VCG_WG_STAMPS
int vcg_debug_wg_stamps(unsigned long long* host_out) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(vcg_wg_stamp_sums), sizeof(unsigned long long) * 256 * G_NWT * 6);
}
#endif

}  // extern "C"
