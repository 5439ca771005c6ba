// Generic bf16 NHWC weight gradient on v_mfma_f32_32x32x16_bf16: the weight / bias gradients of the discriminators'
// Conv2D layers in the bf16 configs (upscaling/upscaler/model.py:839-871, 904-936; PatchGAN) and of the generator's
// Conv2DTranspose (model.py:72), 3x3 / 4x4 kernels, stride 1 or 2, channel counts that are multiples of 64:
//     dW[tap][ci][co] = sum over (n, oy, ox) of  x[n][oy*S + ky - pt][ox*S + kx - pl][ci] * dy[n][oy][ox][co]      (fp32)
// The contraction runs over PIXELS while NHWC keeps channels contiguous: both MFMA operands are needed transposed (8
// consecutive pixels of one channel per lane).  As in the trunk's kernel (bf16_wgrad.hip) they come from the natural
// [pixel][64 channels] LDS image through ds_read_b64_tr_b16 (in a 16-lane group, lane i receives halfword i&3 of the 8 bytes
// addressed by lanes (i>>2)+4j: one read = 4 pixels x 16 channels, two reads = one operand fragment).
//   * workgroup = one (64 output channels) x (64 input channels) block of dW, ALL taps: wave w keeps taps 2w, 2w+1 (two
//     [64 co] x [64 ci] blocks = 128 accumulator registers) for the whole launch; ceil(K*K/2) waves (5 / 8);
//   * persistent over pixel tiles (S = 1: 8 x 16 output pixels, S = 2: 4 x 16); a tile's dy pixels and x halo stream
//     HBM/L2 -> LDS by buffer_load ... lds through a ring of three stages (two where three do not fit), out-of-image pixels zero-filled by the descriptor's range check;
//   * LDS image: 128-byte pixel rows (this block's 64 channels); the two 64-byte halves of a row are swapped when bit 1 of
//     the pixel's column position is set, so the four consecutive pixels of a transposed read fall into four different
//     16-bank quarters.  For stride 2 the halo's columns are stored de-interleaved (even columns, then odd ones): the
//     stride-2 pixel sequence of a tap becomes consecutive positions and reads exactly like stride 1;
//   * every wave dumps its raw accumulators per slab; a second kernel sums the slabs in a fixed order (deterministic)
//     into the caller's kernel layout (strides) -- Keras' (kh,kw,in,out) for Conv2D, (kh,kw,out,in) for Conv2DTranspose.
#include "vcg_common.hpp"
#include <utility>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

template <class F, int... Is>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
    sfor_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ void lds_barrier_g() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ unsigned long long tr_read_g(unsigned addr) {
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

template <int K, int S>
struct GwCfg {
    static constexpr int T = K * K;
    static constexpr int NW = (T + 1) / 2;                      // waves: two taps each
    static constexpr int NTH = NW * 64;
    static constexpr int TR = S == 1 ? 8 : 4, TC = 16;          // output-pixel tile
    static constexpr int KSTEPS = TR * TC / 16;
    static constexpr int XR = (TR - 1) * S + K;                 // halo rows
    static constexpr int XC = (TC - 1) * S + K;                 // halo columns (image space)
    static constexpr int HALF = (XC + 1) / 2;                   // stride 2: number of even columns
    static constexpr int XPOS = S == 2 ? 2 * HALF : XC;         // stored column positions
    static constexpr int PITCH = (XPOS + 1) & ~1;               // even pitch: pixel parity == position parity
    static constexpr int DYB = TR * TC * 128;
    static constexpr int XB = XR * PITCH * 128;
    static constexpr int CHUNKS = (DYB + XB) / 16;
    static constexpr int NDMA = (CHUNKS + NTH - 1) / NTH;
    static constexpr int BUF = NDMA * NTH * 16;                 // one stage, padded to whole DMA instructions
    static constexpr int WAVE_FLOATS = 8 * 16 * 64;             // a wave's dump: 8 tiles x 16 registers x 64 lanes
    static constexpr int NS = 3 * BUF <= 160 * 1024 ? 3 : 2;    // stages: with three, TWO tiles are in flight while one is multiplied
    static_assert(2 * BUF <= 160 * 1024, "stages do not fit the LDS");
    static_assert(NDMA < 64, "vmcnt is a 6-bit counter");
    // s_waitcnt vmcnt(NDMA): everything but the youngest stage's pieces has landed (expcnt / lgkmcnt fields left open)
    static constexpr int WAIT_ONE_BEHIND = (NDMA & 15) | ((NDMA >> 4) << 14) | 0x0F70;
};

struct GwParams {
    const unsigned char* x;      // bf16 NHWC [n][h][w][cin]
    const unsigned char* dy;     // bf16 NHWC [n][oh][ow][cout]
    float* ws;                   // [slab][pair][wave][WAVE_FLOATS]
    float* wsb;                  // [slab][co block][2 lane halves][64]
    int n, h, w_, cin, oh, ow, cout, pt, pl;
    int tiles_x, tiles_y, total, slabs, ci_blocks, co_blocks;
};

template <int K, int S>
__global__ __launch_bounds__((GwCfg<K, S>::NTH), 1) void gwgrad_bf16_kernel(GwParams p) {
    using C = GwCfg<K, S>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)smem;
    int b = blockIdx.x;
    const int cib = b % p.ci_blocks; b /= p.ci_blocks;
    const int cob = b % p.co_blocks; b /= p.co_blocks;
    const int slab = b;

    // lane constants of the transposed reads: pixel position h8 + q (+ tap shift), first channel 16*((l>>4)&1) + 4*(l&3)
    const int h8 = (lane >> 5) * 8, q = (lane & 15) >> 2;
    const int chb = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;          // byte offset inside a 64-byte half
    unsigned abase[2];                                                        // dy: [co half]
#pragma unroll
    for (int coh = 0; coh < 2; ++coh) abase[coh] = (unsigned)((h8 + q) * 128 + ((64 * coh + chb) ^ (64 * ((q >> 1) & 1))));
    // x: [tap of this wave][ci half]; tap t = 2*wv + i -> (ky, kx); stored position of image column c: S == 1 ? c : de-interleaved
    const bool tap1 = 2 * wv + 1 < C::T;
    unsigned bbase[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int t = min(2 * wv + i, C::T - 1), ky = t / K, kx = t % K;
        const int pp0 = S == 2 ? (kx & 1) * C::HALF + (kx >> 1) : kx;         // position of the tile's first pixel for this tap
        const int sw = ((pp0 + q) >> 1) & 1;                                  // (+ multiples of 4 do not change bit 1)
#pragma unroll
        for (int cih = 0; cih < 2; ++cih)
            bbase[i][cih] = (unsigned)(C::DYB + (ky * C::PITCH + pp0 + h8 + q) * 128 + ((64 * cih + chb) ^ (64 * sw)));
    }

    float dbs[2] = {0.f, 0.f};
    f32x16 acc[2][2][2];                     // [tap][ci half][co half]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][c][d][e] = 0.f;

    // Staging.  A slot is one 16-byte chunk; a wave's piece of DMA round k is the 64 slots from k*NTH + wv*64, i.e. 8 consecutive stored
    // pixels of ONE region (the regions start at multiples of 64 slots): whether it is dy or x is wave-uniform.  Everything that depends on
    // (k, lane) only -- the slot's row / column inside the tile, its byte offset from the tile's first pixel, its chunk swizzle -- is loop
    // invariant; per tile a slot costs two adds, two range checks and a select.  (The first version decoded every slot per tile from
    // scratch with 64-bit pointers, per-lane branches and a zero page: ~90 VALU instructions per DMA instruction, 900 per tile against
    // 32 MFMAs -- the kernel was bound by its own address arithmetic.)  Images sit behind one buffer descriptor each: 32-bit offsets,
    // out-of-image and padding slots get the out-of-range offset (zero fill by the range check).
    const int xpix = p.cin * 2, dypix = p.cout * 2;                          // bytes per pixel in HBM
    const long ximg = (long)p.h * p.w_ * xpix, dyimg = (long)p.oh * p.ow * dypix;
    const int l3 = lane >> 3, l7 = lane & 7;
    auto dma = [&](int tile, int buf) {
        const int txi = tile % p.tiles_x, t2 = tile / p.tiles_x, tyi = t2 % p.tiles_y, img = t2 / p.tiles_y;
        const int oy0 = tyi * C::TR, ox0 = txi * C::TC;
        const int gy0 = oy0 * S - p.pt, gx0 = ox0 * S - p.pl;
        const vcg_rsrc rdy = make_rsrc(p.dy + img * dyimg, (unsigned long)dyimg), rx = make_rsrc(p.x + img * ximg, (unsigned long)ximg);
        // byte offsets of the tile's first pixels (the x one may be "negative": it is only used modulo 2^32 under a passed range check)
        const unsigned dybase = (unsigned)(oy0 * p.ow + ox0) * (unsigned)dypix + (unsigned)(cob * 128);
        const unsigned xbase = (unsigned)(gy0 * p.w_ + gx0) * (unsigned)xpix + (unsigned)(cib * 128);
#pragma unroll
        for (int k = 0; k < C::NDMA; ++k) {
            const int sb = k * C::NTH + wv * 64;                              // first slot of this wave's piece (scalar)
            void __attribute__((address_space(3)))* dst = (void __attribute__((address_space(3)))*)(smem + buf * C::BUF + sb * 16);
            if (sb < C::DYB / 16) {
                const int m = sb >> 6, row = m / (C::TC / 8), pos = (m % (C::TC / 8)) * 8 + l3;      // TC = 16: a piece is half a tile row
                const unsigned rel = (unsigned)(row * p.ow + pos) * (unsigned)dypix + (unsigned)((l7 ^ (4 * ((pos >> 1) & 1))) * 16);
                const bool ok = oy0 + row < p.oh && ox0 + pos < p.ow;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, dst, 16, ok ? dybase + rel : VCG_OOB, 0, 0, 0);
            } else {
                const int L = ((sb - C::DYB / 16) >> 3) + l3, row = L / C::PITCH, pos = L - row * C::PITCH;
                const int col = S == 2 ? (pos < C::HALF ? 2 * pos : 2 * (pos - C::HALF) + 1) : pos;
                const unsigned rel = (unsigned)(row * p.w_ + col) * (unsigned)xpix + (unsigned)((l7 ^ (4 * ((pos >> 1) & 1))) * 16);
                const bool ok = row < C::XR && col < C::XC && pos < C::XPOS && (unsigned)(gy0 + row) < (unsigned)p.h && (unsigned)(gx0 + col) < (unsigned)p.w_;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, dst, 16, ok ? xbase + rel : VCG_OOB, 0, 0, 0);
            }
        }
    };

    // ring of NS stages: the pieces of the next NS - 1 tiles are in flight while this one is multiplied (a wave's loads retire in
    // order, so "at most NDMA outstanding" means this tile's stage is complete)
    int tile = slab, buf = 0;
    if (tile < p.total) dma(tile, 0);
    if (C::NS == 3 && tile + p.slabs < p.total) dma(tile + p.slabs, 1);
    for (; tile < p.total; tile += p.slabs, buf = buf + 1 == C::NS ? 0 : buf + 1) {
        if (C::NS == 3 && tile + p.slabs < p.total) __builtin_amdgcn_s_waitcnt(C::WAIT_ONE_BEHIND);
        else __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): this wave's part of the stage has landed
        lds_barrier_g();                             // ... and everyone else's; the stage multiplied last is free again
        const int next = tile + (C::NS - 1) * p.slabs;
        if (next < p.total) dma(next, buf == 0 ? C::NS - 1 : buf - 1);
        const unsigned lb = lds0 + buf * C::BUF;

        sfor<C::KSTEPS>([&](auto ic) {
            constexpr int ks = decltype(ic)::value;          // TC = 16: one k-step per output row
            constexpr int ro = ks;
            unsigned long long fa[2][2], fb[2][2][2];
#pragma unroll
            for (int coh = 0; coh < 2; ++coh)
#pragma unroll
                for (int t = 0; t < 2; ++t) fa[coh][t] = tr_read_g(lb + abase[coh] + (unsigned)((ro * C::TC + 4 * t) * 128));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int cih = 0; cih < 2; ++cih)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        fb[i][cih][t] = tr_read_g(lb + bbase[i][cih] + (unsigned)((ro * S * C::PITCH + 4 * t) * 128));
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fb[0][0][0]), "+v"(fb[0][0][1]),
                           "+v"(fb[0][1][0]), "+v"(fb[0][1][1]), "+v"(fb[1][0][0]), "+v"(fb[1][0][1]), "+v"(fb[1][1][0]), "+v"(fb[1][1][1]));
            if (wv == 0 && cib == 0) {               // wave- and block-uniform: one wave per co block sums dy for the bias gradient
#pragma unroll
                for (int coh = 0; coh < 2; ++coh)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const bf16x4 v4 = __builtin_bit_cast(bf16x4, fa[coh][t]);
                        dbs[coh] += ((float)v4[0] + (float)v4[1]) + ((float)v4[2] + (float)v4[3]);
                    }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (i == 1 && !tap1) continue;       // wave-uniform (odd number of taps: the last wave holds one)
#pragma unroll
                for (int cih = 0; cih < 2; ++cih) {
                    const u64x2 bv = {fb[i][cih][0], fb[i][cih][1]};
                    const bf16x8 bb = __builtin_bit_cast(bf16x8, bv);
#pragma unroll
                    for (int coh = 0; coh < 2; ++coh) {
                        const u64x2 av = {fa[coh][0], fa[coh][1]};
                        acc[i][cih][coh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), bb, acc[i][cih][coh], 0, 0, 0);
                    }
                }
            }
        });
    }

    // raw register dump per wave (coalesced); decoded by the reduction
    float* out = p.ws + (((long)slab * p.co_blocks * p.ci_blocks + (long)cob * p.ci_blocks + cib) * C::NW + wv) * C::WAVE_FLOATS;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int cih = 0; cih < 2; ++cih)
#pragma unroll
            for (int coh = 0; coh < 2; ++coh)
#pragma unroll
                for (int e = 0; e < 16; ++e) out[(((i * 2 + cih) * 2 + coh) * 16 + e) * 64 + lane] = acc[i][cih][coh][e];
    if (wv == 0 && cib == 0 && p.wsb) {              // bias partials: [slab][co block][lane half][64 channels]
        float* bo = p.wsb + (((long)slab * p.co_blocks + cob) * 2 + (lane >> 5)) * 64;
        bo[lane & 31] = dbs[0];
        bo[32 + (lane & 31)] = dbs[1];
    }
}

struct GwReduce {
    const float* ws;
    const float* wsb;
    float* dw;
    float* db;
    int slabs, ci_blocks, co_blocks, nw, T, cin, cout;
    long s_t, s_ci, s_co;        // dw[t*s_t + ci*s_ci + co*s_co]
};

// one thread per element of a (pair, wave) dump: sums the slabs in a fixed order and scatters into the kernel layout
__global__ void gwgrad_reduce_kernel(GwReduce p) {
    const long per_slab = (long)p.co_blocks * p.ci_blocks * p.nw * 8192;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per_slab) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = 0;
    for (; b + 7 < p.slabs; b += 8) {                                   // eight slabs in flight per thread, fixed order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p.ws[(long)(b + u) * per_slab + idx];
        s0 += v[0]; s1 += v[1]; s2 += v[2]; s3 += v[3];
        s0 += v[4]; s1 += v[5]; s2 += v[6]; s3 += v[7];
    }
    for (; b < p.slabs; ++b) s0 += p.ws[(long)b * per_slab + idx];
    // idx -> (pair, wave, tile (i, cih, coh), register e, lane l);  D[row][col]: row = (e&3) + 8*(e>>2) + 4*(l>>5), col = l&31
    const int off = (int)(idx & 8191);
    long r = idx >> 13;
    const int wv = (int)(r % p.nw); r /= p.nw;
    const int cib = (int)(r % p.ci_blocks);
    const int cob = (int)(r / p.ci_blocks);
    const int l = off & 63, e = (off >> 6) & 15, tile = off >> 10, coh = tile & 1, cih = (tile >> 1) & 1, i = tile >> 2;
    const int t = 2 * wv + i;
    if (t >= p.T) return;
    const int co = cob * 64 + 32 * coh + (e & 3) + 8 * (e >> 2) + 4 * (l >> 5), ci = cib * 64 + 32 * cih + (l & 31);
    p.dw[(long)t * p.s_t + (long)ci * p.s_ci + (long)co * p.s_co] = (s0 + s1) + (s2 + s3);
}

__global__ void gwgrad_bias_reduce_kernel(GwReduce p) {
    const int co = blockIdx.x * blockDim.x + threadIdx.x;
    if (co >= p.cout) return;
    const int cob = co >> 6, c = co & 63;
    float s = 0.f;
    int b = 0;
    for (; b + 7 < p.slabs; b += 8) {                                   // eight slabs in flight (one thread per channel: latency-bound otherwise)
        float v[8][2];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float* bo = p.wsb + (((long)(b + u) * p.co_blocks + cob) * 2) * 64;
            v[u][0] = bo[c];
            v[u][1] = bo[64 + c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u][0] + v[u][1];
    }
    for (; b < p.slabs; ++b) {
        const float* bo = p.wsb + (((long)b * p.co_blocks + cob) * 2) * 64;
        s += bo[c] + bo[64 + c];
    }
    p.db[co] = s;
}

constexpr int GW_MAX_WG = 256;       // one 5- or 8-wave workgroup per CU (the ring of stages fills its LDS), ONE round: a second round would pay the
                                     // pipeline ramp and the accumulator dump again and double the partial blocks the reduction reads

struct GwPlan { int tiles_x, tiles_y, total, slabs, ci_blocks, co_blocks, nw, TR; size_t ws_part, ws_bytes; bool ok; };

GwPlan gw_plan(int n, int cin, int oh, int ow, int cout, int k, int s) {
    GwPlan pl{};
    pl.ok = false;
    if (cin % 64 || cout % 64 || k < 3 || k > 4 || (s != 1 && s != 2)) return pl;   // 5x5: 13 waves leave 128 registers each -- fp32 path
    pl.TR = s == 1 ? 8 : 4;
    pl.nw = (k * k + 1) / 2;
    pl.tiles_x = ceil_div(ow, 16);
    pl.tiles_y = ceil_div(oh, pl.TR);
    pl.total = pl.tiles_x * pl.tiles_y * n;
    pl.ci_blocks = cin / 64;
    pl.co_blocks = cout / 64;
    const int pairs = pl.ci_blocks * pl.co_blocks;
    int slabs = GW_MAX_WG / pairs;
    if (slabs < 1) slabs = 1;
    if (slabs > pl.total) slabs = pl.total;
    pl.slabs = slabs;
    pl.ws_part = align_up((size_t)slabs * pairs * pl.nw * 8192 * sizeof(float), 256);
    pl.ws_bytes = pl.ws_part + (size_t)slabs * pl.co_blocks * 128 * sizeof(float);
    pl.ok = true;
    return pl;
}

template <int K, int S>
int launch_gw(const GwParams& p, int grid, hipStream_t st) {
    using C = GwCfg<K, S>;
    auto kern = gwgrad_bf16_kernel<K, S>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::NS * C::BUF);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NTH), C::NS * C::BUF, st, p);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

// x: [n][h][w][cin], dy: [n][oh][ow][cout]; dw[t*s_t + ci*s_ci + co*s_co]
int gw_run(const void* x, const void* dy, float* dw, float* db, int n, int h, int w, int cin, int oh, int ow, int cout, int k, int s,
           int pt, int pl_, long s_t, long s_ci, long s_co, void* ws, size_t ws_bytes, hipStream_t st) {
    const GwPlan pl = gw_plan(n, cin, oh, ow, cout, k, s);
    if (!pl.ok) return VCG_E_UNSUPPORTED;
    if (ws == nullptr || ws_bytes < pl.ws_bytes) return VCG_E_WORKSPACE;
    if ((long)h * w * cin * 2 > 0xFFFFFFE0l || (long)oh * ow * cout * 2 > 0xFFFFFFE0l) return VCG_E_UNSUPPORTED;     // an image behind one buffer descriptor
    GwParams p{};
    p.x = (const unsigned char*)x; p.dy = (const unsigned char*)dy;
    p.ws = (float*)ws; p.wsb = (float*)((char*)ws + pl.ws_part);
    p.n = n; p.h = h; p.w_ = w; p.cin = cin; p.oh = oh; p.ow = ow; p.cout = cout; p.pt = pt; p.pl = pl_;
    p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.total = pl.total; p.slabs = pl.slabs;
    p.ci_blocks = pl.ci_blocks; p.co_blocks = pl.co_blocks;
    const int grid = pl.slabs * pl.ci_blocks * pl.co_blocks;
    int rc = VCG_E_UNSUPPORTED;
#define VCG_GW(K_, S_) if (k == K_ && s == S_) rc = launch_gw<K_, S_>(p, grid, st)
    VCG_GW(3, 1); VCG_GW(3, 2); VCG_GW(4, 1); VCG_GW(4, 2);
#undef VCG_GW
    if (rc != VCG_OK) return rc;
    GwReduce r{};
    r.ws = p.ws; r.wsb = p.wsb; r.dw = dw; r.db = db;
    r.slabs = pl.slabs; r.ci_blocks = pl.ci_blocks; r.co_blocks = pl.co_blocks; r.nw = pl.nw; r.T = k * k; r.cin = cin; r.cout = cout;
    r.s_t = s_t; r.s_ci = s_ci; r.s_co = s_co;
    const long per_slab = (long)pl.co_blocks * pl.ci_blocks * pl.nw * 8192;
    hipLaunchKernelGGL(gwgrad_reduce_kernel, dim3((unsigned)((per_slab + 255) / 256)), dim3(256), 0, st, r);
    VCG_LAUNCH_CHECK();
    if (db) {
        hipLaunchKernelGGL(gwgrad_bias_reduce_kernel, dim3(ceil_div(cout, 256)), dim3(256), 0, st, r);
        VCG_LAUNCH_CHECK();
    }
    return VCG_OK;
}

}  // namespace

extern "C" {

size_t vcg_conv2d_nhwc_bf16_wgrad_workspace_bytes(const vcg_conv_desc* d) {
    if (!d) return 0;
    const GwPlan pl = gw_plan(d->n, d->cin, d->oh, d->ow, d->cout, d->kh, d->stride);
    return pl.ok ? pl.ws_bytes : 0;
}

// Conv2D: dw in Keras' (kh,kw,in,out) layout, dbias [cout] or NULL
int vcg_conv2d_nhwc_bf16_wgrad(const vcg_conv_desc* d, const void* x, const void* dy, float* dw_hwio, float* dbias, void* ws, size_t ws_bytes,
                               hipStream_t stream) {
    VCG_CHECK_PTR(d); VCG_CHECK_PTR(x); VCG_CHECK_PTR(dy); VCG_CHECK_PTR(dw_hwio);
    if (d->kh != d->kw) return VCG_E_UNSUPPORTED;
    return gw_run(x, dy, dw_hwio, dbias, d->n, d->h, d->w, d->cin, d->oh, d->ow, d->cout, d->kh, d->stride, d->pad_top, d->pad_left,
                  (long)d->cin * d->cout, d->cout, 1, ws, ws_bytes, stream);
}

size_t vcg_conv_transpose2d_nhwc_bf16_wgrad_workspace_bytes(const vcg_conv_desc* d) {
    if (!d) return 0;
    // roles exchanged: the transposed convolution's input plays dy, its output gradient plays x
    const GwPlan pl = gw_plan(d->n, d->cout, d->h, d->w, d->cin, d->kh, d->stride);
    return pl.ok ? pl.ws_bytes : 0;
}

// Conv2DTranspose(strides=2,'same') (d: cin,h,w -> cout,oh,ow; pads = crop-before): dW[k][co][ci] = sum_i x[ci][i] * dz[co][2i + k - cb]
// = the weight gradient of the stride-2 convolution dz -> x; dw in Keras' (kh,kw,out,in) layout.  The bias gradient (sum of dz)
// comes from the caller's activation-backward pass.
int vcg_conv_transpose2d_nhwc_bf16_wgrad(const vcg_conv_desc* d, const void* x, const void* dz, float* dw_hwoi, void* ws, size_t ws_bytes,
                                         hipStream_t stream) {
    VCG_CHECK_PTR(d); VCG_CHECK_PTR(x); VCG_CHECK_PTR(dz); VCG_CHECK_PTR(dw_hwoi);
    if (d->kh != d->kw) return VCG_E_UNSUPPORTED;
    // "x" of the kernel = dz [n][oh][ow][cout] (channel index = co), "dy" of the kernel = the layer input [n][h][w][cin] (index = ci)
    // kernel result R[t][ci_k = co][co_k = ci] -> dw[(t*cout + co)*cin + ci]
    return gw_run(dz, x, dw_hwoi, nullptr, d->n, d->oh, d->ow, d->cout, d->h, d->w, d->cin, d->kh, d->stride, d->pad_top, d->pad_left,
                  (long)d->cout * d->cin, d->cin, 1, ws, ws_bytes, stream);
}

}  // extern "C"
