// Generic bf16 NHWC convolution on v_mfma_f32_32x32x16_bf16 -- forward and data gradient of every Conv2D the
// discriminators instantiate in the bf16 configs (BASELINE.json C3/C4): 3x3 / 4x4 / 5x5, stride 1 / 2, 64...512 channels
// (upscaling/upscaler/model.py:839-871 simple_512, :904-936 thin_512; the PatchGAN extension), and the data gradient of
// the generator's Conv2DTranspose (model.py:72).
//
// Implicit GEMM, A = weights (rows = output channels), B = activations (columns = 32 output pixels), k = (tap, 16 input
// channels).  With NHWC a lane's B fragment -- 8 consecutive channels of ITS pixel -- is one 16-byte load, whatever the
// stride, padding or tap: the kernel streams both operands straight into registers, no LDS image, no barrier, every wave on
// its own.  Weights are re-laid out once per optimizer step as MFMA operand fragments ([tap][k-step][32-row block][lane] x
// 16 bytes), so a wave's A fragment is one coalesced 1-KiB load that all waves of the chip share through L1 / L2; the
// activation loads go through a range-checked buffer descriptor (padding and ragged tiles read as zero, vcg_common.hpp).
// The discriminators' tensors are small (<= 67 MB at batch 8) and L2-resident: what bounds the kernel is the vector L1 /
// address path (one 16-byte load per MFMA and wave), i.e. it cannot reach the MFMA roof the LDS-tiled trunk kernel aims
// at -- but it serves every shape with one code path at several times the fp32 kernels' rate, and the data gradient of a
// strided convolution is the same kernel run once per output phase with that phase's tap list.
#include "vcg_common.hpp"
#include <cstdlib>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int GC_MAXTAPS = 25;

struct GcTap { short dy, dx, wt, pad; };

struct GcParams {
    const void* x;          // bf16 NHWC [n][ih][iw][kch]
    const void* wf;         // fragments [wt][kch/16][mblocks][64 lanes] x 16 bytes
    void* y;                // bf16 NHWC [n][oh][ow][mch]
    const float* bias;      // [mch] or null
    const void* mask_src;   // optional bf16 NHWC tensor of the output's shape: out *= (mask_src > 0 ? 1 : mask_slope)
    size_t wbytes;
    int n, ih, iw, kch;     // input tensor
    int oh, ow, mch;        // output tensor
    int loh, low;           // logical output grid of this launch (a phase of a strided data gradient, or the whole output)
    int osy, osx, ooy, oox; // output pixel = (ly*osy + ooy, lx*osx + oox)
    int isy, isx;           // input pixel  = (ly*isy + tap.dy, lx*isx + tap.dx)
    int ntaps, mblocks;
    int act;
    float alpha, mask_slope;
    float* stats;           // LDS-tiled kernel only: per-tile sums / sums of squares of the stored output, [n][tiles per image][2][mch]
    GcTap taps[GC_MAXTAPS];
};

__device__ __forceinline__ void swap32u(float& a, float& b) {
    const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(sw.x);
    b = __uint_as_float(sw.y);
}

__device__ __forceinline__ bf16x8 ld_frag(vcg_rsrc r, unsigned off) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}

// workgroup = 4 waves; wave = 64 output channels (2 MFMA row blocks) x NT tiles of 32 consecutive logical pixels
template <int NT>
__global__ __launch_bounds__(256, (NT == 4 ? 2 : 4)) void gconv_bf16_kernel(const GcParams p) {
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mb0 = blockIdx.y * 2;                                     // first 32-row block of this workgroup
    const bool two = mb0 + 1 < p.mblocks;                               // (mch = 32: only one row block)
    const long total = (long)p.n * p.loh * p.low;
    const long tile0 = ((long)blockIdx.x * 4 + wv) * NT;              // first pixel tile of this wave
    if (tile0 * 32 >= total) return;

    // this lane's pixel in each tile: coordinates in the logical grid
    int img[NT], ly[NT], lx[NT];
    bool pok[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const long P = (tile0 + t) * 32 + r;
        pok[t] = P < total;
        const long Pc = pok[t] ? P : 0;
        lx[t] = (int)(Pc % p.low);
        const long q = Pc / p.low;
        ly[t] = (int)(q % p.loh);
        img[t] = (int)(q / p.loh);
    }
    const vcg_rsrc rx = make_rsrc(p.x, (size_t)p.n * p.ih * p.iw * p.kch * 2);
    const vcg_rsrc rw = make_rsrc(p.wf, p.wbytes);
    const int ksteps = p.kch >> 4;
    const unsigned wlane = (unsigned)lane * 16u;
    const unsigned wstep = (unsigned)p.mblocks * 1024u;                 // bytes between k-steps of one tap

    f32x16 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][t][e] = 0.f;

    for (int ti = 0; ti < p.ntaps; ++ti) {
        const GcTap tp = p.taps[ti];
        unsigned xoff[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int iy = ly[t] * p.isy + tp.dy, ix = lx[t] * p.isx + tp.dx;
            const bool ok = pok[t] && (unsigned)iy < (unsigned)p.ih && (unsigned)ix < (unsigned)p.iw;
            xoff[t] = ok ? (unsigned)((((long)img[t] * p.ih + iy) * p.iw + ix) * p.kch + 8 * hh) * 2u : VCG_OOB;
        }
        unsigned woff = ((unsigned)tp.wt * ksteps * p.mblocks + mb0) * 1024u + wlane;
        // two k-steps per iteration: all loads of both first, then their MFMAs (the other waves of the SIMD cover the latency)
        int ks = 0;
        for (; ks + 2 <= ksteps; ks += 2) {
            bf16x8 a[2][2], b[2][NT];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                a[u][0] = ld_frag(rw, woff + u * wstep);
                a[u][1] = ld_frag(rw, two ? woff + u * wstep + 1024u : VCG_OOB);
#pragma unroll
                for (int t = 0; t < NT; ++t) b[u][t] = ld_frag(rx, xoff[t] == VCG_OOB ? VCG_OOB : xoff[t] + (unsigned)(ks + u) * 32u);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u][0], b[u][t], acc[0][t], 0, 0, 0);
                    acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u][1], b[u][t], acc[1][t], 0, 0, 0);
                }
            woff += 2 * wstep;
        }
        if (ks < ksteps) {                                               // odd number of k-steps (kch = 16, 48, ...)
            const bf16x8 a0 = ld_frag(rw, woff), a1 = ld_frag(rw, two ? woff + 1024u : VCG_OOB);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bf16x8 b = ld_frag(rx, xoff[t] == VCG_OOB ? VCG_OOB : xoff[t] + (unsigned)ks * 32u);
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b, acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b, acc[1][t], 0, 0, 0);
            }
        }
    }

    // epilogue: y = act(acc + bias) [* mask] -> bf16.  An MFMA tile leaves lane (pixel r, half hh) with channels
    // 8g + 4hh + {0..3}; v_permlane32_swap between register groups (2q, 2q+1) of the two half-waves turns that into 8
    // consecutive channels 16q + 8hh + {0..7}: one 16-byte store per (row block, q)
    const float slope = p.act == VCG_ACT_LRELU ? p.alpha : 1.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int oy = ly[t] * p.osy + p.ooy, ox = lx[t] * p.osx + p.oox;
        const bool ok = pok[t] && oy < p.oh && ox < p.ow;
        const long obase = (((long)img[t] * p.oh + oy) * p.ow + ox) * p.mch;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int co = (mb0 + m) * 32 + 16 * q + 8 * hh;
                float v[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float lo = acc[m][t][8 * q + j], hi = acc[m][t][8 * q + 4 + j];
                    swap32u(lo, hi);
                    v[j] = lo;
                    v[4 + j] = hi;
                }
                if (!ok || (m == 1 && !two)) continue;
                bf16x8 mk;
                if (p.mask_src) mk = *(const bf16x8*)((const __bf16*)p.mask_src + obase + co);
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float u = v[j] + (p.bias ? p.bias[co + j] : 0.f);
                    u = u >= 0.f ? u : u * slope;
                    if (p.mask_src) u *= ((float)mk[j] > 0.f ? 1.f : p.mask_slope);
                    o[j] = (__bf16)u;
                }
                *(bf16x8*)((__bf16*)p.y + obase + co) = o;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-tiled variant: 64-channel input chunks, 128 output channels per workgroup, any stride-1 / stride-2 forward convolution
// and EVERY data gradient (a phase of a strided layer's gradient reads dy at unit stride)
// ---------------------------------------------------------------------------------------------------------------
// The streaming kernel above asks the vector L1 for 32 different 128-byte lines per operand load (one per pixel, 32 bytes of
// each used) and re-asks for every tap: at the discriminators' 256->512 layer it runs at 220-270 TFLOP/s whatever its
// occupancy (scripts/kbench_gconv.py).  Here a workgroup stages the (8+SP) x (32+SP)-pixel halo of one 64-channel chunk in LDS
// by `buffer_load ... lds` (whole 128-byte lines, 1 KiB per instruction, zero padding by the range check), double-buffered
// over the (tile, chunk) sequence it walks, and reads it the way the trunk kernel does: XOR-swizzled 16-byte fragments, the
// rows of one (dx, channel group) read once for all dy.  Wave w owns 32 of the workgroup's 128 output channels x the tile's
// 8 rows (128 accumulator registers); its weight fragments stream from L1/L2 as in the kernel above (one coalesced 1-KiB load
// per 8 MFMAs), requested one k-group ahead.  One LDS-only barrier per chunk.
// A stride-s forward convolution is the sum over the s x s PARITY PLANES of its input (x_pq[i][j] = x[s*i + p][s*j + q]) of
// unit-stride convolutions with the taps of that parity (4x4 stride 2: four planes of 2x2 taps): the planes are extra entries of
// the chunk loop, and the DMA gathers a plane's halo with a pixel stride of s (each pixel still one whole 128-byte line).
constexpr int GL_TR = 8, GL_TC = 32;

struct GlParams {
    const void* x;          // bf16 NHWC [n][ih][iw][kch]
    const void* wf;         // fragments [wt][kch/16][mblocks][64 lanes] x 16 bytes
    void* y;                // bf16 NHWC [n][oh][ow][mch]
    const float* bias;
    const void* mask_src;
    size_t wbytes;
    int n, ih, iw, kch, oh, ow, mch, loh, low, osy, osx, ooy, oox, mblocks, act;
    float alpha, mask_slope;
    int isy, isx;           // input pixel of plane (py, px), plane coordinates (i, j): (isy*i + py, isx*j + px)
    int tiles_x, tiles_y, pairs, mgroups, nplanes;
    // the S x S output phases of a strided layer's data gradient merged into ONE launch (phase = fastest index of the pair: the workgroups
    // that read the same dy tile with the four tap sets run side by side); each phase has one plane pl[phase] and its own output placement
    int nphases;
    struct Phase { int ooy, oox, loh, low; } ph[4];
    float* stats;           // optional: [n][tiles_y * tiles_x][2][mch] sums and sums of squares of the stored (bf16-rounded) output per tile
    struct Plane {
        int py, px, dy0, dx0;     // parity; smallest tap offsets in plane coordinates: halo origin of a tile = (ly0 + dy0, lx0 + dx0)
        int wt[25];               // weight tap of offset (dyo, dxo) from the halo origin, row-major (SP+1)^2, or -1  (dwords: scalar loads)
    } pl[4];
};

struct GlSrc { int img, y0, x0, mg, ty, tx, ph; };   // a (tile, output-channel group, phase) pair: image, halo origin, group, tile coordinates, phase

__device__ __forceinline__ void gl_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// HALF: a layer with 64 output channels (the data gradient of a 64-channel layer, the transposed convolution's data gradient) would leave two
// of the four waves without a 32-channel block; there a wave owns one of the TWO blocks x one HALF of the tile's rows instead (no statistics in this form).
// The HALF form (its waves hold half the accumulators: two fit a SIMD) also runs four LOADER waves that do nothing but request the halos:
// a wave's vector-memory loads return in order, so in the one-role form the weight fragments of a k-group queue behind the halo pieces
// requested before them, and a halo request that waits for room in the memory pipe holds up the MFMAs behind it.  The loaders' queue holds
// halo pieces only, which also lets them run a ring of THREE stages (two entries in flight; the same ring in the one-role form was
// measured slower, profiles/r03_gconv_ring_ab.txt).  What it serves: the transposed convolution's data gradient -- 36 MFMAs per wave and
// entry over a 1.07 GB tensor -- and the 64-channel data gradients of the critics.
template <int SP, bool HALF>
__global__ __launch_bounds__(HALF ? 512 : 256, 1) void gconv_lds_bf16_kernel(const GlParams p) {
    constexpr int TRW = HALF ? GL_TR / 2 : GL_TR;                // output rows per wave
    constexpr int HR = GL_TR + SP, HC = GL_TC + SP, ROWB = HC * 128, XB = HR * ROWB, CH16 = XB / 16;
    constexpr int NDMA = (CH16 + 255) / 256, BUF = NDMA * 4096, G = (SP + 1) * 4;
    constexpr bool LOADERS = HALF;
    constexpr int NS = LOADERS && 3 * BUF <= 160 * 1024 ? 3 : 2, AHEAD = NS - 1;
    constexpr int WAIT_ONE_BEHIND = (NDMA & 15) | ((NDMA >> 4) << 14) | 0x0F70;     // s_waitcnt vmcnt(NDMA)
    static_assert(2 * BUF <= 160 * 1024 && NDMA < 64, "gconv_lds: LDS / vmcnt");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mw = HALF ? (wv & 1) : wv, r0 = HALF ? (wv >> 1) * TRW : 0;      // 32-channel block inside the group, first row inside the tile
    const int nchunk = (p.kch >> 6) * p.nplanes, ksteps = p.kch >> 4;      // chunk = (parity plane, 64 input channels), plane-major
    const long img_in = (long)p.ih * p.iw * p.kch * 2, img_out = (long)p.oh * p.ow * p.mch * 2;
    const vcg_rsrc rw = make_rsrc(p.wf, p.wbytes);

    // LDS byte offset of lane (pixel r + dxo, half hh)'s fragment of channel group s of the chunk: boff[dxo] ^ (s << 5)
    int boff[SP + 1];
#pragma unroll
    for (int d = 0; d <= SP; ++d) {
        const int pos = r + d;
        boff[d] = pos * 128 + ((hh ^ ((pos >> 1) & 7)) << 4);
    }
    // halo slot of this lane in DMA round k: pixel (row, col), chunk position -- advanced incrementally (32 pixels per round)
    auto decode = [&](int pair) {
        const int ph = pair % p.nphases, pq = pair / p.nphases;
        const int mg = pq % p.mgroups, t = pq / p.mgroups;
        const int tx = t % p.tiles_x, t2 = t / p.tiles_x, ty = t2 % p.tiles_y, img = t2 / p.tiles_y;
        return GlSrc{img, ty * GL_TR, tx * GL_TC, mg, ty, tx, ph};
    };
    const int cpp = p.kch >> 6;                                  // chunks per plane
    const int dtid = LOADERS ? tid - 256 : tid, dwv = LOADERS ? wv - 4 : wv;     // the staging thread / wave index (loader waves: 4..7)
    auto dma = [&](const GlSrc& sc, int c, int buf, int k, bool live) {
        const int sl = k * 256 + dtid, P = sl >> 3, row = P / HC, col = P - row * HC;
        const int cs = (sl & 7) ^ ((col >> 1) & 7);                                // stored chunk (sl & 7) holds source chunk cs
        const int cc = c % cpp, pli = sc.ph + c / cpp;            // (a merged phase has one plane: c / cpp == 0)
        const int iy = p.isy * (sc.y0 + p.pl[pli].dy0 + row) + p.pl[pli].py, ix = p.isx * (sc.x0 + p.pl[pli].dx0 + col) + p.pl[pli].px;
        const bool ok = live && sl < CH16 && (unsigned)iy < (unsigned)p.ih && (unsigned)ix < (unsigned)p.iw;
        unsigned off = ((unsigned)(iy * p.iw + ix) * (unsigned)p.kch + (unsigned)(cc * 64)) * 2u + (unsigned)(cs * 16);      // < 4 GiB: unsigned arithmetic
        asm volatile("" : "+v"(off));                            // a select, not a branch around the arithmetic
        off = ok ? off : VCG_OOB;
        const vcg_rsrc rx = make_rsrc((const unsigned char*)p.x + sc.img * img_in, (unsigned long)img_in);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (void __attribute__((address_space(3)))*)(smem + buf * BUF + (k * 256 + dwv * 64) * 16), 16, off, 0, 0, 0);
    };
    auto afrag = [&](int wt, int ks, int mtile) {      // (a wave past the last 32-channel block of a ragged group computes zeros)
        return ld_frag(rw, wt < 0 || mtile >= p.mblocks ? VCG_OOB : (unsigned)(((wt * ksteps + ks) * p.mblocks + mtile) * 1024 + lane * 16));
    };

    // the entries (pair, chunk) in the order this workgroup walks them
    auto succ = [&](int& pr, int& ch) {
        if (ch + 1 == nchunk) { pr += gridDim.x; ch = 0; } else ++ch;
    };
    int pair = blockIdx.x;
    if (pair >= p.pairs) return;
    GlSrc cur = decode(pair);
    if (LOADERS && wv >= 4) {
        // loader waves: entry i + AHEAD is requested right behind barrier i (which frees its stage) and entry i + 1 has landed before
        // barrier i + 1.  Every request issues NDMA pieces per wave (past the last entry: zero-record descriptors), so "at most NDMA
        // outstanding" always means "all but the youngest entry's pieces".
        auto request = [&](int pr, int ch, int buf) {
            const bool live = pr < p.pairs;
            const GlSrc sc = decode(live ? pr : (int)blockIdx.x);
#pragma unroll
            for (int k = 0; k < NDMA; ++k) dma(sc, ch, buf, k, live);
        };
        int pr = pair, ch = 0, ap = pair, ac = 0, abuf = 0;      // the entry being multiplied; the next one to request and its stage
        for (int a = 0; a < AHEAD; ++a) {
            request(ap, ac, abuf);
            succ(ap, ac);
            abuf = abuf + 1 == NS ? 0 : abuf + 1;
        }
        if (AHEAD == 2) __builtin_amdgcn_s_waitcnt(WAIT_ONE_BEHIND);
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        gl_barrier();
        while (true) {
            request(ap, ac, abuf);
            succ(ap, ac);
            abuf = abuf + 1 == NS ? 0 : abuf + 1;
            succ(pr, ch);
            if (pr >= p.pairs) break;                            // the compute waves multiply their last entry without another barrier
            if (AHEAD == 2) __builtin_amdgcn_s_waitcnt(WAIT_ONE_BEHIND);
            else __builtin_amdgcn_s_waitcnt(0x0F70);
            gl_barrier();
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                      // (nothing of this wave may still be writing LDS when the workgroup ends)
        return;
    }
    if (!LOADERS) {
#pragma unroll
        for (int k = 0; k < NDMA; ++k) dma(cur, 0, 0, k, true);
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    gl_barrier();

    f32x16 acc[TRW];
#pragma unroll
    for (int n = 0; n < TRW; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
    // statistics for the normalisation behind the convolution (p.stats): per lane [sum | sum of squares][q][j] of its 16 channels
    float sacc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) sacc[i] = 0.f;

    int buf = 0, c = 0;
    while (true) {
        // what comes after this (tile, chunk): the next chunk of the tile, or the first chunk of the workgroup's next pair
        const bool last_chunk = c + 1 == nchunk;
        const int npair = last_chunk ? pair + gridDim.x : pair;
        const bool has_next = npair < p.pairs;
        const GlSrc nxt = last_chunk ? decode(has_next ? npair : pair) : cur;
        const int nc = last_chunk ? 0 : c + 1;
        const unsigned char* xb = smem + buf * BUF;
        const int mtile = cur.mg * (HALF ? 2 : 4) + mw;
        const int cc = c % cpp, pli = cur.ph + c / cpp;

        bf16x8 fb[2][TRW + SP], a[2][SP + 1];
        auto frag = [&](int g, int b) {
            const int d = g >> 2, s = g & 3;
#pragma unroll
            for (int j = 0; j < TRW + SP; ++j) fb[b][j] = *(const bf16x8*)(xb + (r0 + j) * ROWB + (boff[d] ^ (s << 5)));
        };
        auto wfrag = [&](int g, int b) {
            const int d = g >> 2, s = g & 3;
#pragma unroll
            for (int dy = 0; dy <= SP; ++dy) a[b][dy] = afrag(p.pl[pli].wt[dy * (SP + 1) + d], cc * 4 + s, mtile);
        };
        frag(0, 0);
        wfrag(0, 0);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int cb = g & 1;
            if (g + 1 < G) {
                wfrag(g + 1, cb ^ 1);                                  // weights one group ahead (older than this group's DMA pieces)
                frag(g + 1, cb ^ 1);
            }
            if (!LOADERS) {
#pragma unroll
                for (int k = g; k < NDMA; k += G) dma(nxt, nc, buf ^ 1, k, has_next);
            }
#pragma unroll
            for (int dy = 0; dy <= SP; ++dy) {
                if (p.pl[pli].wt[dy * (SP + 1) + (g >> 2)] < 0) continue;     // uniform: a phase / parity plane lacks some offsets of the box
#pragma unroll
                for (int n = 0; n < TRW; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cb][dy], fb[cb][n + dy], acc[n], 0, 0, 0);
            }
        }
        if (last_chunk) {
            // epilogue: y = act(acc + bias) [* mask] -> bf16; v_permlane32_swap makes 8 consecutive channels per lane (16-byte stores)
            const float slope = p.act == VCG_ACT_LRELU ? p.alpha : 1.f;
            const vcg_rsrc ry = make_rsrc((unsigned char*)p.y + cur.img * img_out, (unsigned long)img_out);
            const vcg_rsrc rm = make_rsrc((const unsigned char*)p.mask_src + cur.img * img_out, (unsigned long)(p.mask_src ? img_out : 0));
            const int lx = cur.tx * GL_TC + r, ox = lx * p.osx + p.ph[cur.ph].oox;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int co = mtile * 32 + 16 * q + 8 * hh;
                float bs[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) bs[j] = p.bias && mtile < p.mblocks ? p.bias[co + j] : 0.f;
#pragma unroll
                for (int n = 0; n < TRW; ++n) {
                    const int ly = cur.ty * GL_TR + r0 + n, oy = ly * p.osy + p.ph[cur.ph].ooy;
                    const bool ok = ly < p.ph[cur.ph].loh && lx < p.ph[cur.ph].low && oy < p.oh && ox < p.ow && mtile < p.mblocks;
                    unsigned off = ((unsigned)(oy * p.ow + ox) * (unsigned)p.mch + (unsigned)co) * 2u;
                    asm volatile("" : "+v"(off));
                    off = ok ? off : VCG_OOB;
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float lo = acc[n][8 * q + j], hi = acc[n][8 * q + 4 + j];
                        swap32u(lo, hi);
                        v[j] = lo;
                        v[4 + j] = hi;
                    }
                    bf16x8 mk;
                    if (p.mask_src) mk = ld_frag(rm, off);
                    bf16x8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float u = v[j] + bs[j];
                        u = u >= 0.f ? u : u * slope;
                        if (p.mask_src) u *= ((float)mk[j] > 0.f ? 1.f : p.mask_slope);
                        o[j] = (__bf16)u;
                    }
                    const u32x4 ob = __builtin_bit_cast(u32x4, o);
                    __builtin_amdgcn_raw_buffer_store_b128(ob, ry, (int)off, 0, 0);
                    if (!HALF && p.stats) {                          // (uniform) the values as stored, zero outside the tensor
                        const unsigned m = ok ? 0xFFFFFFFFu : 0u;
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const unsigned w = ob[d] & m;
                            const float lo = __uint_as_float(w << 16), hi = __uint_as_float(w & 0xFFFF0000u);
                            sacc[8 * q + 2 * d] += lo;
                            sacc[16 + 8 * q + 2 * d] = fmaf(lo, lo, sacc[16 + 8 * q + 2 * d]);
                            sacc[8 * q + 2 * d + 1] += hi;
                            sacc[16 + 8 * q + 2 * d + 1] = fmaf(hi, hi, sacc[16 + 8 * q + 2 * d + 1]);
                        }
                    }
                }
            }
            if (!HALF && p.stats) {
                // one record per tile: lane (r, hh) ends up with value r = [stat][q][j] summed over the tile's 8 x 32 pixels
                const float t = half_wave_reduce_scatter32(sacc, r);
                const long tile = ((long)cur.img * p.tiles_y + cur.ty) * p.tiles_x + cur.tx;
                if (mtile < p.mblocks) p.stats[(tile * 2 + (r >> 4)) * p.mch + mtile * 32 + ((r >> 3) & 1) * 16 + 8 * hh + (r & 7)] = t;
#pragma unroll
                for (int i = 0; i < 32; ++i) sacc[i] = 0.f;
            }
#pragma unroll
            for (int n = 0; n < TRW; ++n)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
        }
        if (last_chunk && !has_next) break;
        // the next (tile, chunk)'s pieces were issued under this chunk's MFMAs: retire them, then the one barrier of the chunk
        if (!LOADERS) __builtin_amdgcn_s_waitcnt(0x0F70);
        gl_barrier();
        buf = buf + 1 == NS ? 0 : buf + 1;
        c = nc;
        if (last_chunk) { pair = npair; cur = nxt; }
    }
}

// out[((wt*ksteps + ks)*mblocks + mb)*64 + lane][j] = W(m = mb*32 + (lane&31), k = ks*16 + 8*(lane>>5) + j, tap wt)
//   mode 0: W(m, k, t) = w[(t*kdim + k)*mdim + m]     a Keras (kh,kw,in,out) kernel read as conv forward  (m = out, k = in)
//   mode 1: W(m, k, t) = w[(t*mdim + m)*kdim + k]     the same kernel read for its data gradient          (m = in,  k = out)
__global__ void pack_frag_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int taps, int mdim, int kdim, int mode) {
    const int ksteps = kdim >> 4, mblocks = mdim >> 5;
    const long total = (long)taps * ksteps * mblocks * 64 * 8;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        long q = idx >> 9;
        const int mb = (int)(q % mblocks); q /= mblocks;
        const int ks = (int)(q % ksteps);
        const int t = (int)(q / ksteps);
        const int m = mb * 32 + (lane & 31), k = ks * 16 + 8 * (lane >> 5) + j;
        const float v = mode == 0 ? w[((long)t * kdim + k) * mdim + m] : w[((long)t * mdim + m) * kdim + k];
        out[idx] = (__bf16)v;
    }
}

// both operand copies of one Keras (kh,kw,in,out) kernel in one launch: out_fwd = mode 0 with (mdim, kdim) = (cout, cin), out_dgrad = mode 1 with
// (mdim, kdim) = (cin, cout)
__global__ void pack_frag_pair_kernel(const float* __restrict__ w, __bf16* __restrict__ out_fwd, __bf16* __restrict__ out_dgrad, int taps, int cin, int cout) {
    const long total = (long)taps * cin * cout;
    for (long i2 = blockIdx.x * (long)blockDim.x + threadIdx.x; i2 < 2 * total; i2 += (long)gridDim.x * blockDim.x) {
        const int mode = i2 >= total;
        const long idx = mode ? i2 - total : i2;
        const int mdim = mode ? cin : cout, kdim = mode ? cout : cin;
        const int ksteps = kdim >> 4, mblocks = mdim >> 5;
        const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        long q = idx >> 9;
        const int mb = (int)(q % mblocks); q /= mblocks;
        const int ks = (int)(q % ksteps);
        const int t = (int)(q / ksteps);
        const int m = mb * 32 + (lane & 31), k = ks * 16 + 8 * (lane >> 5) + j;
        // forward: W(m = out, k = in) = w[(t*cin + k)*cout + m];  data gradient: W(m = in, k = out) = w[(t*cin + m)*cout + k]
        const float v = mode == 0 ? w[((long)t * cin + k) * cout + m] : w[((long)t * cin + m) * cout + k];
        (mode ? out_dgrad : out_fwd)[idx] = (__bf16)v;
    }
}

__global__ void bf16_to_f32_kernel(const __bf16* __restrict__ x, float* __restrict__ y, size_t count) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) y[i] = (float)x[i];
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ y, size_t count) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) y[i] = (__bf16)x[i];
}

template <int SP, bool HALF>
int launch_gconv_lds_sp(const GlParams& q, int grid, hipStream_t st) {
    constexpr int HR = GL_TR + SP, HC = GL_TC + SP, CH16 = HR * HC * 8, NDMA = (CH16 + 255) / 256, LDS = (HALF && 3 * NDMA * 4096 <= 160 * 1024 ? 3 : 2) * NDMA * 4096;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)gconv_lds_bf16_kernel<SP, HALF>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    hipLaunchKernelGGL((gconv_lds_bf16_kernel<SP, HALF>), dim3(grid), dim3(HALF ? 512 : 256), LDS, st, q);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

// the LDS-tiled kernel where it applies (input stride 1 or 2, 64-channel input chunks, >= 64 output channels, the taps of a parity plane
// inside a 5x5 box, images below 4 GiB, enough tiles to fill the chip); VCG_GCONV_LDS=0 forces the streaming kernel (A/B aid)
bool plan_gconv_lds(const GcParams& p, GlParams& q, int& spe_out, bool& half_out, int force_spe = 0, bool check_pairs = true) {
    static const bool off = getenv("VCG_GCONV_LDS") && atoi(getenv("VCG_GCONV_LDS")) == 0;
    if (off || p.isy != p.isx || p.isy < 1 || p.isy > 2 || p.kch % 64 || p.mblocks < 2 || p.ntaps < 1) return false;
    if ((long)p.ih * p.iw * p.kch * 2 > 0xFFFFFFE0l || (long)p.oh * p.ow * p.mch * 2 > 0xFFFFFFE0l) return false;
    q = GlParams{};
    // taps by the parity plane of the input they read: dy = isy * a + py
    const int S = p.isy;
    int sp = 0;
    for (int py = 0; py < S; ++py)
        for (int px = 0; px < S; ++px) {
            int a0 = 1 << 20, b0 = 1 << 20, a1 = -(1 << 20), b1 = -(1 << 20), cnt = 0;
            for (int i = 0; i < p.ntaps; ++i) {
                const int dy = p.taps[i].dy, dx = p.taps[i].dx;
                if (dy < -1000) return false;                                  // (the zero-tap sentinel of an empty phase stays on the streaming kernel)
                if (((dy % S) + S) % S != py || ((dx % S) + S) % S != px) continue;
                const int a = (dy - py) / S, b = (dx - px) / S;               // exact: dy - py is a multiple of S
                a0 = a < a0 ? a : a0; a1 = a > a1 ? a : a1; b0 = b < b0 ? b : b0; b1 = b > b1 ? b : b1;
                ++cnt;
            }
            if (!cnt) continue;
            GlParams::Plane& pl = q.pl[q.nplanes++];
            pl.py = py; pl.px = px; pl.dy0 = a0; pl.dx0 = b0;
            const int span = a1 - a0 > b1 - b0 ? a1 - a0 : b1 - b0;
            sp = span > sp ? span : sp;
        }
    if (sp > 4 || q.nplanes < 1) return false;
    if (force_spe && sp > force_spe) return false;
    const int spe = force_spe ? force_spe : (sp < 1 ? 1 : sp);
    for (int k = 0; k < q.nplanes; ++k) {
        GlParams::Plane& pl = q.pl[k];
        for (int i = 0; i < 25; ++i) pl.wt[i] = -1;
        for (int i = 0; i < p.ntaps; ++i) {
            const int dy = p.taps[i].dy, dx = p.taps[i].dx;
            if (((dy % S) + S) % S != pl.py || ((dx % S) + S) % S != pl.px) continue;
            pl.wt[((dy - pl.py) / S - pl.dy0) * (spe + 1) + ((dx - pl.px) / S - pl.dx0)] = p.taps[i].wt;
        }
    }
    q.x = p.x; q.wf = p.wf; q.y = p.y; q.bias = p.bias; q.mask_src = p.mask_src; q.wbytes = p.wbytes;
    q.n = p.n; q.ih = p.ih; q.iw = p.iw; q.kch = p.kch; q.oh = p.oh; q.ow = p.ow; q.mch = p.mch; q.loh = p.loh; q.low = p.low;
    q.osy = p.osy; q.osx = p.osx; q.ooy = p.ooy; q.oox = p.oox; q.mblocks = p.mblocks; q.act = p.act; q.alpha = p.alpha; q.mask_slope = p.mask_slope;
    q.isy = p.isy; q.isx = p.isx; q.stats = p.stats;
    q.tiles_x = ceil_div(p.low, GL_TC); q.tiles_y = ceil_div(p.loh, GL_TR);
    half_out = p.mblocks == 2 && p.stats == nullptr && spe <= 3;     // 64 output channels: waves = 2 channel blocks x 2 row halves (+ loaders, three stages)
    q.mgroups = half_out ? 1 : (p.mblocks + 3) / 4;
    const long pairs = (long)p.n * q.tiles_x * q.tiles_y * q.mgroups;
    if ((check_pairs && pairs < 64) || pairs > 0x7fffffffL) return false;        // too little work for 256 one-workgroup CUs: the streaming kernel's small tiles fill the chip better
    q.pairs = (int)pairs;
    q.nphases = 1;
    q.ph[0] = GlParams::Phase{p.ooy, p.oox, p.loh, p.low};
    spe_out = spe;
    return true;
}

int launch_gconv_lds(const GlParams& q, int spe, bool half, hipStream_t st) {
    const int grid = q.pairs < 256 ? q.pairs : 256;
    switch (spe) {
        case 1: return half ? launch_gconv_lds_sp<1, true>(q, grid, st) : launch_gconv_lds_sp<1, false>(q, grid, st);
        case 2: return half ? launch_gconv_lds_sp<2, true>(q, grid, st) : launch_gconv_lds_sp<2, false>(q, grid, st);
        case 3: return half ? launch_gconv_lds_sp<3, true>(q, grid, st) : launch_gconv_lds_sp<3, false>(q, grid, st);
        default: return half ? launch_gconv_lds_sp<4, true>(q, grid, st) : launch_gconv_lds_sp<4, false>(q, grid, st);
    }
}

// the np (= 4: stride 2) output phases of a data gradient as ONE launch of the LDS-tiled kernel: each phase on its own was a launch of a
// quarter of the tiles (PatchGAN block 3 at C3's shard: 128 tiles for 256 CUs, four times in a row), and the four tap sets of a dy tile
// were fetched by four launches.  false: some phase does not fit the kernel (the caller launches the phases one by one)
bool try_gconv_lds_phases(const GcParams* ps, int np, hipStream_t st, int* rc) {
    if (np < 2 || np > 4) return false;
    GlParams q[4];
    int spe = 1;
    bool half = false;
    for (int i = 0; i < np; ++i) {
        int si = 1;
        bool hi = false;
        if (ps[i].loh <= 0 || ps[i].low <= 0 || !plan_gconv_lds(ps[i], q[i], si, hi, 0, false) || q[i].nplanes != 1) return false;
        spe = si > spe ? si : spe;
    }
    for (int i = 0; i < np; ++i) {
        int si = 1;
        bool hi = false;
        if (!plan_gconv_lds(ps[i], q[i], si, hi, spe, false) || si != spe) return false;
        if (i == 0) half = hi;
        else if (hi != half) return false;
    }
    GlParams m = q[0];
    m.nphases = np;
    for (int i = 0; i < np; ++i) {
        m.pl[i] = q[i].pl[0];
        m.ph[i] = q[i].ph[0];
        m.tiles_x = q[i].tiles_x > m.tiles_x ? q[i].tiles_x : m.tiles_x;
        m.tiles_y = q[i].tiles_y > m.tiles_y ? q[i].tiles_y : m.tiles_y;
    }
    const long pairs = (long)m.n * m.tiles_x * m.tiles_y * m.mgroups * np;
    if (pairs < 64 || pairs > 0x7fffffffL) return false;
    m.pairs = (int)pairs;
    *rc = launch_gconv_lds(m, spe, half, st);
    return true;
}

bool try_gconv_lds(const GcParams& p, hipStream_t st, int* rc) {
    GlParams q;
    int spe = 1;
    bool half = false;
    if (!plan_gconv_lds(p, q, spe, half)) return false;
    *rc = launch_gconv_lds(q, spe, half, st);
    return true;
}

int launch_gconv(GcParams& p, hipStream_t st) {
    const long total = (long)p.n * p.loh * p.low;
    if (total <= 0) return VCG_OK;
    int lrc = VCG_OK;
    if (try_gconv_lds(p, st, &lrc)) return lrc;
    if (p.stats) return VCG_E_UNSUPPORTED;                       // only the LDS-tiled kernel's epilogue writes statistics
    const long tiles = (total + 31) / 32;
    // 4 tiles per wave (128 accumulator registers, half the operand loads per MFMA) once there is enough work to fill the chip
    const int mgroups = (p.mblocks + 1) / 2;
    static const int force_nt = getenv("VCG_GCONV_NT") ? atoi(getenv("VCG_GCONV_NT")) : 0;       // tuning aid (scripts/kbench_gconv.py)
    const bool big = force_nt ? force_nt == 4 : tiles * mgroups >= 4096;
    const int nt = big ? 4 : 2;
    const long wgs = (tiles + 4 * nt - 1) / (4 * nt);
    if (wgs > 0x7fffffffL) return VCG_E_SHAPE;
    const dim3 grid((unsigned)wgs, (unsigned)mgroups);
    if (big) hipLaunchKernelGGL(gconv_bf16_kernel<4>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(gconv_bf16_kernel<2>, grid, dim3(256), 0, st, p);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int check_gdesc(const vcg_conv_desc* d) {
    if (d == nullptr) return VCG_E_NULL;
    if (d->n <= 0 || d->cin <= 0 || d->h <= 0 || d->w <= 0 || d->cout <= 0 || d->oh <= 0 || d->ow <= 0) return VCG_E_SHAPE;
    if (d->kh <= 0 || d->kw <= 0 || d->kh * d->kw > GC_MAXTAPS || d->stride < 1 || d->stride > 3) return VCG_E_UNSUPPORTED;
    if (d->pad_top < 0 || d->pad_left < 0 || d->pad_top >= d->kh || d->pad_left >= d->kw) return VCG_E_SHAPE;
    if (d->cin % 16 || d->cout % 16) return VCG_E_UNSUPPORTED;
    if ((size_t)d->n * d->h * d->w * d->cin * 2 > 0xFFFFFFE0u || (size_t)d->n * d->oh * d->ow * d->cout * 2 > 0xFFFFFFE0u) return VCG_E_SHAPE;
    return VCG_OK;
}

}  // namespace

extern "C" {

size_t vcg_conv_frag_bf16_bytes(int taps, int mdim, int kdim) { return (size_t)taps * mdim * kdim * 2; }

int vcg_pack_conv_frag_bf16(const float* w, int taps, int mdim, int kdim, int mode, void* out, hipStream_t stream) {
    VCG_CHECK_PTR(w); VCG_CHECK_PTR(out);
    if (taps <= 0 || mdim <= 0 || kdim <= 0 || mdim % 32 || kdim % 16 || (mode != 0 && mode != 1)) return VCG_E_UNSUPPORTED;
    const long total = (long)taps * mdim * kdim;
    hipLaunchKernelGGL(pack_frag_kernel, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256), 0, stream, w,
                       (__bf16*)out, taps, mdim, kdim, mode);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_pack_conv_frag_bf16_pair(const float* w, int taps, int cin, int cout, void* out_fwd, void* out_dgrad, hipStream_t stream) {
    VCG_CHECK_PTR(w); VCG_CHECK_PTR(out_fwd); VCG_CHECK_PTR(out_dgrad);
    if (taps <= 0 || cin <= 0 || cout <= 0 || cin % 32 || cout % 32) return VCG_E_UNSUPPORTED;
    const long total = 2l * taps * cin * cout;
    hipLaunchKernelGGL(pack_frag_pair_kernel, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256), 0, stream, w,
                       (__bf16*)out_fwd, (__bf16*)out_dgrad, taps, cin, cout);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_bf16_to_f32(const void* x, float* y, size_t count, hipStream_t stream) {
    VCG_CHECK_PTR(x); VCG_CHECK_PTR(y);
    if (count == 0) return VCG_OK;
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, (const __bf16*)x, y, count);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_f32_to_bf16(const float* x, void* y, size_t count, hipStream_t stream) {
    VCG_CHECK_PTR(x); VCG_CHECK_PTR(y);
    if (count == 0) return VCG_OK;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, x, (__bf16*)y, count);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

static void fwd_params(const vcg_conv_desc* d, GcParams& p) {
    p.wbytes = vcg_conv_frag_bf16_bytes(d->kh * d->kw, d->cout, d->cin);
    p.n = d->n; p.ih = d->h; p.iw = d->w; p.kch = d->cin;
    p.oh = d->oh; p.ow = d->ow; p.mch = d->cout;
    p.loh = d->oh; p.low = d->ow; p.osy = p.osx = 1; p.ooy = p.oox = 0;
    p.isy = p.isx = d->stride;
    p.mblocks = d->cout / 32;
    p.ntaps = 0;
    for (int ky = 0; ky < d->kh; ++ky)
        for (int kx = 0; kx < d->kw; ++kx) p.taps[p.ntaps++] = GcTap{(short)(ky - d->pad_top), (short)(kx - d->pad_left), (short)(ky * d->kw + kx), 0};
}

// records per group (image, or the whole batch) that vcg_conv2d_nhwc_bf16_fwd_stats writes for this layer: one per 8 x 32-pixel output
// tile of the LDS-tiled kernel; a negative VCG_E_* when that kernel does not serve the shape (run vcg_norm_stats_bf16 on the output)
int vcg_conv2d_nhwc_bf16_stats_records(const vcg_conv_desc* d, int stats_mode) {
    int rc = check_gdesc(d);
    if (rc) return rc;
    if (d->cout % 32 || (stats_mode != VCG_STATS_BATCH && stats_mode != VCG_STATS_INSTANCE)) return VCG_E_UNSUPPORTED;
    GcParams p{};
    fwd_params(d, p);
    GlParams q;
    int spe = 1;
    bool half = false;                                          // (the tile grid -- the record count -- is the same in both forms)
    if (!plan_gconv_lds(p, q, spe, half)) return VCG_E_UNSUPPORTED;
    const long per_img = (long)q.tiles_x * q.tiles_y, all = per_img * d->n;
    if (all > 0x3fffffffL) return VCG_E_UNSUPPORTED;
    return (int)(stats_mode == VCG_STATS_INSTANCE ? per_img : all);
}

// vcg_conv2d_nhwc_bf16_fwd (no activation) that also leaves, per output tile, the per-channel sum and sum of squares of the values it
// stored: stats fp32 [n][tiles per image][2][cout] -- read by vcg_norm_finalize_partials as [1][n * tiles][2][cout] (batch statistics) or
// [n][tiles][2][cout] (instance norm).  The statistics pass of the normalisation behind the layer (model.py:840, the PatchGAN blocks) is gone.
int vcg_conv2d_nhwc_bf16_fwd_stats(const vcg_conv_desc* d, const void* x, const void* wfrag, const float* bias, void* y, float* stats,
                                   hipStream_t stream) {
    int rc = check_gdesc(d);
    if (rc) return rc;
    VCG_CHECK_PTR(x); VCG_CHECK_PTR(wfrag); VCG_CHECK_PTR(y); VCG_CHECK_PTR(stats);
    if (d->cout % 32) return VCG_E_UNSUPPORTED;
    GcParams p{};
    p.x = x; p.wf = wfrag; p.y = y; p.bias = bias; p.stats = stats;
    fwd_params(d, p);
    p.act = VCG_ACT_NONE;
    return launch_gconv(p, stream);
}

// y[n][oh][ow][cout] = act(conv(x) + bias): d describes the Keras layer (pads = TF-SAME "before" pads or explicit padding)
int vcg_conv2d_nhwc_bf16_fwd(const vcg_conv_desc* d, const void* x, const void* wfrag, const float* bias, int act, float act_alpha,
                             void* y, hipStream_t stream) {
    int rc = check_gdesc(d);
    if (rc) return rc;
    VCG_CHECK_PTR(x); VCG_CHECK_PTR(wfrag); VCG_CHECK_PTR(y);
    if (d->cout % 32) return VCG_E_UNSUPPORTED;
    if (act != VCG_ACT_NONE && act != VCG_ACT_LRELU) return VCG_E_UNSUPPORTED;
    GcParams p{};
    p.x = x; p.wf = wfrag; p.y = y; p.bias = bias;
    p.wbytes = vcg_conv_frag_bf16_bytes(d->kh * d->kw, d->cout, d->cin);
    p.n = d->n; p.ih = d->h; p.iw = d->w; p.kch = d->cin;
    p.oh = d->oh; p.ow = d->ow; p.mch = d->cout;
    p.loh = d->oh; p.low = d->ow; p.osy = p.osx = 1; p.ooy = p.oox = 0;
    p.isy = p.isx = d->stride;
    p.mblocks = d->cout / 32;
    p.act = act; p.alpha = act_alpha;
    for (int ky = 0; ky < d->kh; ++ky)
        for (int kx = 0; kx < d->kw; ++kx) p.taps[p.ntaps++] = GcTap{(short)(ky - d->pad_top), (short)(kx - d->pad_left), (short)(ky * d->kw + kx), 0};
    return launch_gconv(p, stream);
}

// dx[n][h][w][cin] = data gradient of the layer d describes, from dy[n][oh][ow][cout]; wfrag_t: the kernel packed with mode 1.
// mask_src (optional, bf16 NHWC of dx's shape): dx *= (mask_src > 0 ? 1 : mask_slope) -- the LeakyReLU in front of the layer.
int vcg_conv2d_nhwc_bf16_dgrad(const vcg_conv_desc* d, const void* dy, const void* wfrag_t, const void* mask_src, float mask_slope,
                               void* dx, hipStream_t stream) {
    int rc = check_gdesc(d);
    if (rc) return rc;
    VCG_CHECK_PTR(dy); VCG_CHECK_PTR(wfrag_t); VCG_CHECK_PTR(dx);
    if (d->cin % 32) return VCG_E_UNSUPPORTED;
    GcParams p{};
    p.x = dy; p.wf = wfrag_t; p.y = dx; p.mask_src = mask_src; p.mask_slope = mask_slope;
    p.wbytes = vcg_conv_frag_bf16_bytes(d->kh * d->kw, d->cin, d->cout);
    p.n = d->n; p.ih = d->oh; p.iw = d->ow; p.kch = d->cout;
    p.oh = d->h; p.ow = d->w; p.mch = d->cin;
    p.isy = p.isx = 1;
    p.mblocks = d->cin / 32;
    p.act = VCG_ACT_NONE;
    const int S = d->stride;
    p.osy = p.osx = S;
    // dx[iy][ix] = sum over taps with (iy + pad - ky) % S == 0 of W[ky][kx]^T dy[(iy + pad - ky)/S][(ix + pad - kx)/S]:
    // one problem per output phase (iy % S, ix % S) with that phase's taps -- the four of a stride-2 layer as ONE launch where the LDS-tiled
    // kernel serves them all, else one launch per phase
    GcParams phases[9];
    int np = 0;
    bool all = true;
    for (int py = 0; py < S; ++py)
        for (int px = 0; px < S; ++px) {
            p.ooy = py; p.oox = px;
            p.loh = (d->h - py + S - 1) / S;
            p.low = (d->w - px + S - 1) / S;
            p.ntaps = 0;
            for (int ky = 0; ky < d->kh; ++ky) {
                if ((py + d->pad_top - ky) % S) continue;
                for (int kx = 0; kx < d->kw; ++kx) {
                    if ((px + d->pad_left - kx) % S) continue;
                    // floor division of a multiple of S
                    p.taps[p.ntaps++] = GcTap{(short)((py + d->pad_top - ky) / S), (short)((px + d->pad_left - kx) / S), (short)(ky * d->kw + kx), 0};
                }
            }
            if (p.loh <= 0 || p.low <= 0) { all = false; continue; }
            if (p.ntaps == 0) {   // a phase no tap reaches (k < S): its pixels are zero -- one zero tap keeps the kernel's store path
                p.taps[0] = GcTap{(short)-30000, (short)-30000, 0, 0};
                p.ntaps = 1;
                all = false;
            }
            phases[np++] = p;
        }
    static const bool merge_off = getenv("VCG_GCONV_MERGE_PHASES") && atoi(getenv("VCG_GCONV_MERGE_PHASES")) == 0;      // A/B aid
    if (all && np == 4 && !merge_off) {
        int lrc = VCG_OK;
        if (try_gconv_lds_phases(phases, np, stream, &lrc)) return lrc;
    }
    for (int i = 0; i < np; ++i) {
        rc = launch_gconv(phases[i], stream);
        if (rc) return rc;
    }
    return VCG_OK;
}

// y[n][2h][2w][cout] = act(conv_transpose(x) + bias) for Conv2DTranspose(k, strides 2, 'same') (model.py:72): the data gradient of the
// stride-2 convolution whose Keras kernel (kh, kw, in' = cout, out' = cin) the transposed kernel (kh, kw, out, in) already is -- one
// launch per output phase, through the LDS-tiled kernel where it applies.  d: the transposed layer (cin, h, w -> cout, oh, ow; pads =
// the 'same' crop); wfrag: vcg_pack_conv_frag_bf16(kernel, k*k, mdim = cout, kdim = cin, mode 1).
int vcg_conv_transpose2d_nhwc_bf16_fwd(const vcg_conv_desc* d, const void* x, const void* wfrag, const float* bias, int act, float act_alpha,
                                       void* y, hipStream_t stream) {
    if (d == nullptr) return VCG_E_NULL;
    VCG_CHECK_PTR(x); VCG_CHECK_PTR(wfrag); VCG_CHECK_PTR(y);
    if (d->n <= 0 || d->cin <= 0 || d->cout <= 0 || d->h <= 0 || d->w <= 0 || d->oh <= 0 || d->ow <= 0) return VCG_E_SHAPE;
    if (d->stride != 2 || d->kh != d->kw || d->kh * d->kw > GC_MAXTAPS || d->cin % 16 || d->cout % 32) return VCG_E_UNSUPPORTED;
    if (d->oh > 2 * d->h || d->ow > 2 * d->w || d->pad_top < 0 || d->pad_left < 0 || d->pad_top >= d->kh || d->pad_left >= d->kw) return VCG_E_SHAPE;
    if (act != VCG_ACT_NONE && act != VCG_ACT_LRELU) return VCG_E_UNSUPPORTED;
    if ((size_t)d->n * d->h * d->w * d->cin * 2 > 0xFFFFFFE0u || (size_t)d->n * d->oh * d->ow * d->cout * 2 > 0xFFFFFFE0u) return VCG_E_SHAPE;
    GcParams p{};
    p.x = x; p.wf = wfrag; p.y = y; p.bias = bias;
    p.wbytes = vcg_conv_frag_bf16_bytes(d->kh * d->kw, d->cout, d->cin);
    p.n = d->n; p.ih = d->h; p.iw = d->w; p.kch = d->cin;
    p.oh = d->oh; p.ow = d->ow; p.mch = d->cout;
    p.isy = p.isx = 1;
    p.mblocks = d->cout / 32;
    p.act = act; p.alpha = act_alpha;
    const int S = 2;
    p.osy = p.osx = S;
    int rc = VCG_OK;
    for (int py = 0; py < S; ++py)
        for (int px = 0; px < S; ++px) {
            p.ooy = py; p.oox = px;
            p.loh = (d->oh - py + S - 1) / S;
            p.low = (d->ow - px + S - 1) / S;
            p.ntaps = 0;
            for (int ky = 0; ky < d->kh; ++ky) {
                if ((py + d->pad_top - ky) % S) continue;
                for (int kx = 0; kx < d->kw; ++kx) {
                    if ((px + d->pad_left - kx) % S) continue;
                    p.taps[p.ntaps++] = GcTap{(short)((py + d->pad_top - ky) / S), (short)((px + d->pad_left - kx) / S), (short)(ky * d->kw + kx), 0};
                }
            }
            if (p.loh <= 0 || p.low <= 0) continue;
            if (p.ntaps == 0) {
                p.taps[0] = GcTap{(short)-30000, (short)-30000, 0, 0};
                p.ntaps = 1;
            }
            rc = launch_gconv(p, stream);
            if (rc) return rc;
        }
    return VCG_OK;
}

}  // extern "C"
