// bf16 weight gradient of the layers that read the 3-channel frames -- the generator's initial/conv (Conv2D(64, 9, 'same'),
// upscaling/upscaler/model.py:275) and the critics' first layers (Conv2D(64m, 3, 'same'), model.py:839, 904; the PatchGAN's 4x4 stride-2 layer)
// -- in the bf16 configs (BASELINE.json C3/C4):
//     dW[ky][kx][ci][co] = sum over output pixels p of  dz[p][co] * x[S*p + (ky - pt, kx - pl)][ci]          (fp32 result, Keras HWIO layout)
// These were the last fp32 detours of the bf16 training step (the fp32 wgrad_kernel on an fp32 NCHW copy of dz: 0.55 ms of C3's 14 ms).
// As a GEMM: M = co, N = (tap, ci) = K*K x (3 + 1 zero), K = output pixels -- the mirror image of bf16_wgrad9.hip (there the 3-channel
// tensor is the GRADIENT and the many-channel one the input): both operands are needed pixel-major, i.e. transposed reads
// (ds_read_b64_tr_b16: in a 16-lane group, lane i receives halfword i&3 of the 8 bytes addressed by lanes (i>>2)+4j):
//   * dz tile [4 x 32 output pixels][this workgroup's 64 channels] in LDS, 128-byte pixels whose two 64-byte halves are swapped when bit 1
//     of the pixel index is set (the four pixels of a transposed read fall into four different bank quarters), as in bf16_wgrad.hip.  dz is
//     not shifted, so it is read from HBM exactly once;
//   * the frames as bf16 [pixel][3 channels + 1 zero] = 8 bytes per pixel (packed once per call from the fp32 NCHW frames: the forward
//     kernel multiplies the same bf16 roundings), a halo of ((4-1)*S + K) x ((32-1)*S + K) pixels in LDS.  In a transposed read the 16 lanes
//     of a group supply 16 addresses: lane L addresses pixel (L >> 2) of the k-quad -- at pixel stride S -- shifted for tap slot (L & 3), so
//     one read hands 4 taps x 4 channels their 4 consecutive output pixels and an MFMA column tile is 8 taps x (3 + 1) channels;
//   * four waves = 2 row tiles (32 output channels) x 2 halves of the column tiles; accumulators stay in registers for the whole launch;
//     tiles stream by `buffer_load ... lds` through a ring of three stages, the fragments of the next k-step are read behind this one's MFMAs;
//   * per-wave partial blocks go to the workspace as raw register dumps and are summed in a fixed order (deterministic); the bias gradient
//     (sum of dz) rides along from the dz fragments.
#include "vcg_common.hpp"
#include <utility>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

template <class F, int... Is>
__device__ __forceinline__ void w3_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void w3_static_for(F&& f) {
    w3_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ void w3_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ unsigned long long w3_tr_read(unsigned addr) {
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

template <int K, int S>
struct W3Cfg {
    static constexpr int T = K * K;
    static constexpr int NT = (T + 7) / 8;                      // column tiles: 8 taps x 4 channels each
    static constexpr int NTW = (NT + 1) / 2;                    // ... per wave (two halves)
    static constexpr int TR = 4, TC = 32;                       // output-pixel tile
    static constexpr int AB = TR * TC * 128;                    // dz tile (16384)
    static constexpr int HR = (TR - 1) * S + K;                 // frame halo rows
    static constexpr int HC = (TC - 1) * S + K;                 // ... columns
    static constexpr int HCA = (HC + 2) & ~1;                   // + one column for the pair alignment of an odd origin, even
    static constexpr int XB = HR * HCA * 8;
    static constexpr int CHUNKS = (AB + XB) / 16;
    static constexpr int NDMA = (CHUNKS + 255) / 256;
    static constexpr int BUF = NDMA * 4096;
    static constexpr int NS = 3;
    static constexpr int KS = TR * TC / 16;                     // k-steps of 16 output pixels per tile
    static constexpr int WAVE_FLOATS = NTW * 16 * 64;
    static constexpr int WAIT_ONE_BEHIND = NDMA | 0x0F70;       // s_waitcnt vmcnt(NDMA)
    static_assert(NS * BUF <= 64 * 1024 + 16 * 1024 && NDMA < 16 && NDMA <= NTW * 8, "wgrad3: stages / vmcnt / issue slots");
};

constexpr int W3_GRID = 256;                                     // persistent workgroups per 64-channel block of dz

struct W3Params {
    const unsigned char* dz;     // bf16 NHWC [n][oh][ow][cout]
    const unsigned char* xq;     // bf16 [n][h][w][4] (channel 3 = 0)
    float* ws;                   // [co block][grid][4 waves][WAVE_FLOATS]
    float* wsb;                  // [co block][grid][2 row tiles][2 lane halves][32]
    int n, h, w_, oh, ow, cout, pt, pl;
    int tiles_x, tiles_y, total, grid, co_blocks;
};

// (namespace scope: hipcc emits no host stub for a kernel template whose lambdas return a struct local to the kernel)
struct W3Src { int y0, x0; vcg_rsrc rz, rx; };

template <int K, int S>
__global__ __launch_bounds__(256, 2) void wgrad3_bf16_kernel(W3Params p) {
    using C = W3Cfg<K, S>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mrow = wv & 1, chalf = wv >> 1;                      // row tile (32 of the block's 64 channels), half of the column tiles
    // the channel block is the FASTEST grid index: the workgroups that read the 128-byte pieces of the same dz pixels run side by side
    const int cob = blockIdx.x % p.co_blocks, bx = blockIdx.x / p.co_blocks, nbx = gridDim.x / p.co_blocks;
    const unsigned lds0 = (unsigned)(size_t)smem;
    const long zimg = (long)p.oh * p.ow * p.cout * 2, ximg = (long)p.h * p.w_ * 8;
    const int shift = p.pl & 1;                                    // a tile's halo origin x0*S - pl is odd when pl is: the staged halo starts one column earlier

    // lane constants of the transposed reads.  A (dz): pixel 8*h8 + q (+4t), channels 32*mrow + 16*((l>>4)&1) + 4*(l&3) ..
    const int h8 = (lane >> 5) * 8, q = (lane & 15) >> 2;
    const int chb = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;          // byte offset inside a 64-byte half
    const unsigned abase = (unsigned)((h8 + q) * 128 + ((64 * mrow + chb) ^ (64 * ((q >> 1) & 1))));
    // B (frames): lane L of a 16-group addresses output pixel (L >> 2) of the k-quad, i.e. frame pixel S*(..) + tap offset, tap 8*nt + 4*((lane>>4)&1) + (L & 3)
    unsigned bbase[C::NTW];
#pragma unroll
    for (int j = 0; j < C::NTW; ++j) {
        int tap = 8 * (chalf * C::NTW + j) + 4 * ((lane >> 4) & 1) + (lane & 3);
        tap = tap < C::T ? tap : C::T - 1;                          // dummy columns: any address inside the halo (their sums are discarded)
        const int ky = tap / K, kx = tap - K * ky;
        bbase[j] = (unsigned)(C::AB + (ky * C::HCA + kx + shift + S * (h8 + q)) * 8);
    }

    f32x16 acc[C::NTW];
#pragma unroll
    for (int j = 0; j < C::NTW; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    float dbs = 0.f;                                               // bias gradient: this lane's channel, its 8 pixels per k-step (waves with chalf == 0)

    auto decode = [&](int tile) {
        const int txi = tile % p.tiles_x, t2 = tile / p.tiles_x, tyi = t2 % p.tiles_y, img = t2 / p.tiles_y;
        return W3Src{tyi * C::TR, txi * C::TC, make_rsrc(p.dz + img * zimg, (unsigned long)zimg), make_rsrc(p.xq + img * ximg, (unsigned long)ximg)};
    };
    // one 1-KiB piece (DMA round k) of a tile's stage
    auto dma_piece = [&](const W3Src& sc, int buf, int k) {
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));                                 // keep the slot arithmetic inside the tile loop (no hoisted registers)
        const int s = k * 256 + tid_o;
        unsigned off;
        bool ok;
        if (k < C::AB / 4096) {                                         // dz: pixel P of the tile, 16-byte chunk pos of its 128 bytes
            const int P = s >> 3, pos = s & 7, row = P >> 5, col = P & 31;
            const int cs = pos ^ (4 * ((P >> 1) & 1));                  // stored chunk pos holds source chunk cs
            const int gy = sc.y0 + row, gx = sc.x0 + col;
            ok = gy < p.oh && gx < p.ow;
            off = (unsigned)(gy * p.ow + gx) * (unsigned)(p.cout * 2) + (unsigned)(cob * 128 + cs * 16);
        } else {                                                        // frame halo: two pixels per 16 bytes (w is even, the staged origin is even)
            const int sd = s - C::AB / 16, row = sd / (C::HCA / 2), pc = sd - row * (C::HCA / 2);
            const int gy = sc.y0 * S - p.pt + row, gx = sc.x0 * S - p.pl - shift + 2 * pc;
            ok = sd < C::XB / 16 && (unsigned)gy < (unsigned)p.h && (unsigned)gx < (unsigned)p.w_;
            off = (unsigned)(gy * p.w_ + gx) * 8u;
        }
        asm volatile("" : "+v"(off));
        off = ok ? off : VCG_OOB;
        void __attribute__((address_space(3)))* dst = (void __attribute__((address_space(3)))*)(smem + buf * C::BUF + (k * 256 + wv * 64) * 16);
        if (k < C::AB / 4096) __builtin_amdgcn_raw_ptr_buffer_load_lds(sc.rz, dst, 16, off, 0, 0, 0);      // (no ?: on descriptors: hipcc then drops the
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(sc.rx, dst, 16, off, 0, 0, 0);                      //  kernel template's host stub without a word)
    };

    unsigned long long fa[2][2], fb[2][C::NTW][2];
    auto read_a = [&](unsigned lb, int set, int ks) {
        const int i = ks >> 1, cb = ks & 1;
#pragma unroll
        for (int t = 0; t < 2; ++t) fa[set][t] = w3_tr_read(lb + abase + (unsigned)((i * C::TC + cb * 16 + 4 * t) * 128));
    };
    auto read_b = [&](unsigned lb, int set, int ks, int j) {
        const int i = ks >> 1, cb = ks & 1;
#pragma unroll
        for (int t = 0; t < 2; ++t) fb[set][j][t] = w3_tr_read(lb + bbase[j] + (unsigned)((S * i * C::HCA + S * (cb * 16 + 4 * t)) * 8));
    };
    // s_waitcnt lgkmcnt(0) that also ties the fragment registers of a set to itself (nothing may use them before it)
    static_assert(C::NTW == 1 || C::NTW == 6, "wgrad3: the fragment sets are tied to the wait by name");
#define W3_WAIT_SET(S_)                                                                                                                       \
    do {                                                                                                                                      \
        if constexpr (C::NTW == 1) {                                                                                                          \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[S_][0]), "+v"(fa[S_][1]), "+v"(fb[S_][0][0]), "+v"(fb[S_][0][1]));                 \
        } else {                                                                                                                              \
            asm volatile("s_waitcnt lgkmcnt(0)"                                                                                               \
                         : "+v"(fa[S_][0]), "+v"(fa[S_][1]), "+v"(fb[S_][0][0]), "+v"(fb[S_][0][1]), "+v"(fb[S_][1][0]), "+v"(fb[S_][1][1]), \
                           "+v"(fb[S_][2][0]), "+v"(fb[S_][2][1]), "+v"(fb[S_][3][0]), "+v"(fb[S_][3][1]), "+v"(fb[S_][4][0]),               \
                           "+v"(fb[S_][4][1]), "+v"(fb[S_][C::NTW - 1][0]), "+v"(fb[S_][C::NTW - 1][1]));                                    \
        }                                                                                                                                     \
    } while (0)

    int tile = bx, buf = 0;
    if (tile < p.total) {
        const W3Src first = decode(tile);
#pragma unroll
        for (int k = 0; k < C::NDMA; ++k) dma_piece(first, 0, k);
    }
    {
        const bool h1 = tile + nbx < p.total;                        // (past the last tile: zero-record descriptors, the same number of pieces)
        W3Src second = decode(h1 ? tile + nbx : bx);
        if (!h1) { second.rz = make_rsrc(p.dz, 0); second.rx = make_rsrc(p.xq, 0); }
#pragma unroll
        for (int k = 0; k < C::NDMA; ++k) dma_piece(second, 1, k);
    }
    for (; tile < p.total; tile += nbx, buf = buf + 1 == C::NS ? 0 : buf + 1) {
        // a wave's loads retire in order: "at most NDMA outstanding" = this tile's stage is complete, the next one's may still be in flight
        __builtin_amdgcn_s_waitcnt(C::WAIT_ONE_BEHIND);
        w3_barrier();                                // ... and everyone else's part; the stage multiplied last is free again
        const int next = tile + 2 * nbx;
        const bool has_next = next < p.total;
        W3Src nsrc = decode(has_next ? next : bx);
        if (!has_next) { nsrc.rz = make_rsrc(p.dz, 0); nsrc.rx = make_rsrc(p.xq, 0); }
        const int nbuf = buf == 0 ? C::NS - 1 : buf - 1;
        const unsigned lb = lds0 + buf * C::BUF;
        read_a(lb, 0, 0);
#pragma unroll
        for (int j = 0; j < C::NTW; ++j) read_b(lb, 0, 0, j);
        W3_WAIT_SET(0);
        // k-steps of 16 output pixels: tile row ks >> 1, column half ks & 1
        w3_static_for<C::KS>([&](auto ic) {
            constexpr int ks = decltype(ic)::value, c = ks & 1, n = c ^ 1;
            const u64x2 av = {fa[c][0], fa[c][1]};
            const bf16x8 a = __builtin_bit_cast(bf16x8, av);
            if (chalf == 0) {                        // wave-uniform: these two waves see every pixel of their 32 channels once
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const bf16x4 v4 = __builtin_bit_cast(bf16x4, fa[c][t]);
                    dbs += ((float)v4[0] + (float)v4[1]) + ((float)v4[2] + (float)v4[3]);
                }
            }
#pragma unroll
            for (int j = 0; j < C::NTW; ++j) {
                if (ks + 1 < C::KS) {
                    if (j == 0) read_a(lb, n, ks + 1);
                    read_b(lb, n, ks + 1, j);
                }
                const u64x2 bv = {fb[c][j][0], fb[c][j][1]};
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, bv), acc[j], 0, 0, 0);
                if (ks * C::NTW + j < C::NDMA) dma_piece(nsrc, nbuf, ks * C::NTW + j);      // the tile after next, behind the first MFMAs
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ks + 1 < C::KS) {
                if (n == 0) W3_WAIT_SET(0); else W3_WAIT_SET(1);
            }
        });
    }
#undef W3_WAIT_SET
    __builtin_amdgcn_s_waitcnt(0x0F70);              // (the dead pieces of the last iterations may still be landing in LDS)
    // raw register dump of this wave's tiles (coalesced); decoded by the reduction
    float* out = p.ws + (((long)cob * p.grid + bx) * 4 + wv) * C::WAVE_FLOATS;
#pragma unroll
    for (int j = 0; j < C::NTW; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) out[(j * 16 + e) * 64 + lane] = acc[j][e];
    if (chalf == 0 && p.wsb) p.wsb[((((long)cob * p.grid + bx) * 2 + mrow) * 2 + (lane >> 5)) * 32 + (lane & 31)] = dbs;
}

// dW[tap][ci][co] (and db[co]) = sum over workgroups, in a fixed order, of the wave blocks.  block = 64 consecutive RAW dump elements x 16
// record lanes (coalesced reads, eight records in flight per thread, fixed-order combine through LDS); the decoded (tap, ci, co) position
// is only used for the single store.  The blocks past the weight dump sum the bias partials (two lane halves per record) the same way.
template <int K>
__global__ __launch_bounds__(1024) void wgrad3_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ wsb, int grid, int cout,
                                                             float* __restrict__ dw, float* __restrict__ db) {
    constexpr int T = K * K, NT = (T + 7) / 8, NTW = (NT + 1) / 2, WAVE_FLOATS = NTW * 16 * 64;
    __shared__ float red[16][64];
    const int g = threadIdx.x >> 6, cl = threadIdx.x & 63;
    const int co_blocks = cout >> 6, nraw = co_blocks * 4 * WAVE_FLOATS;
    const int r = blockIdx.x * 64 + cl;
    const bool is_w = r < nraw;                                      // (block-uniform: nraw is a multiple of 64)
    const int rb = r - nraw;                                         // bias: channel index
    const bool is_b = !is_w && db != nullptr && rb < cout;
    const float* src = ws;
    long stride = 0;
    int second = 0, cob = 0, wave = 0, off = 0;
    if (is_w) {
        cob = r / (4 * WAVE_FLOATS);
        const int rem = r - cob * 4 * WAVE_FLOATS;
        wave = rem / WAVE_FLOATS;
        off = rem - wave * WAVE_FLOATS;
        src = ws + (((long)cob * grid) * 4 + wave) * WAVE_FLOATS + off;
        stride = 4l * WAVE_FLOATS;
    } else if (is_b) {
        const int bc = rb >> 6, mrow = (rb >> 5) & 1, m = rb & 31;
        src = wsb + ((((long)bc * grid) * 2 + mrow) * 2) * 32 + m;
        stride = 128;
        second = 32;
    }
    float s = 0.f;
    if (is_w || is_b) {
        int b = g;
        for (; b + 16 * 7 < grid; b += 16 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(long)(b + 16 * u) * stride] + (second ? src[(long)(b + 16 * u) * stride + second] : 0.f);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < grid; b += 16) s += src[(long)b * stride] + (second ? src[(long)b * stride + second] : 0.f);
    }
    red[g][cl] = s;
    __syncthreads();
    if (g != 0 || !(is_w || is_b)) return;
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i][cl];
    if (is_b) {
        db[rb] = t;
        return;
    }
    // raw offset -> column tile j of the wave, register e, lane l;  D[row][col]: row = (e&3) + 8*(e>>2) + 4*(l>>5), col = l&31 = (tap slot, ci)
    const int l = off & 63, e = (off >> 6) & 15, j = off >> 10, mrow = wave & 1, chalf = wave >> 1;
    const int m = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5), col = l & 31, tap = 8 * (chalf * NTW + j) + (col >> 2), ci = col & 3;
    const int co = cob * 64 + mrow * 32 + m;
    if (tap < T && ci < 3) dw[(tap * 3 + ci) * cout + co] = t;
}

// fp32 NCHW frames [n][3][h][w] -> bf16 [n][h][w][4] (channel 3 = 0)
__global__ void pack_frames3_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ out, int n, long hw) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * hw) return;
    const long img = i / hw, px = i - img * hw;
    bf16x4 v;
    v[0] = (__bf16)x[(img * 3 + 0) * hw + px];
    v[1] = (__bf16)x[(img * 3 + 1) * hw + px];
    v[2] = (__bf16)x[(img * 3 + 2) * hw + px];
    v[3] = (__bf16)0.f;
    *(bf16x4*)(out + i * 4) = v;
}

template <int K, int S>
int launch_w3(const W3Params& p, float* dw, float* db, hipStream_t st) {
    using C = W3Cfg<K, S>;
    auto kern = wgrad3_bf16_kernel<K, S>;
    if (C::NS * C::BUF > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::NS * C::BUF);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3(p.grid * p.co_blocks), dim3(256), C::NS * C::BUF, st, p);
    VCG_LAUNCH_CHECK();
    const int total = p.co_blocks * 4 * C::WAVE_FLOATS + p.cout;     // raw dump elements + bias channels
    hipLaunchKernelGGL(wgrad3_reduce_kernel<K>, dim3(ceil_div(total, 64)), dim3(1024), 0, st, (const float*)p.ws, (const float*)p.wsb, p.grid, p.cout, dw, db);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

bool w3_supported(const vcg_conv_desc* d) {
    if (d->cin != 3 || d->cout % 64 || d->kh != d->kw || d->w % 2) return false;
    if ((long)d->h * d->w * 8 > 0xFFFFFFE0l || (long)d->oh * d->ow * d->cout * 2 > 0xFFFFFFE0l) return false;
    if (d->pad_top < 0 || d->pad_left < 0 || d->pad_top >= d->kh || d->pad_left >= d->kw) return false;
    return (d->kh == 9 && d->stride == 1) || (d->kh == 3 && d->stride == 1) || (d->kh == 4 && d->stride == 2);
}

size_t w3_wave_floats(int k) {
    const int T = k * k, NT = (T + 7) / 8, NTW = (NT + 1) / 2;
    return (size_t)NTW * 16 * 64;
}

}  // namespace

extern "C" {

size_t vcg_conv3ch_bf16_wgrad_workspace_bytes(const vcg_conv_desc* d) {
    if (d == nullptr || d->n <= 0 || d->h <= 0 || d->w <= 0 || !w3_supported(d)) return 0;
    const size_t cb = (size_t)(d->cout / 64);
    return align_up(cb * W3_GRID * 4 * w3_wave_floats(d->kh) * sizeof(float), 256) + align_up(cb * W3_GRID * 128 * sizeof(float), 256) +
           (size_t)d->n * d->h * d->w * 8 + 256;
}

// x: the fp32 NCHW frames [n][3][h][w]; dz: bf16 NHWC [n][oh][ow][cout] (the gradient in front of the layer's activation); dw in Keras'
// (kh,kw,3,cout) layout, dbias [cout] or NULL.  VCG_E_UNSUPPORTED for other shapes (odd widths, other kernels): the caller's fp32 path.
int vcg_conv3ch_bf16_wgrad(const vcg_conv_desc* d, const float* x, const void* dz, float* dw_hwio, float* dbias, void* ws, size_t ws_bytes,
                           vcg_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VCG_CHECK_PTR(d); VCG_CHECK_PTR(x); VCG_CHECK_PTR(dz); VCG_CHECK_PTR(dw_hwio); VCG_CHECK_PTR(ws);
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->oh <= 0 || d->ow <= 0) return VCG_E_SHAPE;
    if (!w3_supported(d)) return VCG_E_UNSUPPORTED;
    if (ws_bytes < vcg_conv3ch_bf16_wgrad_workspace_bytes(d)) return VCG_E_WORKSPACE;
    W3Params p{};
    const size_t cb = (size_t)(d->cout / 64);
    p.ws = (float*)ws;
    p.wsb = (float*)((char*)ws + align_up(cb * W3_GRID * 4 * w3_wave_floats(d->kh) * sizeof(float), 256));
    unsigned char* xq = (unsigned char*)p.wsb + align_up(cb * W3_GRID * 128 * sizeof(float), 256);
    xq += (256 - ((size_t)xq & 255)) & 255;
    p.dz = (const unsigned char*)dz; p.xq = xq;
    p.n = d->n; p.h = d->h; p.w_ = d->w; p.oh = d->oh; p.ow = d->ow; p.cout = d->cout; p.pt = d->pad_top; p.pl = d->pad_left;
    p.tiles_x = ceil_div(d->ow, 32); p.tiles_y = ceil_div(d->oh, 4);
    p.total = p.n * p.tiles_x * p.tiles_y;
    p.grid = p.total < W3_GRID ? p.total : W3_GRID;
    p.co_blocks = (int)cb;
    const long px = (long)d->n * d->h * d->w;
    pack_frames3_bf16_kernel<<<(unsigned)((px + 255) / 256), 256, 0, stream>>>(x, (__bf16*)xq, d->n, (long)d->h * d->w);
    VCG_LAUNCH_CHECK();
    if (d->kh == 9) return launch_w3<9, 1>(p, dw_hwio, dbias, stream);
    if (d->kh == 3) return launch_w3<3, 1>(p, dw_hwio, dbias, stream);
    return launch_w3<4, 2>(p, dw_hwio, dbias, stream);
}

}  // extern "C"
