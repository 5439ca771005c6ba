// bf16-storage kernels (configs C3-C5 of BASELINE.json): activations bf16 NHWC, weights bf16 packed
// [tap][out-channel][in-channel], fp32 accumulation on v_mfma_f32_32x32x16_bf16, fp32 epilogue.
//
// Why NHWC here while the fp32 path is NCHW: the bf16 MFMA takes 8 consecutive k per lane, and k of the
// implicit GEMM is the input channel -- with channels innermost one lane's operand fragment is ONE 16-byte LDS
// read, and a pixel's 64 channels are one 128-byte line in HBM.
//
// Reference ops served: Conv2D 3x3 'same' of residual_block / prefinal conv (upscaling/upscaler/model.py:19,22,283)
// with the inference-mode BatchNormalization folded into a per-channel scale/shift (model.py:20,23,284),
// PReLU (model.py:21) and the block's Add (model.py:25,285) fused into the epilogue.
#include "vcg_common.hpp"
#include <cstdlib>
#include <utility>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    // lane l (r = l&31, h = l>>5): A[row r][k = 8h+j], B[k = 8h+j][col r], j = 0..7; D as the f32 form
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// v_permlane32_swap: lanes 32-63 of a <-> lanes 0-31 of b
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void swap32(float& a, float& b) {
    const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(sw.x);
    b = __uint_as_float(sw.y);
}

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N-1>{})
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// ---------------------------------------------------------------------------------------------------------------
// layout / packing helpers
// ---------------------------------------------------------------------------------------------------------------
__global__ void pack_kernel_bf16(const float* __restrict__ w, __bf16* __restrict__ out, int taps, int a, int b,
                                 int transpose, int flip) {
    // out[tap'][i][j] (j contiguous) = transpose ? w[tap][j][i] : w[tap][i][j];  tap' = flip ? taps-1-tap : tap
    const long total = (long)taps * a * b;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int j = (int)(idx % b);
        const int i = (int)((idx / b) % a);
        const int t = (int)(idx / ((long)a * b));
        const int ts = flip ? taps - 1 - t : t;
        const float v = transpose ? w[((long)ts * b + j) * a + i] : w[((long)ts * a + i) * b + j];
        out[idx] = (__bf16)v;
    }
}

// Every 3x3 64 -> 64 kernel of a model in ONE launch (the trunk re-derives 2 x 19 bf16 copies after each optimizer step: 38 launches of
// 4.7 us otherwise): blockIdx.y = layer, out[layer][0] = forward pack [tap][co][ci], out[layer][1] = data-gradient pack [8 - tap][ci][co].
constexpr int PACK_BATCH_MAX = 48;
struct PackBatch {
    const float* w[PACK_BATCH_MAX];
};
__global__ void pack3x3_c64_batch_kernel(PackBatch pb, __bf16* __restrict__ out) {
    const float* w = pb.w[blockIdx.y];
    __bf16* o = out + (long)blockIdx.y * 2 * 9 * 64 * 64;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < 2 * 9 * 64 * 64; idx += gridDim.x * blockDim.x) {
        const int which = idx / (9 * 64 * 64), r = idx - which * 9 * 64 * 64;
        const int j = r & 63, i = (r >> 6) & 63, t = r >> 12;
        // Keras (3,3,in,out): w[tap][ci][co].  forward: out[t][co=i][ci=j] (transpose); data gradient: out[t][ci=i][co=j] of tap 8 - t (flip)
        const float v = which == 0 ? w[(t * 64 + j) * 64 + i] : w[((8 - t) * 64 + i) * 64 + j];
        o[idx] = (__bf16)v;
    }
}

__global__ void f32_nchw_to_bf16_nhwc_kernel(const float* __restrict__ x, __bf16* __restrict__ y, int n, int c, int hw) {
    // one block per (n, 64-pixel segment): coalesced reads along pixels, coalesced writes along channels
    __shared__ float tile[64][65];
    const int p0 = blockIdx.x * 64, img = blockIdx.y;
    for (int c0 = 0; c0 < c; c0 += 64) {
        for (int e = threadIdx.x; e < 64 * 64; e += blockDim.x) {
            const int cc = e >> 6, pp = e & 63;
            float v = 0.f;
            if (c0 + cc < c && p0 + pp < hw) v = x[((long)img * c + c0 + cc) * hw + p0 + pp];
            tile[cc][pp] = v;
        }
        __syncthreads();
        if ((c & 7) == 0) {
            // 16-byte stores: a thread packs 8 consecutive channels of one pixel (8 lanes = one pixel's 128 bytes)
            for (int e = threadIdx.x; e < 64 * 8; e += blockDim.x) {
                const int pp = e >> 3, ch = (e & 7) * 8;
                if (c0 + ch < c && p0 + pp < hw) {
                    bf16x8 v;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (__bf16)tile[ch + j][pp];
                    *(bf16x8*)(y + ((long)img * hw + p0 + pp) * c + c0 + ch) = v;
                }
            }
        } else {
            for (int e = threadIdx.x; e < 64 * 64; e += blockDim.x) {
                const int pp = e >> 6, cc = e & 63;
                if (c0 + cc < c && p0 + pp < hw) y[((long)img * hw + p0 + pp) * c + c0 + cc] = (__bf16)tile[cc][pp];
            }
        }
        __syncthreads();
    }
}

__global__ void bf16_nhwc_to_f32_nchw_kernel(const __bf16* __restrict__ x, float* __restrict__ y, int n, int c, int hw) {
    __shared__ float tile[64][65];
    const int p0 = blockIdx.x * 64, img = blockIdx.y;
    for (int c0 = 0; c0 < c; c0 += 64) {
        if ((c & 7) == 0) {
            // 16-byte loads: a thread takes 8 consecutive channels of one pixel
            for (int e = threadIdx.x; e < 64 * 8; e += blockDim.x) {
                const int pp = e >> 3, ch = (e & 7) * 8;
                bf16x8 v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.f;
                if (c0 + ch < c && p0 + pp < hw) v = *(const bf16x8*)(x + ((long)img * hw + p0 + pp) * c + c0 + ch);
#pragma unroll
                for (int j = 0; j < 8; ++j) tile[ch + j][pp] = (float)v[j];
            }
        } else {
            for (int e = threadIdx.x; e < 64 * 64; e += blockDim.x) {
                const int pp = e >> 6, cc = e & 63;
                float v = 0.f;
                if (c0 + cc < c && p0 + pp < hw) v = (float)x[((long)img * hw + p0 + pp) * c + c0 + cc];
                tile[cc][pp] = v;
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * 64; e += blockDim.x) {
            const int cc = e >> 6, pp = e & 63;
            if (c0 + cc < c && p0 + pp < hw) y[((long)img * c + c0 + cc) * hw + p0 + pp] = tile[cc][pp];
        }
        __syncthreads();
    }
}

// PReLU backward at the entry of the bf16 trunk (initial/prelu, model.py:276): the two gradients that meet at its output -- the trunk's and
// the long skip's (model.py:285), both bf16 NHWC -- are added, multiplied by the activation's derivative (sign of the stored pre-activation
// z) and written as the fp32 NCHW tensor the 3-channel convolution's weight-gradient kernel reads; the slope gradient sum(d * z, z < 0)
// leaves as one record of c floats per workgroup (summed in a fixed order by sum_records_kernel).  c % 8 == 0, c <= 64 * gridDim-free loop.
__global__ __launch_bounds__(256) void prelu_bwd_bf16_to_f32_nchw_kernel(const __bf16* __restrict__ d1, const __bf16* __restrict__ d2,
                                                                         const __bf16* __restrict__ z, const float* __restrict__ alpha,
                                                                         float* __restrict__ dz, float* __restrict__ rec, int c, int hw, int tiles) {
    __shared__ float tile[64][65];
    __shared__ float red[32][64];
    const int img = blockIdx.y, tid = threadIdx.x;
    const int ch = (tid & 7) * 8;                       // this thread's channel octet inside a 64-channel chunk (the same in both passes below)
    for (int c0 = 0; c0 < c; c0 += 64) {
        float da[8], al[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            da[j] = 0.f;
            al[j] = c0 + ch + j < c ? alpha[c0 + ch + j] : 0.f;
        }
        for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
            const int p0 = t * 64;
#pragma unroll
            for (int e = tid; e < 64 * 8; e += 256) {
                const int pp = e >> 3;
                float g[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) g[j] = 0.f;
                if (c0 + ch < c && p0 + pp < hw) {
                    const long o = ((long)img * hw + p0 + pp) * c + c0 + ch;
                    const bf16x8 a = *(const bf16x8*)(d1 + o), zz = *(const bf16x8*)(z + o);
                    bf16x8 b;
#pragma unroll
                    for (int j = 0; j < 8; ++j) b[j] = (__bf16)0.f;
                    if (d2) b = *(const bf16x8*)(d2 + o);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float d = (float)a[j] + (float)b[j], zf = (float)zz[j];
                        g[j] = zf >= 0.f ? d : d * al[j];
                        da[j] += zf >= 0.f ? 0.f : d * zf;
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) tile[ch + j][pp] = g[j];
            }
            __syncthreads();
            for (int e = tid; e < 64 * 64; e += 256) {
                const int cc = e >> 6, pp = e & 63;
                if (c0 + cc < c && p0 + pp < hw) dz[((long)img * c + c0 + cc) * hw + p0 + pp] = tile[cc][pp];
            }
            __syncthreads();
        }
        // the 32 threads that share a channel octet add up in a fixed order
#pragma unroll
        for (int j = 0; j < 8; ++j) red[tid >> 3][ch + j] = da[j];
        __syncthreads();
        if (tid < 64 && c0 + tid < c) {
            float s = 0.f;
            for (int k = 0; k < 32; ++k) s += red[k][tid];
            rec[((long)img * gridDim.x + blockIdx.x) * c + c0 + tid] = s;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3 stride-1 'same' convolution, 64 -> 64 channels: the generator trunk
// ---------------------------------------------------------------------------------------------------------------
// At bf16 this convolution is HBM-bound even at the full MFMA rate (per 16x32 pixels: 142 KB of traffic against
// 9.2 k MFMA cycles per SIMD), so the kernel is organised around keeping loads AND stores in flight under the MFMAs:
//   * persistent workgroups of 8 waves: 6 compute waves + 2 loader waves (wave-specialised: vmcnt is per wave
//     and retires in order, so a wave that both prefetches and stores ends up draining its stores before it may
//     touch the prefetched registers; with the roles split, neither side ever waits for the other's traffic);
//   * LDS: all 9x64x64 weights (72 KiB, loaded once) + one 14x34-pixel halo tile (59.5 KiB) + 768 B of epilogue
//     parameters.  Both images hold 128-byte rows (one pixel / one out-channel x 64 in-channels) whose eight 16-byte
//     chunks are XOR-swizzled with (index>>1)&7: the ds_read_b128 of the 16-lane groups {0-3,12-15,20-27},
//     {4-11,16-19,28-31} (MI355X_MICROARCH.md, LDS) then touches all 64 banks once, for every tap shift;
//   * compute wave w owns output rows 2w, 2w+1 of the 12x32 tile: a 64-channel x 64-pixel accumulator (4 MFMA tiles,
//     64 VGPRs), operand fragments double-buffered in registers (reads of k-step i+1 issued before the MFMAs of i);
//   * loader waves fetch the next tile's halo into registers during the MFMA phase and write it to LDS between the
//     two barriers that end a tile, while the compute waves run their epilogue.
constexpr int NCW = 6, NLW = 2;             // compute / loader waves
constexpr int TR = 2 * NCW, TC = 32, HR = TR + 2, HC = TC + 2;
constexpr int ROWB = HC * 128;              // bytes per halo row
constexpr int XB = HR * ROWB;               // 60928
constexpr int WB = 9 * 64 * 128;            // 73728
constexpr int PB = 3 * 64 * 4;              // per-channel epilogue parameters: scale, shift, negative-side slope
constexpr int NCHUNK = HR * HC * 8;         // 16-byte chunks per halo tile (3808)
constexpr int NT = (NCW + NLW) * 64;
constexpr int NPRE = (NCHUNK + NLW * 64 - 1) / (NLW * 64);   // per-loader-thread prefetch registers (30)

struct C3Params {
    const uint4* x;
    const uint4* w;
    __bf16* y;
    const float* scale;
    const float* shift;
    const float* alpha;
    const __bf16* res;
    int n, h, w_, tiles_x, tiles_y, total;
    int act;
    float act_alpha;
    float* stats;            // v2 with STATS: per-channel sum / sum of squares of the stored (bf16-rounded) output, [unit][row group][2][64]
    int stats_per_tile;      // 0: unit = workgroup (one record pair per launch: batch statistics); 1: unit = tile (instance norm)
};

// workgroup barrier ordering LDS only: global stores stay in flight across it
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <bool AFF, bool SLOPE, bool RES>
__global__ __launch_bounds__(NT, 1) void conv3x3_c64_bf16_kernel(C3Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wl = smem;
    unsigned char* xl = smem + WB;
    float* prm = (float*)(smem + WB + XB);
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int c = tid; c < 9 * 64 * 8; c += NT) {
        const int chunk = c & 7, co = (c >> 3) & 63, tap = c >> 9;
        *(uint4*)(wl + tap * 8192 + co * 128 + ((chunk ^ ((co >> 1) & 7)) << 4)) = p.w[c];
    }
    if (tid < 64) {
        prm[tid] = p.scale ? p.scale[tid] : 1.f;
        prm[64 + tid] = p.shift ? p.shift[tid] : 0.f;
        prm[128 + tid] = p.act == VCG_ACT_PRELU ? p.alpha[tid] : (p.act == VCG_ACT_LRELU ? p.act_alpha : 1.f);
    }

    if (wv >= NCW) {
        // ------------------------------------------------------------------------------------------ loader waves
        const int lt = tid - NCW * 64;
        uint4 pre[NPRE];
        unsigned okmask = 0;
        auto fetch = [&](int tile) {
            const int txi = tile % p.tiles_x, t2 = tile / p.tiles_x, tyi = t2 % p.tiles_y, img = t2 / p.tiles_y;
            const int y0 = tyi * TR - 1, x0 = txi * TC - 1;
            okmask = 0;
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                const int c = min(lt + NLW * 64 * i, NCHUNK - 1);     // the tail re-reads the last chunk: same data, same slot
                const int pix = c >> 3, row = pix / HC, col = pix - row * HC;
                const int gy = y0 + row, gx = x0 + col;
                const bool ok = (unsigned)gy < (unsigned)p.h && (unsigned)gx < (unsigned)p.w_;
                const int cy = min(max(gy, 0), p.h - 1), cx = min(max(gx, 0), p.w_ - 1);
                pre[i] = p.x[((long)(img * p.h + cy) * p.w_ + cx) * 8 + (c & 7)];
                okmask |= ok ? (1u << i) : 0u;
            }
        };
        auto stash = [&]() {
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                const int c = min(lt + NLW * 64 * i, NCHUNK - 1);
                const int pix = c >> 3, row = pix / HC, col = pix - row * HC;
                const uint4 v = (okmask >> i) & 1u ? pre[i] : make_uint4(0, 0, 0, 0);
                *(uint4*)(xl + pix * 128 + (((c & 7) ^ ((col >> 1) & 7)) << 4)) = v;
            }
        };
        int tile = blockIdx.x;
        fetch(tile);
        stash();
        lds_barrier();                                   // B0: weights, parameters and the first tile are in LDS
        for (; tile < p.total; tile += gridDim.x) {
            const int next = tile + gridDim.x;
            if (next < p.total) fetch(next);
            lds_barrier();                               // A: the compute waves have read the current tile
            if (next < p.total) stash();
            lds_barrier();                               // B: the next tile is in LDS
        }
        return;
    }

    // --------------------------------------------------------------------------------------------- compute waves
    int aoff[4], boff[3][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) aoff[s] = r * 128 + (((2 * s + hh) ^ ((r >> 1) & 7)) << 4);
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int pos = r + dx;
            boff[dx][s] = pos * 128 + (((2 * s + hh) ^ ((pos >> 1) & 7)) << 4);
        }
    const unsigned char* xb = xl + (wv * 2) * ROWB;
    lds_barrier();                                       // B0

    for (int tile = blockIdx.x; tile < p.total; tile += gridDim.x) {
        const int txi = tile % p.tiles_x, t2 = tile / p.tiles_x, tyi = t2 % p.tiles_y, img = t2 / p.tiles_y;
        const int gx = txi * TC + r, gy0 = tyi * TR + wv * 2;
        const bool okx = gx < p.w_;

        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

        // 36 k-steps (9 taps x 4 channel groups of 16)
        bf16x8 fa[2][2], fb[2][2];
        bf16x8 rr[2][2][2];
        // k-step at which the residual tile is requested: under the last 8 k-steps.  Requesting it at k-step 3 (under 32 k-steps, to cover a
        // whole HBM round trip) measured SLOWER: 73 against 61 us (data gradient + skip) and 57.6 against 55.6 us at batch 8, 194.5 against 190
        // at batch 32 -- VCG_RES_AT re-defines it for A/B builds
#ifndef VCG_RES_AT
#define VCG_RES_AT 27
#endif
        constexpr int RES_AT = VCG_RES_AT;
        auto frag = [&](int i, int buf) {
            const int tap = i >> 2, s = i & 3, dy = tap / 3, dx = tap - 3 * dy;
            const unsigned char* wa = wl + tap * 8192 + aoff[s];
            fa[buf][0] = *(const bf16x8*)(wa);
            fa[buf][1] = *(const bf16x8*)(wa + 4096);
            fb[buf][0] = *(const bf16x8*)(xb + dy * ROWB + boff[dx][s]);
            fb[buf][1] = *(const bf16x8*)(xb + (dy + 1) * ROWB + boff[dx][s]);
        };
        frag(0, 0);
#pragma unroll
        for (int i = 0; i < 36; ++i) {
            const int cur = i & 1;
            if (i + 1 < 36) frag(i + 1, cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);          // keep the next step's reads ahead of this step's MFMAs
            acc[0][0] = mfma_bf16(fa[cur][0], fb[cur][0], acc[0][0]);
            acc[0][1] = mfma_bf16(fa[cur][0], fb[cur][1], acc[0][1]);
            acc[1][0] = mfma_bf16(fa[cur][1], fb[cur][0], acc[1][0]);
            acc[1][1] = mfma_bf16(fa[cur][1], fb[cur][1], acc[1][1]);
            __builtin_amdgcn_sched_barrier(0);
            if (i == RES_AT && RES) {
                // the residual tile: requested under the remaining k-steps, 16 bytes (8 channels) per lane and group
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int pt = 0; pt < 2; ++pt) {
                            const int cy = min(gy0 + pt, p.h - 1), cx = min(gx, p.w_ - 1);
                            rr[mt][q][pt] = *(const bf16x8*)(p.res + ((long)(img * p.h + cy) * p.w_ + cx) * 64 + mt * 32 + 16 * q + 8 * hh);
                        }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        lds_barrier();                                   // A: this tile's LDS image may be overwritten

        // epilogue: y = act(acc * scale + shift) + residual -> bf16.  An MFMA tile leaves lane (pixel, h) with channels
        // 8g+4h+{0..3}; v_permlane32_swap between the register groups (2q, 2q+1) of the two half-waves turns that into
        // 8 consecutive channels 16q+8h+{0..7}: 16-byte residual loads and stores.  Phase 1 computes all final values
        // (consuming every outstanding load), phase 2 is nothing but the 8 stores, which then drain under the next
        // tile's MFMAs.
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int co = mt * 32 + 16 * q + 8 * hh;
                float sc[8], sh[8], al[8];
                if (AFF) {
                    *(f32x4*)&sc[0] = *(const f32x4*)(prm + co);
                    *(f32x4*)&sc[4] = *(const f32x4*)(prm + co + 4);
                    *(f32x4*)&sh[0] = *(const f32x4*)(prm + 64 + co);
                    *(f32x4*)&sh[4] = *(const f32x4*)(prm + 64 + co + 4);
                }
                if (SLOPE) {
                    *(f32x4*)&al[0] = *(const f32x4*)(prm + 128 + co);
                    *(f32x4*)&al[4] = *(const f32x4*)(prm + 128 + co + 4);
                }
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float lo = acc[mt][pt][8 * q + j], hi = acc[mt][pt][8 * q + 4 + j];
                        swap32(lo, hi);
                        v[j] = lo;
                        v[4 + j] = hi;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float u = v[j];
                        if (AFF) u = u * sc[j] + sh[j];
                        if (SLOPE) u = u >= 0.f ? u : u * al[j];
                        if (RES) u += (float)rr[mt][q][pt][j];
                        acc[mt][pt][8 * q + j] = u;
                    }
                }
            }
        // pin phase 1 here (otherwise its arithmetic is sunk into the conditional store blocks, and with it the waits)
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
                    bf16x8 ov;
#pragma unroll
                    for (int j = 0; j < 8; ++j) ov[j] = (__bf16)acc[mt][pt][8 * q + j];
                    const int gy = gy0 + pt;
                    if (gy < p.h && okx) *(bf16x8*)(p.y + ((long)(img * p.h + gy) * p.w_ + gx) * 64 + mt * 32 + 16 * q + 8 * hh) = ov;
                }
        lds_barrier();                                   // B: the next tile is in LDS
    }
}

// ---------------------------------------------------------------------------------------------------------------
// v2 of the trunk convolution: weights in REGISTERS, halo tiles double-buffered by LDS-DMA, one wave per SIMD
// ---------------------------------------------------------------------------------------------------------------
// v1 keeps all 9x64x64 weights in LDS (72 KiB), which leaves room for ONE halo tile: hence its loader waves, its two
// barriers per tile, the 2+1+1+2 placement of six compute waves on four SIMDs and one LDS read per MFMA.  Here the
// workgroup is four waves, one per SIMD, each with the 512-register budget that buys:
//   * a wave owns 32 of the 64 output channels x 8 of the 16 tile rows and keeps ITS 36 weight fragments (9 taps x 4
//     channel groups) in 144 registers for the whole launch -- no weights in LDS at all;
//   * LDS holds two 18x34-pixel halo tiles (2 x 76.5 KiB), filled by `buffer_load_dwordx4 ... lds` (range-checked by the
//     buffer descriptor: the zero padding costs no branch and no select on data);
//   * a software pipeline over HALF tiles (two accumulator sets of 4 rows): the epilogue of one half is issued in the MFMA
//     shadow of the other, the next tile's DMA pieces in phase A's -- see the tile loop;
//   * one barrier per tile; it orders LDS only, so the stores drain under the following MFMAs.
// Measured (scripts/micro/v2_stamps.py, batch 32 at 256x256, sustained): 13.6 k cycles per tile for 9.2 k of MFMA issue
// (serial epilogue: 15.4 k; v1's pipe is busy 28 % of the time) at the 1.6 GHz the chip holds under this load -- a bare
// v_mfma_f32_32x32x16_bf16 stream on random operands holds 1.5-1.75 GHz = 1.5-1.75 PFLOP/s (scripts/micro/mfma_peak_bf16.hip):
// the kernel is power-limited, and cycles saved come back as a lower clock.  The variant WITH a
// residual input stays on v1: it moves 1.5x the bytes, and with one wave per SIMD every stalled vector-memory issue also
// stalls that SIMD's MFMA stream (measured 0.234 ms against 0.213).
constexpr int V2_TR = 16, V2_TC = 32, V2_HR = V2_TR + 2, V2_HC = V2_TC + 2;
constexpr int V2_ROWB = V2_HC * 128;
constexpr int V2_XB = V2_HR * V2_ROWB;                          // 78336
constexpr int V2_NT = 256;
constexpr int V2_CHUNKS = V2_XB / 16;                           // 4896 = 19 * 256 + 32
constexpr int V2_NDMA = (V2_CHUNKS + V2_NT - 1) / V2_NT;        // 20 rounds; the last one is half of wave 0
constexpr int V2_PAD = 512;                                     // what the other half of that wave writes (zeros) past the tile
constexpr int V2_BUF = V2_XB + V2_PAD;
constexpr int V2_LDS = 2 * V2_BUF + PB;
static_assert(V2_LDS <= 160 * 1024, "v2 trunk kernel: LDS");
static_assert(V2_NDMA == 20 && V2_CHUNKS - 19 * V2_NT == 32, "v2 trunk kernel: DMA schedule");

// (namespace scope: hipcc emits no host stub for a kernel template whose lambdas return a struct local to the kernel)
struct TileSrc { unsigned base; int x0; vcg_rsrc rs; };          // a tile's halo: byte offset of its origin in the image, first column, image descriptor
struct OutPos { vcg_rsrc rs; int gx, gy0; bool okx; };           // where a half's accumulators go: image descriptor, column, first row

// Diagnostic build only (-DVCG_V2_STAMPS, scripts/micro/v2_stamps.sh): s_memtime brackets around the four segments of a
// tile, summed per wave in scalar registers and written to a buffer of their own after the loop.  No stamp executes in
// the shipped library; read the SHARES of such a build, not its run time (cdna_hip_programming.md, In-kernel stamps).
#ifdef VCG_V2_STAMPS
__device__ unsigned long long vcg_v2_stamp_sums[256 * 4 * 6];
#define V2_STAMP(t)                                                                   \
    do {                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");     \
        __builtin_amdgcn_sched_barrier(0);                                            \
    } while (0)
#define V2_STAMP_ADD(sum, a, b) sum += (b) - (a)
#else
#define V2_STAMP(t) do { } while (0)
#define V2_STAMP_ADD(sum, a, b) do { } while (0)
#endif

// STATS: the epilogue also accumulates, per lane, the sum and the sum of squares of the values it stores (as rounded to bf16: the
// statistics are those of the tensor the next kernel reads) for the training-mode BatchNormalization / instance norm behind the
// convolution (model.py:20,23,284) -- the separate statistics pass over the output (one more read of the tensor, two more launches
// per normalisation) is gone.  16 values of a store unit cost 24 vector instructions in two more stages of the drain; the 32 partial
// sums of a lane are reduced over the 32 pixels of the wave once per launch (once per tile for per-image statistics) and written as
// one record per (workgroup | tile, row group); vcg_norm_finalize_partials sums the records in a fixed order.
template <bool AFF, bool SLOPE, bool STATS = false>
__global__ __launch_bounds__(V2_NT, 1) void conv3x3_c64_bf16_v2_kernel(C3Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef VCG_V2_STAMPS
    const unsigned long long k_c0 = __builtin_amdgcn_s_memtime(), k_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    float* prm = (float*)(smem + 2 * V2_BUF);
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int coh = wv & 1, rg = wv >> 1;                       // channel half, row group (rows 8rg .. 8rg+7 of the tile)

    if (tid < 64) {
        prm[tid] = p.scale ? p.scale[tid] : 1.f;
        prm[64 + tid] = p.shift ? p.shift[tid] : 0.f;
        prm[128 + tid] = p.act == VCG_ACT_PRELU ? p.alpha[tid] : (p.act == VCG_ACT_LRELU ? p.act_alpha : 1.f);
    }
    // LDS byte offset of lane (pixel r + dx, half hh)'s fragment of channel group s: boff[dx] ^ (s << 5) -- the swizzle XORs
    // the chunk index 2s + hh with (pos >> 1) & 7, and 2s only touches bits 5-6 of the 128-byte row
    int boff[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int pos = r + dx;
        boff[dx] = rg * 8 * V2_ROWB + pos * 128 + ((hh ^ ((pos >> 1) & 7)) << 4);
    }
    const long img_bytes = (long)p.h * p.w_ * 128;

    // one 4-KiB piece (round k of 20) of a tile's halo: slot = 16-byte chunk of the LDS image, in image order.  Issued between MFMAs,
    // so its address arithmetic has to fit an MFMA's shadow: per lane and round the byte offset of the slot's source RELATIVE to the
    // tile's halo origin ((row * w + col) * 128 + chunk * 16) and its halo column are launch constants (20 + 5 registers); a piece is
    // then  offset = tile base + constant,  one column test (left / right image edge)  and a select.  Rows above / below the image
    // need no test: their offsets fall outside the image's buffer descriptor (negative ones wrap to > 4 GiB - 4 MiB); without a
    // next tile the descriptor has zero records and every piece writes zeros into the idle buffer.
    unsigned dma_c[V2_NDMA], dma_colp[(V2_NDMA + 3) / 4];
#pragma unroll
    for (int k = 0; k < V2_NDMA; ++k) {
        const int sl = k * V2_NT + tid, P = sl >> 3, row = P / V2_HC, col = P - row * V2_HC;
        const int cs = (sl & 7) ^ ((col >> 1) & 7);                                // stored chunk (sl & 7) holds source chunk cs
        dma_c[k] = (unsigned)(row * p.w_ + col) * 128u + (unsigned)(cs * 16);
        if ((k & 3) == 0) dma_colp[k >> 2] = 0;
        dma_colp[k >> 2] |= (unsigned)(sl < V2_CHUNKS ? col : 255) << (8 * (k & 3));     // 255: the slots past the tile (round 19, lanes 32-63)
    }
    auto locate = [&](int tile, bool live) {
        const int txi = tile % p.tiles_x, t2 = tile / p.tiles_x, tyi = t2 % p.tiles_y, img = t2 / p.tiles_y;
        const int y0 = tyi * V2_TR - 1, x0 = txi * V2_TC - 1;
        return TileSrc{(unsigned)(y0 * p.w_ + x0) * 128u, x0, make_rsrc((const unsigned char*)p.x + img * img_bytes, (unsigned long)(live ? img_bytes : 0))};
    };
    auto dma = [&](const TileSrc& ts, int buf, int k) {
        if (k == V2_NDMA - 1 && wv != 0) return;                 // wave-uniform
        const int col = (int)((dma_colp[k >> 2] >> (8 * (k & 3))) & 255u);
        unsigned off = ts.base + dma_c[k];
        asm volatile("" : "+v"(off));                            // a select, not a branch around the arithmetic (it would split the schedule)
        off = (unsigned)(ts.x0 + col) < (unsigned)p.w_ ? off : VCG_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ts.rs, (void __attribute__((address_space(3)))*)(smem + buf * V2_BUF + (k * V2_NT + wv * 64) * 16),
                                                 16, off, 0, 0, 0);
    };

    // the first tile's halo goes out before anything else (it is the HBM round trip every workgroup starts with), the weights behind it
    int tile = blockIdx.x, buf = 0;
    if (tile < p.total) {
        const TileSrc tp = locate(tile, true);
#pragma unroll
        for (int k = 0; k < V2_NDMA; ++k) dma(tp, 0, k);
    }
    // this wave's 36 weight fragments: A[row = co][k = 8hh + j] of (tap, channel group s) = packed [tap][co][ci] chunk 2s + hh
    bf16x8 wa[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) {
        const int tap = i >> 2, s = i & 3;
        wa[i] = __builtin_bit_cast(bf16x8, p.w[(tap * 64 + coh * 32 + r) * 8 + 2 * s + hh]);
    }

    // ---- the pipelined tile loop ---------------------------------------------------------------------------------------
    // A wave's 8 rows are two HALVES of 4 rows with an accumulator set each (2 x 64 registers): while the 144 MFMAs of one half
    // run, the epilogue of the OTHER half (finished 144 MFMAs ago) is issued between them, one 16-byte store unit (8 channels of
    // one row) per k-group -- with one wave per SIMD nothing else could fill the MFMA shadow, and a serial epilogue was 26 % of
    // the tile (profiles/r02_v2_stamps.txt).  Phase A of a tile computes half 0 and drains half 1 of the PREVIOUS tile (whose
    // position is carried in `pv`); phase B computes half 1 and drains half 0.  Per (dx, channel group) a half reads its six halo
    // rows once for the MFMAs of all three dy.
    bf16x8 fb[2][6];
    auto frag = [&](const unsigned char* xb, int h, int g, int b) {
        const int dx = g >> 2, s = g & 3;
#pragma unroll
        for (int j = 0; j < 6; ++j) fb[b][j] = *(const bf16x8*)(xb + (4 * h + j) * V2_ROWB + (boff[dx] ^ (s << 5)));
    };
    auto out_off = [&](const OutPos& o, int n, int q) {
        const int gy = o.gy0 + n;
        unsigned off = (unsigned)(gy * p.w_ + o.gx) * 128u + (unsigned)((coh * 32 + 16 * q + 8 * hh) * 2);
        asm volatile("" : "+v"(off));                            // a select, not a branch around the arithmetic (it would split the schedule)
        return gy < p.h && o.okx ? off : VCG_OOB;
    };
    // one store unit u = (q, n): channels 16q + 8hh + {0..7} of row n.  An MFMA tile leaves lane (pixel, h) with channels
    // 8g+4h+{0..3}; v_permlane32_swap between the register groups (2q, 2q+1) of the two half-waves makes them 8 consecutive ones.
    float sc[8], sh[8], al[8];
    auto epi_params = [&](int q) {
        int prm_o = 0;
        asm volatile("" : "+v"(prm_o));                  // re-read per use: 24 registers not to be held across tiles
        const float* prm_t = prm + prm_o + coh * 32 + 16 * q + 8 * hh;
        if (AFF) {
            *(f32x4*)&sc[0] = *(const f32x4*)(prm_t);
            *(f32x4*)&sc[4] = *(const f32x4*)(prm_t + 4);
            *(f32x4*)&sh[0] = *(const f32x4*)(prm_t + 64);
            *(f32x4*)&sh[4] = *(const f32x4*)(prm_t + 68);
        }
        if (SLOPE) {
            *(f32x4*)&al[0] = *(const f32x4*)(prm_t + 128);
            *(f32x4*)&al[4] = *(const f32x4*)(prm_t + 132);
        }
    };
    // the unit in seven stages, so that a stage fits the shadow of one MFMA (32 cycles = about seven VALU instructions):
    //   0, 1: accumulator reads + permlane swaps of channels 0-3 / 4-7;  2..5: scale / shift / slope of two values each;  6: pack + store
    float ev[8];
    float sacc[32];                                              // STATS: [sum | sum of squares][q][j] of this lane's 16 channels
    unsigned ovb[4];                                             // STATS: the unit's packed output between its store and its two statistics stages
#pragma unroll
    for (int i = 0; i < 32; ++i) sacc[i] = 0.f;
    auto epi_stage = [&](f32x16 (&acc)[4], const OutPos& o, int u, int st) {
        const int q = u >> 2, n = u & 3;
        if (st == 0 && n == 0) epi_params(q);
        if (st < 2) {
#pragma unroll
            for (int j = 2 * st; j < 2 * st + 2; ++j) {
                float lo = acc[n][8 * q + j], hi = acc[n][8 * q + 4 + j];
                swap32(lo, hi);
                ev[j] = lo;
                ev[4 + j] = hi;
            }
            // pin the stage where it is written: LLVM sinks side-effect-free arithmetic across sched_barrier down to its use
            asm volatile("" : "+v"(ev[2 * st]), "+v"(ev[2 * st + 1]), "+v"(ev[2 * st + 4]), "+v"(ev[2 * st + 5]));
        } else if (st < 6) {
#pragma unroll
            for (int j = 2 * (st - 2); j < 2 * (st - 2) + 2; ++j) {
                float t = ev[j];
                if (AFF) t = t * sc[j] + sh[j];
                if (SLOPE) t = t >= 0.f ? t : t * al[j];
                ev[j] = t;
            }
            asm volatile("" : "+v"(ev[2 * (st - 2)]), "+v"(ev[2 * (st - 2) + 1]));
        } else if (st == 6) {
            bf16x8 ov;
#pragma unroll
            for (int j = 0; j < 8; ++j) ov[j] = (__bf16)ev[j];
            // through the image's buffer descriptor: an out-of-image lane (or a half with nothing pending) gets the out-of-range
            // offset instead of an exec mask -- no branch to split the schedule, and hipcc can count the stores in its vmcnt waits
            const u32x4 ob = __builtin_bit_cast(u32x4, ov);
            __builtin_amdgcn_raw_buffer_store_b128(ob, o.rs, (int)out_off(o, n, q), 0, 0);
            if (STATS) {
                // what the statistics stages read: zero for a pixel outside the image (its accumulators hold the bias)
                const unsigned m = (o.gy0 + n < p.h && o.okx) ? 0xFFFFFFFFu : 0u;
#pragma unroll
                for (int d = 0; d < 4; ++d) ovb[d] = ob[d] & m;
                asm volatile("" : "+v"(ovb[0]), "+v"(ovb[1]), "+v"(ovb[2]), "+v"(ovb[3]));
            }
        } else if (STATS) {
            // stages 7, 8: channels 4(st-7) .. 4(st-7)+3 of the unit, unpacked from the stored bf16 pairs
#pragma unroll
            for (int d = 2 * (st - 7); d < 2 * (st - 7) + 2; ++d) {
                const float lo = __uint_as_float(ovb[d] << 16), hi = __uint_as_float(ovb[d] & 0xFFFF0000u);
                sacc[8 * q + 2 * d] += lo;
                sacc[16 + 8 * q + 2 * d] = fmaf(lo, lo, sacc[16 + 8 * q + 2 * d]);
                sacc[8 * q + 2 * d + 1] += hi;
                sacc[16 + 8 * q + 2 * d + 1] = fmaf(hi, hi, sacc[16 + 8 * q + 2 * d + 1]);
            }
            asm volatile("" : "+v"(sacc[8 * q + 4 * (st - 7)]), "+v"(sacc[8 * q + 4 * (st - 7) + 1]), "+v"(sacc[8 * q + 4 * (st - 7) + 2]),
                         "+v"(sacc[8 * q + 4 * (st - 7) + 3]), "+v"(sacc[16 + 8 * q + 4 * (st - 7)]), "+v"(sacc[16 + 8 * q + 4 * (st - 7) + 1]),
                         "+v"(sacc[16 + 8 * q + 4 * (st - 7) + 2]), "+v"(sacc[16 + 8 * q + 4 * (st - 7) + 3]));
        }
    };
    // STATS: the lane sums -> one record: lane (r, hh) ends up with value r = [stat][q][j] summed over the wave's 32 pixels
    auto stats_flush = [&](int unit) {
        const float t = half_wave_reduce_scatter32(sacc, r);
        p.stats[(((long)unit * 2 + rg) * 2 + (r >> 4)) * 64 + coh * 32 + ((r >> 3) & 1) * 16 + 8 * hh + (r & 7)] = t;
#pragma unroll
        for (int i = 0; i < 32; ++i) sacc[i] = 0.f;
    };
    // one phase: 12 k-groups of 12 MFMAs into `acc` (rows 4h..4h+3), the next group's six rows read under them, `drain`'s eight
    // store units in groups 2..9, and (phase A only) two DMA pieces of the next tile in groups 0..9.  The order inside a group is
    // written out and pinned (sched_barrier after every MFMA): left to the scheduler, the drain ends up behind the MFMAs.
    auto phase = [&](f32x16 (&acc)[4], f32x16 (&drain)[4], const OutPos& dpos, const unsigned char* xb, int h, const TileSrc& np, int nbuf,
                     bool has_next) {
        (void)has_next;
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 12; ++g) {
            const int cur = g & 1, dx = g >> 2, s = g & 3;
            const bool dr = g >= 2 && g < 10, dm = h == 0 && g < 10;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int dy = i >> 2, n = i & 3;
                acc[n] = mfma_bf16(wa[(dy * 3 + dx) * 4 + s], fb[cur][n + dy], acc[n]);
                if (i == 0) {
                    if (g + 1 < 12) frag(xb, h, g + 1, cur ^ 1);
                    else if (h == 0) frag(xb, 1, 0, cur ^ 1);         // phase B's first rows, under phase A's last group
                }
                // unconditional DMA (no branch to split the group): without a next tile its descriptor has no records
                if (i == 1 && dm) dma(np, nbuf, 2 * g);
                if (i == 3 && dm) dma(np, nbuf, 2 * g + 1);
                if (dr && i == 2) epi_stage(drain, dpos, g - 2, 0);
                if (dr && i >= 4 && i <= 9) epi_stage(drain, dpos, g - 2, i - 3);
                if (STATS && dr && i >= 10) epi_stage(drain, dpos, g - 2, i - 3);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0): this wave's pieces (and weights) have landed
    lds_barrier();
    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, sum_pa = 0, sum_pb = 0, sum_wt = 0, sum_br = 0;
    (void)st0, (void)st1, (void)st2, (void)st3, (void)st4, (void)sum_pa, (void)sum_pb, (void)sum_wt, (void)sum_br;

    f32x16 acc0[4], acc1[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc1[n][e] = 0.f;
    OutPos pv{make_rsrc(p.y, 0), 0, 0, false};                   // nothing pending before the first tile: every offset out of range

    for (; tile < p.total; tile += gridDim.x, buf ^= 1) {
        V2_STAMP(st0);
        const int next = tile + gridDim.x;
        const bool has_next = next < p.total;
        const TileSrc np = locate(has_next ? next : tile, has_next);
        const unsigned char* xb = smem + buf * V2_BUF;
        const int txi = tile % p.tiles_x, t2 = tile / p.tiles_x, tyi = t2 % p.tiles_y, img = t2 / p.tiles_y;
        const int gx = txi * V2_TC + r;
        const vcg_rsrc yrs = make_rsrc((const unsigned char*)p.y + img * img_bytes, (unsigned long)img_bytes);
        const OutPos p0{yrs, gx, tyi * V2_TR + rg * 8, gx < p.w_}, p1{yrs, gx, tyi * V2_TR + rg * 8 + 4, gx < p.w_};
        frag(xb, 0, 0, 0);
        phase(acc0, acc1, pv, xb, 0, np, buf ^ 1, has_next);      // half 0; drains the previous tile's half 1
        if (STATS && p.stats_per_tile && tile != (int)blockIdx.x) stats_flush(tile - (int)gridDim.x);     // the previous tile is complete
        V2_STAMP(st1);
        phase(acc1, acc0, p0, xb, 1, np, buf ^ 1, has_next);      // half 1; drains this tile's half 0
        pv = p1;
        V2_STAMP(st2);
        // The next tile's pieces were issued in phase A: retire them -- vmcnt(8) leaves phase B's eight younger stores in flight
        // (vector memory operations retire in order) -- then the one barrier of the tile: every wave's pieces are in LDS and
        // every wave is done reading this tile's image.
        __builtin_amdgcn_s_waitcnt(0x0F78);
        V2_STAMP(st3);
        lds_barrier();
        V2_STAMP(st4);
        V2_STAMP_ADD(sum_pa, st0, st1);
        V2_STAMP_ADD(sum_pb, st1, st2);
        V2_STAMP_ADD(sum_wt, st2, st3);
        V2_STAMP_ADD(sum_br, st3, st4);
    }
    // the last tile's half 1
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int st = 0; st < (STATS ? 9 : 7); ++st) epi_stage(acc1, pv, u, st);
    if (STATS) stats_flush(p.stats_per_tile ? tile - (int)gridDim.x : (int)blockIdx.x);
#ifdef VCG_V2_STAMPS
    if (lane == 0) {
        unsigned long long* o = vcg_v2_stamp_sums + (blockIdx.x * 4 + wv) * 6;
        o[0] = sum_pa, o[1] = sum_pb, o[2] = sum_wt, o[3] = sum_br;
        o[4] = __builtin_amdgcn_s_memtime() - k_c0, o[5] = __builtin_amdgcn_s_memrealtime() - k_r0;     // whole kernel: core clock / 100 MHz
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// 9x9 stride-1 'same' convolution, 3 -> 64 channels (+bias, PReLU): the generator's initial/conv (model.py:275-276)
// ---------------------------------------------------------------------------------------------------------------
// Input: the fp32 NCHW frames as they arrive; output: bf16 NHWC, i.e. the kernel is also the entry into the bf16
// layout.  In LDS a pixel is RGB0 in bf16 (8 bytes), so 4 consecutive pixels x 4 channels are one 16-wide k-step and
// a lane's operand fragment (2 pixels) is one 8-byte-aligned ds_read2_b64: k-steps = 9 ky x 3 groups of 4 kx (the
// taps kx = 9..11 carry zero weights) = 27, against 36 of the 64-channel 3x3 convolution.  Same skeleton as
// conv3x3_c64_bf16_kernel: 6 compute + 2 loader waves, weights (54 KiB of operand fragments) resident in LDS,
// 12x32-pixel tiles, permlane-swapped 16-byte stores.  The kernel is bound by its 128 bytes of output per pixel.
// The same kernel serves every convolution ON THREE INPUT CHANNELS (template <KH, NG, S>: KH kernel rows, NG groups of 4 kernel columns,
// stride S): 9x9 stride 1 above (KH 9, NG 3), the critics' first layers -- 4x4 stride 2 (PatchGAN block 1: KH 4, NG 1, S 2) and 3x3 stride 1
// (simple_512 / thin_512 block 1, model.py:839: KH 3, NG 1) -- with bias + LeakyReLU / PReLU in the epilogue.  An output tile of 12x32
// pixels reads a halo of (12 S + KH - S) x (31 S + 4 NG + 1) input pixels; lane r's fragment sits at column r S + 4 j + 2 hh.
template <int KH, int NG, int S>
struct I3Cfg {
    static constexpr int HR = TR * S + KH - S;                    // halo rows           (9x9: 20)
    static constexpr int HC = (31 * S + 4 * NG + 2) & ~1;         // halo columns, even  (9x9: 44)
    static constexpr int ROWB = HC * 8;
    static constexpr int XB = (HR * ROWB + 15) & ~15;             // 9x9: 7040 B
    static constexpr int NK = KH * NG;                            // k-steps of 16
    static constexpr int WB = NK * 64 * 32;                       // 9x9: 55296 B: [k-step][out-channel][half][8 bf16]
    static constexpr int NPIX = HR * HC;                          // 9x9: 880 pixels per halo tile
    static constexpr int NPRE = (NPIX + NLW * 64 - 1) / (NLW * 64);       // pixels per loader lane (9x9: 7)
    static constexpr int LDS = WB + XB + 512;
};

struct I9Params {
    const float* x;          // fp32 NCHW [n][3][h][w]
    const uint4* w;          // packed fragments (vcg_pack_first9x9_bf16)
    const float* bias;       // [cout] or null
    const float* alpha;      // PReLU slopes [cout] or null (none)
    __bf16* y;               // bf16 NHWC [n][h][w][cout]
    __bf16* z;               // optional bf16 NHWC [n][h][w][cout]: the value in front of the PReLU (its backward needs the sign and, for the slope gradient, the value)
    const __bf16* mask;      // optional bf16 NHWC [n][h][w][cout]: y *= (mask > 0 ? 1 : mask_slope)  (data gradient in front of a LeakyReLU)
    float mask_slope;
    float* chsum;            // optional [workgroups per channel block * 6][cout]: per-wave sums of the stored output per channel (a bias gradient)
    int n, h, w_, cout, tiles_x, tiles_y, total;        // cout = 64 * nblk; workgroup b serves channel block b % nblk; h, w_: INPUT size
    int oh, ow, pad_top, pad_left;                      // output size and the 'before' pads (9x9 'same': h, w_, 4, 4)
    float slope;                                        // without alpha: LeakyReLU slope (1 = no activation)
    int xcd_group;                                      // block_and_stream
};

// Several workgroups write different 128-byte channel blocks of the SAME pixels (cout = 64 nblk): mapped so that the nblk workgroups of one
// tile stream sit on ONE XCD (workgroups b and b + 8 share an XCD under round-robin placement -- speed only, never correctness) and run side by
// side, their pieces of a pixel's 128 nblk bytes meet in that XCD's L2 and leave it together.  An experiment (VCG_XCD_GROUP=1), measured
// neutral: convT 389 vs 387 us, the final-conv data gradient 717 vs 712 us at C3's shard -- the default is the plain b % nblk mapping.
__device__ __forceinline__ void block_and_stream(int nblk, int xcd_group, int& cb, int& wg) {
    const int b = blockIdx.x;
    if (xcd_group && gridDim.x % (8 * nblk) == 0) {
        const int slot = b >> 3;
        cb = slot % nblk;
        wg = (slot / nblk) * 8 + (b & 7);
    } else {
        cb = b % nblk;
        wg = b / nblk;
    }
}

// Diagnostic build only (-DVCG_I9_STAMPS, scripts/micro/i9_stamps.sh): s_memtime sums per wave [MFMA loop, barrier after it, epilogue, barrier
// after it, tiles, kernel clocks] (compute waves) / [fetch issue, barrier 1, stash, barrier 2, tiles, kernel clocks] (loader waves)
#ifdef VCG_I9_STAMPS
__device__ unsigned long long vcg_i9_stamp_sums[512 * 8 * 6];
#define I9_STAMP(t)                                                                   \
    do {                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");     \
        __builtin_amdgcn_sched_barrier(0);                                            \
    } while (0)
#define I9_ADD(sum, a, b) sum += (b) - (a)
#else
#define I9_STAMP(t) do { } while (0)
#define I9_ADD(sum, a, b) do { } while (0)
#endif

template <int KH, int NG, int S>
__global__ __launch_bounds__(NT, 1) void conv_c3to64_bf16_kernel(I9Params p) {
    using C = I3Cfg<KH, NG, S>;
    constexpr int I_HC = C::HC, I_ROWB = C::ROWB, I_XB = C::XB, I_WB = C::WB, I_NPIX = C::NPIX, I_NPRE = C::NPRE, NK = C::NK;
#ifdef VCG_I9_STAMPS
    unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0, ntl = 0, st0, st1, st2, st3, st4;
    const unsigned long long k_c0 = __builtin_amdgcn_s_memtime();
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wl = smem;
    unsigned char* xl = smem + I_WB;
    float* prm = (float*)(smem + I_WB + I_XB);       // bias[64], slope[64]
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk = p.cout >> 6, nwg = gridDim.x / nblk;
    int cb, wg0;
    block_and_stream(nblk, p.xcd_group, cb, wg0);

    for (int c = tid; c < I_WB / 16; c += NT) ((uint4*)wl)[c] = p.w[(long)cb * (I_WB / 16) + c];
    if (tid < 64) {
        prm[tid] = p.bias ? p.bias[cb * 64 + tid] : 0.f;
        prm[64 + tid] = p.alpha ? p.alpha[cb * 64 + tid] : p.slope;
    }
    const long plane = (long)p.h * p.w_;

    if (wv >= NCW) {
        const int lt = tid - NCW * 64;
        float pre[I_NPRE][3];
        auto fetch = [&](int tile) {
            const int txi = tile % p.tiles_x, t2 = tile / p.tiles_x, tyi = t2 % p.tiles_y, img = t2 / p.tiles_y;
            const int y0 = tyi * TR * S - p.pad_top, x0 = txi * TC * S - p.pad_left;
            const float* xi = p.x + (long)img * 3 * plane;
#pragma unroll
            for (int i = 0; i < I_NPRE; ++i) {
                const int pix = min(lt + NLW * 64 * i, I_NPIX - 1);
                const int row = pix / I_HC, col = pix - row * I_HC;
                const int gy = y0 + row, gx = x0 + col;
                const bool ok = (unsigned)gy < (unsigned)p.h && (unsigned)gx < (unsigned)p.w_;
                const long o = (long)min(max(gy, 0), p.h - 1) * p.w_ + min(max(gx, 0), p.w_ - 1);
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const float v = xi[ch * plane + o];
                    pre[i][ch] = ok ? v : 0.f;
                }
            }
        };
        auto stash = [&]() {
#pragma unroll
            for (int i = 0; i < I_NPRE; ++i) {
                const int pix = min(lt + NLW * 64 * i, I_NPIX - 1);
                bf16x4 v;
                v[0] = (__bf16)pre[i][0];
                v[1] = (__bf16)pre[i][1];
                v[2] = (__bf16)pre[i][2];
                v[3] = (__bf16)0.f;
                *(bf16x4*)(xl + pix * 8) = v;
            }
        };
        int tile = wg0;
        if (tile < p.total) fetch(tile);
        stash();
        lds_barrier();
        for (; tile < p.total; tile += nwg) {
            const int next = tile + nwg;
            I9_STAMP(st0);
            if (next < p.total) fetch(next);
            I9_STAMP(st1);
            lds_barrier();
            I9_STAMP(st2);
            if (next < p.total) stash();
            I9_STAMP(st3);
            lds_barrier();
            I9_STAMP(st4);
            I9_ADD(s0, st0, st1); I9_ADD(s1, st1, st2); I9_ADD(s2, st2, st3); I9_ADD(s3, st3, st4);
#ifdef VCG_I9_STAMPS
            ++ntl;
#endif
        }
#ifdef VCG_I9_STAMPS
        if (lane == 0 && blockIdx.x < 512) {
            unsigned long long* o = vcg_i9_stamp_sums + (blockIdx.x * 8 + wv) * 6;
            o[0] = s0, o[1] = s1, o[2] = s2, o[3] = s3, o[4] = ntl, o[5] = __builtin_amdgcn_s_memtime() - k_c0;
        }
#endif
        return;
    }

    const int aoff = r * 32 + hh * 16;                         // weight fragment of (co = r [+32], half)
    const unsigned char* xb = xl + (wv * 2 * S) * I_ROWB + (r * S + 2 * hh) * 8;
    float csum[32];                                            // p.chsum: this lane's 32 channels [mt][q][j], summed over its pixels
#pragma unroll
    for (int i = 0; i < 32; ++i) csum[i] = 0.f;
    lds_barrier();

    for (int tile = wg0; tile < p.total; tile += nwg) {
        const int txi = tile % p.tiles_x, t2 = tile / p.tiles_x, tyi = t2 % p.tiles_y, img = t2 / p.tiles_y;
        const int gx = txi * TC + r, gy0 = tyi * TR + wv * 2;
        const bool okx = gx < p.ow;

        // the LeakyReLU mask of this wave's 2 x 32 pixels (data gradient in front of an activation): requested here, read in the epilogue --
        // under the MFMA loop.  (Loaded where it was used, each of the eight 16-byte loads waited out an HBM round trip: 6.6k of a tile's
        // 21k cycles, profiles/r03_i9_stamps.txt.)
        bf16x8 mk[2][2][2];
        if (p.mask) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt) {
                        const int gy = gy0 + pt;
                        const long o = ((long)(img * p.oh + min(gy, p.oh - 1)) * p.ow + min(gx, p.ow - 1)) * p.cout + cb * 64 + mt * 32 + 16 * q + 8 * hh;
                        mk[mt][q][pt] = *(const bf16x8*)(p.mask + o);
                    }
        }

        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

        bf16x8 fa[2][2], fb[2][2];
        auto frag = [&](auto ic) {
            constexpr int i = decltype(ic)::value, ky = i / NG, j = i - NG * ky, buf = i & 1;
            const unsigned char* wa = wl + i * 2048 + aoff;
            fa[buf][0] = *(const bf16x8*)(wa);
            fa[buf][1] = *(const bf16x8*)(wa + 1024);
            const unsigned char* b0 = xb + ky * I_ROWB + j * 32;
            bf16x4 lo = *(const bf16x4*)(b0), hi = *(const bf16x4*)(b0 + 8);
            fb[buf][0] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            lo = *(const bf16x4*)(b0 + S * I_ROWB);
            hi = *(const bf16x4*)(b0 + S * I_ROWB + 8);
            fb[buf][1] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        };
        I9_STAMP(st0);
        frag(std::integral_constant<int, 0>{});
        static_for<NK>([&](auto ic) {
            constexpr int i = decltype(ic)::value, cur = i & 1;
            if constexpr (i + 1 < NK) frag(std::integral_constant<int, i + 1>{});
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = mfma_bf16(fa[cur][0], fb[cur][0], acc[0][0]);
            acc[0][1] = mfma_bf16(fa[cur][0], fb[cur][1], acc[0][1]);
            acc[1][0] = mfma_bf16(fa[cur][1], fb[cur][0], acc[1][0]);
            acc[1][1] = mfma_bf16(fa[cur][1], fb[cur][1], acc[1][1]);
            __builtin_amdgcn_sched_barrier(0);
        });
        I9_STAMP(st1);
        lds_barrier();
        I9_STAMP(st2);

#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int co = mt * 32 + 16 * q + 8 * hh;
                float sh[8], al[8];
                *(f32x4*)&sh[0] = *(const f32x4*)(prm + co);
                *(f32x4*)&sh[4] = *(const f32x4*)(prm + co + 4);
                *(f32x4*)&al[0] = *(const f32x4*)(prm + 64 + co);
                *(f32x4*)&al[4] = *(const f32x4*)(prm + 64 + co + 4);
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float lo = acc[mt][pt][8 * q + j], hi = acc[mt][pt][8 * q + 4 + j];
                        swap32(lo, hi);
                        v[j] = lo;
                        v[4 + j] = hi;
                    }
                    const int gy = gy0 + pt;
                    const bool ok = gy < p.oh && okx;
                    const long o = ((long)(img * p.oh + min(gy, p.oh - 1)) * p.ow + min(gx, p.ow - 1)) * p.cout + cb * 64 + co;
                    bf16x8 ov, zv;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float u = v[j] + sh[j];
                        zv[j] = (__bf16)u;
                        u = u >= 0.f ? u : u * al[j];
                        if (p.mask) u = (float)mk[mt][q][pt][j] > 0.f ? u : u * p.mask_slope;
                        ov[j] = (__bf16)u;
                    }
                    if (ok) *(bf16x8*)(p.y + o) = ov;
                    if (p.z && ok) *(bf16x8*)(p.z + o) = zv;
                    if (p.chsum && ok) {                             // the values as stored
#pragma unroll
                        for (int j = 0; j < 8; ++j) csum[mt * 16 + q * 8 + j] += (float)ov[j];
                    }
                }
            }
        I9_STAMP(st3);
        lds_barrier();
        I9_STAMP(st4);
        I9_ADD(s0, st0, st1); I9_ADD(s1, st1, st2); I9_ADD(s2, st2, st3); I9_ADD(s3, st3, st4);
#ifdef VCG_I9_STAMPS
        ++ntl;
#endif
    }
#ifdef VCG_I9_STAMPS
    if (lane == 0 && blockIdx.x < 512) {
        unsigned long long* o = vcg_i9_stamp_sums + (blockIdx.x * 8 + wv) * 6;
        o[0] = s0, o[1] = s1, o[2] = s2, o[3] = s3, o[4] = ntl, o[5] = __builtin_amdgcn_s_memtime() - k_c0;
    }
#endif
    if (p.chsum) {
        // lane (r, hh) ends up with the sum of value r = [mt][q][j] over the wave's 32 pixel lanes: channel mt*32 + 16q + 8hh + j
        const float t = half_wave_reduce_scatter32(csum, r);
        p.chsum[(long)(wg0 * NCW + wv) * p.cout + cb * 64 + (r >> 4) * 32 + ((r >> 3) & 1) * 16 + 8 * hh + (r & 7)] = t;
    }
}

__global__ void pack_first9x9_kernel(const float* __restrict__ w, uint4* __restrict__ out, int cout, int dgrad, int kh, int kw, int ng) {
    // out[channel block][k-step = ky*ng+j][co in block][half] = 8 bf16: (kx = 4j+2*half, RGB0), (kx+1, RGB0)
    //   dgrad == 0: w is Keras (kh,kw,3,cout), the forward kernel of a 3 -> cout convolution
    //   dgrad == 1: w is Keras (kh,kw,cout,3), the kernel of a cout -> 3 convolution; packed for its DATA GRADIENT
    //               (a 3 -> cout convolution with the taps flipped):  W'[ky][kx][c][m] = w[kh-1-ky][kw-1-kx][m][c]
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int nk = kh * ng;
    if (idx >= (cout >> 6) * nk * 64 * 2) return;
    const int h = idx & 1, co = (idx >> 1) & 63, ks = (idx >> 7) % nk, cb = idx / (nk * 128), ky = ks / ng, j = ks - ng * ky;
    const int m = cb * 64 + co;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int kx = 4 * j + 2 * h + (e >> 2), c = e & 3;
        float x = 0.f;
        if (kx < kw && c < 3) x = dgrad ? w[(((kh - 1 - ky) * kw + (kw - 1 - kx)) * cout + m) * 3 + c] : w[((ky * kw + kx) * 3 + c) * cout + m];
        v[e] = (__bf16)x;
    }
    out[idx] = __builtin_bit_cast(uint4, v);
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3 stride-2 'same' transposed convolution, 64 -> 64*m channels (+LeakyReLU): upsampling_block (model.py:70-75)
// ---------------------------------------------------------------------------------------------------------------
// TF 'same' with k=3, s=2 crops nothing at the top/left: out[o] = sum_i x[i] w[o-2i]  =>  the four sub-pixel phases
//   even output rows (2q):   taps ky=0 (input row q) and ky=2 (input row q-1);   odd rows (2q+1): ky=1 (row q)
// and the same along x: 4 + 2 + 2 + 1 = 9 taps, the same 36 k-steps per 64x64 wave tile as the 3x3 convolution, but
// four accumulator tiles' worth of output.  Same organisation as conv3x3_c64_bf16_kernel: persistent workgroups of
// 6 compute + 2 loader waves, weights of ONE 64-channel output block resident in LDS (blockIdx & (m-1) picks the
// block, so the grid is a multiple of m), input halo tile = 1 row above / 1 column left of 12x32 input pixels.
constexpr int TH = TR + 1;                   // halo rows of the transposed convolution (row -1 .. TR-1)
constexpr int TXB = TH * ROWB;               // 13 rows x 34 columns (column 33 unused) x 128 B
constexpr int TNCHUNK = TH * HC * 8;
constexpr int TNPRE = (TNCHUNK + NLW * 64 - 1) / (NLW * 64);

struct CTParams {
    const uint4* x;
    const uint4* w;          // packed [9][cout][64]
    const float* shift;      // bias [cout] or null
    __bf16* y;
    int n, h, w_, cout, tiles_x, tiles_y, total;    // total = n * tiles_y * tiles_x (per output-channel block)
    float slope;             // LeakyReLU slope (1 = none)
    int xcd_group;           // block_and_stream
};

// k-step tables (compile time): phase-major tap order; ky*3+kx and the halo row / column offset of the tap
// (1 = same input pixel, 0 = the previous one); taps [0,4) -> phase (0,0), [4,6) -> (0,1), [6,8) -> (1,0), 8 -> (1,1)
__host__ __device__ constexpr int ct_tap(int t) { constexpr int T[9] = {0, 2, 6, 8, 1, 7, 3, 5, 4}; return T[t]; }
__host__ __device__ constexpr int ct_rdy(int t) { constexpr int T[9] = {1, 1, 0, 0, 1, 0, 1, 1, 1}; return T[t]; }
__host__ __device__ constexpr int ct_rdx(int t) { constexpr int T[9] = {1, 0, 1, 0, 1, 1, 1, 0, 1}; return T[t]; }
__host__ __device__ constexpr int ct_phase(int t) { return t < 4 ? 0 : t < 6 ? 1 : t < 8 ? 2 : 3; }
__host__ __device__ constexpr bool ct_first(int t) { return t == 0 || t == 4 || t == 6 || t == 8; }
__host__ __device__ constexpr bool ct_last(int t) { return t == 3 || t == 5 || t == 7 || t == 8; }


// CT_DEFER (default; -DCT_NO_DEFER builds the immediate-store form for A/B): see the tile loop
#if !defined(CT_NO_DEFER) && !defined(CT_DEFER)
#define CT_DEFER
#endif
// Diagnostic build only (-DVCG_CT_STAMPS, scripts/micro/ct_stamps.sh): per compute wave, s_memtime sums of [tile body = MFMAs + interleaved
// phase epilogues, barrier A, barrier B, tiles, kernel clocks]
#ifdef VCG_CT_STAMPS
__device__ unsigned long long vcg_ct_stamp_sums[256 * 8 * 5];
#define CT_STAMP(t)                                                                   \
    do {                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");     \
        __builtin_amdgcn_sched_barrier(0);                                            \
    } while (0)
#else
#define CT_STAMP(t) do { } while (0)
#endif

__global__ __launch_bounds__(NT, 1) void convt3x3_c64_bf16_kernel(CTParams p) {
#ifdef VCG_CT_STAMPS
    unsigned long long cs0 = 0, cs1 = 0, cs2 = 0, cnt = 0, ct0, ct1, ct2, ct3;
    const unsigned long long ck0 = __builtin_amdgcn_s_memtime();
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wl = smem;
    unsigned char* xl = smem + WB;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk = p.cout >> 6, nwg = gridDim.x / nblk;
    int cb, wg;
    block_and_stream(nblk, p.xcd_group, cb, wg);
    float* prm = (float*)(smem + WB + TXB);          // bias of this block's 64 channels
    if (tid < 64) prm[tid] = p.shift ? p.shift[cb * 64 + tid] : 0.f;

    for (int c = tid; c < 9 * 64 * 8; c += NT) {
        const int chunk = c & 7, co = (c >> 3) & 63, tap = c >> 9;
        *(uint4*)(wl + tap * 8192 + co * 128 + ((chunk ^ ((co >> 1) & 7)) << 4)) = p.w[((long)tap * p.cout + cb * 64 + co) * 8 + chunk];
    }

    if (wv >= NCW) {
        const int lt = tid - NCW * 64;
        uint4 pre[TNPRE];
        unsigned okmask = 0;
        auto fetch = [&](int tile) {
            const int txi = tile % p.tiles_x, t2 = tile / p.tiles_x, tyi = t2 % p.tiles_y, img = t2 / p.tiles_y;
            const int y0 = tyi * TR - 1, x0 = txi * TC - 1;
            okmask = 0;
#pragma unroll
            for (int i = 0; i < TNPRE; ++i) {
                const int c = min(lt + NLW * 64 * i, TNCHUNK - 1);
                const int pix = c >> 3, row = pix / HC, col = pix - row * HC;
                const int gy = y0 + row, gx = x0 + col;
                const bool ok = (unsigned)gy < (unsigned)p.h && (unsigned)gx < (unsigned)p.w_;
                const int cy = min(max(gy, 0), p.h - 1), cx = min(max(gx, 0), p.w_ - 1);
                pre[i] = p.x[((long)(img * p.h + cy) * p.w_ + cx) * 8 + (c & 7)];
                okmask |= ok ? (1u << i) : 0u;
            }
        };
        auto stash = [&]() {
#pragma unroll
            for (int i = 0; i < TNPRE; ++i) {
                const int c = min(lt + NLW * 64 * i, TNCHUNK - 1);
                const int pix = c >> 3, row = pix / HC, col = pix - row * HC;
                const uint4 v = (okmask >> i) & 1u ? pre[i] : make_uint4(0, 0, 0, 0);
                *(uint4*)(xl + pix * 128 + (((c & 7) ^ ((col >> 1) & 7)) << 4)) = v;
            }
        };
        int tile = wg;
        if (tile < p.total) fetch(tile);
        stash();
        lds_barrier();
        for (; tile < p.total; tile += nwg) {
            const int next = tile + nwg;
            if (next < p.total) fetch(next);
            lds_barrier();
            if (next < p.total) stash();
            lds_barrier();
        }
        return;
    }

    int aoff[4], boff[2][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) aoff[s] = r * 128 + (((2 * s + hh) ^ ((r >> 1) & 7)) << 4);
#pragma unroll
    for (int dx = 0; dx < 2; ++dx)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int pos = r + dx;
            boff[dx][s] = pos * 128 + (((2 * s + hh) ^ ((pos >> 1) & 7)) << 4);
        }
    const unsigned char* xb = xl + (wv * 2) * ROWB;
    const int ow = 2 * p.w_, oh = 2 * p.h;
    lds_barrier();

#ifdef CT_DEFER
    // Deferred stores: a phase's eight 16-byte stores per lane are not issued behind its MFMAs but one per k-step under the NEXT phase's MFMAs
    // (the last phase's under the next tile's first k-steps).  Stamps showed a tile as 6.5 k cycles of MFMA issue + 17-23 k cycles in which the
    // wave sits in store instructions the memory pipe accepts at ~600 cycles each (all six waves at once, ~4.2 TB/s while it lasts) + 6.3 k at
    // the barriers, i.e. nothing is being stored during 40 % of the time; spread over the whole tile the stores run under the MFMAs.
    bf16x8 pend[8];
    long pbase = 0;
    unsigned pok = 0;                  // bit 0/1: row gy0 / gy0 + 1 of the pending phase is inside the image (and the column is)
    const long prow = 2l * (2 * p.w_) * p.cout;                        // two output rows further down
    auto issue = [&](auto jc) {
        constexpr int j = decltype(jc)::value, mt = j >> 2, q = (j >> 1) & 1, pt = j & 1;
        if ((pok >> pt) & 1u) *(bf16x8*)(p.y + pbase + pt * prow + mt * 32 + 16 * q) = pend[j];
    };
#endif
    for (int tile = wg; tile < p.total; tile += nwg) {
        const int txi = tile % p.tiles_x, t2 = tile / p.tiles_x, tyi = t2 % p.tiles_y, img = t2 / p.tiles_y;
        const int gx = txi * TC + r, gy0 = tyi * TR + wv * 2;
        const bool okx = gx < p.w_;

        f32x16 acc[2][2];
        bf16x8 fa[2][2], fb[2][2];
        CT_STAMP(ct0);
        auto frag = [&](auto ic) {
            constexpr int i = decltype(ic)::value, t = i >> 2, s = i & 3, buf = i & 1;
            const unsigned char* wa = wl + ct_tap(t) * 8192 + aoff[s];
            fa[buf][0] = *(const bf16x8*)(wa);
            fa[buf][1] = *(const bf16x8*)(wa + 4096);
            fb[buf][0] = *(const bf16x8*)(xb + ct_rdy(t) * ROWB + boff[ct_rdx(t)][s]);
            fb[buf][1] = *(const bf16x8*)(xb + (ct_rdy(t) + 1) * ROWB + boff[ct_rdx(t)][s]);
        };
        frag(std::integral_constant<int, 0>{});
        static_for<36>([&](auto ic) {
            constexpr int i = decltype(ic)::value, cur = i & 1, t = i >> 2;
            constexpr bool first = (i & 3) == 0 && ct_first(t), last = (i & 3) == 3 && ct_last(t);
            if constexpr (i + 1 < 36) frag(std::integral_constant<int, i + 1>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (first) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
            }
            acc[0][0] = mfma_bf16(fa[cur][0], fb[cur][0], acc[0][0]);
            acc[0][1] = mfma_bf16(fa[cur][0], fb[cur][1], acc[0][1]);
#ifdef CT_DEFER
            // the pending phase's stores: phase 0 (16 k-steps) takes one every second k-step, phases 1 and 2 (8 k-steps) one per k-step,
            // phase 3 (4 k-steps) two
            if constexpr (i < 16) { if constexpr ((i & 1) == 0) issue(std::integral_constant<int, i / 2>{}); }
            else if constexpr (i < 32) issue(std::integral_constant<int, (i - 16) & 7>{});
            else { issue(std::integral_constant<int, 2 * (i - 32)>{}); }
#endif
            acc[1][0] = mfma_bf16(fa[cur][1], fb[cur][0], acc[1][0]);
            acc[1][1] = mfma_bf16(fa[cur][1], fb[cur][1], acc[1][1]);
#ifdef CT_DEFER
            if constexpr (i >= 32) issue(std::integral_constant<int, 2 * (i - 32) + 1>{});
#endif
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (last) {
                // this phase's 64 channels x 64 output pixels: LeakyReLU, bf16, 16-byte stores
                constexpr int py = ct_phase(t) >> 1, px = ct_phase(t) & 1;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int pt = 0; pt < 2; ++pt) {
                            float v[8], sh[8];
                            *(f32x4*)&sh[0] = *(const f32x4*)(prm + mt * 32 + 16 * q + 8 * hh);
                            *(f32x4*)&sh[4] = *(const f32x4*)(prm + mt * 32 + 16 * q + 8 * hh + 4);
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                float lo = acc[mt][pt][8 * q + j], hi = acc[mt][pt][8 * q + 4 + j];
                                swap32(lo, hi);
                                v[j] = lo;
                                v[4 + j] = hi;
                            }
                            bf16x8 ov;
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const float b = v[j] + sh[j];
                                const float u = b >= 0.f ? b : b * p.slope;
                                ov[j] = (__bf16)u;
                            }
#ifdef CT_DEFER
                            pend[mt * 4 + q * 2 + pt] = ov;
#else
                            const int gy = gy0 + pt;
                            if (gy < p.h && okx)
                                *(bf16x8*)(p.y + ((long)(img * oh + 2 * gy + py) * ow + 2 * gx + px) * p.cout + cb * 64 + mt * 32 + 16 * q + 8 * hh) = ov;
#endif
                        }
#ifdef CT_DEFER
                pbase = ((long)(img * oh + 2 * gy0 + py) * ow + 2 * gx + px) * p.cout + cb * 64 + 8 * hh;
                pok = okx ? ((gy0 < p.h ? 1u : 0u) | (gy0 + 1 < p.h ? 2u : 0u)) : 0u;
#endif
#ifndef CT_NO_SCHED
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
        });
        CT_STAMP(ct1);
        lds_barrier();
        CT_STAMP(ct2);
        lds_barrier();
        CT_STAMP(ct3);
#ifdef VCG_CT_STAMPS
        cs0 += ct1 - ct0, cs1 += ct2 - ct1, cs2 += ct3 - ct2, ++cnt;
#endif
    }
#ifdef CT_DEFER
    static_for<8>([&](auto jc) { issue(jc); });            // the last tile's last phase
#endif
#ifdef VCG_CT_STAMPS
    if (lane == 0 && blockIdx.x < 256) {
        unsigned long long* o = vcg_ct_stamp_sums + (blockIdx.x * 8 + wv) * 5;
        o[0] = cs0, o[1] = cs1, o[2] = cs2, o[3] = cnt, o[4] = __builtin_amdgcn_s_memtime() - ck0;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// 9x9 stride-1 'same' convolution, 256 -> 3 channels (+bias, tanh): the generator's final/conv (model.py:290-291)
// ---------------------------------------------------------------------------------------------------------------
// Three output channels would waste 29 of the MFMA's 32 rows, so the rows carry (ky, co) instead:
//   row 4*ky+co (ky < 8)  and  row 4*co+3 (ky = 8);      k = (kx, ci);      columns = 32 consecutive x.
// One MFMA pass over ONE input row yi then yields, for every ky, that row's contribution to output row yi+4-ky.
// A wave marches down the image and carries the partial sums of the 9 output rows in flight IN THE ACCUMULATOR:
// before the next input row the accumulator is shifted by one ky-group (4 rows = half a register group: a
// v_permlane32_swap + select per register) and handed to the MFMA as its C operand; the group that falls off the end
// (ky = 7) lands in the spare rows 3,7,11, where the ky = 8 products complete it.  After the pass those three rows
// hold the finished output row yi-4 -- no atomics, no partial tensors, every input row is read from HBM once.
//   * the 256 input channels are split over the 4 waves of a workgroup (64 each): the 9x4 weight fragments of a
//     wave stay in 144 VGPRs; the four partial output rows meet in LDS once per row (768 B per wave);
//   * the wave's 72-pixel x 64-channel slice of the input row goes HBM -> LDS by global_load_lds (no VGPRs), double
//     buffered, XOR-swizzled on the global side so that the shifted ds_read_b128 of all 9 kx are conflict-free;
//   * work item = (image, 64-column strip, segment of 32..128 output rows); 2 workgroups per CU.
constexpr int F_PIX = 72;                    // 64 output columns + 4 + 4
constexpr int F_ROWB = F_PIX * 128;          // one wave's slice of one input row in LDS (9216 B)
constexpr int F_LDS = 4 * 2 * F_ROWB + 2 * 4 * 3 * 64 * 4;
constexpr int F_NFRAG = 4 * 9 * 4 * 64;      // 16-byte weight fragments; the packed buffer holds 4 more (zeros)

struct F9Params {
    const unsigned char* x;      // bf16 NHWC [n][h][w][256]
    const uint4* wfrag;          // packed [4 chunks][9 kx][4 s][64 lanes] x 16 B, followed by 64 zero bytes
    const float* bias;           // [3] or null
    float* y;                    // fp32 NCHW [n][3][h][w]
    int n, h, w_, strips, segs, sh, total;      // sh: output rows per work item (8 halo rows are recomputed per item)
    int tanh_act;
};

// Diagnostic build only (-DVCG_F9_STAMPS, scripts/micro/f9_stamps.sh): s_memtime brackets around the five segments of an input row, summed
// per wave and written to a buffer of their own; no stamp executes in the shipped library.
#ifdef VCG_F9_STAMPS
__device__ unsigned long long vcg_f9_stamp_sums[512 * 4 * 8];
#define F9_STAMP(t) V2_STAMP_RAW(t)
#define F9_ADD(sum, a, b) sum += (b) - (a)
#else
#define F9_STAMP(t) do { } while (0)
#define F9_ADD(sum, a, b) do { } while (0)
#endif
#define V2_STAMP_RAW(t)                                                               \
    do {                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");     \
        __builtin_amdgcn_sched_barrier(0);                                            \
    } while (0)

// tanh from one v_exp_f32 and one v_rcp_f32 (tanhf's libm expansion was ~100 instructions per output value on the three waves that
// finish a row while the fourth idles): 1 - 2 / (exp(2|x|) + 1), odd; below 2^-6 the cubic, where the quotient form would cancel.
// Absolute error < 2e-7, far below the bf16 operands' own rounding.
__device__ __forceinline__ float fast_tanh(float x) {
    const float a = fabsf(x);
    const float t = 1.f - 2.f * __frcp_rn(__expf(2.f * a) + 1.f);
    const float small = a * (1.f - a * a * (1.f / 3.f));
    return copysignf(a < 0.015625f ? small : t, x);
}

// BUF: the row slices are fetched through the image's buffer descriptor (pixels outside the image read as zero by the range check) with two
// lane constants; the pointer form (BUF = false: images too large for a descriptor) keeps nine 64-bit lane addresses, which at this kernel's
// 256-register budget are SPILLED -- hipcc then waits vmcnt(0) in front of every reload, i.e. for every earlier piece of the row: the nine
// pieces went out one HBM round trip after the other, 8.0 k of the 15.4 k cycles of a row (profiles/r03_f9_stamps.txt).
template <bool BUF>
__global__ __launch_bounds__(256, 2) void conv9x9_c256to3_bf16_kernel(F9Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long f0 = 0, f1 = 0, f2 = 0, f3 = 0, f4 = 0, f5 = 0, f6 = 0, fs_shift = 0, fs_wait = 0, fs_dma = 0, fs_mfma = 0, fs_bar = 0, fs_out = 0, f_rows = 0;
    (void)f6, (void)fs_shift, (void)f0, (void)f1, (void)f2, (void)f3, (void)f4, (void)f5, (void)fs_wait, (void)fs_dma, (void)fs_mfma, (void)fs_bar, (void)fs_out, (void)f_rows;
#ifdef VCG_F9_STAMPS
    const unsigned long long k_c0 = __builtin_amdgcn_s_memtime(), k_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int c = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave = input-channel chunk
    unsigned char* rowbuf = smem + c * 2 * F_ROWB;
    float* part = (float*)(smem + 4 * 2 * F_ROWB);                    // [2][4][3][64]

    // weights: 36 fragments of 16 B per lane
    bf16x8 wf[9][4];
#pragma unroll
    for (int kx = 0; kx < 9; ++kx)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint4 v = p.wfrag[((c * 9 + kx) * 4 + s) * 64 + lane];
            wf[kx][s] = __builtin_bit_cast(bf16x8, v);
        }
    // B-fragment addresses: pixel (r+kx) of the slice, chunk (2s+h) ^ f(pixel);  addr = T[kx] ^ (s << 5)
    int T[9];
#pragma unroll
    for (int kx = 0; kx < 9; ++kx) {
        const int pos = r + kx;
        T[kx] = (pos * 128) | ((((pos >> 1) & 7) ^ hh) << 4);
    }
    // DMA slots of this lane: slot = k*64 + lane -> pixel = slot >> 3, stored chunk = slot & 7 holds channel chunk
    // (slot & 7) ^ f(pixel)
    // (recomputed per row from three lane constants: 18 more live VGPRs would spill)
    const int l3 = lane >> 3, l4 = lane >> 4, l7 = lane & 7;
    const float bias = (p.bias && tid < 192) ? p.bias[tid >> 6] : 0.f;
    const unsigned char* zeros = (const unsigned char*)(p.wfrag + F_NFRAG);      // padding pixels are fetched from here
    // BUF: byte offset of this lane's slot of piece k relative to the piece's first pixel: pixel l3, source chunk of parity k & 1
    unsigned lc[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) lc[par] = (unsigned)(l3 * 512 + (c * 8 + (l7 ^ ((4 * par + l4) & 7))) * 16);
    const long img_bytes = (long)p.h * p.w_ * 512;

    for (int item = blockIdx.x; item < p.total; item += gridDim.x) {
        const int seg = item % p.segs, i2 = item / p.segs, strip = i2 % p.strips, img = i2 / p.strips;
        const int x0 = strip * 64, y0 = seg * p.sh, y1 = min(y0 + p.sh, p.h);
        const vcg_rsrc rs = make_rsrc(p.x + img * img_bytes, (unsigned long)img_bytes);

        auto dma = [&](int yi, int buf) {
            if (BUF) {
                // offset = (row, first pixel of the strip's halo) + piece + lane constant, in 32-bit wrap-around arithmetic: a row above the
                // image lands just below 4 GiB, a row below it just past the image -- both outside the descriptor (host guard)
                const unsigned row_off = (unsigned)(yi * p.w_ + x0 - 4) * 512u;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const int gx = x0 - 4 + k * 8 + l3;
                    unsigned off = row_off + (unsigned)(k * 4096) + lc[k & 1];
                    asm volatile("" : "+v"(off));                    // a select, not a branch around the arithmetic
                    off = (unsigned)gx < (unsigned)p.w_ ? off : VCG_OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (void __attribute__((address_space(3)))*)(rowbuf + buf * F_ROWB + k * 1024), 16, off, 0, 0, 0);
                }
                return;
            }
            const bool rowok = (unsigned)yi < (unsigned)p.h;
            const unsigned char* rowp = p.x + ((long)(img * p.h + (rowok ? yi : 0)) * p.w_) * 512;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int gx = x0 - 4 + k * 8 + l3;
                const bool ok = rowok && (unsigned)gx < (unsigned)p.w_;
                const int dsrc = (c * 8 + (l7 ^ ((4 * k + l4) & 7))) * 16;
                const unsigned char* src = ok ? rowp + (long)gx * 512 + dsrc : zeros;
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                                 (void __attribute__((address_space(3)))*)(rowbuf + buf * F_ROWB + k * 1024), 16, 0, 0);
            }
        };

        f32x16 acc[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[tt][e] = 0.f;

        dma(y0 - 4, 0);
        for (int yi = y0 - 4; yi < y1 + 4; ++yi) {
            const int buf = (yi - (y0 - 4)) & 1;
            F9_STAMP(f0);
            __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): this row's slice has landed in LDS
            asm volatile("" ::: "memory");
            F9_STAMP(f1);
            if (yi + 1 < y1 + 4) dma(yi + 1, buf ^ 1);
            const unsigned char* xb = rowbuf + buf * F_ROWB;
            F9_STAMP(f6);

            // shift the partial sums by one ky group and use them as the C operand.  Register by register, carrying the previous group's
            // upper halves: three temporaries instead of 32 (with all 16 swaps first the kernel spilled, and every scratch reload behind
            // an LDS-DMA costs a vmcnt(0))
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                float ph[3] = {0.f, 0.f, 0.f};                   // upper halves of the previous group (group 2jq-1 -> 2jq)
#pragma unroll
                for (int jq = 0; jq < 4; ++jq)
#pragma unroll
                    for (int sl = 0; sl < 3; ++sl) {
                        float a = acc[tt][4 * jq + sl], b = a;
                        swap32(a, b);                            // a = (lower, lower), b = (upper, upper)
                        acc[tt][4 * jq + sl] = hh ? a : ph[sl];  // group 2jq -> 2jq+1
                        ph[sl] = b;
                    }
                acc[tt][3] = hh ? ph[1] : ph[0];                 // rows 3 / 7  <- group 7, co 0 / 1
                acc[tt][7] = hh ? 0.f : ph[2];                   // row 11      <- group 7, co 2
                acc[tt][11] = 0.f;
                acc[tt][15] = 0.f;
            }

            F9_STAMP(f2);
            bf16x8 fb[2][2];
            // an opaque zero in every fragment address: without it hipcc hoists all 36 (kx, s) addresses out of the row loop and keeps them
            // live in 36 registers -- this kernel has 9 to give (144 of its 256 hold the weights) and spilled the rest
            int opq = 0;
            asm volatile("" : "+v"(opq));
            auto frag = [&](auto ic) {
                constexpr int i = decltype(ic)::value, kx = i >> 2, s = i & 3, bq = i & 1;
                const unsigned char* a = xb + ((T[kx] + opq) ^ (s << 5));
                fb[bq][0] = *(const bf16x8*)(a);
                fb[bq][1] = *(const bf16x8*)(a + 32 * 128);
            };
            frag(std::integral_constant<int, 0>{});
            static_for<36>([&](auto ic) {
                constexpr int i = decltype(ic)::value, kx = i >> 2, s = i & 3, cur = i & 1;
                if constexpr (i + 1 < 36) frag(std::integral_constant<int, i + 1>{});
                __builtin_amdgcn_sched_barrier(0);
                acc[0] = mfma_bf16(wf[kx][s], fb[cur][0], acc[0]);
                acc[1] = mfma_bf16(wf[kx][s], fb[cur][1], acc[1]);
                __builtin_amdgcn_sched_barrier(0);
            });

            F9_STAMP(f3);
            // finished output row yo = yi - 4: this wave's partial (its 64 input channels) -> LDS
            const int yo = yi - 4, slot = yo & 1;
            if (yo >= y0) {
                float* pp = part + ((slot * 4 + c) * 3) * 64;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    // rows 3 / 7 / 11 = co 0 (lower half) / 1 (upper half) / 2 (lower half): every lane stores register 3 to its co's row, the
                    // lower half register 7 as well (written as selects: hipcc turned the two-sided branch into a 16-way register select)
                    pp[hh * 64 + tt * 32 + r] = acc[tt][3];
                    if (hh == 0) pp[2 * 64 + tt * 32 + r] = acc[tt][7];
                }
            }
            // the partial sums are ordinary LDS stores: wait for them and meet.  NOT lds_barrier(): its fence makes hipcc drain vmcnt(0),
            // i.e. wait here for the NEXT row's slice, which nothing reads before the wait at the top of the next iteration
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            F9_STAMP(f4);
            if (yo >= y0 && tid < 192) {
                const int co = tid >> 6, col = tid & 63;
                const float* q = part + slot * 4 * 3 * 64 + co * 64 + col;
                float v = ((q[0] + q[3 * 64]) + (q[2 * 3 * 64] + q[3 * 3 * 64])) + bias;
                if (p.tanh_act) v = fast_tanh(v);
                if (x0 + col < p.w_) p.y[((long)(img * 3 + co) * p.h + yo) * p.w_ + x0 + col] = v;
            }
            F9_STAMP(f5);
            F9_ADD(fs_wait, f0, f1); F9_ADD(fs_dma, f1, f6); F9_ADD(fs_shift, f6, f2); F9_ADD(fs_mfma, f2, f3); F9_ADD(fs_bar, f3, f4); F9_ADD(fs_out, f4, f5);
            F9_ADD(f_rows, 0ull, 1ull);
        }
        lds_barrier();       // the next item's first partial slot / row buffers are free
    }
#ifdef VCG_F9_STAMPS
    if (lane == 0 && blockIdx.x < 512) {
        unsigned long long* o = vcg_f9_stamp_sums + (blockIdx.x * 4 + c) * 8;
        o[0] = fs_wait, o[1] = fs_dma, o[2] = fs_mfma, o[3] = fs_bar, o[4] = fs_out, o[5] = f_rows;
        o[6] = __builtin_amdgcn_s_memtime() - k_c0, o[7] = fs_shift;
    }
#endif
}

__global__ void pack_final9x9_kernel(const float* __restrict__ w, uint4* __restrict__ out) {
    // w: Keras (9,9,256,3) -> out[chunk][kx][s][lane] = 8 bf16: A[row = lane&31][k = 8*(lane>>5) + j] of k-step (kx, s)
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= F_NFRAG + 4) return;
    if (idx >= F_NFRAG) {
        out[idx] = make_uint4(0, 0, 0, 0);
        return;
    }
    const int lane = idx & 63, s = (idx >> 6) & 3, kx = (idx >> 8) % 9, c = idx / (9 * 256);
    const int row = lane & 31, h = lane >> 5, g = row >> 2, sl = row & 3;
    int ky = -1, co = 0;
    if (sl < 3) { ky = g; co = sl; }
    else if (g < 3) { ky = 8; co = g; }
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = c * 64 + 16 * s + 8 * h + j;
        v[j] = (__bf16)(ky >= 0 ? w[((ky * 9 + kx) * 256 + ci) * 3 + co] : 0.f);
    }
    out[idx] = __builtin_bit_cast(uint4, v);
}

}  // namespace

template <int KH, int NG, int S>
static int launch_conv3ch(I9Params p, hipStream_t stream) {
    using C = I3Cfg<KH, NG, S>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_c3to64_bf16_kernel<KH, NG, S>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int nblk = p.cout / 64;
    int per = 1024 / nblk;
    if (per > p.total) per = p.total;
    conv_c3to64_bf16_kernel<KH, NG, S><<<per * nblk, NT, C::LDS, stream>>>(p);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

static int vcg_xcd_group() {
    // measured neutral (profiles/r03_xcd_group_ab.txt): off unless VCG_XCD_GROUP=1
    static const int v = [] { const char* e = getenv("VCG_XCD_GROUP"); return e != nullptr && e[0] == '1' ? 1 : 0; }();
    return v;
}

extern "C" {

int vcg_pack_conv_kernel_bf16(const void* w, int32_t taps, int32_t a, int32_t b, int32_t transpose, int32_t flip, void* out,
                              hipStream_t stream) {
    VCG_CHECK_PTR(w);
    VCG_CHECK_PTR(out);
    if (taps <= 0 || a <= 0 || b <= 0) return VCG_E_SHAPE;
    const long total = (long)taps * a * b;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    pack_kernel_bf16<<<blocks, 256, 0, stream>>>((const float*)w, (__bf16*)out, taps, a, b, transpose, flip);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_pack_conv3x3_c64_bf16_batch(const void* const* w_host_array, int32_t count, void* out, hipStream_t stream) {
    VCG_CHECK_PTR(w_host_array);
    VCG_CHECK_PTR(out);
    if (count <= 0 || count > PACK_BATCH_MAX) return VCG_E_SHAPE;
    PackBatch pb;
    for (int i = 0; i < count; ++i) {
        VCG_CHECK_PTR(w_host_array[i]);
        pb.w[i] = (const float*)w_host_array[i];
    }
    pack3x3_c64_batch_kernel<<<dim3(36, count), 256, 0, stream>>>(pb, (__bf16*)out);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_f32_nchw_to_bf16_nhwc(const void* x, void* y, int32_t n, int32_t c, int32_t h, int32_t w, hipStream_t stream) {
    VCG_CHECK_PTR(x);
    VCG_CHECK_PTR(y);
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0) return VCG_E_SHAPE;
    dim3 grid(ceil_div(h * w, 64), n);
    f32_nchw_to_bf16_nhwc_kernel<<<grid, 256, 0, stream>>>((const float*)x, (__bf16*)y, n, c, h * w);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_bf16_nhwc_to_f32_nchw(const void* x, void* y, int32_t n, int32_t c, int32_t h, int32_t w, hipStream_t stream) {
    VCG_CHECK_PTR(x);
    VCG_CHECK_PTR(y);
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0) return VCG_E_SHAPE;
    dim3 grid(ceil_div(h * w, 64), n);
    bf16_nhwc_to_f32_nchw_kernel<<<grid, 256, 0, stream>>>((const __bf16*)x, (float*)y, n, c, h * w);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

// the v2 kernel finds its halo rows through 32-bit byte offsets against the image's buffer descriptor: the image plus the halo rows
// below it (and a row of slack for the wrapped offsets of the row above it) must stay below 4 GiB, or a halo offset wraps into the image
static bool v2_image_fits(int h, int w) { return ((long)h + V2_HR + 2) * w * 128 + 4096 <= 0xFFFFFFE0l; }

int vcg_conv2d_bf16_stats_records(const vcg_conv_desc* d, int32_t stats_mode) {
    if (d == nullptr) return VCG_E_NULL;
    if (d->n <= 0 || d->h <= 0 || d->w <= 0) return VCG_E_SHAPE;
    if (stats_mode != VCG_STATS_BATCH && stats_mode != VCG_STATS_INSTANCE) return VCG_E_UNSUPPORTED;
    if (!(d->cin == 64 && d->cout == 64 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad_top == 1 && d->pad_left == 1)) return VCG_E_UNSUPPORTED;
    if (getenv("VCG_CONV3X3_V1") != nullptr || !v2_image_fits(d->h, d->w)) return VCG_E_UNSUPPORTED;
    const long tiles_img = (long)ceil_div(d->w, V2_TC) * ceil_div(d->h, V2_TR), total = tiles_img * d->n;
    if (stats_mode == VCG_STATS_INSTANCE) return (int)(2 * tiles_img);
    return (int)(2 * (total < 256 ? total : 256));
}

int vcg_conv2d_bf16_fwd(const vcg_conv_desc* d, const void* x, const void* w_packed, void* y, const vcg_epilogue_bf16* ep,
                        hipStream_t stream) {
    VCG_CHECK_PTR(d);
    VCG_CHECK_PTR(x);
    VCG_CHECK_PTR(w_packed);
    VCG_CHECK_PTR(y);
    if (d->n <= 0 || d->h <= 0 || d->w <= 0) return VCG_E_SHAPE;
    if (d->oh != d->h || d->ow != d->w) return VCG_E_SHAPE;
    const int act = ep ? ep->act : VCG_ACT_NONE;
    if (act == VCG_ACT_TANH) return VCG_E_UNSUPPORTED;
    if (act == VCG_ACT_PRELU && (!ep || !ep->prelu_alpha)) return VCG_E_NULL;
    if (d->cin == 64 && d->cout == 64 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad_top == 1 && d->pad_left == 1) {
        C3Params p;
        p.x = (const uint4*)x;
        p.w = (const uint4*)w_packed;
        p.y = (__bf16*)y;
        p.scale = ep ? (const float*)ep->scale : nullptr;
        p.shift = ep ? (const float*)ep->shift : nullptr;
        p.alpha = (ep && act == VCG_ACT_PRELU) ? (const float*)ep->prelu_alpha : nullptr;
        p.res = ep ? (const __bf16*)ep->residual : nullptr;
        p.n = d->n;
        p.h = d->h;
        p.w_ = d->w;
        p.tiles_x = ceil_div(d->w, TC);
        p.tiles_y = ceil_div(d->h, TR);
        p.total = p.n * p.tiles_x * p.tiles_y;
        p.act = act;
        p.act_alpha = ep ? ep->act_alpha : 0.f;
        p.stats = (ep && ep->stats_mode != VCG_STATS_NONE) ? (float*)ep->stats : nullptr;
        p.stats_per_tile = ep && ep->stats_mode == VCG_STATS_INSTANCE;
        if (ep && ep->stats_mode != VCG_STATS_NONE) {
            // statistics come out of the v2 kernel's drain: no activation, no residual input (what stands in front of a normalisation)
            if (!ep->stats) return VCG_E_NULL;
            if (vcg_conv2d_bf16_stats_records(d, ep->stats_mode) <= 0 || act != VCG_ACT_NONE || ep->residual) return VCG_E_UNSUPPORTED;
        }
        static bool attr_set = false;
        if (!attr_set) {
            for (auto f : {(const void*)conv3x3_c64_bf16_kernel<false, false, false>, (const void*)conv3x3_c64_bf16_kernel<false, false, true>,
                           (const void*)conv3x3_c64_bf16_kernel<false, true, false>, (const void*)conv3x3_c64_bf16_kernel<false, true, true>,
                           (const void*)conv3x3_c64_bf16_kernel<true, false, false>, (const void*)conv3x3_c64_bf16_kernel<true, false, true>,
                           (const void*)conv3x3_c64_bf16_kernel<true, true, false>, (const void*)conv3x3_c64_bf16_kernel<true, true, true>}) {
                hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, WB + XB + PB);
                if (e != hipSuccess) return (int)e;
            }
            attr_set = true;
        }
        const bool aff = p.scale || p.shift, slope = act != VCG_ACT_NONE, res = p.res != nullptr;
        // v2 (one wave per SIMD, weights in registers) unless there is a residual input or the image is too large for one
        // buffer descriptor; VCG_CONV3X3_V1=1 forces v1 (A/B aid: scripts/gpu_v2_ab.sh)
        static const bool use_v1 = getenv("VCG_CONV3X3_V1") != nullptr;
        if (!use_v1 && !res && v2_image_fits(d->h, d->w)) {
            p.tiles_x = ceil_div(d->w, V2_TC);
            p.tiles_y = ceil_div(d->h, V2_TR);
            p.total = p.n * p.tiles_x * p.tiles_y;
            static bool attr2_set = false;
            if (!attr2_set) {
                for (auto f : {(const void*)conv3x3_c64_bf16_v2_kernel<false, false>, (const void*)conv3x3_c64_bf16_v2_kernel<false, true>,
                               (const void*)conv3x3_c64_bf16_v2_kernel<true, false>, (const void*)conv3x3_c64_bf16_v2_kernel<true, true>,
                               (const void*)conv3x3_c64_bf16_v2_kernel<true, false, true>}) {
                    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, V2_LDS);
                    if (e != hipSuccess) return (int)e;
                }
                attr2_set = true;
            }
            const int grid2 = p.total < 256 ? p.total : 256;
#define VCG_C3V2_LAUNCH(A, S) conv3x3_c64_bf16_v2_kernel<A, S><<<grid2, V2_NT, V2_LDS, stream>>>(p)
            if (p.stats) {
                conv3x3_c64_bf16_v2_kernel<true, false, true><<<grid2, V2_NT, V2_LDS, stream>>>(p);      // a null scale / shift reads as 1 / 0
            } else if (aff) {
                if (slope) VCG_C3V2_LAUNCH(true, true); else VCG_C3V2_LAUNCH(true, false);
            } else {
                if (slope) VCG_C3V2_LAUNCH(false, true); else VCG_C3V2_LAUNCH(false, false);
            }
#undef VCG_C3V2_LAUNCH
            VCG_LAUNCH_CHECK();
            return VCG_OK;
        }
        const int grid = p.total < 256 ? p.total : 256;
#define VCG_C3_LAUNCH(A, S, R) conv3x3_c64_bf16_kernel<A, S, R><<<grid, NT, WB + XB + PB, stream>>>(p)
        if (aff) {
            if (slope) { if (res) VCG_C3_LAUNCH(true, true, true); else VCG_C3_LAUNCH(true, true, false); }
            else { if (res) VCG_C3_LAUNCH(true, false, true); else VCG_C3_LAUNCH(true, false, false); }
        } else {
            if (slope) { if (res) VCG_C3_LAUNCH(false, true, true); else VCG_C3_LAUNCH(false, true, false); }
            else { if (res) VCG_C3_LAUNCH(false, false, true); else VCG_C3_LAUNCH(false, false, false); }
        }
#undef VCG_C3_LAUNCH
        VCG_LAUNCH_CHECK();
        return VCG_OK;
    }
    return VCG_E_UNSUPPORTED;
}

int vcg_conv_transpose2d_bf16_fwd(const vcg_conv_desc* d, const void* x, const void* w_packed, void* y, const vcg_epilogue_bf16* ep,
                                  hipStream_t stream) {
    VCG_CHECK_PTR(d);
    VCG_CHECK_PTR(x);
    VCG_CHECK_PTR(w_packed);
    VCG_CHECK_PTR(y);
    if (d->n <= 0 || d->h <= 0 || d->w <= 0) return VCG_E_SHAPE;
    if (d->oh != 2 * d->h || d->ow != 2 * d->w || d->stride != 2) return VCG_E_SHAPE;
    const int act = ep ? ep->act : VCG_ACT_NONE;
    if (ep && (ep->scale || ep->residual || ep->stats_mode != VCG_STATS_NONE)) return VCG_E_UNSUPPORTED;
    if (act != VCG_ACT_NONE && act != VCG_ACT_LRELU) return VCG_E_UNSUPPORTED;
    const int nblk = d->cout / 64;
    if (d->cin != 64 || d->cout % 64 != 0 || (nblk & (nblk - 1)) != 0 || nblk > 8 || d->kh != 3 || d->kw != 3) return VCG_E_UNSUPPORTED;
    CTParams p;
    p.x = (const uint4*)x;
    p.w = (const uint4*)w_packed;
    p.shift = ep ? (const float*)ep->shift : nullptr;
    p.y = (__bf16*)y;
    p.n = d->n;
    p.h = d->h;
    p.w_ = d->w;
    p.cout = d->cout;
    p.tiles_x = ceil_div(d->w, TC);
    p.tiles_y = ceil_div(d->h, TR);
    p.total = p.n * p.tiles_x * p.tiles_y;
    p.slope = act == VCG_ACT_LRELU ? ep->act_alpha : 1.f;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)convt3x3_c64_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WB + TXB + 256);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    int per = 256 / nblk;                       // workgroups per output-channel block
    if (per > p.total) per = p.total;
    p.xcd_group = vcg_xcd_group();
    convt3x3_c64_bf16_kernel<<<per * nblk, NT, WB + TXB + 256, stream>>>(p);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_pack_final9x9_bf16(const void* w, void* out, hipStream_t stream) {
    VCG_CHECK_PTR(w);
    VCG_CHECK_PTR(out);
    pack_final9x9_kernel<<<(F_NFRAG + 4 + 255) / 256, 256, 0, stream>>>((const float*)w, (uint4*)out);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_conv9x9_to3_bf16_fwd(const vcg_conv_desc* d, const void* x, const void* wfrag, const void* bias, int32_t tanh_act, void* y,
                             hipStream_t stream) {
    VCG_CHECK_PTR(d);
    VCG_CHECK_PTR(x);
    VCG_CHECK_PTR(wfrag);
    VCG_CHECK_PTR(y);
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->oh != d->h || d->ow != d->w) return VCG_E_SHAPE;
    if (d->cin != 256 || d->cout != 3 || d->kh != 9 || d->kw != 9 || d->stride != 1 || d->pad_top != 4 || d->pad_left != 4) return VCG_E_UNSUPPORTED;
    F9Params p;
    p.x = (const unsigned char*)x;
    p.wfrag = (const uint4*)wfrag;
    p.bias = (const float*)bias;
    p.y = (float*)y;
    p.n = d->n;
    p.h = d->h;
    p.w_ = d->w;
    p.strips = ceil_div(d->w, 64);
    static const int max_grid = getenv("VCG_F9_GRID") ? atoi(getenv("VCG_F9_GRID")) : 512;      // tuning aid (scripts/micro/f9_stamps.py)
    // segments per column strip: the split that minimises the rows the busiest workgroup marches through -- rounds of items per
    // workgroup x (segment height + 8 recomputed halo rows).  (Round 2 halved the height until the items filled the grid twice: 25 % halo
    // rows at C3's shape, and at C4's 1080 items on 512 workgroups = a third round for 56 of them.)
    {
        long best = -1;
        const int colstrips = p.n * p.strips;
        for (int segs = 1; segs <= ceil_div(d->h, 16); ++segs) {
            const int sh = ceil_div(d->h, segs);
            if (ceil_div(d->h, sh) != segs) continue;                             // (the same height reached with fewer segments)
            const long items = (long)colstrips * segs, rounds = (items + max_grid - 1) / max_grid, cost = rounds * (sh + 8);
            if (best < 0 || cost < best) { best = cost; p.sh = sh; p.segs = segs; }
        }
    }
    p.total = p.n * p.strips * p.segs;
    p.tanh_act = tanh_act;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv9x9_c256to3_bf16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute((const void*)conv9x9_c256to3_bf16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int grid = p.total < max_grid ? p.total : max_grid;
    // the descriptor form needs the image, four rows above and four below it inside 32-bit offsets
    const bool buf = ((long)d->h + 8) * d->w * 512 + 65536 <= 0xFFFFFFE0l && getenv("VCG_F9_PTR") == nullptr;
    if (buf) conv9x9_c256to3_bf16_kernel<true><<<grid, 256, F_LDS, stream>>>(p);
    else conv9x9_c256to3_bf16_kernel<false><<<grid, 256, F_LDS, stream>>>(p);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_pack_first9x9_bf16(const void* w, void* out, hipStream_t stream) {
    return vcg_pack_conv9x9_3ch_bf16(w, 64, 0, out, stream);
}

int vcg_pack_conv9x9_3ch_bf16(const void* w, int32_t cout, int32_t dgrad, void* out, hipStream_t stream) {
    VCG_CHECK_PTR(w);
    VCG_CHECK_PTR(out);
    if (cout <= 0 || cout % 64 != 0) return VCG_E_SHAPE;
    const int total = (cout >> 6) * 27 * 64 * 2;
    pack_first9x9_kernel<<<(total + 255) / 256, 256, 0, stream>>>((const float*)w, (uint4*)out, cout, dgrad, 9, 9, 3);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

// Conv2D(cout, k, strides s) on 3 input channels, fp32 NCHW frames -> bf16 NHWC (+bias, LeakyReLU): the critics' block 1
// (model.py:839 3x3 stride 1; PatchGAN 4x4 stride 2).  wfrag: vcg_conv3ch_bf16_wfrag_bytes(kh, cout) bytes from vcg_pack_conv3ch_bf16.
static bool conv3ch_supported(int kh, int kw, int stride) { return (kh == 4 && kw == 4 && stride == 2) || (kh == 3 && kw == 3 && stride == 1); }

size_t vcg_conv3ch_bf16_wfrag_bytes(int32_t kh, int32_t kw, int32_t cout) {
    if (kh <= 0 || kw <= 0 || kw > 4 || cout <= 0 || cout % 64) return 0;
    return (size_t)(cout >> 6) * kh * 64 * 2 * 16;
}

int vcg_pack_conv3ch_bf16(const void* w, int32_t kh, int32_t kw, int32_t cout, void* out, hipStream_t stream) {
    VCG_CHECK_PTR(w);
    VCG_CHECK_PTR(out);
    if (cout <= 0 || cout % 64 != 0 || kh <= 0 || kw <= 0) return VCG_E_SHAPE;
    if (kw > 4) return VCG_E_UNSUPPORTED;
    const int total = (cout >> 6) * kh * 64 * 2;
    pack_first9x9_kernel<<<(total + 255) / 256, 256, 0, stream>>>((const float*)w, (uint4*)out, cout, 0, kh, kw, 1);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

static int conv9x9_3ch_wgs(const vcg_conv_desc* d, int cout) {          // workgroups per output-channel block
    const int nblk = cout / 64;
    const long total = (long)d->n * ceil_div(d->w, TC) * ceil_div(d->h, TR);
    const int per = 512 / nblk;                                   // 62 KiB of LDS: two workgroups per CU
    return (int)(per > total ? total : per);
}

static int launch_conv9x9_3ch(const vcg_conv_desc* d, int cout, const void* x, const void* wfrag, const void* bias, const void* prelu_alpha,
                              const void* mask, float mask_slope, void* y, float* chsum, hipStream_t stream, void* z = nullptr) {
    const int nblk = cout / 64;
    if (cout % 64 != 0 || nblk < 1 || nblk > 8 || (nblk & (nblk - 1))) return VCG_E_UNSUPPORTED;
    I9Params p;
    p.chsum = chsum;
    p.x = (const float*)x;
    p.w = (const uint4*)wfrag;
    p.bias = (const float*)bias;
    p.alpha = (const float*)prelu_alpha;
    p.y = (__bf16*)y;
    p.z = (__bf16*)z;
    p.mask = (const __bf16*)mask;
    p.mask_slope = mask_slope;
    p.n = d->n;
    p.h = d->h;
    p.w_ = d->w;
    p.cout = cout;
    p.tiles_x = ceil_div(d->w, TC);
    p.tiles_y = ceil_div(d->h, TR);
    p.total = p.n * p.tiles_x * p.tiles_y;
    p.oh = d->h; p.ow = d->w; p.pad_top = 4; p.pad_left = 4; p.slope = 1.f;
    p.xcd_group = vcg_xcd_group();
    using C = I3Cfg<9, 3, 1>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_c3to64_bf16_kernel<9, 3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int per = conv9x9_3ch_wgs(d, cout);
    conv_c3to64_bf16_kernel<9, 3, 1><<<per * nblk, NT, C::LDS, stream>>>(p);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_conv3ch_bf16_fwd(const vcg_conv_desc* d, const void* x, const void* wfrag, const void* bias, float lrelu_slope, void* y, hipStream_t stream) {
    VCG_CHECK_PTR(d);
    VCG_CHECK_PTR(x);
    VCG_CHECK_PTR(wfrag);
    VCG_CHECK_PTR(y);
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->oh <= 0 || d->ow <= 0 || d->pad_top < 0 || d->pad_left < 0) return VCG_E_SHAPE;
    if (d->cin != 3 || !conv3ch_supported(d->kh, d->kw, d->stride)) return VCG_E_UNSUPPORTED;
    const int nblk = d->cout / 64;
    if (d->cout % 64 != 0 || nblk < 1 || nblk > 8 || (nblk & (nblk - 1))) return VCG_E_UNSUPPORTED;
    I9Params p;
    p.chsum = nullptr;
    p.x = (const float*)x;
    p.w = (const uint4*)wfrag;
    p.bias = (const float*)bias;
    p.alpha = nullptr;
    p.xcd_group = vcg_xcd_group();
    p.slope = lrelu_slope;
    p.y = (__bf16*)y;
    p.z = nullptr;
    p.mask = nullptr;
    p.mask_slope = 0.f;
    p.n = d->n; p.h = d->h; p.w_ = d->w; p.cout = d->cout;
    p.oh = d->oh; p.ow = d->ow; p.pad_top = d->pad_top; p.pad_left = d->pad_left;
    p.tiles_x = ceil_div(d->ow, TC);
    p.tiles_y = ceil_div(d->oh, TR);
    p.total = p.n * p.tiles_x * p.tiles_y;
    if (d->kh == 4) return launch_conv3ch<4, 1, 2>(p, stream);
    return launch_conv3ch<3, 1, 1>(p, stream);
}

int vcg_conv9x9_from3_bf16_fwd(const vcg_conv_desc* d, const void* x, const void* wfrag, const void* bias, const void* prelu_alpha,
                               void* y, hipStream_t stream) {
    VCG_CHECK_PTR(d);
    VCG_CHECK_PTR(x);
    VCG_CHECK_PTR(wfrag);
    VCG_CHECK_PTR(y);
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->oh != d->h || d->ow != d->w) return VCG_E_SHAPE;
    if (d->cin != 3 || d->kh != 9 || d->kw != 9 || d->stride != 1 || d->pad_top != 4 || d->pad_left != 4) return VCG_E_UNSUPPORTED;
    return launch_conv9x9_3ch(d, d->cout, x, wfrag, bias, prelu_alpha, nullptr, 0.f, y, nullptr, stream);
}

static int prelu_bwd_gridx(int hw) { const int t = ceil_div(hw, 64); return t < 128 ? t : 128; }

// the same with the result left in the bf16 NHWC layout (for the bf16 weight gradient of the 3-channel convolution, bf16_wgrad3.hip): no
// transposition, a thread owns 8 channels of a pixel; same record layout
__global__ __launch_bounds__(256) void prelu_bwd_bf16_nhwc_kernel(const __bf16* __restrict__ d1, const __bf16* __restrict__ d2, const __bf16* __restrict__ z,
                                                                  const float* __restrict__ alpha, __bf16* __restrict__ dz, float* __restrict__ rec, int c,
                                                                  int hw, int tiles) {
    __shared__ float red[32][64];
    const int img = blockIdx.y, tid = threadIdx.x;
    const int ch = (tid & 7) * 8;
    for (int c0 = 0; c0 < c; c0 += 64) {
        float da[8], al[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            da[j] = 0.f;
            al[j] = c0 + ch + j < c ? alpha[c0 + ch + j] : 0.f;
        }
        for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
            const int p0 = t * 64;
#pragma unroll
            for (int e = tid; e < 64 * 8; e += 256) {
                const int pp = e >> 3;
                if (c0 + ch < c && p0 + pp < hw) {
                    const long o = ((long)img * hw + p0 + pp) * c + c0 + ch;
                    const bf16x8 a = *(const bf16x8*)(d1 + o), zz = *(const bf16x8*)(z + o);
                    bf16x8 b;
#pragma unroll
                    for (int j = 0; j < 8; ++j) b[j] = (__bf16)0.f;
                    if (d2) b = *(const bf16x8*)(d2 + o);
                    bf16x8 g;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float d = (float)a[j] + (float)b[j], zf = (float)zz[j];
                        g[j] = (__bf16)(zf >= 0.f ? d : d * al[j]);
                        da[j] += zf >= 0.f ? 0.f : d * zf;
                    }
                    *(bf16x8*)(dz + o) = g;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) red[tid >> 3][ch + j] = da[j];
        __syncthreads();
        if (tid < 64 && c0 + tid < c) {
            float s = 0.f;
            for (int k = 0; k < 32; ++k) s += red[k][tid];
            rec[((long)img * gridDim.x + blockIdx.x) * c + c0 + tid] = s;
        }
        __syncthreads();
    }
}

int vcg_prelu_bwd_nhwc_bf16_records(int n, int hw) {
    if (n <= 0 || hw <= 0) return VCG_E_SHAPE;
    return n * prelu_bwd_gridx(hw);
}

int vcg_prelu_bwd_nhwc_bf16(const void* d1, const void* d2, const void* z, const float* prelu_alpha, int n, int c, int hw, float* dz_nchw,
                            float* records, hipStream_t stream) {
    VCG_CHECK_PTR(d1); VCG_CHECK_PTR(z); VCG_CHECK_PTR(prelu_alpha); VCG_CHECK_PTR(dz_nchw); VCG_CHECK_PTR(records);
    if (n <= 0 || c <= 0 || hw <= 0 || n > 65535) return VCG_E_SHAPE;
    if (c % 8) return VCG_E_UNSUPPORTED;
    prelu_bwd_bf16_to_f32_nchw_kernel<<<dim3(prelu_bwd_gridx(hw), n), 256, 0, stream>>>((const __bf16*)d1, (const __bf16*)d2, (const __bf16*)z, prelu_alpha,
                                                                                     dz_nchw, records, c, hw, ceil_div(hw, 64));
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_prelu_bwd_nhwc_bf16_to_bf16(const void* d1, const void* d2, const void* z, const float* prelu_alpha, int n, int c, int hw, void* dz_nhwc,
                                    float* records, hipStream_t stream) {
    VCG_CHECK_PTR(d1); VCG_CHECK_PTR(z); VCG_CHECK_PTR(prelu_alpha); VCG_CHECK_PTR(dz_nhwc); VCG_CHECK_PTR(records);
    if (n <= 0 || c <= 0 || hw <= 0 || n > 65535) return VCG_E_SHAPE;
    if (c % 8) return VCG_E_UNSUPPORTED;
    prelu_bwd_bf16_nhwc_kernel<<<dim3(prelu_bwd_gridx(hw), n), 256, 0, stream>>>((const __bf16*)d1, (const __bf16*)d2, (const __bf16*)z, prelu_alpha,
                                                                              (__bf16*)dz_nhwc, records, c, hw, ceil_div(hw, 64));
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_conv9x9_from3_bf16_fwd_train(const vcg_conv_desc* d, const void* x, const void* wfrag, const void* bias, const void* prelu_alpha,
                                     void* y, void* z, hipStream_t stream) {
    VCG_CHECK_PTR(d);
    VCG_CHECK_PTR(x);
    VCG_CHECK_PTR(wfrag);
    VCG_CHECK_PTR(y);
    VCG_CHECK_PTR(z);
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->oh != d->h || d->ow != d->w) return VCG_E_SHAPE;
    if (d->cin != 3 || d->kh != 9 || d->kw != 9 || d->stride != 1 || d->pad_top != 4 || d->pad_left != 4) return VCG_E_UNSUPPORTED;
    return launch_conv9x9_3ch(d, d->cout, x, wfrag, bias, prelu_alpha, nullptr, 0.f, y, nullptr, stream, z);
}

int vcg_conv9x9_to3_bf16_dgrad(const vcg_conv_desc* d, const void* dy, const void* wfrag, const void* y_prev, float lrelu_slope, void* dx,
                               hipStream_t stream) {
    VCG_CHECK_PTR(d);
    VCG_CHECK_PTR(dy);
    VCG_CHECK_PTR(wfrag);
    VCG_CHECK_PTR(dx);
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->oh != d->h || d->ow != d->w) return VCG_E_SHAPE;
    if (d->cout != 3 || d->kh != 9 || d->kw != 9 || d->stride != 1 || d->pad_top != 4 || d->pad_left != 4) return VCG_E_UNSUPPORTED;
    return launch_conv9x9_3ch(d, d->cin, dy, wfrag, nullptr, nullptr, y_prev, lrelu_slope, dx, nullptr, stream);
}

// the same, also leaving per-channel sums of the stored dx as records [vcg_conv9x9_to3_bf16_dgrad_chsum_records(d)][cin] -- the bias gradient of
// the layer that produced the convolution's input (upsampling_block's Conv2DTranspose, model.py:72) without another pass over dx; add the
// records up with vcg_sum_records
int vcg_conv9x9_to3_bf16_dgrad_chsum_records(const vcg_conv_desc* d) {
    if (d == nullptr) return VCG_E_NULL;
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->cin <= 0 || d->cin % 64) return VCG_E_SHAPE;
    return conv9x9_3ch_wgs(d, d->cin) * NCW;
}

int vcg_conv9x9_to3_bf16_dgrad_chsum(const vcg_conv_desc* d, const void* dy, const void* wfrag, const void* y_prev, float lrelu_slope, void* dx,
                                     float* records, hipStream_t stream) {
    VCG_CHECK_PTR(d);
    VCG_CHECK_PTR(dy);
    VCG_CHECK_PTR(wfrag);
    VCG_CHECK_PTR(dx);
    VCG_CHECK_PTR(records);
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->oh != d->h || d->ow != d->w) return VCG_E_SHAPE;
    if (d->cout != 3 || d->kh != 9 || d->kw != 9 || d->stride != 1 || d->pad_top != 4 || d->pad_left != 4) return VCG_E_UNSUPPORTED;
    return launch_conv9x9_3ch(d, d->cin, dy, wfrag, nullptr, nullptr, y_prev, lrelu_slope, dx, records, stream);
}

}  // extern "C"

#ifdef VCG_F9_STAMPS
extern "C" int vcg_debug_f9_stamps(unsigned long long* host_out) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(vcg_f9_stamp_sums), sizeof(unsigned long long) * 512 * 4 * 8);
}
#endif
#ifdef VCG_CT_STAMPS
extern "C" int vcg_debug_ct_stamps(unsigned long long* host_out) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(vcg_ct_stamp_sums), sizeof(unsigned long long) * 256 * 8 * 5);
}
#endif
#ifdef VCG_I9_STAMPS
extern "C" int vcg_debug_i9_stamps(unsigned long long* host_out) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(vcg_i9_stamp_sums), sizeof(unsigned long long) * 512 * 8 * 6);
}
#endif
#ifdef VCG_V2_STAMPS
// diagnostic build only: copy the per-wave sums [256 workgroups][4 waves][phase A, phase B, vmcnt wait, barrier, kernel core clocks, kernel 100-MHz ticks] to the host
extern "C" int vcg_debug_v2_stamps(unsigned long long* host_out) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(vcg_v2_stamp_sums), sizeof(unsigned long long) * 256 * 4 * 6);
}
#endif
