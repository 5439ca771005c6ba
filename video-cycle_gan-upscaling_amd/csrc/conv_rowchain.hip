// fp32 9x9 stride-1 'same' convolution from 256 to <= 3 channels (+bias, tanh): the generator's final/conv
// (upscaling/upscaler/model.py:290-291) at image widths that are multiples of 64.
//
// Three output channels would waste 29 of the 32 MFMA rows, so the rows carry (ky, co):
//   row 4*ky+co (ky < 8)  and  row 4*co+3 (ky = 8);      k = (kx, ci);      columns = 32 consecutive x.
// One pass over ONE input row yi yields, for every ky, that row's contribution to output row yi+4-ky.  A wave marches
// down the image and carries the partial sums of the 9 output rows in flight IN THE ACCUMULATOR: before the next
// input row the accumulator is shifted by one ky-group (4 rows = half a register group: v_permlane32_swap + select
// per register) and handed to the MFMA as its C operand; the group that falls off the end (ky = 7) lands in the spare
// rows 3,7,11 where the ky = 8 products complete it.  After the pass those rows hold the finished output row yi-4:
// no partial tensors, 27 of 32 rows and all 64 columns of every MFMA are useful (the (co,kx)-row kernel in conv_fwd.hip
// uses 27/32 x 56/64), and every input row is read from HBM once per 64-column strip.
//   * the 256 input channels are split over the 8 waves of a workgroup; a wave's 9 x 16 weight values per lane stay in
//     144 VGPRs for the whole launch; the eight partial output rows meet in LDS once per row;
//   * the wave's [32 channels][72 columns] slice of the input row goes HBM -> LDS by global_load_lds (16 B per lane,
//     no VGPRs), double buffered; B operands are conflict-free ds_read_b32 at compile-time offsets;
//   * work item = (image, 64-column strip, segment of output rows); one workgroup per CU, 2 waves per SIMD.
#include "vcg_common.hpp"
#include <utility>

namespace {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void swap32(float& a, float& b) {      // lanes 32-63 of a <-> lanes 0-31 of b
    const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(sw.x);
    b = __uint_as_float(sw.y);
}

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

constexpr int R_NW = 8;                      // waves = input-channel slices
constexpr int R_CI = 32;                     // channels per wave
constexpr int R_PIX = 72;                    // 64 output columns + 4 + 4
constexpr int R_ROWB = R_CI * R_PIX * 4;     // 9216 B
constexpr int R_PART = 2 * R_NW * 3 * 64 * 4;
constexpr int R_LDS = R_NW * 2 * R_ROWB + R_PART;      // 159744 B

struct RowChainParams {
    const float* x;          // fp32 NCHW [n][256][h][w]
    const float* w;          // element (tap, co, ci) at w[tap*ws_t + co*ws_m + ci*ws_k]
    const float* bias;
    float* y;                // fp32 NCHW [n][cout][h][w]
    int n, h, w_, cout, strips, segs, sh, total;
    int ws_t, ws_m, ws_k;
    int act;
};

__global__ __launch_bounds__(R_NW * 64, 1) void conv9x9_rowchain_f32_kernel(RowChainParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int c = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* rowbuf = smem + c * 2 * R_ROWB;
    float* part = (float*)(smem + R_NW * 2 * R_ROWB);          // [2][8][3][64]

    // A operand of k-step (kx, pr): A[row = lane&31][k = h] = W[ky(row)][kx][ci = 32c + 2pr + h][co(row)]
    const int g = r >> 2, sl = r & 3;
    int ky = -1, co = 0;
    if (sl < 3) { ky = g; co = sl; }
    else if (g < 3) { ky = 8; co = g; }
    if (co >= p.cout) { ky = -1; co = 0; }
    float wf[9][16];
#pragma unroll
    for (int kx = 0; kx < 9; ++kx)
#pragma unroll
        for (int pr = 0; pr < 16; ++pr) {
            const int ci = c * R_CI + 2 * pr + hh;
            // (unconditional load at a clamped index, then select: a conditional load would become 144 branches)
            const float v = p.w[(long)(max(ky, 0) * 9 + kx) * p.ws_t + (long)co * p.ws_m + (long)ci * p.ws_k];
            wf[kx][pr] = ky >= 0 ? v : 0.f;
        }
    const int bbase = hh * (R_PIX * 4) + r * 4;                // B[k = h][col r]: channel 2pr+h, column r + kx (+32 tt)
    const float bias = (p.bias && tid < 192 && (tid >> 6) < p.cout) ? p.bias[tid >> 6] : 0.f;
    const long plane = (long)p.h * p.w_;

    for (int item = blockIdx.x; item < p.total; item += gridDim.x) {
        const int seg = item % p.segs, i2 = item / p.segs, strip = i2 % p.strips, img = i2 / p.strips;
        const int x0 = strip * 64, y0 = seg * p.sh, y1 = min(y0 + p.sh, p.h);
        const float* xin = p.x + ((long)img * 256 + c * R_CI) * plane;

        // The wave's [32 channels][72 columns] slice of a row, nine 1-KiB pieces through the slab's buffer descriptor.  The lane's chunk
        // (channel slot/18, columns 4*(slot%18)..+3) is recomputed per row from an opaque lane index: nine hoisted 64-bit lane pointers
        // were SPILLED at this kernel's 256-register budget, and hipcc waits vmcnt(0) in front of every scratch reload -- i.e. for every
        // piece issued before it: the pieces went out one HBM round trip after the other (the same defect, measured, in
        // bf16_conv.hip: conv9x9_c256to3_bf16_kernel, profiles/r03_f9_stamps.txt).
        const vcg_rsrc rs = make_rsrc(xin, (size_t)R_CI * plane * 4);
        auto dma = [&](int yi, int buf) {
            const bool rowok = (unsigned)yi < (unsigned)p.h;
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const unsigned row_off = (unsigned)(yi * p.w_ + x0 - 4) * 4u;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int slot = k * 64 + ln;
                const int cl = slot / 18, ch = slot - cl * 18;
                const int gx = x0 - 4 + 4 * ch;
                const bool ok = rowok && gx >= 0 && gx + 3 < p.w_;
                unsigned off = row_off + (unsigned)cl * (unsigned)(plane * 4) + (unsigned)(ch * 16);
                asm volatile("" : "+v"(off));                   // a select, not a branch around the arithmetic
                off = ok ? off : VCG_OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (void __attribute__((address_space(3)))*)(rowbuf + buf * R_ROWB + k * 1024), 16, off, 0, 0, 0);
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
            __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): this row's slice has landed in LDS
            asm volatile("" ::: "memory");
            if (yi + 1 < y1 + 4) dma(yi + 1, buf ^ 1);
            const unsigned char* xb = rowbuf + buf * R_ROWB + bbase;

            // shift the partial sums by one ky group; they become the C operand of this row's MFMAs.  Register by register, carrying the
            // previous group's upper halves (three temporaries, not 32: the kernel has no registers to spare)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                float ph3[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int jq = 0; jq < 4; ++jq)
#pragma unroll
                    for (int s3 = 0; s3 < 3; ++s3) {
                        float a = acc[tt][4 * jq + s3], b = a;
                        swap32(a, b);                              // a = (lower, lower), b = (upper, upper)
                        acc[tt][4 * jq + s3] = hh ? a : ph3[s3];   // group 2jq -> 2jq+1; group 2jq-1 -> 2jq
                        ph3[s3] = b;
                    }
                acc[tt][3] = hh ? ph3[1] : ph3[0];             // rows 3 / 7  <- group 7, co 0 / 1
                acc[tt][7] = hh ? 0.f : ph3[2];                // row 11      <- group 7, co 2
                acc[tt][11] = 0.f;
                acc[tt][15] = 0.f;
            }

            // 36 groups of 8 MFMAs: (kx, half of the 16 channel pairs, column tile); B values double-buffered
            float fb[2][8];
            auto loadb = [&](auto ic) {
                constexpr int i = decltype(ic)::value, kx = i >> 2, ph = (i >> 1) & 1, tt = i & 1, bq = i & 1;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    fb[bq][q] = *(const float*)(xb + (ph * 8 + q) * (2 * R_PIX * 4) + kx * 4 + tt * 128);
            };
            loadb(std::integral_constant<int, 0>{});
            static_for<36>([&](auto ic) {
                constexpr int i = decltype(ic)::value, kx = i >> 2, ph = (i >> 1) & 1, tt = i & 1, cur = i & 1;
                if constexpr (i + 1 < 36) loadb(std::integral_constant<int, i + 1>{});
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[tt] = mfma32(wf[kx][ph * 8 + q], fb[cur][q], acc[tt]);
                __builtin_amdgcn_sched_barrier(0);
            });

            // finished output row yo = yi - 4: this wave's partial (its 32 input channels) -> LDS, summed by 192 threads
            const int yo = yi - 4, slot = yo & 1;
            if (yo >= y0) {
                float* pp = part + ((slot * R_NW + c) * 3) * 64;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    // rows 3 / 7 / 11 = co 0 (lower half) / 1 (upper half) / 2 (lower half)
                    pp[hh * 64 + tt * 32 + r] = acc[tt][3];
                    if (hh == 0) pp[2 * 64 + tt * 32 + r] = acc[tt][7];
                }
            }
            // the partial sums are ordinary LDS stores: wait for them and meet.  NOT lds_barrier(): its fence makes hipcc drain vmcnt(0),
            // i.e. wait here for the NEXT row's slice, which nothing reads before the wait at the top of the next iteration
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (yo >= y0 && tid < 192) {
                const int oc = tid >> 6, col = tid & 63;
                if (oc < p.cout) {
                    const float* q = part + slot * R_NW * 3 * 64 + oc * 64 + col;
                    float v = (((q[0] + q[192]) + (q[2 * 192] + q[3 * 192])) + ((q[4 * 192] + q[5 * 192]) + (q[6 * 192] + q[7 * 192]))) + bias;
                    if (p.act == VCG_ACT_TANH) v = tanhf(v);
                    p.y[((long)(img * p.cout + oc) * p.h + yo) * p.w_ + x0 + col] = v;
                }
            }
        }
        lds_barrier();
    }
}

}  // namespace

// returns VCG_E_UNSUPPORTED when the shape is not served (the caller falls back to the (co,kx)-row kernel)
int vcg_internal_conv9_rowchain(const float* x, const float* w, float* y, int n, int cin, int h, int wd, int cout, const float* bias, int act,
                                int ws_t, int ws_m, int ws_k, hipStream_t st) {
    if (cin != 256 || cout < 1 || cout > 3 || (wd & 63) != 0 || h < 1 || n < 1) return VCG_E_UNSUPPORTED;
    if (act != VCG_ACT_NONE && act != VCG_ACT_TANH) return VCG_E_UNSUPPORTED;
    if ((long)R_CI * h * wd * 4 > 0xFFFFFFE0l) return VCG_E_UNSUPPORTED;          // a wave's 32-channel slab behind one buffer descriptor
    RowChainParams p;
    p.x = x; p.w = w; p.bias = bias; p.y = y;
    p.n = n; p.h = h; p.w_ = wd; p.cout = cout;
    p.strips = wd / 64;
    // segments per column strip: the split that minimises the rows the busiest workgroup marches through (rounds of items per
    // workgroup x (segment height + 8 recomputed halo rows))
    {
        long best = -1;
        for (int segs = 1; segs <= ceil_div(h, 16); ++segs) {
            const int sh = ceil_div(h, segs);
            if (ceil_div(h, sh) != segs) continue;
            const long items = (long)n * p.strips * segs, rounds = (items + 255) / 256, cost = rounds * (sh + 8);
            if (best < 0 || cost < best) { best = cost; p.sh = sh; p.segs = segs; }
        }
    }
    p.total = n * p.strips * p.segs;
    p.ws_t = ws_t; p.ws_m = ws_m; p.ws_k = ws_k;
    p.act = act;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv9x9_rowchain_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int grid = p.total < 256 ? p.total : 256;
    conv9x9_rowchain_f32_kernel<<<grid, R_NW * 64, R_LDS, st>>>(p);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}
