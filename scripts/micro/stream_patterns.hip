// Micro-benchmark: what HBM rate do the bf16 kernels' ACCESS PATTERNS allow on a tensor far larger than the Infinity Cache?
// Every kernel on the 1.07 GB up-sampling tensor ([8, 512, 512, 256] bf16, 512-byte pixels) runs at 2.4-3.2 TB/s while a float4 copy
// reaches ~6.3 (MI355X_MICROARCH.md).  Patterns, each as a persistent 256-thread workgroup per CU x OCC, 16 bytes per lane and access:
//   contig      a wave instruction covers 1 KiB contiguous (the copy kernel's pattern)
//   pix128      a wave instruction covers 32 pixels x 32 bytes at a 128-byte pixel stride; 4 instructions complete the 128-byte pixels
//               (the trunk kernels' epilogue on 64-channel tensors)
//   blk512      the same on 512-byte pixels: a workgroup owns ONE 128-byte channel block of the pixels, the other three blocks belong to
//               other workgroups that run concurrently (convt3x3_c64_bf16_kernel's stores, the 3-channel kernel's mask loads + stores)
//   row512      512-byte pixels, a wave instruction covers 2 whole pixels (32 lanes x 16 bytes each): full pixels by one wave
//   blk512s2    blk512 with the lanes on every SECOND pixel, the pixels between written in a later pass of the same wave (the transposed
//               convolution's column phases)
// for stores, loads, and loads + stores (read one tensor, write another).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/stream_patterns.hip -o /tmp/stream_patterns && /tmp/stream_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

enum { CONTIG = 0, PIX128 = 1, BLK512 = 2, ROW512 = 3, BLK512S2 = 4 };
enum { ST = 1, LD = 2 };

// byte offset of (unit u, instruction i, lane l) inside a "chunk" of 4 KiB x ... ; each pattern walks the buffer in units
template <int PAT>
__device__ __forceinline__ size_t offset_of(size_t unit, int i, int lane, int blk) {
    if (PAT == CONTIG) return unit * 4096 + (size_t)i * 1024 + lane * 16;                     // 4 instructions = 4 KiB contiguous
    if (PAT == PIX128) {                                                                        // unit = 32 pixels of 128 B
        const int r = lane & 31, hh = lane >> 5;
        return unit * 4096 + (size_t)r * 128 + i * 32 + hh * 16;
    }
    if (PAT == BLK512) {                                                                        // unit = 32 pixels of 512 B, this workgroup's 128-B block
        const int r = lane & 31, hh = lane >> 5;
        return unit * 16384 + (size_t)r * 512 + blk * 128 + i * 32 + hh * 16;
    }
    if (PAT == BLK512S2) {                                                                      // unit = 64 pixels of 512 B: lanes address every
        const int r = lane & 31, hh = lane >> 5;                                                // SECOND pixel (the transposed convolution's column
        const int sub = (int)(unit & 1);                                                        // phases): odd units fill in the pixels between
        return (unit >> 1) * 32768 + (size_t)(2 * r + sub) * 512 + blk * 128 + i * 32 + hh * 16;
    }
    // ROW512: unit = 8 pixels of 512 B, instruction i covers pixels 2i, 2i+1 entirely
    return unit * 4096 + (size_t)i * 1024 + lane * 16;
}

template <int PAT, int MODE>
__global__ __launch_bounds__(256) void k(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, size_t units, unsigned* sink) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // BLK512: workgroup b owns channel block b % 4 and unit stream b / 4 (so the four blocks of a unit are touched by four workgroups)
    constexpr bool B4 = PAT == BLK512 || PAT == BLK512S2;
    const int blk = B4 ? blockIdx.x & 3 : 0;
    const size_t w0 = (B4 ? blockIdx.x >> 2 : blockIdx.x) * 4 + wv, nw = (size_t)(B4 ? gridDim.x >> 2 : gridDim.x) * 4;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (size_t u = w0; u < units; u += nw) {
        u32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const size_t o = offset_of<PAT>(u, i, lane, blk);
            if (MODE & LD) v[i] = *(const u32x4*)(src + o);
            else v[i] = u32x4{(unsigned)u, (unsigned)i, (unsigned)lane, 1u};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const size_t o = offset_of<PAT>(u, i, lane, blk);
            if (MODE & ST) *(u32x4*)(dst + o) = v[i];
            else acc += v[i];
        }
    }
    if (!(MODE & ST) && acc[0] + acc[1] + acc[2] + acc[3] == 0x12345678u) sink[0] = 1;         // keep the loads
}

template <int PAT, int MODE>
void run(const char* name, unsigned char* a, unsigned char* b, size_t bytes, int occ, unsigned* sink) {
    // units: CONTIG / PIX128 / ROW512 move 4 KiB per unit; BLK512 moves 4 KiB per (unit, block) of a 16 KiB unit
    const size_t units = (PAT == BLK512 || PAT == BLK512S2) ? bytes / 16384 : bytes / 4096;
    const int grid = 256 * occ;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<PAT, MODE><<<grid, 256>>>(a, b, units, sink);
    hipDeviceSynchronize();
    const int reps = 5;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) k<PAT, MODE><<<grid, 256>>>(a, b, units, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double moved = (double)bytes * ((MODE & LD ? 1 : 0) + (MODE & ST ? 1 : 0)) * reps;
    printf("%-8s %-12s occ=%d  %8.1f us per pass  %6.2f TB/s\n", name, MODE == ST ? "store" : MODE == LD ? "load" : "load+store", occ, ms * 1e3 / reps,
           moved / (ms * 1e-3) / 1e12);
}

int main() {
    const size_t bytes = (size_t)8 * 512 * 512 * 256 * 2;          // 1.07 GB: the up-sampling tensor of C3's shard
    unsigned char *a, *b;
    unsigned* sink;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    hipMemset(a, 1, bytes);
    hipMemset(b, 2, bytes);
    for (int occ : {1, 2}) {
        run<CONTIG, ST>("contig", a, b, bytes, occ, sink);
        run<PIX128, ST>("pix128", a, b, bytes, occ, sink);
        run<BLK512, ST>("blk512", a, b, bytes, occ, sink);
        run<ROW512, ST>("row512", a, b, bytes, occ, sink);
        run<BLK512S2, ST>("blk512s2", a, b, bytes, occ, sink);
        run<CONTIG, LD>("contig", a, b, bytes, occ, sink);
        run<PIX128, LD>("pix128", a, b, bytes, occ, sink);
        run<BLK512, LD>("blk512", a, b, bytes, occ, sink);
        run<CONTIG, ST | LD>("contig", a, b, bytes, occ, sink);
        run<PIX128, ST | LD>("pix128", a, b, bytes, occ, sink);
        run<BLK512, ST | LD>("blk512", a, b, bytes, occ, sink);
    }
    return 0;
}
