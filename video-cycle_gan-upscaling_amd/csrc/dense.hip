// Dense head of the discriminators (model.py:876,880,884): batch <= 64 rows, weight-streaming bound.
// y[b][o] = bias[o] + sum_i x[b][i]*W[i][o];  W is Keras' (in,out) layout, out contiguous.
#include "vcg_common.hpp"

namespace {

constexpr int kMaxBatch = 64;
constexpr int kBT = 8;  // batch rows per accumulator group

// partial sums reduced through LDS in a fixed order (deterministic).  grid.x = ceil(out/64), grid.y = ceil(batch/kBT)
// block = 64 output columns x kSL slices of the input dimension (1024 threads); a thread keeps four weight loads in flight.  (The first
// version had 4 slices and one load in flight: Dense_1 of simple_512, 2048 -> 1024 at batch 8, is 16 blocks of 512 serial loads each --
// 0.27 ms for an 8 MB weight read.)
constexpr int kSL = 16;
__global__ __launch_bounds__(64 * kSL) void dense_fwd_kernel(const float* x, const float* w, const float* bias, float* y,
                                                             int batch, int in, int out) {
    __shared__ float red[kSL][kBT][64];
    const int col = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int o = blockIdx.x * 64 + col;
    const int b0 = blockIdx.y * kBT;
    float acc[kBT];
#pragma unroll
    for (int b = 0; b < kBT; ++b) acc[b] = 0.f;
    if (o < out) {
        int i = slice;
        for (; i + 3 * kSL < in; i += 4 * kSL) {
            float wv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) wv[u] = w[(size_t)(i + u * kSL) * out + o];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int b = 0; b < kBT; ++b)
                    if (b0 + b < batch) acc[b] += x[(size_t)(b0 + b) * in + i + u * kSL] * wv[u];
        }
        for (; i < in; i += kSL) {
            const float wv = w[(size_t)i * out + o];
#pragma unroll
            for (int b = 0; b < kBT; ++b)
                if (b0 + b < batch) acc[b] += x[(size_t)(b0 + b) * in + i] * wv;
        }
    }
#pragma unroll
    for (int b = 0; b < kBT; ++b) red[slice][b][col] = acc[b];
    __syncthreads();
    if (slice == 0 && o < out) {
#pragma unroll
        for (int b = 0; b < kBT; ++b) {
            if (b0 + b >= batch) continue;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < kSL; ++k) s += red[k][b][col];          // fixed order
            if (bias) s += bias[o];
            y[(size_t)(b0 + b) * out + o] = s;
        }
    }
}

// dx[b][i] = sum_o dy[b][o]*W[i][o]: one wave per input row i, lanes stride over o
__global__ __launch_bounds__(256) void dense_dgrad_kernel(const float* dy, const float* w, float* dx, int batch, int in,
                                                          int out) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= in) return;
    for (int b0 = 0; b0 < batch; b0 += kBT) {
        float acc[kBT];
#pragma unroll
        for (int b = 0; b < kBT; ++b) acc[b] = 0.f;
        for (int o = lane; o < out; o += 64) {
            const float wv = w[(size_t)i * out + o];
#pragma unroll
            for (int b = 0; b < kBT; ++b)
                if (b0 + b < batch) acc[b] += dy[(size_t)(b0 + b) * out + o] * wv;
        }
#pragma unroll
        for (int b = 0; b < kBT; ++b) {
            const float s = wave_sum(acc[b]);
            if (lane == 0 && b0 + b < batch) dx[(size_t)(b0 + b) * in + i] = s;
        }
    }
}

// dW[i][o] = sum_b x[b][i]*dy[b][o]
__global__ __launch_bounds__(256) void dense_wgrad_kernel(const float* x, const float* dy, float* dw, int batch, int in,
                                                          int out) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)in * out) return;
    const int i = (int)(idx / out), o = (int)(idx % out);
    float s = 0.f;
    for (int b = 0; b < batch; ++b) s += x[(size_t)b * in + i] * dy[(size_t)b * out + o];
    dw[idx] = s;
}

__global__ void dense_dbias_kernel(const float* dy, float* db, int batch, int out) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= out) return;
    float s = 0.f;
    for (int b = 0; b < batch; ++b) s += dy[(size_t)b * out + o];
    db[o] = s;
}

}  // namespace

extern "C" {

int vcg_dense_fwd(const float* x, const float* w_io, const float* bias, float* y, int batch, int in, int out,
                  vcg_stream_t stream) {
    VCG_CHECK_PTR(x); VCG_CHECK_PTR(w_io); VCG_CHECK_PTR(y);
    if (batch <= 0 || in <= 0 || out <= 0) return VCG_E_SHAPE;
    hipLaunchKernelGGL(dense_fwd_kernel, dim3(ceil_div(out, 64), ceil_div(batch, kBT)), dim3(64 * kSL), 0, (hipStream_t)stream,
                       x, w_io, bias, y, batch, in, out);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_dense_dgrad(const float* dy, const float* w_io, float* dx, int batch, int in, int out, vcg_stream_t stream) {
    VCG_CHECK_PTR(dy); VCG_CHECK_PTR(w_io); VCG_CHECK_PTR(dx);
    if (batch <= 0 || in <= 0 || out <= 0) return VCG_E_SHAPE;
    hipLaunchKernelGGL(dense_dgrad_kernel, dim3(ceil_div(in, 4)), dim3(256), 0, (hipStream_t)stream, dy, w_io, dx, batch,
                       in, out);
    VCG_LAUNCH_CHECK();
    return VCG_OK;
}

int vcg_dense_wgrad(const float* x, const float* dy, float* dw_io, float* dbias, int batch, int in, int out,
                    vcg_stream_t stream) {
    VCG_CHECK_PTR(x); VCG_CHECK_PTR(dy); VCG_CHECK_PTR(dw_io);
    if (batch <= 0 || in <= 0 || out <= 0) return VCG_E_SHAPE;
    const size_t total = (size_t)in * out;
    hipLaunchKernelGGL(dense_wgrad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       dy, dw_io, batch, in, out);
    VCG_LAUNCH_CHECK();
    if (dbias) {
        hipLaunchKernelGGL(dense_dbias_kernel, dim3(ceil_div(out, 256)), dim3(256), 0, (hipStream_t)stream, dy, dbias,
                           batch, out);
        VCG_LAUNCH_CHECK();
    }
    return VCG_OK;
}

}  // extern "C"
