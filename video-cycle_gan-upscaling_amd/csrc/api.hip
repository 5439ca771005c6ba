// C-ABI wrappers for the convolution family: map each Keras layer operation (forward, data
// gradient, weight gradient) onto the three MFMA kernels (conv_fwd.hip, conv_transpose.hip,
// conv_wgrad.hip).  See include/vcg.h for the contract.
#include "vcg_common.hpp"

int vcg_internal_conv(const float* x, const float* w, float* y, int n, int cin, int h, int wd, int cout, int oh,
                      int ow, int kh, int kw, int stride, int pad_top, int pad_left, int flip, const vcg_epilogue* ep,
                      int smallm, int ws_t, int ws_m, int ws_k, hipStream_t st, float* stats = nullptr);
int vcg_internal_convt(const float* x, const float* w, float* y, int n, int cin, int h, int wd, int cout, int oh,
                       int ow, int k, int cby, int cbx, const vcg_epilogue* ep, hipStream_t st);
size_t vcg_internal_wgrad_ws(int n, int mtot, int ah, int aw, int jctot, int kh, int kw, int S);
int vcg_internal_wgrad(const float* A, const float* B, float* dw, float* db, int n, int mtot, int ah, int aw,
                       int jctot, int bh, int bw, int kh, int kw, int S, int pt, int pl_, int flip, int ts, int sm,
                       int sj, void* ws, size_t ws_bytes, hipStream_t st);

namespace {

int check_desc(const vcg_conv_desc* d) {
    if (d == nullptr) return VCG_E_NULL;
    if (d->n <= 0 || d->cin <= 0 || d->h <= 0 || d->w <= 0 || d->cout <= 0 || d->oh <= 0 || d->ow <= 0)
        return VCG_E_SHAPE;
    if (d->kh <= 0 || d->kw <= 0 || d->stride < 1 || d->stride > 3) return VCG_E_UNSUPPORTED;
    if (d->pad_top < 0 || d->pad_left < 0 || d->pad_top >= d->kh || d->pad_left >= d->kw) return VCG_E_SHAPE;
    return VCG_OK;
}

// the small-M (channel,kx)-in-rows kernel pays off when <= 32/KW output channels are produced
// the small-M kernel (<= 32 / kw result channels) is instantiated for the square kernels only
inline bool use_smallm(int mch, int kh, int kw, int stride) { return stride == 1 && kh == kw && kh >= 3 && mch * kw <= 32; }

}  // namespace

extern "C" {

const char* vcg_version(void) { return "vcg-hip 0.1 (gfx950, fp32 MFMA 32x32x2)"; }

const char* vcg_error_string(int code) {
    switch (code) {
        case VCG_OK: return "ok";
        case VCG_E_NULL: return "required pointer is NULL";
        case VCG_E_SHAPE: return "inconsistent or unsupported dimensions";
        case VCG_E_UNSUPPORTED: return "kernel size / stride / padding combination not instantiated";
        case VCG_E_WORKSPACE: return "workspace missing or too small";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

// ---------------------------------------------------------------------------------------------------
// Conv2D
// ---------------------------------------------------------------------------------------------------
int vcg_conv2d_fwd(const vcg_conv_desc* d, const float* x, const float* w_hwio, float* y, const vcg_epilogue* ep,
                   vcg_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    VCG_CHECK_PTR(x); VCG_CHECK_PTR(w_hwio); VCG_CHECK_PTR(y);
    // output extent must be consistent with the pads: last tap of the last output stays < h + k
    if ((d->oh - 1) * d->stride - d->pad_top >= d->h || (d->ow - 1) * d->stride - d->pad_left >= d->w) return VCG_E_SHAPE;
    const bool sm = use_smallm(d->cout, d->kh, d->kw, d->stride);
    // HWIO = [tap][cin][cout]: element (mch=co, kc=ci, tap) at tap*cin*cout + ci*cout + co
    return vcg_internal_conv(x, w_hwio, y, d->n, d->cin, d->h, d->w, d->cout, d->oh, d->ow, d->kh, d->kw, d->stride,
                             d->pad_top, d->pad_left, 0, ep, sm ? 1 : 0, d->cin * d->cout, 1, d->cout,
                             (hipStream_t)stream);
}

// the tile grid of conv_fwd_kernel for this layer (conv_fwd.hip: 8 output rows x 32 or 64 columns per workgroup)
static bool conv_stats_tiles(const vcg_conv_desc* d, int* tiles_per_image) {
    const bool k3 = d->kh == 3 && d->kw == 3, k4 = d->kh == 4 && d->kw == 4;
    if (!(k3 || k4) || d->stride > 2 || d->cin <= 3) return false;
    const bool sm = use_smallm(d->cout, d->kh, d->kw, d->stride);
    if (sm) return false;
    const int xt = (k3 && d->stride == 1 && d->ow > 32) ? 2 : 1;
    *tiles_per_image = ceil_div(d->ow, 32 * xt) * ceil_div(d->oh, 8);
    return true;
}

// records per group (image, or the whole batch) that vcg_conv2d_fwd_stats writes for this layer; a negative VCG_E_* when the layer's kernel
// has no statistics epilogue (run vcg_norm_stats on the output)
int vcg_conv2d_stats_records(const vcg_conv_desc* d, int stats_mode) {
    int rc = check_desc(d);
    if (rc) return rc;
    int tpi = 0;
    if (!conv_stats_tiles(d, &tpi) || (stats_mode != VCG_STATS_BATCH && stats_mode != VCG_STATS_INSTANCE)) return VCG_E_UNSUPPORTED;
    const long all = (long)tpi * d->n;
    if (all > 0x3fffffffL) return VCG_E_UNSUPPORTED;
    return (int)(stats_mode == VCG_STATS_INSTANCE ? tpi : all);
}

// vcg_conv2d_fwd (bias, no activation) that also leaves, per output tile, the per-channel sum of (y - bias) and of its square:
// stats fp32 [n][tiles per image][2][cout] -- read by vcg_norm_finalize_partials_shifted(kshift = bias) as [1][n * tiles] (batch statistics)
// or [n][tiles] (instance norm).  The statistics pass of the normalisation behind the layer (model.py:20, 23, 284, 840) is gone.
int vcg_conv2d_fwd_stats(const vcg_conv_desc* d, const float* x, const float* w_hwio, const float* bias, float* y, float* stats,
                         vcg_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    VCG_CHECK_PTR(x); VCG_CHECK_PTR(w_hwio); VCG_CHECK_PTR(y); VCG_CHECK_PTR(stats);
    if ((d->oh - 1) * d->stride - d->pad_top >= d->h || (d->ow - 1) * d->stride - d->pad_left >= d->w) return VCG_E_SHAPE;
    int tpi = 0;
    if (!conv_stats_tiles(d, &tpi)) return VCG_E_UNSUPPORTED;
    vcg_epilogue ep{};
    ep.bias = bias;
    ep.act = VCG_ACT_NONE;
    return vcg_internal_conv(x, w_hwio, y, d->n, d->cin, d->h, d->w, d->cout, d->oh, d->ow, d->kh, d->kw, d->stride, d->pad_top, d->pad_left, 0,
                             &ep, 0, d->cin * d->cout, 1, d->cout, (hipStream_t)stream, stats);
}

int vcg_conv2d_dgrad(const vcg_conv_desc* d, const float* dy, const float* w_hwio, const float* w_hwoi, float* dx,
                     const float* residual, vcg_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    VCG_CHECK_PTR(dy); VCG_CHECK_PTR(dx);
    vcg_epilogue ep{};
    ep.residual = residual;
    hipStream_t st = (hipStream_t)stream;
    if (d->stride == 1) {
        // dx[ci][i] = sum_{co,k'} dy[co][i + k' - (K-1-p)] * W[K-1-k'][ci][co]
        const int pt = d->kh - 1 - d->pad_top, pl = d->kw - 1 - d->pad_left;
        if (use_smallm(d->cin, d->kh, d->kw, 1)) {
            VCG_CHECK_PTR(w_hwio);
            // rows = (ci, kx); element (mch=ci, kc=co, tap) of HWIO at tap*cin*cout + ci*cout + co
            return vcg_internal_conv(dy, w_hwio, dx, d->n, d->cout, d->oh, d->ow, d->cin, d->h, d->w, d->kh, d->kw, 1,
                                     pt, pl, 1, &ep, 1, d->cin * d->cout, d->cout, 1, st);
        }
        VCG_CHECK_PTR(w_hwoi);
        // HWOI = [tap][kc=co][m=ci]
        return vcg_internal_conv(dy, w_hwoi, dx, d->n, d->cout, d->oh, d->ow, d->cin, d->h, d->w, d->kh, d->kw, 1, pt,
                                 pl, 1, &ep, 0, 0, 0, 0, st);
    }
    // stride 2: dx[ci][i] = sum_{co,j,k: 2j+k-p=i} dy[co][j] * W[k][ci][co]  == transposed conv, crop p
    // (stride 3 -- sparse_512 only -- is served by the caller: vcg_dilate2d of dy, then this function with stride 1)
    if (d->kh != d->kw || d->stride != 2) return VCG_E_UNSUPPORTED;
    VCG_CHECK_PTR(w_hwoi);
    return vcg_internal_convt(dy, w_hwoi, dx, d->n, d->cout, d->oh, d->ow, d->cin, d->h, d->w, d->kh, d->pad_top,
                              d->pad_left, &ep, st);
}

size_t vcg_conv2d_wgrad_workspace_bytes(const vcg_conv_desc* d) {
    if (check_desc(d)) return 0;
    size_t a;
    if (d->stride == 1 && d->cout <= 3)
        a = vcg_internal_wgrad_ws(d->n, d->cin, d->h, d->w, d->cout, d->kh, d->kw, 1);
    else
        a = vcg_internal_wgrad_ws(d->n, d->cout, d->oh, d->ow, d->cin, d->kh, d->kw, d->stride);
    const size_t b = vcg_channel_sum_workspace_bytes(d->n, d->cout, d->oh * d->ow);
    return a > b ? a : b;
}

int vcg_conv2d_wgrad(const vcg_conv_desc* d, const float* x, const float* dy, float* dw_hwio, float* dbias, void* ws,
                     size_t ws_bytes, vcg_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    VCG_CHECK_PTR(x); VCG_CHECK_PTR(dy); VCG_CHECK_PTR(dw_hwio);
    hipStream_t st = (hipStream_t)stream;
    const int cc = d->cin * d->cout;
    if (d->stride == 1 && d->cout <= 3) {
        // swapped orientation: A = x (m = ci), B = dy (jc = co), flipped taps, pads K-1-p
        rc = vcg_internal_wgrad(x, dy, dw_hwio, nullptr, d->n, d->cin, d->h, d->w, d->cout, d->oh, d->ow, d->kh, d->kw,
                                1, d->kh - 1 - d->pad_top, d->kw - 1 - d->pad_left, 1, cc, d->cout, 1, ws, ws_bytes, st);
    } else {
        // normal: A = dy (m = co), B = x (jc = ci);  dw[tap][ci][co]; the bias gradient (per-channel sum of
        // dy) is accumulated from the staged dy tiles inside the same kernel
        rc = vcg_internal_wgrad(dy, x, dw_hwio, dbias, d->n, d->cout, d->oh, d->ow, d->cin, d->h, d->w, d->kh, d->kw,
                                d->stride, d->pad_top, d->pad_left, 0, cc, 1, d->cout, ws, ws_bytes, st);
        dbias = nullptr;
    }
    if (rc) return rc;
    if (dbias) {
        // the wgrad partial buffer is consumed by then (same stream); reuse the workspace
        if (ws_bytes < vcg_channel_sum_workspace_bytes(d->n, d->cout, d->oh * d->ow)) return VCG_E_WORKSPACE;
        rc = vcg_channel_sum(dy, d->n, d->cout, d->oh * d->ow, dbias, ws, ws_bytes, stream);
    }
    return rc;
}

// ---------------------------------------------------------------------------------------------------
// Conv2DTranspose (stride 2, 'same')
// ---------------------------------------------------------------------------------------------------
int vcg_conv_transpose2d_fwd(const vcg_conv_desc* d, const float* x, const float* w_hwio_t, float* y,
                             const vcg_epilogue* ep, vcg_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    VCG_CHECK_PTR(x); VCG_CHECK_PTR(w_hwio_t); VCG_CHECK_PTR(y);
    if (d->stride != 2 || d->kh != d->kw) return VCG_E_UNSUPPORTED;
    if (d->oh > 2 * d->h || d->ow > 2 * d->w) return VCG_E_SHAPE;
    // w_hwio_t = per-tap transpose of the Keras (kh,kw,out,in) kernel = [tap][kc=in][m=out]
    return vcg_internal_convt(x, w_hwio_t, y, d->n, d->cin, d->h, d->w, d->cout, d->oh, d->ow, d->kh, d->pad_top,
                              d->pad_left, ep, (hipStream_t)stream);
}

int vcg_conv_transpose2d_dgrad(const vcg_conv_desc* d, const float* dy, const float* w_hwoi, float* dx,
                               const float* residual, vcg_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    VCG_CHECK_PTR(dy); VCG_CHECK_PTR(w_hwoi); VCG_CHECK_PTR(dx);
    if (d->stride != 2) return VCG_E_UNSUPPORTED;
    vcg_epilogue ep{};
    ep.residual = residual;
    // dx[ci][i] = sum_{co,k} dy[co][2i + k - cb] * W[k][co][ci]: stride-2 conv over dy, pad = crop,
    // kernel (kh,kw,out,in) = [tap][kc=co][m=ci] as stored
    return vcg_internal_conv(dy, w_hwoi, dx, d->n, d->cout, d->oh, d->ow, d->cin, d->h, d->w, d->kh, d->kw, 2,
                             d->pad_top, d->pad_left, 0, &ep, 0, 0, 0, 0, (hipStream_t)stream);
}

size_t vcg_conv_transpose2d_wgrad_workspace_bytes(const vcg_conv_desc* d) {
    if (check_desc(d)) return 0;
    size_t a = vcg_internal_wgrad_ws(d->n, d->cin, d->h, d->w, d->cout, d->kh, d->kw, 2);
    size_t b = vcg_channel_sum_workspace_bytes(d->n, d->cout, d->oh * d->ow);
    return a > b ? a : b;
}

int vcg_conv_transpose2d_wgrad(const vcg_conv_desc* d, const float* x, const float* dy, float* dw_hwoi, float* dbias,
                               void* ws, size_t ws_bytes, vcg_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    VCG_CHECK_PTR(x); VCG_CHECK_PTR(dy); VCG_CHECK_PTR(dw_hwoi);
    if (d->stride != 2) return VCG_E_UNSUPPORTED;
    // dW[k][co][ci] = sum_i x[ci][i] * dy[co][2i + k - cb]: A = x (m = ci), B = dy (jc = co), stride 2
    rc = vcg_internal_wgrad(x, dy, dw_hwoi, nullptr, d->n, d->cin, d->h, d->w, d->cout, d->oh, d->ow, d->kh, d->kw, 2,
                            d->pad_top, d->pad_left, 0, d->cin * d->cout, 1, d->cin, ws, ws_bytes, (hipStream_t)stream);
    if (rc) return rc;
    if (dbias) {
        if (ws_bytes < vcg_channel_sum_workspace_bytes(d->n, d->cout, d->oh * d->ow)) return VCG_E_WORKSPACE;
        rc = vcg_channel_sum(dy, d->n, d->cout, d->oh * d->ow, dbias, ws, ws_bytes, stream);
    }
    return rc;
}

}  // extern "C"
