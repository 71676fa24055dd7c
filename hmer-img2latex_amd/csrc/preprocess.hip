// Image preprocessing of the inference / data path on the device (reference img2latex/data/utils.py:18-90 `load_image`
// after PIL has decoded the file, with data/transforms.py:26-56 `ResizeWithAspectRatio`): mode conversion, LANCZOS
// resize to the target height (aspect kept), right-pad / centre-crop to the target width, /255 and normalisation,
// for a ragged batch of uint8 images in ONE pair of launches.  The resize is Pillow's (libImaging/Resample.c, 8 bits
// per channel): separable, horizontal pass first into an 8-bit intermediate, per-output-pixel normalised Lanczos-3
// weights in 22-bit fixed point, int32 accumulation from 2^21, arithmetic shift, clamp.  The weights are computed on
// the HOST in double precision with libm's sin -- i2l_lanczos_coeffs below, the same arithmetic Pillow runs -- so the
// device work is pure integer and the result is bit-identical to the reference's.
#include <math.h>
#include <string.h>

#include <thread>
#include <vector>

#include "common.h"

namespace {

constexpr int PREC = 32 - 8 - 2;

double sinc_filter(double x) {
    if (x == 0.0) return 1.0;
    x = x * M_PI;
    return sin(x) / x;
}
double lanczos_filter(double x) {
    if (-3.0 <= x && x < 3.0) return sinc_filter(x) * sinc_filter(x / 3);
    return 0.0;
}
// Pillow's BICUBIC (Keys, a = -0.5, support 2): what Image.resize() uses when no filter is named (predictor.py:439)
double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}
double filter_support(int filter) { return filter == I2L_FILTER_BICUBIC ? 2.0 : 3.0; }

// ---- the same tables built ON THE DEVICE (r04): one thread per (table, output sample) repeats the host function's
// double-precision arithmetic operation by operation (no contraction into FMAs: the host build has none).  The one
// difference is sin(): the device's (ocml) and libm's are both faithfully rounded but not the same function, so a
// weight can differ by one unit of 2^-22 where the normalised value sits within ~1e-16 of a rounding boundary (odds
// ~1e-9 per weight).  The host function stays the Pillow-identical reference; tests hold the device tables equal to it.
#pragma clang fp contract(off)
__device__ double sinc_filter_d(double x) {
    if (x == 0.0) return 1.0;
    x = x * M_PI;
    return sin(x) / x;
}
__device__ double resample_weight_d(int filter, double x) {
    if (filter == I2L_FILTER_BICUBIC) {
        const double a = -0.5;
        if (x < 0.0) x = -x;
        if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
        if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
        return 0.0;
    }
    if (-3.0 <= x && x < 3.0) return sinc_filter_d(x) * sinc_filter_d(x / 3);
    return 0.0;
}
__global__ __launch_bounds__(64) void resample_coeffs_kernel(int filter, const int32_t* __restrict__ in_sizes,
                                                             const int32_t* __restrict__ out_sizes,
                                                             const int64_t* __restrict__ offsets, int32_t* __restrict__ tables) {
    const int i = blockIdx.y;
    const int in_size = in_sizes[i], out_size = out_sizes[i];
    const float in0 = 0.0f, in1 = (float)in_size;
    double filterscale, scale;
    filterscale = scale = (double)(in1 - in0) / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = (filter == I2L_FILTER_BICUBIC ? 2.0 : 3.0) * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    const double ss = 1.0 / filterscale;
    int32_t* bounds = tables + offsets[i];
    int32_t* kk_all = bounds + 2 * (int64_t)out_size;
    for (int xx = blockIdx.x * 64 + threadIdx.x; xx < out_size; xx += gridDim.x * 64) {
        const double center = in0 + (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) ww += resample_weight_d(filter, (x + xmin - center + 0.5) * ss);
        int32_t* kk = kk_all + (size_t)xx * ksize;
        int x;
        for (x = 0; x < xmax; ++x) {
            double k = resample_weight_d(filter, (x + xmin - center + 0.5) * ss);
            if (ww != 0.0) k /= ww;
            kk[x] = k < 0 ? (int)(-0.5 + k * (1 << PREC)) : (int)(0.5 + k * (1 << PREC));
        }
        for (; x < ksize; ++x) kk[x] = 0;
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
}
#pragma clang fp contract(fast)

// value of converted-mode pixel (y, x, band c) of the source image: utils.py:45-49 img.convert("L" / "RGB")
__device__ __forceinline__ int src_px(const uint8_t* __restrict__ img, int w, int src_c, int out_c, int y, int x, int c) {
    const size_t p = ((size_t)y * w + x) * src_c;
    if (src_c == out_c) return img[p + c];
    if (src_c == 3) return (img[p] * 19595 + img[p + 1] * 38470 + img[p + 2] * 7471 + 0x8000) >> 16;   // RGB -> L
    return img[p];                                                                                        // L -> RGB
}
__device__ __forceinline__ int clip8(int acc) { return min(max(acc >> PREC, 0), 255); }

// horizontal pass: tmp[row][xx][c] for rows ybox_first .. ybox_first + tmp_rows
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ pixels, const i2l_resize_plan* __restrict__ plans,
                                                       const int32_t* __restrict__ tables, uint8_t* __restrict__ ws, int out_c) {
    const i2l_resize_plan pl = plans[blockIdx.y];
    if (!pl.need_h) return;
    const long total = (long)pl.tmp_rows * pl.new_w;
    const uint8_t* img = pixels + pl.src_offset;
    uint8_t* tmp = ws + pl.tmp_offset;
    const int32_t* bh = tables + pl.bh_offset;
    const int32_t* kh = tables + pl.kh_offset;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int row = (int)(idx / pl.new_w), xx = (int)(idx - (long)row * pl.new_w);
        const int xmin = bh[2 * xx], n = bh[2 * xx + 1];
        const int32_t* k = kh + (size_t)xx * pl.kh_ksize;
        // one pass over the taps for all bands: a source pixel (and its mode conversion) is read once, not once per band
        int acc0 = 1 << (PREC - 1), acc1 = acc0, acc2 = acc0;
        const uint8_t* srow = img + ((size_t)(pl.ybox_first + row) * pl.src_w + xmin) * pl.src_c;
        if (pl.src_c == 1) {                                   // L source: the bands of an RGB output are equal
            for (int x = 0; x < n; ++x) acc0 += srow[x] * k[x];
            acc1 = acc2 = acc0;
        } else if (out_c == 1) {                               // RGB -> L (Convert.c rgb2l)
            for (int x = 0; x < n; ++x)
                acc0 += ((srow[3 * x] * 19595 + srow[3 * x + 1] * 38470 + srow[3 * x + 2] * 7471 + 0x8000) >> 16) * k[x];
        } else {
            for (int x = 0; x < n; ++x) {
                const int kx = k[x];
                acc0 += srow[3 * x] * kx; acc1 += srow[3 * x + 1] * kx; acc2 += srow[3 * x + 2] * kx;
            }
        }
        uint8_t* o = tmp + ((size_t)row * pl.new_w + xx) * out_c;
        o[0] = (uint8_t)clip8(acc0);
        if (out_c == 3) { o[1] = (uint8_t)clip8(acc1); o[2] = (uint8_t)clip8(acc2); }
    }
}

// vertical pass + pad / crop + float conversion: out[b][c][y][x]
__global__ __launch_bounds__(256) void resize_v_norm_kernel(const uint8_t* __restrict__ pixels, const i2l_resize_plan* __restrict__ plans,
                                                            const int32_t* __restrict__ tables, const uint8_t* __restrict__ ws,
                                                            int out_c, int out_h, int out_w, int normalize, float* __restrict__ out) {
    const int b = blockIdx.y;
    const i2l_resize_plan pl = plans[b];
    const uint8_t* img = pixels + pl.src_offset;
    const uint8_t* tmp = ws + pl.tmp_offset;
    const int32_t* bv = tables + pl.bv_offset;
    const int32_t* kv = tables + pl.kv_offset;
    const int left = pl.new_w > out_w ? (pl.new_w - out_w) / 2 : 0;        // transforms.py:50-55 centre crop
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < (long)out_h * out_w; idx += (long)gridDim.x * 256) {
        const int y = (int)(idx / out_w), x = (int)(idx - (long)y * out_w);
        const int xs = x + left;
        int v[3];
        if (xs >= pl.new_w) {
            v[0] = 255; v[1] = v[2] = (out_c == 1) ? 255 : 0;     // Image.new(mode, size, 255): white for L, (255,0,0) for RGB
        } else if (pl.need_v) {
            const int y0 = bv[2 * y] - (pl.need_h ? pl.ybox_first : 0), n = bv[2 * y + 1];   // Resample.c "shift bounds for vertical pass"
            const int32_t* k = kv + (size_t)y * pl.kv_ksize;
            int a0 = 1 << (PREC - 1), a1 = a0, a2 = a0;
            if (pl.need_h) {
                const uint8_t* col = tmp + ((size_t)y0 * pl.new_w + xs) * out_c;
                const size_t rs = (size_t)pl.new_w * out_c;
                if (out_c == 1) for (int j = 0; j < n; ++j) a0 += col[j * rs] * k[j];
                else for (int j = 0; j < n; ++j) { const int kj = k[j]; a0 += col[j * rs] * kj; a1 += col[j * rs + 1] * kj; a2 += col[j * rs + 2] * kj; }
            } else {
                for (int j = 0; j < n; ++j) {
                    const int kj = k[j];
                    a0 += src_px(img, pl.src_w, pl.src_c, out_c, y0 + j, xs, 0) * kj;
                    if (out_c == 3) { a1 += src_px(img, pl.src_w, pl.src_c, out_c, y0 + j, xs, 1) * kj; a2 += src_px(img, pl.src_w, pl.src_c, out_c, y0 + j, xs, 2) * kj; }
                }
            }
            v[0] = clip8(a0); v[1] = clip8(a1); v[2] = clip8(a2);
        } else {
            for (int c = 0; c < out_c; ++c)
                v[c] = pl.need_h ? tmp[((size_t)y * pl.new_w + xs) * out_c + c] : src_px(img, pl.src_w, pl.src_c, out_c, y, xs, c);
        }
        for (int c = 0; c < out_c; ++c) {
            float t = __fdiv_rn((float)v[c], 255.0f);                       // utils.py:68
            if (normalize) {
                if (out_c == 1 || normalize == 2) t = __fsub_rn(__fmul_rn(t, 2.0f), 1.0f);   // utils.py:74; predictor.py:449-451
                else t = __fdiv_rn(__fsub_rn(t, mean[c]), stdv[c]);         // utils.py:77-79
            }
            out[(((size_t)b * out_c + c) * out_h + y) * out_w + x] = t;
        }
    }
}

// torch.nn.functional.interpolate(mode="bilinear", align_corners=False) of fp32 planes (predictor.py:483-491):
// source index = max(0, (dst + 0.5) * in / out - 0.5), the neighbour clamped to the last sample
__global__ __launch_bounds__(256) void bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, long planes,
                                                       int ih, int iw, int oh, int ow) {
    const float sh = (float)ih / (float)oh, sw = (float)iw / (float)ow;
    const long total = planes * oh * ow;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int x = (int)(idx % ow), y = (int)((idx / ow) % oh);
        const long p = idx / ((long)ow * oh);
        const float fy = fmaxf(__fsub_rn(__fmul_rn(sh, (float)y + 0.5f), 0.5f), 0.0f);
        const float fx = fmaxf(__fsub_rn(__fmul_rn(sw, (float)x + 0.5f), 0.5f), 0.0f);
        const int y0 = min((int)fy, ih - 1), x0 = min((int)fx, iw - 1);
        const int y1 = min(y0 + 1, ih - 1), x1 = min(x0 + 1, iw - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float* src = in + p * (long)ih * iw;
        const float top = __fadd_rn(__fmul_rn(1.0f - lx, src[(long)y0 * iw + x0]), __fmul_rn(lx, src[(long)y0 * iw + x1]));
        const float bot = __fadd_rn(__fmul_rn(1.0f - lx, src[(long)y1 * iw + x0]), __fmul_rn(lx, src[(long)y1 * iw + x1]));
        out[idx] = __fadd_rn(__fmul_rn(1.0f - ly, top), __fmul_rn(ly, bot));
    }
}

}  // namespace

extern "C" int i2l_resize_bilinear_f32(const float* in, float* out, int64_t planes, int in_h, int in_w, int out_h,
                                       int out_w, i2l_stream_t stream) {
    if (!in || !out || planes <= 0 || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0) return I2L_ERR_ARG;
    long total = (long)planes * out_h * out_w;
    int bx = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(bilinear_kernel, dim3(bx), dim3(256), 0, i2l_s(stream), in, out, (long)planes, in_h, in_w, out_h, out_w);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}

// ---- host: Pillow's precompute_coeffs + normalize_coeffs_8bpc for the LANCZOS / BICUBIC filter over the full source range
extern "C" int i2l_resample_ksize(int filter, int in_size, int out_size) {
    if (in_size <= 0 || out_size <= 0 || (filter != I2L_FILTER_LANCZOS && filter != I2L_FILTER_BICUBIC)) return 0;
    double filterscale = (double)((float)in_size - 0.0f) / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    return (int)ceil(filter_support(filter) * filterscale) * 2 + 1;
}
extern "C" int i2l_lanczos_ksize(int in_size, int out_size) { return i2l_resample_ksize(I2L_FILTER_LANCZOS, in_size, out_size); }

extern "C" int i2l_resample_coeffs(int filter, int in_size, int out_size, int32_t* bounds_out, int32_t* kk_out) {
    if (in_size <= 0 || out_size <= 0 || !bounds_out || !kk_out) return I2L_ERR_ARG;
    if (filter != I2L_FILTER_LANCZOS && filter != I2L_FILTER_BICUBIC) return I2L_ERR_UNSUPPORTED;
    const float in0 = 0.0f, in1 = (float)in_size;
    double filterscale, scale;
    filterscale = scale = (double)(in1 - in0) / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = filter_support(filter) * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = in0 + (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double k[512];
        if (xmax > 512) return I2L_ERR_UNSUPPORTED;           // down-scaling by more than ~80x
        int x;
        for (x = 0; x < xmax; ++x) {
            const double arg = (x + xmin - center + 0.5) * ss;
            const double w = filter == I2L_FILTER_BICUBIC ? bicubic_filter(arg) : lanczos_filter(arg);
            k[x] = w;
            ww += w;
        }
        int32_t* kk = kk_out + (size_t)xx * ksize;
        for (x = 0; x < xmax; ++x) {
            if (ww != 0.0) k[x] /= ww;
            kk[x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << PREC)) : (int)(0.5 + k[x] * (1 << PREC));
        }
        for (; x < ksize; ++x) kk[x] = 0;
        bounds_out[2 * xx] = xmin;
        bounds_out[2 * xx + 1] = xmax;
    }
    return I2L_OK;
}

extern "C" int i2l_lanczos_coeffs(int in_size, int out_size, int32_t* bounds_out, int32_t* kk_out) {
    return i2l_resample_coeffs(I2L_FILTER_LANCZOS, in_size, out_size, bounds_out, kk_out);
}

// n tables in one call, computed by up to `threads` host threads (the double-precision sin() of the Lanczos weights is
// ~0.1 ms per image on one core: the whole evaluate batch would wait ~25 ms for its tables).  Entry i resamples
// in_sizes[i] -> out_sizes[i] and writes bounds (out, 2) at out + offsets[i], weights (out, ksize) right behind them.
extern "C" int i2l_resample_coeffs_batch(int filter, int n, const int32_t* in_sizes, const int32_t* out_sizes,
                                         const int64_t* offsets, int32_t* out, int threads) {
    if (n < 0 || (n > 0 && (!in_sizes || !out_sizes || !offsets || !out))) return I2L_ERR_ARG;
    if (filter != I2L_FILTER_LANCZOS && filter != I2L_FILTER_BICUBIC) return I2L_ERR_UNSUPPORTED;
    std::vector<int> rc((size_t)(n > 0 ? n : 1), I2L_OK);
    auto work = [&](int lo, int hi) {
        for (int i = lo; i < hi; ++i)
            rc[i] = i2l_resample_coeffs(filter, in_sizes[i], out_sizes[i], out + offsets[i], out + offsets[i] + 2 * (int64_t)out_sizes[i]);
    };
    const int nt = threads < 1 ? 1 : (threads > n ? (n > 0 ? n : 1) : threads);
    if (nt <= 1) work(0, n);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; ++t) pool.emplace_back(work, (int)((long long)n * t / nt), (int)((long long)n * (t + 1) / nt));
        for (auto& th : pool) th.join();
    }
    for (int i = 0; i < n; ++i) if (rc[i] != I2L_OK) return rc[i];
    return I2L_OK;
}

extern "C" int i2l_resample_coeffs_device(int filter, int n, const int32_t* in_sizes, const int32_t* out_sizes,
                                          const int64_t* offsets, int32_t* tables, int max_out_size, i2l_stream_t stream) {
    if (n <= 0 || n > 65535 || !in_sizes || !out_sizes || !offsets || !tables || max_out_size <= 0) return I2L_ERR_ARG;
    if (filter != I2L_FILTER_LANCZOS && filter != I2L_FILTER_BICUBIC) return I2L_ERR_UNSUPPORTED;
    int bx = (max_out_size + 63) / 64;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(resample_coeffs_kernel, dim3(bx, n), dim3(64), 0, i2l_s(stream), filter, in_sizes, out_sizes, offsets, tables);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}

// n host buffers -> one (pinned) block, on up to `threads` host threads that live for the call: the ragged pages of a batch
// are separate arrays on the reference's side (PIL decodes them one by one); one thread copies 16 MB in ~3 ms
extern "C" int i2l_pack_host(const void* const* srcs, const int64_t* sizes, const int64_t* offsets, int n, void* dst, int threads) {
    if (n < 0 || (n > 0 && (!srcs || !sizes || !offsets || !dst))) return I2L_ERR_ARG;
    for (int i = 0; i < n; ++i)
        if (sizes[i] < 0 || offsets[i] < 0 || (sizes[i] > 0 && !srcs[i])) return I2L_ERR_ARG;
    int64_t total = 0;
    for (int i = 0; i < n; ++i) total += sizes[i];
    int nt = threads < 1 ? 1 : threads;
    if (total < (1 << 20) || n < 2) nt = 1;
    if (nt > n) nt = n;
    auto work = [&](int t) {                                  // contiguous runs of images with about total / nt bytes each
        const int64_t lo = total * t / nt, hi = total * (t + 1) / nt;
        int64_t acc = 0;
        for (int i = 0; i < n; ++i) {
            if (acc >= lo && acc < hi && sizes[i] > 0) memcpy(static_cast<char*>(dst) + offsets[i], srcs[i], (size_t)sizes[i]);
            acc += sizes[i];
        }
    };
    if (nt == 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; ++t) pool.emplace_back(work, t);
        for (auto& th : pool) th.join();
    }
    return I2L_OK;
}

extern "C" int i2l_preprocess_images(const uint8_t* pixels, const i2l_resize_plan* plans, const int32_t* tables, int n,
                                     int max_tmp_px, int out_c, int out_h, int out_w, int normalize, void* workspace,
                                     float* out, i2l_stream_t stream) {
    if (!pixels || !plans || !tables || !out || n <= 0 || (out_c != 1 && out_c != 3) || out_h <= 0 || out_w <= 0 || n > 65535)
        return I2L_ERR_ARG;
    if (max_tmp_px > 0 && !workspace) return I2L_ERR_WORKSPACE;
    hipStream_t s = i2l_s(stream);
    if (max_tmp_px > 0) {
        int bx = (max_tmp_px + 255) / 256;
        if (bx > 1024) bx = 1024;
        hipLaunchKernelGGL(resize_h_kernel, dim3(bx, n), dim3(256), 0, s, pixels, plans, tables, static_cast<uint8_t*>(workspace), out_c);
        I2L_CHECK_LAUNCH();
    }
    int bx = (out_h * out_w + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(resize_v_norm_kernel, dim3(bx, n), dim3(256), 0, s, pixels, plans, tables,
                       static_cast<const uint8_t*>(workspace), out_c, out_h, out_w, normalize, out);
    I2L_CHECK_LAUNCH();
    return I2L_OK;
}
