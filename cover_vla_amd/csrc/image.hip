// Image resampling on the device (the pre-processing either side of the hot path).
//
//   resample_axis_k   one separable pass out[o] = sum_j coef[o][j] * in[start[o] + j] along H or W of an HWC image, with the
//                     span table computed on the host by the restated filter-bank code (cover_vla_amd/imaging.py):
//                       FIXED = true   Pillow's 8-bit path (Resample.c: 22-bit fixed-point coefficients, accumulator seeded
//                                      with 1 << 21, clip8 of ss >> 22) -- what torchvision.transforms.Resize does to a PIL
//                                      image, i.e. open_clip's SigLIP2 preprocess (efficient_ensemble_merged.py:338);
//                       FIXED = false  TensorFlow's ScaleAndTranslate float path (rows first, then columns; sequential
//                                      fp32 accumulation in span order; the final uint8 cast truncates) -- tf.image.resize(
//                                      bilinear, antialias=True) of process_raw_image_to_jpg (eval_utils.py:273-283).
//   u8_to_chw_norm_k  ToTensor + Normalize: ((x / 255) - mean) / std in fp32, HWC uint8 -> CHW fp32.
//   bilinear_pad_k    torch.nn.functional.interpolate(mode="bilinear", align_corners=False) followed by the left/top
//                     padding of resize_with_pad (modeling_pi0.py:131-150), fp32 NCHW.
// All of it is byte / small-float work on <= 1.2 MB images: one thread per output element, coalesced along the channel-
// interleaved row; nothing here is worth LDS or MFMA.
#include <hip/hip_runtime.h>
#include "common.h"
#include "kernels.h"

// OpenCV's 8-bit fixed-point resize (imgproc resize.cpp, INTER_LANCZOS4 and the other non-area 8-bit paths): coefficients are
// shorts with INTER_RESIZE_COEF_BITS = 11 fractional bits, the first pass keeps exact int32 sums, the second applies
// FixedPtCast<int, uchar, 22>: saturate((sum + (1 << 21)) >> 22). LAST = false: uint8 -> int32; LAST = true: int32 -> uint8.
template <typename TIn, typename TOut, bool LAST>
__global__ __launch_bounds__(256) void resample_axis_cv_k(const TIn* __restrict__ in, TOut* __restrict__ out, int Hin, int Win, int C, int Hout,
                                                          int Wout, int axis, const int* __restrict__ bounds, const int* __restrict__ coefs, int ksize) {
    const long long total = (long long)Hout * Wout * C;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const int x = (int)((idx / C) % Wout);
    const int y = (int)(idx / ((long long)C * Wout));
    const int o = axis == 0 ? y : x;
    const int start = bounds[2 * o], cnt = bounds[2 * o + 1];
    const TIn* p = axis == 0 ? in + ((size_t)start * Win + x) * C + c : in + ((size_t)y * Win + start) * C + c;
    const size_t step = axis == 0 ? (size_t)Win * C : (size_t)C;
    const int* k = coefs + (size_t)o * ksize;
    int ss = 0;
    for (int j = 0; j < cnt; ++j) ss += (int)p[j * step] * k[j];
    if (LAST) {
        int v = (ss + (1 << 21)) >> 22;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        out[idx] = (TOut)v;
    } else {
        out[idx] = (TOut)ss;
    }
}

template <typename TIn, typename TOut, bool FIXED>
__global__ __launch_bounds__(256) void resample_axis_k(const TIn* __restrict__ in, TOut* __restrict__ out, int Hin, int Win, int C,
                                                       int Hout, int Wout, int axis, const int* __restrict__ bounds,
                                                       const void* __restrict__ coefs, int ksize, int trunc_u8) {
    const long long total = (long long)Hout * Wout * C;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const int x = (int)((idx / C) % Wout);
    const int y = (int)(idx / ((long long)C * Wout));
    const int o = axis == 0 ? y : x;
    const int start = bounds[2 * o], cnt = bounds[2 * o + 1];
    // element (start + j) along the resampled axis, same coordinate on the other one
    const TIn* p = axis == 0 ? in + ((size_t)start * Win + x) * C + c : in + ((size_t)y * Win + start) * C + c;
    const size_t step = axis == 0 ? (size_t)Win * C : (size_t)C;
    if (FIXED) {
        const int* k = (const int*)coefs + (size_t)o * ksize;
        int ss = 1 << 21;                                   // 1 << (PRECISION_BITS - 1)
        for (int j = 0; j < cnt; ++j) ss += (int)p[j * step] * k[j];
        int v = ss >> 22;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        out[idx] = (TOut)v;
    } else {
        // separate IEEE multiply and add, as the host form (numpy) evaluates them: HIP's __fmul_rn / __fadd_rn are plain
        // operators that hipcc contracts into v_fma_f32 (-ffp-contract=fast), which differs in the last bit
#pragma clang fp contract(off)
        const float* k = (const float*)coefs + (size_t)o * ksize;
        float acc = 0.0f;
        for (int j = 0; j < cnt; ++j) {
            const float prod = (float)p[j * step] * k[j];
            acc = acc + prod;
        }
        if (trunc_u8) {
            acc = acc < 0.f ? 0.f : (acc > 255.f ? 255.f : acc);
            out[idx] = (TOut)(int)acc;                      // static_cast<uint8>(float): truncation
        } else {
            out[idx] = (TOut)acc;
        }
    }
}

hipError_t launch_resample_axis(const void* in, int in_kind, void* out, int out_kind, int Hin, int Win, int C, int Hout, int Wout,
                                int axis, const int* bounds, const void* coefs, int ksize, int fixed, hipStream_t st) {
    const long long total = (long long)Hout * Wout * C;
    if (total <= 0) return hipSuccess;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    // kinds: 0 = uint8, 1 = float, 2 = int32 (OpenCV fixed-point mode only)
    if (fixed == 2) {
        if (in_kind == 0 && out_kind == 2)
            hipLaunchKernelGGL((resample_axis_cv_k<uint8_t, int, false>), grid, block, 0, st, (const uint8_t*)in, (int*)out, Hin, Win, C, Hout, Wout, axis,
                               bounds, (const int*)coefs, ksize);
        else if (in_kind == 2 && out_kind == 0)
            hipLaunchKernelGGL((resample_axis_cv_k<int, uint8_t, true>), grid, block, 0, st, (const int*)in, (uint8_t*)out, Hin, Win, C, Hout, Wout, axis,
                               bounds, (const int*)coefs, ksize);
        else return hipErrorInvalidValue;
    } else if (fixed) {
        if (in_kind != 0 || out_kind != 0) return hipErrorInvalidValue;
        hipLaunchKernelGGL((resample_axis_k<uint8_t, uint8_t, true>), grid, block, 0, st, (const uint8_t*)in, (uint8_t*)out, Hin, Win, C,
                           Hout, Wout, axis, bounds, coefs, ksize, 0);
    } else if (in_kind == 0 && out_kind == 1) {
        hipLaunchKernelGGL((resample_axis_k<uint8_t, float, false>), grid, block, 0, st, (const uint8_t*)in, (float*)out, Hin, Win, C,
                           Hout, Wout, axis, bounds, coefs, ksize, 0);
    } else if (in_kind == 1 && out_kind == 1) {
        hipLaunchKernelGGL((resample_axis_k<float, float, false>), grid, block, 0, st, (const float*)in, (float*)out, Hin, Win, C, Hout,
                           Wout, axis, bounds, coefs, ksize, 0);
    } else if (in_kind == 1 && out_kind == 0) {
        hipLaunchKernelGGL((resample_axis_k<float, uint8_t, false>), grid, block, 0, st, (const float*)in, (uint8_t*)out, Hin, Win, C,
                           Hout, Wout, axis, bounds, coefs, ksize, 1);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void u8_to_chw_norm_k(const uint8_t* __restrict__ in, float* __restrict__ out, int H, int W, int C,
                                                        float m0, float m1, float m2, float s0, float s1, float s2) {
#pragma clang fp contract(off)
    const long long total = (long long)H * W * C;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // CHW index: writes coalesced
    if (idx >= total) return;
    const int x = (int)(idx % W), y = (int)((idx / W) % H), c = (int)(idx / ((long long)W * H));
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    const float t = __fdiv_rn((float)in[((size_t)y * W + x) * C + c], 255.0f);   // ToTensor: .div(255)
    out[idx] = __fdiv_rn(__fsub_rn(t, mean), sd);                                 // Normalize: (t - mean) / std
}
// rescale-by-multiplication form: ((float)x * scale - mean) / std, the arithmetic of INT-ACT's process_images (src/utils/pipeline.py:55-67:
// `image * rescale_factor`, then (image - mean) / std in fp32)
__global__ __launch_bounds__(256) void u8_to_chw_scale_norm_k(const uint8_t* __restrict__ in, float* __restrict__ out, int H, int W, int C, float scale,
                                                              float m0, float m1, float m2, float s0, float s1, float s2) {
#pragma clang fp contract(off)
    const long long total = (long long)H * W * C;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % W), y = (int)((idx / W) % H), c = (int)(idx / ((long long)W * H));
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    // plain operators under `fp contract(off)`: __fmul_rn / __fsub_rn are inline header functions whose multiply and subtract hipcc
    // contracts into ONE v_fma_f32 (a single rounding: 1 ulp away from torch's separate multiply and subtract)
    const float t = (float)in[((size_t)y * W + x) * C + c] * scale;
    const float d = t - mean;
    out[idx] = d / sd;
}
hipError_t launch_u8_to_chw_scale_norm(const uint8_t* in, float* out, int H, int W, int C, float scale, const float* mean, const float* stdv, hipStream_t st) {
    if (C != 3) return hipErrorInvalidValue;
    const long long total = (long long)H * W * C;
    hipLaunchKernelGGL(u8_to_chw_scale_norm_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in, out, H, W, C, scale, mean[0], mean[1], mean[2],
                       stdv[0], stdv[1], stdv[2]);
    return hipGetLastError();
}
hipError_t launch_u8_to_chw_norm(const uint8_t* in, float* out, int H, int W, int C, const float* mean, const float* stdv, hipStream_t st) {
    if (C != 3) return hipErrorInvalidValue;
    const long long total = (long long)H * W * C;
    hipLaunchKernelGGL(u8_to_chw_norm_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in, out, H, W, C, mean[0], mean[1], mean[2],
                       stdv[0], stdv[1], stdv[2]);
    return hipGetLastError();
}

// out[n][c][pad_top + y][pad_left + x] = bilinear(in[n][c], y, x); everything above / left of the resized image = pad_value.
// Source index arithmetic of ATen's area_pixel_compute_source_index (align_corners = false): src = scale * (dst + 0.5) - 0.5,
// clamped at 0; i1 = i0 + (i0 < size - 1); value = h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11).
__global__ __launch_bounds__(256) void bilinear_pad_k(const float* __restrict__ in, float* __restrict__ out, int NC, int Hin, int Win,
                                                      int Hr, int Wr, int Hout, int Wout, int pad_top, int pad_left, float rh, float rw,
                                                      float pad_value) {
#pragma clang fp contract(off)
    const long long total = (long long)NC * Hout * Wout;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int X = (int)(idx % Wout), Y = (int)((idx / Wout) % Hout);
    const int nc = (int)(idx / ((long long)Wout * Hout));
    const int y = Y - pad_top, x = X - pad_left;
    if (y < 0 || x < 0 || y >= Hr || x >= Wr) { out[idx] = pad_value; return; }
    float sy = __fsub_rn(__fmul_rn(rh, (float)y + 0.5f), 0.5f), sx = __fsub_rn(__fmul_rn(rw, (float)x + 0.5f), 0.5f);
    sy = sy < 0.f ? 0.f : sy;
    sx = sx < 0.f ? 0.f : sx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int yp = y0 < Hin - 1 ? 1 : 0, xp = x0 < Win - 1 ? 1 : 0;
    const float h1 = sy - (float)y0, h0 = 1.0f - h1, w1 = sx - (float)x0, w0 = 1.0f - w1;
    const float* p = in + ((size_t)nc * Hin + y0) * Win + x0;
    const float top = __fadd_rn(__fmul_rn(w0, p[0]), __fmul_rn(w1, p[xp]));
    const float bot = __fadd_rn(__fmul_rn(w0, p[(size_t)yp * Win]), __fmul_rn(w1, p[(size_t)yp * Win + xp]));
    out[idx] = __fadd_rn(__fmul_rn(h0, top), __fmul_rn(h1, bot));
}
hipError_t launch_bilinear_pad(const float* in, float* out, int NC, int Hin, int Win, int Hr, int Wr, int Hout, int Wout, int pad_top,
                               int pad_left, float pad_value, hipStream_t st) {
    if (NC <= 0 || Hr <= 0 || Wr <= 0 || pad_top + Hr > Hout || pad_left + Wr > Wout) return hipErrorInvalidValue;
    const long long total = (long long)NC * Hout * Wout;
    const float rh = (float)Hin / (float)Hr, rw = (float)Win / (float)Wr;   // area_pixel_compute_scale: input_size / output_size
    hipLaunchKernelGGL(bilinear_pad_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in, out, NC, Hin, Win, Hr, Wr, Hout, Wout,
                       pad_top, pad_left, rh, rw, pad_value);
    return hipGetLastError();
}
