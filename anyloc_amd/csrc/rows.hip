// Row-wise HBM-bound kernels: L2 normalisation, LayerNorm, patch gather (im2col),
// CLS row, facet slice (+ F.normalize).  One 256-thread block per row, float4
// coalesced accesses; rows are re-read from L1/L2 for the second pass.
#include <cmath>

#include "common.hpp"

namespace anyloc {

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const float t = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return t;
}

// out[r,:] = x[r,:] / max(||x[r,:]||, eps)
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* x, int64_t ldx, float* out,
                                                          int64_t ldo, int64_t dim,
                                                          float eps) {
  __shared__ float red[4];
  const float* xr = x + (int64_t)blockIdx.x * ldx;
  float* orow = out + (int64_t)blockIdx.x * ldo;
  float ss = 0.f;
  if ((dim & 3) == 0 && (ldx & 3) == 0 && (ldo & 3) == 0) {
    const int64_t n4 = dim >> 2;
    for (int64_t i = threadIdx.x; i < n4; i += 256) {
      const f32x4 v = reinterpret_cast<const f32x4*>(xr)[i];
      ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    const float nrm = fmaxf(sqrtf(block_sum(ss, red)), eps);
    for (int64_t i = threadIdx.x; i < n4; i += 256) {
      f32x4 v = reinterpret_cast<const f32x4*>(xr)[i];
      v[0] /= nrm; v[1] /= nrm; v[2] /= nrm; v[3] /= nrm;
      reinterpret_cast<f32x4*>(orow)[i] = v;
    }
  } else {
    for (int64_t i = threadIdx.x; i < dim; i += 256) ss += xr[i] * xr[i];
    const float nrm = fmaxf(sqrtf(block_sum(ss, red)), eps);
    for (int64_t i = threadIdx.x; i < dim; i += 256) orow[i] = xr[i] / nrm;
  }
}

// y = (x - mean) / sqrt(var + eps) * w + b      (torch LayerNorm, biased variance)
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        int dim, float eps) {
  __shared__ float red[4];
  const f32x4* xr = reinterpret_cast<const f32x4*>(x + (int64_t)blockIdx.x * dim);
  f32x4* yr = reinterpret_cast<f32x4*>(y + (int64_t)blockIdx.x * dim);
  const int n4 = dim >> 2;
  float s = 0.f;
  for (int i = threadIdx.x; i < n4; i += 256) {
    const f32x4 v = xr[i];
    s += (v[0] + v[1]) + (v[2] + v[3]);
  }
  const float mean = block_sum(s, red) / (float)dim;
  float q = 0.f;
  for (int i = threadIdx.x; i < n4; i += 256) {
    const f32x4 v = xr[i];
    const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
    q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  }
  const float rstd = 1.0f / sqrtf(block_sum(q, red) / (float)dim + eps);
  for (int i = threadIdx.x; i < n4; i += 256) {
    const f32x4 v = xr[i];
    const f32x4 wv = reinterpret_cast<const f32x4*>(w)[i], bv = reinterpret_cast<const f32x4*>(b)[i];
    f32x4 o;
    o[0] = (v[0] - mean) * rstd * wv[0] + bv[0];
    o[1] = (v[1] - mean) * rstd * wv[1] + bv[1];
    o[2] = (v[2] - mean) * rstd * wv[2] + bv[2];
    o[3] = (v[3] - mean) * rstd * wv[3] + bv[3];
    yr[i] = o;
  }
}

// col[(b*Np + py*gw + px), c*P*P + i*P + j] = img[b, c, py*P + i, px*P + j]
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, float* __restrict__ col,
                                                     int H, int W, int P, int gh, int gw, int kpad) {
  const int64_t prow = blockIdx.x;            // b*Np + patch
  const int np = gh * gw;
  const int64_t b = prow / np;
  const int pi = (int)(prow - b * np), py = pi / gw, px = pi - py * gw;
  const int kk = 3 * P * P;
  float* dst = col + prow * kpad;
  for (int k = threadIdx.x; k < kpad; k += 256) {
    float v = 0.f;
    if (k < kk) {
      const int c = k / (P * P), rem = k - c * P * P, i = rem / P, j = rem - i * P;
      v = img[((b * 3 + c) * H + (py * P + i)) * (int64_t)W + (px * P + j)];
    }
    dst[k] = v;
  }
}

// x[b*T, :] = cls + pos[0, :]
__global__ __launch_bounds__(256) void cls_row_kernel(float* __restrict__ x, const float* __restrict__ cls,
                                                      const float* __restrict__ pos, int T, int dim) {
  float* dst = x + (int64_t)blockIdx.x * T * dim;
  for (int i = threadIdx.x; i < dim; i += 256) dst[i] = cls[i] + pos[i];
}

// out[b, n, ooff + d] = src[(b*T + skip + n), coff + d]   (optionally / max(||.||, eps))
__global__ __launch_bounds__(256) void facet_rows_kernel(const float* __restrict__ src, int64_t lds_, int coff,
                                                         float* __restrict__ out, int64_t ldo, int ooff, int T,
                                                         int skip, int rows_per_img, int dim, int normalize,
                                                         float eps) {
  __shared__ float red[4];
  const int64_t orow = blockIdx.x;
  const int64_t b = orow / rows_per_img;
  const int n = (int)(orow - b * rows_per_img);
  const f32x4* s = reinterpret_cast<const f32x4*>(src + (b * T + skip + n) * lds_ + coff);
  f32x4* o = reinterpret_cast<f32x4*>(out + orow * ldo + ooff);
  const int n4 = dim >> 2;
  float nrm = 1.0f;
  if (normalize) {
    float ss = 0.f;
    for (int i = threadIdx.x; i < n4; i += 256) {
      const f32x4 v = s[i];
      ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    nrm = fmaxf(sqrtf(block_sum(ss, red)), eps);
  }
  for (int i = threadIdx.x; i < n4; i += 256) {
    f32x4 v = s[i];
    if (normalize) { v[0] /= nrm; v[1] /= nrm; v[2] /= nrm; v[3] /= nrm; }
    o[i] = v;
  }
}

// uint8 HWC -> float CHW, centre crop, (x/255 - mean)/std : ToTensor + Normalize + CenterCrop
// (reference dvgl_benchmark/datasets_ws.py:20-23, scripts/dino_v2_vlad.py:173-176)
__global__ __launch_bounds__(256) void preprocess_u8_kernel(const unsigned char* __restrict__ img,
                                                            float* __restrict__ out, int H, int W, int y0, int x0,
                                                            int Ho, int Wo, float m0, float m1, float m2, float s0,
                                                            float s1, float s2) {
  const int64_t b = blockIdx.z;
  const int y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
  if (x >= Wo) return;
  const unsigned char* px = img + ((b * H + (y0 + y)) * (int64_t)W + (x0 + x)) * 3;
  float* o = out + (b * 3 * Ho + y) * (int64_t)Wo + x;
  const int64_t plane = (int64_t)Ho * Wo;
  o[0] = ((float)px[0] / 255.0f - m0) / s0;
  o[plane] = ((float)px[1] / 255.0f - m1) / s1;
  o[2 * plane] = ((float)px[2] / 255.0f - m2) / s2;
}

}  // namespace

int preprocess_u8(const unsigned char* img, float* out, int64_t batch, int H, int W, int Ho, int Wo, const float* mean,
                  const float* stdv, hipStream_t stream) {
  // torchvision center_crop: top = int(round((H - th) / 2.0)) with Python's round-half-to-even
  const int y0 = (int)nearbyint((H - Ho) / 2.0), x0 = (int)nearbyint((W - Wo) / 2.0);
  ProfScope prof("preprocess_u8", stream, 6.0 * batch * Ho * Wo, (double)batch * (3.0 * H * W + 12.0 * Ho * Wo));
  hipLaunchKernelGGL(preprocess_u8_kernel, dim3((Wo + 255) / 256, Ho, (unsigned)batch), dim3(256), 0, stream, img, out,
                     H, W, y0, x0, Ho, Wo, mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2]);
  return launch_status("preprocess_u8_kernel");
}

int l2norm_rows(const float* x, int64_t ldx, float* out, int64_t ldo, int64_t rows, int64_t dim, float eps,
                hipStream_t stream) {
  if (rows == 0 || dim == 0) return ANYLOC_OK;
  ANYLOC_CHECK_ARG(x && out, "l2norm_rows: null pointer");
  ANYLOC_CHECK_ARG(rows < (1ll << 31), "l2norm_rows: too many rows");
  ProfScope prof("l2norm_rows", stream, 3.0 * rows * dim, 8.0 * rows * dim);
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3((unsigned)rows), dim3(256), 0, stream, x, ldx, out, ldo, dim, eps);
  return launch_status("l2norm_rows_kernel");
}

int layernorm(const float* x, float* y, const float* w, const float* b, int64_t rows, int dim, float eps,
              hipStream_t stream) {
  ANYLOC_CHECK_ARG(dim % 4 == 0, "layernorm: dim %% 4 != 0");
  ProfScope prof("layernorm", stream, 8.0 * rows * dim, 8.0 * rows * dim);
  hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)rows), dim3(256), 0, stream, x, y, w, b, dim, eps);
  return launch_status("layernorm_kernel");
}

int im2col(const float* img, float* col, int64_t batch, int H, int W, int P, int kpad, hipStream_t stream) {
  const int gh = H / P, gw = W / P;
  ProfScope prof("im2col", stream, 0.0, 4.0 * batch * (3.0 * H * W + (double)gh * gw * kpad));
  hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)(batch * gh * gw)), dim3(256), 0, stream, img, col, H, W, P, gh,
                     gw, kpad);
  return launch_status("im2col_kernel");
}

int cls_rows(float* x, const float* cls, const float* pos, int64_t batch, int T, int dim, hipStream_t stream) {
  ProfScope prof("cls_rows", stream, 0.0, 12.0 * batch * dim);
  hipLaunchKernelGGL(cls_row_kernel, dim3((unsigned)batch), dim3(256), 0, stream, x, cls, pos, T, dim);
  return launch_status("cls_row_kernel");
}

int facet_rows(const float* src, int64_t lds_, int coff, float* out, int64_t ldo, int ooff, int64_t batch, int T,
               int skip, int rows_per_img, int dim, int normalize, float eps, hipStream_t stream) {
  ANYLOC_CHECK_ARG(dim % 4 == 0 && coff % 4 == 0 && ooff % 4 == 0 && lds_ % 4 == 0 && ldo % 4 == 0,
                   "facet_rows: alignment");
  ProfScope prof("facet_rows", stream, 3.0 * batch * rows_per_img * dim, 8.0 * batch * rows_per_img * dim);
  hipLaunchKernelGGL(facet_rows_kernel, dim3((unsigned)(batch * rows_per_img)), dim3(256), 0, stream, src, lds_, coff,
                     out, ldo, ooff, T, skip, rows_per_img, dim, normalize, eps);
  return launch_status("facet_rows_kernel");
}

}  // namespace anyloc

extern "C" int anyloc_preprocess_u8(const unsigned char* img_hwc, int64_t batch, int64_t height, int64_t width,
                                    int64_t crop_h, int64_t crop_w, const float* mean3, const float* std3,
                                    float* out, void* stream) {
  ANYLOC_CHECK_ARG(img_hwc && out && mean3 && std3, "preprocess_u8: null pointer");
  ANYLOC_CHECK_ARG(batch > 0 && batch < 65536 && height > 0 && width > 0 && height < 65536,
                   "preprocess_u8: bad batch/size");
  ANYLOC_CHECK_ARG(crop_h > 0 && crop_w > 0 && crop_h <= height && crop_w <= width,
                   "preprocess_u8: crop %lldx%lld outside image %lldx%lld", (long long)crop_h, (long long)crop_w,
                   (long long)height, (long long)width);
  return anyloc::preprocess_u8(img_hwc, out, batch, (int)height, (int)width, (int)crop_h, (int)crop_w, mean3, std3,
                               static_cast<hipStream_t>(stream));
}


namespace anyloc {
namespace {

// torch's bicubic kernel (A = -0.75), align_corners = False, no antialias: the convolution torchvision's
// resize(..., BICUBIC) applies to a float tensor (reference demo/anyloc_vlad_generate.py:175-176)
__device__ __forceinline__ void cubic_coeffs(float t, float c[4]) {
  const float A = -0.75f;
  float x = t + 1.0f;
  c[0] = ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A;
  x = t;
  c[1] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
  x = 1.0f - t;
  c[2] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
  x = 2.0f - t;
  c[3] = ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A;
}

// out[b,c,y,x] = resized[b,c,top+y,left+x], resized = bicubic(in[b,c], size (rh, rw)); one thread per output pixel
__global__ __launch_bounds__(256) void resize_bicubic_kernel(const float* __restrict__ in, int H, int W, int rh, int rw,
                                                             int top, int left, int ch, int cw, float* __restrict__ out) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= cw) return;
  const int64_t plane = blockIdx.z;
  const float sy = (float)H / (float)rh, sx = (float)W / (float)rw;
  const float fy = sy * ((float)(top + y) + 0.5f) - 0.5f, fx = sx * ((float)(left + x) + 0.5f) - 0.5f;
  const float y0f = floorf(fy), x0f = floorf(fx);
  float cy[4], cx[4];
  cubic_coeffs(fy - y0f, cy);
  cubic_coeffs(fx - x0f, cx);
  const int y0 = (int)y0f, x0 = (int)x0f;
  const float* src = in + plane * (int64_t)H * W;
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float* r = src + (int64_t)min(max(y0 - 1 + i, 0), H - 1) * W;
    float row = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) row += r[min(max(x0 - 1 + j, 0), W - 1)] * cx[j];
    acc += row * cy[i];
  }
  out[(plane * ch + y) * (int64_t)cw + x] = acc;
}

}  // namespace
}  // namespace anyloc

extern "C" int anyloc_resize_bicubic(const float* in, int64_t planes, int64_t height, int64_t width, int64_t out_h,
                                     int64_t out_w, int64_t crop_top, int64_t crop_left, int64_t crop_h, int64_t crop_w,
                                     float* out, void* stream) {
  ANYLOC_CHECK_ARG(in && out, "resize_bicubic: null pointer");
  ANYLOC_CHECK_ARG(planes > 0 && planes < 65536 && height > 0 && width > 0 && out_h > 0 && out_w > 0 && out_h < 65536,
                   "resize_bicubic: bad size");
  ANYLOC_CHECK_ARG(crop_top >= 0 && crop_left >= 0 && crop_h > 0 && crop_w > 0 && crop_top + crop_h <= out_h &&
                       crop_left + crop_w <= out_w, "resize_bicubic: crop window outside the resized image");
  hipStream_t s = static_cast<hipStream_t>(stream);
  anyloc::ProfScope prof("resize_bicubic", s, 40.0 * planes * crop_h * crop_w, 4.0 * planes * (height * width + crop_h * crop_w));
  hipLaunchKernelGGL(anyloc::resize_bicubic_kernel, dim3((unsigned)((crop_w + 255) / 256), (unsigned)crop_h, (unsigned)planes),
                     dim3(256), 0, s, in, (int)height, (int)width, (int)out_h, (int)out_w, (int)crop_top, (int)crop_left,
                     (int)crop_h, (int)crop_w, out);
  return anyloc::launch_status("resize_bicubic_kernel");
}

extern "C" int anyloc_l2norm_rows(const float* x, float* out, int64_t rows, int64_t dim, float eps, void* stream) {
  return anyloc::l2norm_rows(x, dim, out, dim, rows, dim, eps, static_cast<hipStream_t>(stream));
}
