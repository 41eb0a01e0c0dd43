// Image input producer on the device: prep_im_for_blob (lib/utils/blob.py:32-54) + im_list_to_blob (:16-29) for one
// frame — uint8 HxWx3 (cv2.imread order) -> float32 NHWC blob, resized by im_scale with bilinear interpolation,
// channels re-arranged, mean-subtracted and divided by the std-dev, optionally zero-padded to 4 channels so the stem
// convolution reads 16-byte pixels (first of the "next" rows; the reference does this on the CPU with
// cv2 + numpy per frame, lib/roi_data_layer/minibatch.py:518-676).
//
// cv2.resize(INTER_LINEAR) semantics restated (cv2 is not vendored: PARITY UNPINNED): output size = round-half-even
// of size*scale; source coordinate (dst + 0.5)/scale - 0.5 in fp32, floor, weight = fraction; indices below 0 /
// above size-1 clamp with weight 0; horizontal blend first, then vertical.  numpy then computes
// float32(float64(px) - mean) and float32(float64(.) / std).  HBM-bound elementwise kernel, one thread per pixel.
#include "common.h"

namespace {

struct PrepParams {
  double mean[3], stdv[3];
  int arrange[3];
};

__device__ __forceinline__ void resize_tap(int d, float inv_scale, int size, int& i0, int& i1, float& w0, float& w1) {
  float f = (float)(((double)d + 0.5) * (double)inv_scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { f = 0.f; s = 0; }
  if (s >= size - 1) { f = 0.f; s = size - 1; }
  i0 = s;
  i1 = min(s + 1, size - 1);
  w0 = 1.f - f;
  w1 = f;
}

__global__ __launch_bounds__(256) void prep_image_kernel(const uint8_t* __restrict__ img, int H, int W, int Ho, int Wo,
                                                        float inv_scale_x, float inv_scale_y, PrepParams prm, int c_out,
                                                        float* __restrict__ blob) {
  const int total = Ho * Wo;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int x = i % Wo, y = i / Wo;
    int x0, x1, y0, y1;
    float a0, a1, b0, b1;
    resize_tap(x, inv_scale_x, W, x0, x1, a0, a1);
    resize_tap(y, inv_scale_y, H, y0, y1, b0, b1);
    const uint8_t* r0 = img + (size_t)y0 * W * 3;
    const uint8_t* r1 = img + (size_t)y1 * W * 3;
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < 3; ++c) {
      const int sc = prm.arrange[c];   // im[:, :, pixel_arrange]
      const float top = (float)r0[x0 * 3 + sc] * a0 + (float)r0[x1 * 3 + sc] * a1;
      const float bot = (float)r1[x0 * 3 + sc] * a0 + (float)r1[x1 * 3 + sc] * a1;
      const float v = top * b0 + bot * b1;
      const float centred = (float)((double)v - prm.mean[c]);
      o[c] = (float)((double)centred / prm.stdv[c]);
    }
    float* dst = blob + (size_t)i * c_out;
    for (int c = 0; c < c_out; ++c) dst[c] = o[c];
  }
}

int round_half_even(double v) { return (int)nearbyint(v); }

}  // namespace

extern "C" int frcnn_prep_image_out_size(int h, int w, float scale, int* out_h, int* out_w) {
  FRCNN_REQUIRE(h > 0 && w > 0 && scale > 0.f && out_h && out_w, "prep_image_out_size: bad arguments");
  *out_h = round_half_even((double)h * (double)scale);
  *out_w = round_half_even((double)w * (double)scale);
  return FRCNN_OK;
}

extern "C" int frcnn_prep_image(const uint8_t* img_hwc3, int h, int w, float scale, const double* means_host,
                                const double* stddevs_host, const int* arrange_host, int c_out, float* blob,
                                void* stream_) {
  FRCNN_REQUIRE(img_hwc3 && means_host && stddevs_host && arrange_host && blob && h > 0 && w > 0 && scale > 0.f &&
                    (c_out == 3 || c_out == 4),
                "prep_image: bad arguments (c_out is 3 or 4)");
  PrepParams prm;
  for (int c = 0; c < 3; ++c) {
    prm.mean[c] = means_host[c];
    prm.stdv[c] = stddevs_host[c];
    prm.arrange[c] = arrange_host[c];
    FRCNN_REQUIRE(prm.arrange[c] >= 0 && prm.arrange[c] < 3 && prm.stdv[c] != 0.0, "prep_image: bad channel order / std-dev");
  }
  const int ho = round_half_even((double)h * (double)scale), wo = round_half_even((double)w * (double)scale);
  FRCNN_REQUIRE(ho > 0 && wo > 0, "prep_image: empty output");
  // cv2 derives the inverse scale from fx/fy themselves (1/fx), not from the rounded sizes
  const float inv = (float)(1.0 / (double)scale);
  hipLaunchKernelGGL(prep_image_kernel, dim3(std::min((ho * wo + 255) / 256, 4096)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), img_hwc3, h, w, ho, wo, inv, inv, prm, c_out, blob);
  return frcnn::check_launch("prep_image_kernel");
}
