// Input pipeline in front of the attack: Pillow-exact bicubic resize of an 8-bit RGB image + ToTensor + Normalize,
// written straight into the image's slot of the (B, 3, S, S) fp32 batch tensor (gfx950 / MI355X).
//
// Arithmetic = Pillow's src/libImaging/Resample.c 8-bit path: int32 fixed-point taps (22 fractional bits, computed
// on the host in double precision exactly like precompute_coeffs/normalize_coeffs_8bpc), horizontal pass, clip to
// uint8, vertical pass, clip to uint8; then (u8 / 255 - mean) / std in fp32.  Integer work, bit-exact by construction.
// Pass 1 reads the HWC source rows (coalesced along W*C) and writes a uint8 intermediate (H_in, W_out, C); pass 2 reads
// it column-wise per output row with lanes along W (coalesced) and writes planar fp32.  Both are tiny next to one PGD
// iteration (a 640x480 image is 0.9 MB); the host->device copy of the source dominates, hence the staging stream in
// vqattack_amd/preprocess.py.
#include "common.hpp"

namespace vqa {

constexpr int kPrecisionBits = 32 - 8 - 2;

__device__ __forceinline__ uint8_t clip8(int acc) {
  int v = acc >> kPrecisionBits;
  return static_cast<uint8_t>(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// dst[y, xo, c] = clip8( 2^21 + sum_k kk[xo, k] * src[y, xmin(xo) + k, c] )
__global__ __launch_bounds__(kBlock) void resize_h_kernel(const uint8_t* __restrict__ src, int h, int w_in, int c,
                                                          const int32_t* __restrict__ kk,
                                                          const int32_t* __restrict__ bounds, int ksize, int w_out,
                                                          uint8_t* __restrict__ dst) {
  const long total = static_cast<long>(h) * w_out * c;
  for (long i = static_cast<long>(blockIdx.x) * kBlock + threadIdx.x; i < total;
       i += static_cast<long>(gridDim.x) * kBlock) {
    const int ch = static_cast<int>(i % c);
    const int xo = static_cast<int>((i / c) % w_out);
    const long y = i / (static_cast<long>(c) * w_out);
    const int x0 = bounds[2 * xo], n = bounds[2 * xo + 1];
    const int32_t* k = kk + static_cast<long>(xo) * ksize;
    const uint8_t* row = src + (y * w_in + x0) * c + ch;
    int acc = 1 << (kPrecisionBits - 1);
    for (int t = 0; t < n; ++t) acc += static_cast<int>(row[static_cast<long>(t) * c]) * k[t];
    dst[i] = clip8(acc);
  }
}

// out[ch, yo, x] = (clip8( 2^21 + sum_k kk[yo, k] * src[ymin(yo) + k, x, ch] ) / 255 - mean) / std
// IDENTITY: no vertical resampling (H_in == H_out), only the conversion.
template <bool IDENTITY>
__global__ __launch_bounds__(kBlock) void resize_v_normalize_kernel(const uint8_t* __restrict__ src, int h_in, int w,
                                                                    int c, const int32_t* __restrict__ kk,
                                                                    const int32_t* __restrict__ bounds, int ksize,
                                                                    int h_out, float mean, float stdv,
                                                                    float* __restrict__ dst) {
  const long total = static_cast<long>(c) * h_out * w;
  for (long i = static_cast<long>(blockIdx.x) * kBlock + threadIdx.x; i < total;
       i += static_cast<long>(gridDim.x) * kBlock) {
    const int x = static_cast<int>(i % w);
    const int yo = static_cast<int>((i / w) % h_out);
    const int ch = static_cast<int>(i / (static_cast<long>(w) * h_out));
    uint8_t u;
    if (IDENTITY) {
      u = src[(static_cast<long>(yo) * w + x) * c + ch];
    } else {
      const int y0 = bounds[2 * yo], n = bounds[2 * yo + 1];
      const int32_t* k = kk + static_cast<long>(yo) * ksize;
      const uint8_t* col = src + (static_cast<long>(y0) * w + x) * c + ch;
      int acc = 1 << (kPrecisionBits - 1);
      for (int t = 0; t < n; ++t) acc += static_cast<int>(col[static_cast<long>(t) * w * c]) * k[t];
      u = clip8(acc);
    }
    const float v = static_cast<float>(u) / 255.0f;     // ToTensor
    dst[i] = (v - mean) / stdv;                         // Normalize
  }
}

}  // namespace vqa

using namespace vqa;

extern "C" {

int vqa_resize_bicubic_h_u8(const uint8_t* src, int h, int w_in, int c, const int32_t* kk, const int32_t* bounds,
                            int ksize, int w_out, uint8_t* dst, vqa_stream_t stream) {
  clear_stale_error();
  if (!src || !kk || !bounds || !dst) return VQA_ERR_NULL;
  if (h <= 0 || w_in <= 0 || c <= 0 || c > 4 || ksize <= 0 || w_out <= 0) return VQA_ERR_SHAPE;
  const long total = static_cast<long>(h) * w_out * c;
  resize_h_kernel<<<blocks_for(total, kBlock), kBlock, 0, static_cast<hipStream_t>(stream)>>>(src, h, w_in, c, kk, bounds,
                                                                                             ksize, w_out, dst);
  return launch_status();
}

int vqa_resize_bicubic_v_normalize(const uint8_t* src, int h_in, int w, int c, const int32_t* kk,
                                   const int32_t* bounds, int ksize, int h_out, float mean, float stdv, float* dst,
                                   vqa_stream_t stream) {
  clear_stale_error();
  if (!src || !dst) return VQA_ERR_NULL;
  if (h_in <= 0 || w <= 0 || c <= 0 || c > 4 || h_out <= 0 || stdv == 0.0f) return VQA_ERR_SHAPE;
  if (!aligned4(dst)) return VQA_ERR_ALIGN;
  const long total = static_cast<long>(c) * h_out * w;
  const int grid = blocks_for(total, kBlock);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!kk || !bounds) {
    if (h_in != h_out) return VQA_ERR_NULL;
    resize_v_normalize_kernel<true><<<grid, kBlock, 0, st>>>(src, h_in, w, c, nullptr, nullptr, 0, h_out, mean, stdv, dst);
  } else {
    if (ksize <= 0) return VQA_ERR_SHAPE;
    resize_v_normalize_kernel<false><<<grid, kBlock, 0, st>>>(src, h_in, w, c, kk, bounds, ksize, h_out, mean, stdv, dst);
  }
  return launch_status();
}

}  // extern "C"
