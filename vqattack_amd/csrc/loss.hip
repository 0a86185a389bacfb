// Cross-modal feature-loss reduction: sum over rows of -cos(a_row, b_row), with the gradient w.r.t. `a`
// produced in the same pass (gfx950 / MI355X).
//
// One wavefront owns one row: D <= 2048 floats are held in registers as NCH 16-byte pieces per lane per
// operand (D = 768 -> 3, D = 1024 -> 4), so a and b are read from HBM exactly once and the gradient row is
// written once: 12*D bytes per row against 20*D for a separate forward and backward and far more for the
// reference's unfused chain (norm, clamp, div, mul, sum, neg, sum, sum + their autograd backward).
// dot / |a|^2 / |b|^2 are reduced across the 64 lanes with DPP shuffles; no LDS traffic except the 4-wave
// combine of the per-workgroup loss partial.  Row order per wave is fixed (grid-stride), partials are folded
// in index order by vqa_sum_partials, so the loss is bitwise reproducible for a given grid.
#include "common.hpp"

namespace vqa {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kLossBlocks = 256 * 8;          // workgroups (4 rows in flight each)
constexpr int kWavesPerBlock = kBlock / kWave;

struct RowAddr {
  long rows0, rows1;
  long a0, a1, b0, b1, g0, g1;
  long mask_period;
};

// Up to kMaxLayers (a, b, grad) base pointers passed BY VALUE in the kernel arguments: one launch covers every
// per-layer feature map of a LayerFeatures list (13 for the base encoders, 25 for VLMo-large) without packing them.
constexpr int kMaxLayers = 32;
struct LayerTable {
  const float* a[kMaxLayers];
  const float* b[kMaxLayers];
  float* g[kMaxLayers];
  int n;
};

template <int NCH, bool GRAD>
__global__ __launch_bounds__(kBlock) void neg_cos_rows_kernel(LayerTable tab, float* __restrict__ partial,
                                                              const uint8_t* __restrict__ row_mask, RowAddr ra,
                                                              int D, float gscale, float cos_eps) {
  __shared__ float lds[kWavesPerBlock];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const long per_layer = ra.rows0 * ra.rows1;
  const long total = per_layer * tab.n;
  const long wstride = static_cast<long>(gridDim.x) * kWavesPerBlock;
  float acc = 0.0f;   // this wave's sum of -cos over its rows (same value in every lane)
  for (long rr = static_cast<long>(blockIdx.x) * kWavesPerBlock + wave; rr < total; rr += wstride) {
    // wave-uniform by construction; readfirstlane makes it a scalar so the table lookup is one s_load, not a waterfall
    const int layer = __builtin_amdgcn_readfirstlane(static_cast<int>(rr / per_layer));
    const long r = rr - layer * per_layer;
    const float* __restrict__ a = tab.a[layer];
    const float* __restrict__ b = tab.b[layer];
    float* __restrict__ ga = tab.g[layer];
    const long o = r / ra.rows1, i = r - o * ra.rows1;
    const float w = row_mask ? static_cast<float>(row_mask[(o % ra.mask_period) * ra.rows1 + i]) : 1.0f;
    const bool live = w != 0.0f;   // wave-uniform: weight-0 rows (padded tokens) are never loaded
    const float* pa = a + o * ra.a0 + i * ra.a1;
    const float* pb = b + o * ra.b0 + i * ra.b1;
    f32x4 va[NCH], vb[NCH];
    if (live) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int d = (c * kWave + lane) * 4;
        if (d < D) {
          va[c] = *reinterpret_cast<const f32x4*>(pa + d);
          vb[c] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pb + d));
        } else {
          va[c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
          vb[c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
      }
    }
    float dot = 0.0f, na2 = 0.0f, nb2 = 0.0f;
    if (live) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          dot += va[c][k] * vb[c][k];
          na2 += va[c][k] * va[c][k];
          nb2 += vb[c][k] * vb[c][k];
        }
      }
    }
    dot = wave_sum(dot);
    na2 = wave_sum(na2);
    nb2 = wave_sum(nb2);
    const float na = sqrtf(na2), nb = sqrtf(nb2);
    const float dna = fmaxf(na, cos_eps), dnb = fmaxf(nb, cos_eps);
    const float inv = 1.0f / (dna * dnb);
    if (live) acc += -(w * (dot * inv));
    if (GRAD) {
      // d(-cos)/da = -( b/(dna*dnb) - [na > eps] * dot/(dna*dnb) * a/na^2 ), scaled by the upstream gscale
      const float gs = gscale * w;
      const float kb = -gs * inv;
      const float ka = (na > cos_eps) ? gs * dot * inv / (na * na) : 0.0f;
      float* pg = ga + o * ra.g0 + i * ra.g1;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int d = (c * kWave + lane) * 4;
        if (d < D) {
          f32x4 r;
          if (live) {
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = kb * vb[c][k] + ka * va[c][k];
          } else {
            r = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
          }
          *reinterpret_cast<f32x4*>(pg + d) = r;
        }
      }
    }
  }
  if (lane == 0) lds[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = lds[0];
#pragma unroll
    for (int w = 1; w < kWavesPerBlock; ++w) s += lds[w];
    partial[blockIdx.x] = s;
  }
}

__global__ __launch_bounds__(kBlock) void sum_partials_kernel(const float* __restrict__ partial, int count,
                                                              float* __restrict__ dst, int accumulate,
                                                              float scale) {
  __shared__ float lds[kWavesPerBlock];
  float acc = 0.0f;
  for (int i = threadIdx.x; i < count; i += kBlock) acc += partial[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & (kWave - 1)) == 0) lds[threadIdx.x / kWave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = lds[0];
#pragma unroll
    for (int w = 1; w < kWavesPerBlock; ++w) s += lds[w];
    s *= scale;
    dst[0] = accumulate ? dst[0] + s : s;
  }
}

template <int NCH>
static void launch_cos(bool grad, int grid, hipStream_t st, const LayerTable& tab, float* partial,
                       const uint8_t* mask, const RowAddr& ra, int D, float gscale, float eps) {
  if (grad)
    neg_cos_rows_kernel<NCH, true><<<grid, kBlock, 0, st>>>(tab, partial, mask, ra, D, gscale, eps);
  else
    neg_cos_rows_kernel<NCH, false><<<grid, kBlock, 0, st>>>(tab, partial, mask, ra, D, gscale, eps);
}

static int launch_cos_table(const LayerTable& tab, bool grad, float* partial, const uint8_t* row_mask,
                            const RowAddr& ra, int D, float gscale, float cos_eps, hipStream_t st) {
  // always launch the full grid: every partial slot is (re)written, so vqa_sum_partials can fold a fixed count
  const int grid = kLossBlocks;
  switch ((D + 255) / 256) {
    case 1: launch_cos<1>(grad, grid, st, tab, partial, row_mask, ra, D, gscale, cos_eps); break;
    case 2: launch_cos<2>(grad, grid, st, tab, partial, row_mask, ra, D, gscale, cos_eps); break;
    case 3: launch_cos<3>(grad, grid, st, tab, partial, row_mask, ra, D, gscale, cos_eps); break;
    case 4: launch_cos<4>(grad, grid, st, tab, partial, row_mask, ra, D, gscale, cos_eps); break;
    case 5: launch_cos<5>(grad, grid, st, tab, partial, row_mask, ra, D, gscale, cos_eps); break;
    case 6: launch_cos<6>(grad, grid, st, tab, partial, row_mask, ra, D, gscale, cos_eps); break;
    case 7: launch_cos<7>(grad, grid, st, tab, partial, row_mask, ra, D, gscale, cos_eps); break;
    default: launch_cos<8>(grad, grid, st, tab, partial, row_mask, ra, D, gscale, cos_eps); break;
  }
  return launch_status();
}

}  // namespace vqa

using namespace vqa;

extern "C" {

int vqa_neg_cos_partials(void) { return kLossBlocks; }

int vqa_neg_cos_rows(const float* a, const float* b, float* ga, float* partial, const uint8_t* row_mask,
                     long mask_period, long rows0, long rows1, int D, long a_stride0, long a_stride1,
                     long b_stride0, long b_stride1, long g_stride0, long g_stride1, float gscale, float cos_eps,
                     vqa_stream_t stream) {
  if (!a || !b || !partial) return VQA_ERR_NULL;
  if (rows0 < 0 || rows1 < 0 || D <= 0 || D > 2048 || (D & 3)) return VQA_ERR_SHAPE;
  if ((a_stride0 | a_stride1 | b_stride0 | b_stride1) & 3) return VQA_ERR_SHAPE;
  if (ga && ((g_stride0 | g_stride1) & 3)) return VQA_ERR_SHAPE;
  if (row_mask && mask_period <= 0) return VQA_ERR_SHAPE;
  if (!aligned16(a) || !aligned16(b) || (ga && !aligned16(ga))) return VQA_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  RowAddr ra{rows0, rows1 > 0 ? rows1 : 1, a_stride0, a_stride1, b_stride0, b_stride1, g_stride0, g_stride1,
             row_mask ? mask_period : 1};
  if (rows1 == 0) ra.rows0 = 0;
  LayerTable tab{};
  tab.a[0] = a;
  tab.b[0] = b;
  tab.g[0] = ga;
  tab.n = 1;
  return launch_cos_table(tab, ga != nullptr, partial, row_mask, ra, D, gscale, cos_eps, st);
}

int vqa_neg_cos_max_layers(void) { return kMaxLayers; }

int vqa_neg_cos_rows_multi(const float* const* a, const float* const* b, float* const* ga, int n_layers,
                           float* partial, const uint8_t* row_mask, long mask_period, long rows0, long rows1, int D,
                           long a_stride0, long a_stride1, long b_stride0, long b_stride1, long g_stride0,
                           long g_stride1, float gscale, float cos_eps, vqa_stream_t stream) {
  if (!a || !b || !partial) return VQA_ERR_NULL;
  if (n_layers < 1 || n_layers > kMaxLayers) return VQA_ERR_SHAPE;
  if (rows0 < 0 || rows1 < 0 || D <= 0 || D > 2048 || (D & 3)) return VQA_ERR_SHAPE;
  if ((a_stride0 | a_stride1 | b_stride0 | b_stride1) & 3) return VQA_ERR_SHAPE;
  if (ga && ((g_stride0 | g_stride1) & 3)) return VQA_ERR_SHAPE;
  if (row_mask && mask_period <= 0) return VQA_ERR_SHAPE;
  LayerTable tab{};
  tab.n = n_layers;
  for (int l = 0; l < n_layers; ++l) {
    if (!a[l] || !b[l] || (ga && !ga[l])) return VQA_ERR_NULL;
    if (!aligned16(a[l]) || !aligned16(b[l]) || (ga && !aligned16(ga[l]))) return VQA_ERR_ALIGN;
    tab.a[l] = a[l];
    tab.b[l] = b[l];
    tab.g[l] = ga ? ga[l] : nullptr;
  }
  RowAddr ra{rows0, rows1 > 0 ? rows1 : 1, a_stride0, a_stride1, b_stride0, b_stride1, g_stride0, g_stride1,
             row_mask ? mask_period : 1};
  if (rows1 == 0) ra.rows0 = 0;
  return launch_cos_table(tab, ga != nullptr, partial, row_mask, ra, D, gscale, cos_eps,
                          static_cast<hipStream_t>(stream));
}

int vqa_sum_partials(const float* partial, int count, float* dst, int accumulate, float scale,
                     vqa_stream_t stream) {
  if (!partial || !dst) return VQA_ERR_NULL;
  if (count < 0) return VQA_ERR_SHAPE;
  sum_partials_kernel<<<1, kBlock, 0, static_cast<hipStream_t>(stream)>>>(partial, count, dst, accumulate, scale);
  return launch_status();
}

}  // extern "C"
