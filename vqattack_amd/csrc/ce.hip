// Masked-LM cross entropy over the 30522-word vocabulary: loss AND d loss / d logits in one launch (gfx950 / MI355X).
//
// One 256-thread workgroup per logits row (V = 30522 floats = 119 KB): sweep 1 streams the row from HBM with 8-byte
// loads and keeps a per-lane online (max, sum-exp); the 4 waves combine through DPP shuffles + LDS; sweep 2 re-reads
// the row -- L2-resident, the row is far below the 4 MB XCD L2 -- and writes the gradient
//     g_j = gscale * ( softmax_j * sum_k w_k  -  sum_k w_k [j == label_k] ),   w_k = 1/n_valid_k or 0 when ignored
// for ALL K label sets at once (the reference sums K separate F.cross_entropy calls over the same logits:
// A-ch/attacks/fast_gradient_method.py:136-139).  Algorithmic bytes: 4*V read + 4*V written per row (8*V), against
// ~6 full passes per label set for log_softmax + nll_loss + their autograd backward.
#include "common.hpp"

namespace vqa {

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int kWaves = kBlock / kWave;

// inv_count[k] = 1 / #{r : labels[k][r] != ignore}   (F.cross_entropy's mean over non-ignored targets)
__global__ __launch_bounds__(kBlock) void ce_count_kernel(const int64_t* __restrict__ labels, long rows,
                                                          long ignore_index, float* __restrict__ inv_count) {
  __shared__ float lds[kWaves];
  const int64_t* lab = labels + static_cast<long>(blockIdx.x) * rows;
  float c = 0.0f;
  for (long r = threadIdx.x; r < rows; r += kBlock) c += (lab[r] != ignore_index) ? 1.0f : 0.0f;
  c = wave_sum(c);
  if ((threadIdx.x & (kWave - 1)) == 0) lds[threadIdx.x / kWave] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.0f;
    for (int w = 0; w < kWaves; ++w) s += lds[w];
    inv_count[blockIdx.x] = 1.0f / s;     // all-ignored -> inf -> NaN loss, like torch's 0/0
  }
}

__device__ __forceinline__ void online_merge(float& m, float& s, float m2, float s2) {
  const float mm = fmaxf(m, m2);
  s = s * expf(m - mm) + s2 * expf(m2 - mm);
  m = mm;
}

template <bool GRAD, int MAXK>
__global__ __launch_bounds__(kBlock) void ce_rows_kernel(const float* __restrict__ logits, long row_stride,
                                                         const int64_t* __restrict__ labels, int K, long rows, int V,
                                                         long ignore_index, const float* __restrict__ inv_count,
                                                         float* __restrict__ grad, float* __restrict__ row_loss,
                                                         float gscale) {
  __shared__ float lds_m[kWaves], lds_s[kWaves];
  const long r = blockIdx.x;
  const float* x = logits + r * row_stride;
  const bool vec = ((reinterpret_cast<uintptr_t>(x) & 7u) == 0);
  // ---- sweep 1: online softmax statistics
  float m = -INFINITY, s = 0.0f;
  const int v2 = vec ? V / 2 : 0;
  for (int j = threadIdx.x; j < v2; j += kBlock) {
    f32x2 v = reinterpret_cast<const f32x2*>(x)[j];
    const float mm = fmaxf(m, fmaxf(v[0], v[1]));
    s = s * expf(m - mm) + expf(v[0] - mm) + expf(v[1] - mm);
    m = mm;
  }
  for (int j = 2 * v2 + threadIdx.x; j < V; j += kBlock) {
    const float v = x[j];
    const float mm = fmaxf(m, v);
    s = s * expf(m - mm) + expf(v - mm);
    m = mm;
  }
  if (s == 0.0f) m = -INFINITY;   // lanes that saw nothing
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float m2 = __shfl_xor(m, off, kWave), s2 = __shfl_xor(s, off, kWave);
    if (m2 != -INFINITY || m != -INFINITY) online_merge(m, s, m2, s2);
  }
  if ((threadIdx.x & (kWave - 1)) == 0) {
    lds_m[threadIdx.x / kWave] = m;
    lds_s[threadIdx.x / kWave] = s;
  }
  __syncthreads();
  m = lds_m[0];
  s = lds_s[0];
#pragma unroll
  for (int w = 1; w < kWaves; ++w)
    if (lds_m[w] != -INFINITY || m != -INFINITY) online_merge(m, s, lds_m[w], lds_s[w]);
  const float lse = m + logf(s);
  // ---- per-row label bookkeeping (K <= MAXK label sets)
  long lab[MAXK];
  float wk[MAXK];
  float wsum = 0.0f, loss = 0.0f;
#pragma unroll
  for (int k = 0; k < MAXK; ++k) {
    lab[k] = -1;
    wk[k] = 0.0f;
    if (k < K) {
      const long t = labels[static_cast<long>(k) * rows + r];
      if (t != ignore_index && t >= 0 && t < V) {
        lab[k] = t;
        wk[k] = inv_count[k];
        wsum += wk[k];
        loss += wk[k] * (lse - x[t]);
      }
    }
  }
  if (threadIdx.x == 0) row_loss[r] = loss;
  if (!GRAD) return;
  // ---- sweep 2: gradient (row re-read from L2)
  float* g = grad + r * static_cast<long>(V);
  const float inv_s = 1.0f / s;
  const bool gvec = vec && ((reinterpret_cast<uintptr_t>(g) & 7u) == 0);
  const int g2 = gvec ? V / 2 : 0;
  for (int j = threadIdx.x; j < g2; j += kBlock) {
    f32x2 v = reinterpret_cast<const f32x2*>(x)[j];
    f32x2 o;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      float val = wsum * (expf(v[e] - m) * inv_s);
#pragma unroll
      for (int k = 0; k < MAXK; ++k) val -= (lab[k] == 2 * j + e) ? wk[k] : 0.0f;
      o[e] = gscale * val;
    }
    __builtin_nontemporal_store(o, reinterpret_cast<f32x2*>(g) + j);
  }
  for (int j = 2 * g2 + threadIdx.x; j < V; j += kBlock) {
    float val = wsum * (expf(x[j] - m) * inv_s);
#pragma unroll
    for (int k = 0; k < MAXK; ++k) val -= (lab[k] == j) ? wk[k] : 0.0f;
    g[j] = gscale * val;
  }
}

}  // namespace vqa

using namespace vqa;

extern "C" {

int vqa_ce_max_label_sets(void) { return 8; }

int vqa_ce_rows(const float* logits, long row_stride, const int64_t* labels, int K, long rows, int V,
                long ignore_index, float* inv_count, float* grad, float* row_loss, float gscale,
                vqa_stream_t stream) {
  if (!logits || !labels || !inv_count || !row_loss) return VQA_ERR_NULL;
  if (K < 1 || K > 8 || rows < 0 || V <= 0 || row_stride < V) return VQA_ERR_SHAPE;
  if (!aligned4(logits) || (grad && !aligned4(grad))) return VQA_ERR_ALIGN;
  if (rows == 0) return VQA_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  ce_count_kernel<<<K, kBlock, 0, st>>>(labels, rows, ignore_index, inv_count);
  const int grid = static_cast<int>(rows);
  if (grad)
    ce_rows_kernel<true, 8><<<grid, kBlock, 0, st>>>(logits, row_stride, labels, K, rows, V, ignore_index, inv_count,
                                                     grad, row_loss, gscale);
  else
    ce_rows_kernel<false, 8><<<grid, kBlock, 0, st>>>(logits, row_stride, labels, K, rows, V, ignore_index, inv_count,
                                                      grad, row_loss, gscale);
  return launch_status();
}

}  // extern "C"
