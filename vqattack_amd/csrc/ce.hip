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

// Register-resident variant for V <= 32768 (the MLM vocabulary is 30522): the whole row is held as 32 float4 per
// lane, so the logits cross HBM exactly once and nothing is re-read, not even from L2; max and sum-exp are plain
// two-step block reductions (no serial online-rescale chain), exp() is evaluated once per element.
// Rows of 30522 floats start 8 bytes off a 16-byte boundary every other row, so a row is split into a <= 3 element
// head, a 16-byte aligned body of float4 and a <= 3 element tail (8-byte accesses run at 0.54-0.70x the 16-byte rate
// on this chip, MI355X_MICROARCH.md).
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kRegFloats = 32768;     // capacity of one workgroup's registers: THREADS * QUADS * 4
static int g_ce_threads = 256;        // vqa_set_option(4, 256 | 512 | 1024); 256 measured fastest at 2 WG/CU

// launch bounds: two workgroups per CU whatever THREADS is (2 * THREADS / 256 waves per SIMD), so that one row's
// reduce/exp phase overlaps the other's HBM phase; that caps the kernel at 64 VGPRs for THREADS = 1024
constexpr int ce_blocks_per_cu(int) { return 2; }   // 3 per CU forces spills at 256 threads: 181 us vs 130 us

template <bool GRAD, int MAXK, int THREADS>
__global__ __launch_bounds__(THREADS, ce_blocks_per_cu(THREADS) * THREADS / 256) void ce_rows_reg_kernel(const float* __restrict__ logits, long row_stride,
                                                             const int64_t* __restrict__ labels, int K, long rows,
                                                             int V, long ignore_index,
                                                             const float* __restrict__ inv_count,
                                                             float* __restrict__ grad, float* __restrict__ row_loss,
                                                             float gscale) {
  constexpr int kQuads = kRegFloats / 4 / THREADS;
  constexpr int kWavesT = THREADS / kWave;
  __shared__ float lds[kWavesT];
  const long r = blockIdx.x;
  const float* x = logits + r * row_stride;
  // logits base is 16-byte aligned (host check): alignment of the row start follows from its element offset
  const int head = static_cast<int>((4 - ((r * row_stride) & 3)) & 3);
  const int nquad = (V - head) / 4;
  const int tail0 = head + 4 * nquad;              // first tail element
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x + head);
  f32x4 v[kQuads];
#pragma unroll
  for (int i = 0; i < kQuads; ++i) {
    const int j = i * THREADS + threadIdx.x;
    v[i] = (j < nquad) ? x4[j] : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  }
  // the <= 6 leftover elements: lane t < head owns x[t]; lane 8 + t owns x[tail0 + t]
  int edge = -1;
  if (threadIdx.x < head) edge = threadIdx.x;
  else if (threadIdx.x >= 8 && threadIdx.x < 8 + (V - tail0)) edge = tail0 + (threadIdx.x - 8);
  float ve = (edge >= 0) ? x[edge] : -INFINITY;
  float m = ve;
#pragma unroll
  for (int i = 0; i < kQuads; ++i) m = fmaxf(m, fmaxf(fmaxf(v[i][0], v[i][1]), fmaxf(v[i][2], v[i][3])));
  m = wave_max(m);
  if ((threadIdx.x & (kWave - 1)) == 0) lds[threadIdx.x / kWave] = m;
  __syncthreads();
  m = lds[0];
#pragma unroll
  for (int w = 1; w < kWavesT; ++w) m = fmaxf(m, lds[w]);
  __syncthreads();
  ve = __expf(ve - m);                              // exp(-inf) = 0 for lanes without an edge element
  float s = ve;
#pragma unroll
  for (int i = 0; i < kQuads; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[i][e] = __expf(v[i][e] - m);
      s += v[i][e];
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & (kWave - 1)) == 0) lds[threadIdx.x / kWave] = s;
  __syncthreads();
  s = lds[0];
#pragma unroll
  for (int w = 1; w < kWavesT; ++w) s += lds[w];
  const float lse = m + logf(s);
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
  float* g = grad + r * static_cast<long>(V);       // (rows, V) contiguous: same head/tail split as the logits row
  const float c = wsum / s;
  if (edge >= 0) {
    float val = c * ve;
#pragma unroll
    for (int k = 0; k < MAXK; ++k) val -= (lab[k] == edge) ? wk[k] : 0.0f;
    g[edge] = gscale * val;
  }
  f32x4* g4 = reinterpret_cast<f32x4*>(g + head);
#pragma unroll
  for (int i = 0; i < kQuads; ++i) {
    const int j = i * THREADS + threadIdx.x;
    if (j < nquad) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float val = c * v[i][e];
#pragma unroll
        for (int k = 0; k < MAXK; ++k) val -= (lab[k] == head + 4 * j + e) ? wk[k] : 0.0f;
        o[e] = gscale * val;
      }
      __builtin_nontemporal_store(o, g4 + j);
    }
  }
}

}  // namespace vqa

using namespace vqa;

extern "C" {

int vqa_ce_max_label_sets(void) { return 8; }

int vqa_ce_set_threads(int threads) {   // reached through vqa_set_option(4, threads)
  if (threads != 256 && threads != 512 && threads != 1024) return VQA_ERR_SHAPE;
  g_ce_threads = threads;
  return VQA_OK;
}

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
  // register path: row_stride == V keeps the gradient row's alignment phase equal to the logits row's
  const bool reg_path = V >= 8 && V <= kRegFloats - 8 && row_stride == V && aligned16(logits) && (!grad || aligned16(grad));
  if (reg_path) {
#define VQA_CE_REG(G, MK, T) \
  ce_rows_reg_kernel<G, MK, T><<<grid, T, 0, st>>>(logits, row_stride, labels, K, rows, V, ignore_index, inv_count, \
                                                   grad, row_loss, gscale)
#define VQA_CE_K(G, T) \
  do { if (K == 1) VQA_CE_REG(G, 1, T); else if (K <= 4) VQA_CE_REG(G, 4, T); else VQA_CE_REG(G, 8, T); } while (0)
    if (g_ce_threads == 256) { if (grad) VQA_CE_K(true, 256); else VQA_CE_K(false, 256); }
    else if (g_ce_threads == 512) { if (grad) VQA_CE_K(true, 512); else VQA_CE_K(false, 512); }
    else { if (grad) VQA_CE_K(true, 1024); else VQA_CE_K(false, 1024); }
#undef VQA_CE_K
#undef VQA_CE_REG
    return launch_status();
  }
  if (grad)
    ce_rows_kernel<true, 8><<<grid, kBlock, 0, st>>>(logits, row_stride, labels, K, rows, V, ignore_index, inv_count,
                                                     grad, row_loss, gscale);
  else
    ce_rows_kernel<false, 8><<<grid, kBlock, 0, st>>>(logits, row_stride, labels, K, rows, V, ignore_index, inv_count,
                                                      grad, row_loss, gscale);
  return launch_status();
}

}  // extern "C"
