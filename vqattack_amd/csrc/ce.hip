// Masked-LM cross entropy over the 30522-word vocabulary: loss AND d loss / d logits in one launch (gfx950 / MI355X).
//
// Register-resident kernel (ce_rows_reg_kernel, V <= 32760): one 256-thread workgroup per logits row keeps the whole
// row in registers (32 float4 per lane), so the logits cross HBM exactly once and nothing is re-read, not even from
// L2; max and sum-exp are two-step block reductions (DPP within a wave, LDS across the 4 waves), exp() is evaluated
// once per element, and the gradient
//     g_j = gscale * ( softmax_j * sum_k w_k  -  sum_k w_k [j == label_k] ),   w_k = 1/n_valid_k or 0 when ignored
// is written for ALL K label sets at once (the reference sums K separate F.cross_entropy calls over the same logits:
// A-ch/attacks/fast_gradient_method.py:136-139).  The workgroup that finishes last folds the per-row losses in row
// order into the scalar loss (relaxed two-level arrival counters, common.hpp): no second launch.
// exp is the hardware v_exp_f32 path (__expf, ~2 ulp); the tests state 1e-4 relative against torch.
// A label that is neither ignore_index nor in [0, V) poisons the loss with NaN and ORs VQA_FLAG_BAD_LABEL into *flag
// (torch device-asserts on such a label; it is never silently dropped).
// Algorithmic bytes: 4*V read + 4*V written per live row (8*V); a dead row (all labels ignore_index: zero gradient,
// logits untouched) costs 4*V written -- or NOTHING when the caller keeps the gradient buffer across launches and passes
// `row_state`: one byte per row that remembers whether the row's gradient in that buffer is non-zero, so a dead row is
// zeroed only if an earlier launch left a live gradient there (in an attack the labels are fixed: never) -- against ~6
// full passes per label set for log_softmax + nll_loss + their autograd backward.
#include "common.hpp"

extern "C" int vqa_sum_partials(const float* partial, int count, float* dst, int accumulate, float scale,
                                vqa_stream_t stream);   // loss.hip

namespace vqa {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kWaves = kBlock / kWave;

constexpr int kCeMaxK = 8;
// scratch layout: the fold's arrival counters (kArriveWords), then K x groups reciprocal valid-label counts

// Normalisation of the mean: rows are split into `groups` consecutive groups of `rpg` rows (one group = one SAMPLE of
// a batched attack; groups == 1 is F.cross_entropy's mean over the whole batch) and label set k of group g is divided
// by its own number of non-ignored labels -- the batch-1 reference's loss, summed over the samples of the batch.
struct CeNorm {
  const float* inv_count;   // [K][groups]
  long rpg, groups;
};

// inv_count[k][g] = 1 / #{r in group g : labels[k][r] != ignore}   (F.cross_entropy's mean over non-ignored targets);
// block 0 also zeroes the arrival counters of the in-kernel loss fold.  One workgroup per (label set, group).
__global__ __launch_bounds__(kBlock) void ce_count_kernel(const int64_t* __restrict__ labels, long rows,
                                                          long ignore_index, unsigned* __restrict__ counters,
                                                          float* __restrict__ inv_count, long rpg, long groups) {
  __shared__ float lds[kWaves];
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < kArriveWords; i += kBlock) counters[i] = 0u;
  const long k = blockIdx.x / groups, g = blockIdx.x - k * groups;
  const int64_t* lab = labels + k * rows;
  const long r1 = (g + 1) * rpg < rows ? (g + 1) * rpg : rows;
  float c = 0.0f;
  for (long r = g * rpg + threadIdx.x; r < r1; r += kBlock) c += (lab[r] != ignore_index) ? 1.0f : 0.0f;
  c = wave_sum(c);
  if ((threadIdx.x & (kWave - 1)) == 0) lds[threadIdx.x / kWave] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.0f;
    for (int w = 0; w < kWaves; ++w) s += lds[w];
    inv_count[blockIdx.x] = 1.0f / s;     // all-ignored -> inf, never multiplied in: such a (set, group) adds 0
  }
}

// Per-row label bookkeeping shared by the kernels: weights of the K label sets, their sum, the row's loss.
template <int MAXK>
struct RowLabels {
  long lab[MAXK];
  float wk[MAXK];
  float wsum, loss;
  bool bad;
};

template <int MAXK>
__device__ __forceinline__ RowLabels<MAXK> read_labels(const int64_t* __restrict__ labels, int K, long rows, long r,
                                                       int V, long ignore_index, const CeNorm& norm,
                                                       const float* __restrict__ x, float lse) {
  RowLabels<MAXK> L;
  L.wsum = 0.0f;
  L.loss = 0.0f;
  L.bad = false;
#pragma unroll
  for (int k = 0; k < MAXK; ++k) {
    L.lab[k] = -1;
    L.wk[k] = 0.0f;
    if (k < K) {
      const long t = labels[static_cast<long>(k) * rows + r];
      if (t != ignore_index) {
        if (t >= 0 && t < V) {
          L.lab[k] = t;
          L.wk[k] = norm.inv_count[k * norm.groups + r / norm.rpg];
          L.wsum += L.wk[k];
          L.loss += L.wk[k] * (lse - x[t]);
        } else {
          L.bad = true;
        }
      }
    }
  }
  if (L.bad) L.loss = __builtin_nanf("");
  return L;
}

// A row none of whose K labels is a target (all ignore_index) has loss 0 and an identically zero gradient whatever its
// logits are.  In the reference's workload that is every position except the [MASK]-ed answer pieces
// (adv_attack.py:433-558: labels are -100 elsewhere), i.e. >= 90 % of the B x L rows of a dense (B, L, V) logits tensor:
// such a row's logits are never loaded and no exponential is evaluated -- it costs its zero gradient store only
// (4 V bytes instead of 8 V).  The labels of a row are the same for the whole workgroup: the branch is scalar.
__device__ __forceinline__ bool row_is_dead(const int64_t* __restrict__ labels, int K, long rows, long r,
                                            long ignore_index) {
  bool live = false;
  for (int k = 0; k < K; ++k) live |= (labels[static_cast<long>(k) * rows + r] != ignore_index);
  return !live;
}

struct CeFold {          // in-kernel fold of row_loss into the scalar loss (loss_out == NULL: row losses only)
  unsigned* counter;
  float* loss_out;
  int accumulate;
  float scale;
};

// Called by every thread of every workgroup at the end of a kernel whose thread 0 wrote row_loss with agent-scope
// stores: the workgroup that arrives last sums row_loss[0 .. rows) in row order.
template <int THREADS>
__device__ __forceinline__ void fold_row_losses(const CeFold& fold, const float* row_loss, long rows) {
  if (!fold.loss_out) return;
  __shared__ int lds_last;
  __shared__ float lds_f[THREADS / kWave];
  if (threadIdx.x == 0) lds_last = arrive_is_last(fold.counter) ? 1 : 0;   // row losses were stored write-through
  __syncthreads();
  if (!lds_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  float acc = 0.0f;
  for (long i = threadIdx.x; i < rows; i += THREADS)
    acc += __hip_atomic_load(row_loss + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  acc = wave_sum(acc);
  if ((threadIdx.x & (kWave - 1)) == 0) lds_f[threadIdx.x / kWave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = lds_f[0];
#pragma unroll
    for (int w = 1; w < THREADS / kWave; ++w) s += lds_f[w];
    s *= fold.scale;
    fold.loss_out[0] = fold.accumulate ? fold.loss_out[0] + s : s;
  }
}

__device__ __forceinline__ void online_merge(float& m, float& s, float m2, float s2) {
  const float mm = fmaxf(m, m2);
  s = s * expf(m - mm) + s2 * expf(m2 - mm);
  m = mm;
}

// ---- streaming fallback (any V, strided / unaligned rows): two sweeps, the second re-reads the row from L2 ----
template <bool GRAD, int MAXK>
__global__ __launch_bounds__(kBlock) void ce_rows_kernel(const float* __restrict__ logits, long row_stride,
                                                         const int64_t* __restrict__ labels, int K, long rows, int V,
                                                         long ignore_index, CeNorm norm,
                                                         float* __restrict__ grad, float* __restrict__ row_loss,
                                                         float gscale, int* __restrict__ flag,
                                                         unsigned char* __restrict__ row_state) {
  __shared__ float lds_m[kWaves], lds_s[kWaves];
  const long r = blockIdx.x;
  if (row_is_dead(labels, K, rows, r, ignore_index)) {       // nothing to learn from this row: zero loss, zero gradient
    if (threadIdx.x == 0) row_loss[r] = 0.0f;
    if (GRAD && (!row_state || row_state[r])) {              // workgroup-uniform: the buffer holds a stale live gradient
      float* g = grad + r * static_cast<long>(V);
      for (int j = threadIdx.x; j < V; j += kBlock) g[j] = 0.0f;
      __syncthreads();                                       // every thread has read row_state[r]
      if (row_state && threadIdx.x == 0) row_state[r] = 0;
    }
    return;
  }
  if (GRAD && row_state && threadIdx.x == 0) row_state[r] = 1;
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
  const RowLabels<MAXK> L = read_labels<MAXK>(labels, K, rows, r, V, ignore_index, norm, x, lse);
  if (threadIdx.x == 0) {
    row_loss[r] = L.loss;
    if (L.bad && flag) atomicOr(flag, VQA_FLAG_BAD_LABEL);
  }
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
      float val = L.wsum * (expf(v[e] - m) * inv_s);
#pragma unroll
      for (int k = 0; k < MAXK; ++k) val -= (L.lab[k] == 2 * j + e) ? L.wk[k] : 0.0f;
      o[e] = gscale * val;
    }
    __builtin_nontemporal_store(o, reinterpret_cast<f32x2*>(g) + j);
  }
  for (int j = 2 * g2 + threadIdx.x; j < V; j += kBlock) {
    float val = L.wsum * (expf(x[j] - m) * inv_s);
#pragma unroll
    for (int k = 0; k < MAXK; ++k) val -= (L.lab[k] == j) ? L.wk[k] : 0.0f;
    g[j] = gscale * val;
  }
}

// ---- register-resident persistent kernel ----------------------------------------------------------------------
// Rows of 30522 floats start 8 bytes off a 16-byte boundary every other row, so a row is split into a <= 3 element
// head, a 16-byte aligned body of float4 and a <= 3 element tail (8-byte accesses run at 0.54-0.70x the 16-byte rate
// on this chip, MI355X_MICROARCH.md).  Lane t < head owns x[t]; lane 8 + t owns tail element t.
constexpr int kRegFloats = 32768;     // capacity of one workgroup's registers: THREADS * QUADS * 4
VQA_KNOB g_ce_threads = 512;        // option 4 (256 | 512 | 1024); 512 measured 2-3 % ahead of 256 (profiles/r02)
VQA_KNOB g_ce_variant = 2;          // option 5 (2 | 3): 2 = size-aware non-temporal logits loads, 3 = always

struct RowGeom {
  const f32x4* x4;      // aligned body
  const float* x;       // row start
  int head, nquad, tail0;
};

__device__ __forceinline__ RowGeom row_geom(const float* logits, long r, int V) {
  RowGeom g;
  // logits base is 16-byte aligned and rows are dense (host check): the alignment phase follows from the offset
  g.head = static_cast<int>((4 - ((r * V) & 3)) & 3);
  g.nquad = (V - g.head) / 4;
  g.tail0 = g.head + 4 * g.nquad;
  g.x = logits + r * V;
  g.x4 = reinterpret_cast<const f32x4*>(g.x + g.head);
  return g;
}

__device__ __forceinline__ int edge_index(const RowGeom& g, int V) {
  if (static_cast<int>(threadIdx.x) < g.head) return threadIdx.x;
  if (threadIdx.x >= 8 && static_cast<int>(threadIdx.x) < 8 + (V - g.tail0)) return g.tail0 + (threadIdx.x - 8);
  return -1;
}

// One workgroup per row; WGPC workgroups are resident per CU (launch bounds), so one row's reduce / exp phase overlaps
// the HBM phases of the others.  (A persistent variant that streamed row r's gradient out while loading row r + grid
// into the same registers measured 5-8 % slower: the loads then issue behind the store's VALU work instead of in one
// burst at workgroup start -- profiles/r02/kernel_roofline_ab_b64.jsonl; three workgroups per CU at a 168-VGPR budget
// measured 5-7 % slower than two -- profiles/r02/kernel_roofline_ab2_b64.jsonl.)
template <bool GRAD, int MAXK, int THREADS, int WGPC, bool NTL>
__global__ __launch_bounds__(THREADS, WGPC * THREADS / 256) void ce_rows_reg_kernel(
    const float* __restrict__ logits, const int64_t* __restrict__ labels, int K, long rows, int V, long ignore_index,
    CeNorm norm, float* __restrict__ grad, float* __restrict__ row_loss, float gscale,
    int* __restrict__ flag, CeFold fold, unsigned char* __restrict__ row_state) {
  constexpr int kQuads = kRegFloats / 4 / THREADS;
  constexpr int kWavesT = THREADS / kWave;
  __shared__ float lds[kWavesT];
  const long r = blockIdx.x;
  const RowGeom cur = row_geom(logits, r, V);
  if (row_is_dead(labels, K, rows, r, ignore_index)) {       // workgroup-uniform: no logits load, no exp, zeros out
    if (threadIdx.x == 0) __hip_atomic_store(row_loss + r, 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // row_state given: the gradient buffer persists across launches and this row's slice is known to be zero unless an
    // earlier launch wrote a live gradient there -> nothing is stored (workgroup-uniform branch)
    if (GRAD && (!row_state || row_state[r])) {
      float* g = grad + r * static_cast<long>(V);
      const int edge = edge_index(cur, V);
      if (edge >= 0) g[edge] = 0.0f;
      f32x4* g4 = reinterpret_cast<f32x4*>(g + cur.head);
      const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int i = 0; i < kQuads; ++i) {
        const int j = i * THREADS + threadIdx.x;
        if (j < cur.nquad) __builtin_nontemporal_store(zero, g4 + j);
      }
      __syncthreads();                                       // every thread has read row_state[r]
      if (row_state && threadIdx.x == 0) row_state[r] = 0;
    }
    fold_row_losses<THREADS>(fold, row_loss, rows);
    return;
  }
  if (GRAD && row_state && threadIdx.x == 0) row_state[r] = 1;   // this launch leaves a live gradient in the row
  f32x4 v[kQuads];
#pragma unroll
  for (int i = 0; i < kQuads; ++i) {
    const int j = i * THREADS + threadIdx.x;
    v[i] = (j < cur.nquad) ? (NTL ? __builtin_nontemporal_load(cur.x4 + j) : cur.x4[j])
                           : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  }
  const int edge = edge_index(cur, V);
  float ve = (edge >= 0) ? cur.x[edge] : -INFINITY;
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
  const RowLabels<MAXK> L = read_labels<MAXK>(labels, K, rows, r, V, ignore_index, norm, cur.x, lse);
  if (threadIdx.x == 0) {
    __hip_atomic_store(row_loss + r, L.loss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (L.bad && flag) atomicOr(flag, VQA_FLAG_BAD_LABEL);
  }
  if (GRAD) {
    float* g = grad + r * static_cast<long>(V);
    const float c = L.wsum / s;
    if (edge >= 0) {
      float val = c * ve;
#pragma unroll
      for (int k = 0; k < MAXK; ++k) val -= (L.lab[k] == edge) ? L.wk[k] : 0.0f;
      g[edge] = gscale * val;
    }
    f32x4* g4 = reinterpret_cast<f32x4*>(g + cur.head);
#pragma unroll
    for (int i = 0; i < kQuads; ++i) {
      const int j = i * THREADS + threadIdx.x;
      if (j < cur.nquad) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float val = c * v[i][e];
#pragma unroll
          for (int k = 0; k < MAXK; ++k) val -= (L.lab[k] == cur.head + 4 * j + e) ? L.wk[k] : 0.0f;
          o[e] = gscale * val;
        }
        __builtin_nontemporal_store(o, g4 + j);
      }
    }
  }
  fold_row_losses<THREADS>(fold, row_loss, rows);
}

template <bool GRAD, int MAXK, int THREADS>
static void launch_reg(int variant, long rows, hipStream_t st, const float* logits, const int64_t* labels, int K, int V,
                       long ignore_index, const CeNorm& norm, float* grad, float* row_loss, float gscale,
                       int* flag, const CeFold& fold, unsigned char* row_state) {
  // logits are read exactly once: beyond twice the 256 MB Infinity Cache a non-temporal load is 5 % ahead (2.5 GB:
  // 474 vs 497 us), below it the plain load still finds part of the producer GEMM's output in the cache (625 MB: 119
  // vs 122 us) -- profiles/r02/kernel_roofline_nt_b{64,256}.jsonl.  variant 3 forces nt (A/B).
  const bool big = static_cast<size_t>(rows) * V * sizeof(float) > (512ull << 20);
  if (variant == 3 || big)
    ce_rows_reg_kernel<GRAD, MAXK, THREADS, 2, true><<<static_cast<int>(rows), THREADS, 0, st>>>(
        logits, labels, K, rows, V, ignore_index, norm, grad, row_loss, gscale, flag, fold, row_state);
  else
    ce_rows_reg_kernel<GRAD, MAXK, THREADS, 2, false><<<static_cast<int>(rows), THREADS, 0, st>>>(
        logits, labels, K, rows, V, ignore_index, norm, grad, row_loss, gscale, flag, fold, row_state);
}

template <int THREADS>
static void launch_reg_k(bool want_grad, int variant, long rows, hipStream_t st, const float* logits,
                         const int64_t* labels, int K, int V, long ignore_index, const CeNorm& norm, float* grad,
                         float* row_loss, float gscale, int* flag, const CeFold& fold, unsigned char* row_state) {
#define VQA_CE_GO(G, MK) \
  launch_reg<G, MK, THREADS>(variant, rows, st, logits, labels, K, V, ignore_index, norm, grad, row_loss, gscale, \
                             flag, fold, row_state)
  if (want_grad) {
    if (K == 1) VQA_CE_GO(true, 1); else if (K <= 4) VQA_CE_GO(true, 4); else VQA_CE_GO(true, 8);
  } else {
    if (K == 1) VQA_CE_GO(false, 1); else if (K <= 4) VQA_CE_GO(false, 4); else VQA_CE_GO(false, 8);
  }
#undef VQA_CE_GO
}

}  // namespace vqa

using namespace vqa;

extern "C" {

int vqa_ce_max_label_sets(void) { return kCeMaxK; }

long vqa_ce_scratch_floats(int K, long groups) {
  if (K < 1 || groups < 1) return 0;
  return kArriveWords + static_cast<long>(K) * groups;
}

#ifdef VQA_TUNING
int vqa_ce_set_threads(int threads) {   // reached through vqa_set_option(4, threads)
  if (threads != 256 && threads != 512 && threads != 1024) return VQA_ERR_SHAPE;
  g_ce_threads = threads;
  return VQA_OK;
}

int vqa_ce_set_variant(int variant) {   // reached through vqa_set_option(5, variant)
  if (variant != 2 && variant != 3) return VQA_ERR_SHAPE;
  g_ce_variant = variant;
  return VQA_OK;
}
#endif  // VQA_TUNING

int vqa_ce_rows(const float* logits, long row_stride, const int64_t* labels, int K, long rows, int V,
                long ignore_index, long rows_per_group, float* scratch, float* grad, float* row_loss, float gscale,
                float* loss_out, int accumulate, int* flag, unsigned char* row_state, vqa_stream_t stream) {
  clear_stale_error();
  if (!logits || !labels || !scratch || !row_loss) return VQA_ERR_NULL;
  if (row_state && !grad) return VQA_ERR_NULL;               // row_state describes the gradient buffer
  if (K < 1 || K > kCeMaxK || rows < 0 || V <= 0 || row_stride < V || rows_per_group < 0) return VQA_ERR_SHAPE;
  const long rpg = (rows_per_group == 0 || rows_per_group > rows) ? (rows > 0 ? rows : 1) : rows_per_group;
  const long groups = rows > 0 ? (rows + rpg - 1) / rpg : 1;
  if (static_cast<long>(K) * groups > 0x7fffffffL) return VQA_ERR_SHAPE;
  if (!aligned4(logits) || (grad && !aligned4(grad)) || !aligned4(scratch)) return VQA_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (rows == 0) return loss_out ? vqa_sum_partials(row_loss, 0, loss_out, accumulate, gscale, stream) : VQA_OK;
  unsigned* counters = reinterpret_cast<unsigned*>(scratch);
  float* inv_count = scratch + kArriveWords;
  ce_count_kernel<<<static_cast<int>(K * groups), kBlock, 0, st>>>(labels, rows, ignore_index, counters, inv_count, rpg,
                                                                   groups);
  const CeNorm norm{inv_count, rpg, groups};
  // register path: row_stride == V keeps the gradient row's alignment phase equal to the logits row's
  const bool reg_path = V >= 8 && V <= kRegFloats - 8 && row_stride == V && aligned16(logits) && (!grad || aligned16(grad));
  if (reg_path) {
    const CeFold fold{counters, loss_out, accumulate, gscale};
    const bool g = grad != nullptr;
#ifdef VQA_TUNING
    if (g_ce_threads == 256)
      launch_reg_k<256>(g, g_ce_variant, rows, st, logits, labels, K, V, ignore_index, norm, grad, row_loss, gscale, flag, fold, row_state);
    else if (g_ce_threads == 1024)
      launch_reg_k<1024>(g, g_ce_variant, rows, st, logits, labels, K, V, ignore_index, norm, grad, row_loss, gscale, flag, fold, row_state);
    else
#endif
    launch_reg_k<512>(g, g_ce_variant, rows, st, logits, labels, K, V, ignore_index, norm, grad, row_loss, gscale, flag, fold, row_state);
    return launch_status();
  }
  const int grid = static_cast<int>(rows);
  if (grad)
    ce_rows_kernel<true, kCeMaxK><<<grid, kBlock, 0, st>>>(logits, row_stride, labels, K, rows, V, ignore_index, norm,
                                                           grad, row_loss, gscale, flag, row_state);
  else
    ce_rows_kernel<false, kCeMaxK><<<grid, kBlock, 0, st>>>(logits, row_stride, labels, K, rows, V, ignore_index,
                                                            norm, grad, row_loss, gscale, flag, nullptr);
  const int rc = launch_status();
  if (rc != VQA_OK || !loss_out) return rc;
  if (rows > 0x7fffffffL) return VQA_ERR_SHAPE;
  return vqa_sum_partials(row_loss, static_cast<int>(rows), loss_out, accumulate, gscale, stream);
}

}  // extern "C"
