// Per-sample norm reductions and the L2 / L1 image updates (gfx950 / MI355X).
//
// Reductions are two-stage and deterministic: stage 1 (grid = chunks x batch) reduces a contiguous chunk of one
// sample with 16-byte loads -> wave shuffle (DPP) reduction -> 4-wave LDS combine and writes ONE partial per
// workgroup; stage 2 (one workgroup per sample) folds the partials in index order.  No float atomics, so the
// norms -- and therefore the perturbations -- are bitwise reproducible from run to run.
// The update kernels re-read the gradient; at the path's sizes (<= 453 MB per tensor) that second read is mostly
// served by the 256 MiB Infinity Cache when the batch is small and by HBM otherwise (algorithmic bytes are
// counted as HBM bytes either way).
#include "common.hpp"

namespace vqa {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRedU = 4;                 // 16-byte loads in flight per lane
constexpr int kMaxChunks = 512;          // partials per sample (stage 2 folds them with one workgroup)

static int chunks_for(int batch, size_t n_per) {
  size_t per_block = static_cast<size_t>(kBlock) * kRedU * 4;   // elements one workgroup covers per sweep
  size_t c = (n_per + per_block - 1) / per_block;
  size_t cap = static_cast<size_t>(grid_cap()) * 2 / (batch > 0 ? batch : 1);
  if (cap < 1) cap = 1;
  if (cap > kMaxChunks) cap = kMaxChunks;
  if (c > cap) c = cap;
  if (c < 1) c = 1;
  return static_cast<int>(c);
}

__device__ __forceinline__ float block_sum(float v, float* lds) {
  v = wave_sum(v);
  const int w = threadIdx.x / kWave;
  if ((threadIdx.x & (kWave - 1)) == 0) lds[w] = v;
  __syncthreads();
  float r = lds[0];
#pragma unroll
  for (int i = 1; i < kBlock / kWave; ++i) r += lds[i];
  __syncthreads();
  return r;
}

__device__ __forceinline__ float block_max(float v, float* lds) {
  v = wave_max(v);
  const int w = threadIdx.x / kWave;
  if ((threadIdx.x & (kWave - 1)) == 0) lds[w] = v;
  __syncthreads();
  float r = lds[0];
#pragma unroll
  for (int i = 1; i < kBlock / kWave; ++i) r = fmaxf(r, lds[i]);
  __syncthreads();
  return r;
}

// A workgroup's share of one sample: elements [lo, hi) with lo a multiple of 4.
__device__ __forceinline__ void chunk_bounds(size_t n_per, int chunks, size_t& lo, size_t& hi) {
  size_t n4 = (n_per + 3) / 4;
  size_t per = (n4 + chunks - 1) / chunks;
  lo = static_cast<size_t>(blockIdx.x) * per * 4;
  hi = lo + per * 4;
  if (lo > n_per) lo = n_per;
  if (hi > n_per) hi = n_per;
}

// ---- stage 1: sum of squares of (t - sub) -------------------------------------------------------
template <bool VEC, bool SUB>
__global__ __launch_bounds__(kBlock) void sumsq_stage1(const float* __restrict__ t, const float* __restrict__ sub,
                                                       float* __restrict__ ws, size_t n_per, int chunks) {
  __shared__ float lds[kBlock / kWave];
  const size_t base = static_cast<size_t>(blockIdx.y) * n_per;
  size_t lo, hi;
  chunk_bounds(n_per, chunks, lo, hi);
  float acc = 0.0f;
  if (VEC) {
    const f32x4* t4 = reinterpret_cast<const f32x4*>(t + base);
    const f32x4* s4 = reinterpret_cast<const f32x4*>(sub + (SUB ? base : 0));
    const size_t lo4 = lo / 4, hi4 = hi / 4;   // VEC requires n_per % 4 == 0
    for (size_t i = lo4 + threadIdx.x; i < hi4; i += static_cast<size_t>(kBlock) * kRedU) {
      f32x4 v[kRedU], s[kRedU];
#pragma unroll
      for (int u = 0; u < kRedU; ++u) {
        const size_t j = i + static_cast<size_t>(u) * kBlock;
        if (j < hi4) {
          v[u] = t4[j];
          if (SUB) s[u] = s4[j];
        }
      }
#pragma unroll
      for (int u = 0; u < kRedU; ++u) {
        const size_t j = i + static_cast<size_t>(u) * kBlock;
        if (j < hi4) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float d = SUB ? (v[u][k] - s[u][k]) : v[u][k];
            acc += d * d;
          }
        }
      }
    }
  } else {
    for (size_t i = lo + threadIdx.x; i < hi; i += kBlock) {
      float d = SUB ? (t[base + i] - sub[base + i]) : t[base + i];
      acc += d * d;
    }
  }
  float r = block_sum(acc, lds);
  if (threadIdx.x == 0) ws[static_cast<size_t>(blockIdx.y) * chunks + blockIdx.x] = r;
}

__global__ __launch_bounds__(kBlock) void sum_stage2(const float* __restrict__ ws, float* __restrict__ out,
                                                     int chunks) {
  __shared__ float lds[kBlock / kWave];
  const float* p = ws + static_cast<size_t>(blockIdx.x) * chunks;
  float acc = 0.0f;
  for (int i = threadIdx.x; i < chunks; i += kBlock) acc += p[i];
  float r = block_sum(acc, lds);
  if (threadIdx.x == 0) out[blockIdx.x] = r;
}

// ---- stage 1/2: max |g| and the number of elements attaining it ---------------------------------
// Stage 1 reads its chunk from HBM exactly once: the chunk is swept in register-resident tiles (kTieU 16-byte pieces per
// lane); per tile the workgroup reduces the maximum and counts the elements that attain it from the SAME registers,
// then merges (max, count) into the chunk's running pair -- no second sweep through L2.
constexpr int kTieU = 8;
template <bool VEC>
__global__ __launch_bounds__(kBlock) void absmax_stage1(const float* __restrict__ g, float* __restrict__ ws,
                                                        size_t n_per, int chunks, int batch) {
  __shared__ float lds[kBlock / kWave];
  const size_t base = static_cast<size_t>(blockIdx.y) * n_per;
  size_t lo, hi;
  chunk_bounds(n_per, chunks, lo, hi);
  // |g| >= 0.  fmaxf drops NaN; the reference's torch.max PROPAGATES it (the sample's maximum is NaN, nothing ties with
  // it, the result is 0 / 0 and the self-check assert of optimize_linear fires, utils.py:96-104): a NaN anywhere in the
  // chunk is tracked separately and makes the chunk's -- and so the sample's -- maximum NaN.
  float run_max = 0.0f, run_cnt = 0.0f, seen_nan = 0.0f;
  if (VEC) {
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g + base);
    const size_t hi4 = hi / 4;
    for (size_t t0 = lo / 4; t0 < hi4; t0 += static_cast<size_t>(kBlock) * kTieU) {
      f32x4 v[kTieU];
      float m = 0.0f;
#pragma unroll
      for (int u = 0; u < kTieU; ++u) {
        const size_t j = t0 + static_cast<size_t>(u) * kBlock + threadIdx.x;
        v[u] = (j < hi4) ? g4[j] : f32x4{-1.0f, -1.0f, -1.0f, -1.0f};      // fabsf(-1) never ties with a tile max of 0
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (j < hi4) {
            m = fmaxf(m, fabsf(v[u][k]));
            seen_nan = (v[u][k] != v[u][k]) ? 1.0f : seen_nan;
          }
      }
      const float tm = block_max(m, lds);
      float c = 0.0f;
#pragma unroll
      for (int u = 0; u < kTieU; ++u) {
        const size_t j = t0 + static_cast<size_t>(u) * kBlock + threadIdx.x;
        if (j < hi4) {
#pragma unroll
          for (int k = 0; k < 4; ++k) c += (fabsf(v[u][k]) == tm) ? 1.0f : 0.0f;
        }
      }
      const float tc = block_sum(c, lds);
      if (tm > run_max) { run_max = tm; run_cnt = tc; }
      else if (tm == run_max) run_cnt += tc;
    }
  } else {
    float m = 0.0f;
    for (size_t i = lo + threadIdx.x; i < hi; i += kBlock) {
      m = fmaxf(m, fabsf(g[base + i]));
      seen_nan = (g[base + i] != g[base + i]) ? 1.0f : seen_nan;
    }
    run_max = block_max(m, lds);
    float c = 0.0f;
    for (size_t i = lo + threadIdx.x; i < hi; i += kBlock) c += (fabsf(g[base + i]) == run_max) ? 1.0f : 0.0f;
    run_cnt = block_sum(c, lds);
  }
  if (block_max(seen_nan, lds) > 0.0f) run_max = __builtin_nanf("");
  if (threadIdx.x == 0) {
    const size_t slot = static_cast<size_t>(blockIdx.y) * chunks + blockIdx.x;
    ws[slot] = run_max;
    ws[static_cast<size_t>(batch) * chunks + slot] = (lo < hi) ? run_cnt : 0.0f;
  }
}

__global__ __launch_bounds__(kBlock) void absmax_stage2(const float* __restrict__ ws, float* __restrict__ amax,
                                                        float* __restrict__ ties, int chunks, int batch) {
  __shared__ float lds[kBlock / kWave];
  const float* pm = ws + static_cast<size_t>(blockIdx.x) * chunks;
  const float* pc = ws + static_cast<size_t>(batch) * chunks + static_cast<size_t>(blockIdx.x) * chunks;
  float m = 0.0f, seen_nan = 0.0f;
  for (int i = threadIdx.x; i < chunks; i += kBlock) {
    m = fmaxf(m, pm[i]);
    seen_nan = (pm[i] != pm[i]) ? 1.0f : seen_nan;
  }
  float gm = block_max(m, lds);
  if (block_max(seen_nan, lds) > 0.0f) gm = __builtin_nanf("");       // torch.max propagates NaN
  float c = 0.0f;
  for (int i = threadIdx.x; i < chunks; i += kBlock) c += (pm[i] == gm) ? pc[i] : 0.0f;
  const float gc = block_sum(c, lds);
  if (threadIdx.x == 0) {
    amax[blockIdx.x] = gm;
    ties[blockIdx.x] = gc;
  }
}

// ---- per-sample elementwise updates -------------------------------------------------------------
struct NormParams {
  float eps, cmin, cmax;
  unsigned mode;
};

enum Kind { kL2Fgm, kL2Project, kL1Fgm, kClipEtaL2, kOptLinL2, kOptLinL1 };

// s0/s1: the two element streams (x,g | adv,x0 | t,-); st/st2: per-sample statistics.
template <int KIND>
__device__ __forceinline__ float norm_apply(const NormParams& p, float a, float b, float st, float st2, bool& bad) {
  if (KIND == kL2Fgm) {
    if (p.mode & VQA_CHECK_RANGE) bad |= out_of_range(a, p.cmin, p.cmax);
    float v = a + p.eps * (b / sqrtf(fmaxf(1e-12f, st)));
    return (p.mode & VQA_CLIP) ? clamp_torch(v, p.cmin, p.cmax) : v;
  } else if (KIND == kL2Project) {
    float factor = fminf(1.0f, p.eps / sqrtf(fmaxf(1e-12f, st)));
    float v = b + (a - b) * factor;
    return (p.mode & VQA_CLIP) ? clamp_torch(v, p.cmin, p.cmax) : v;
  } else if (KIND == kL1Fgm) {
    if (p.mode & VQA_CHECK_RANGE) bad |= out_of_range(a, p.cmin, p.cmax);
    float hit = (fabsf(b) == st) ? 1.0f : 0.0f;
    float v = a + p.eps * (sign_torch(b) * hit / st2);
    return (p.mode & VQA_CLIP) ? clamp_torch(v, p.cmin, p.cmax) : v;
  } else if (KIND == kClipEtaL2) {
    return a * fminf(1.0f, p.eps / sqrtf(fmaxf(1e-12f, st)));
  } else if (KIND == kOptLinL2) {
    return p.eps * (a / sqrtf(fmaxf(1e-12f, st)));
  } else {
    float hit = (fabsf(a) == st) ? 1.0f : 0.0f;
    return p.eps * (sign_torch(a) * hit / st2);
  }
}

template <int KIND, bool VEC, bool TWO>
__global__ __launch_bounds__(kBlock) void per_sample_kernel(const float* s0, const float* __restrict__ s1,
                                                            const float* __restrict__ stat,
                                                            const float* __restrict__ stat2, float* out,
                                                            size_t n_per, NormParams p, int* __restrict__ flag) {
  const size_t base = static_cast<size_t>(blockIdx.y) * n_per;
  const float st = stat[blockIdx.y];
  const float st2 = stat2 ? stat2[blockIdx.y] : 1.0f;
  bool bad = false;
  if (VEC) {
    const f32x4* a4 = reinterpret_cast<const f32x4*>(s0 + base);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(s1 + (TWO ? base : 0));
    f32x4* o4 = reinterpret_cast<f32x4*>(out + base);
    const size_t n4 = n_per / 4;
    const size_t tile = static_cast<size_t>(kBlock) * kRedU;
    for (size_t t0 = static_cast<size_t>(blockIdx.x) * tile; t0 < n4; t0 += static_cast<size_t>(gridDim.x) * tile) {
      f32x4 va[kRedU], vb[kRedU];
#pragma unroll
      for (int u = 0; u < kRedU; ++u) {
        const size_t i = t0 + static_cast<size_t>(u) * kBlock + threadIdx.x;
        if (i < n4) {
          va[u] = a4[i];
          if (TWO) vb[u] = b4[i];
        }
      }
#pragma unroll
      for (int u = 0; u < kRedU; ++u) {
        const size_t i = t0 + static_cast<size_t>(u) * kBlock + threadIdx.x;
        if (i < n4) {
          f32x4 r;
#pragma unroll
          for (int k = 0; k < 4; ++k) r[k] = norm_apply<KIND>(p, va[u][k], TWO ? vb[u][k] : 0.0f, st, st2, bad);
          o4[i] = r;
        }
      }
    }
  } else {
    for (size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; i < n_per;
         i += static_cast<size_t>(gridDim.x) * kBlock)
      out[base + i] = norm_apply<KIND>(p, s0[base + i], TWO ? s1[base + i] : 0.0f, st, st2, bad);
  }
  if ((p.mode & VQA_CHECK_RANGE) && bad) atomicOr(flag, VQA_FLAG_RANGE);
  // The reference's optimize_linear asserts that its result has unit norm (A-ch/utils.py:101-104 for L1, :110-116 for
  // L2) -- a host sync per call.  The assert can only fire when a sample's statistic is degenerate: L1: max|g| is 0 (all
  // entries tie with sign 0) or NaN; L2: the sum of squares is inf or NaN.  Reported as a flag bit instead.
  if (flag && blockIdx.x == 0 && threadIdx.x == 0) {
    bool degenerate = false;
    if (KIND == kL2Fgm || KIND == kOptLinL2) degenerate = !(st < INFINITY);
    if (KIND == kL1Fgm || KIND == kOptLinL1) degenerate = !(st > 0.0f) || !(st2 >= 1.0f);
    if (degenerate) atomicOr(flag, VQA_FLAG_DEGENERATE);
  }
}

template <int KIND, bool TWO>
static int launch_per_sample(const float* s0, const float* s1, const float* stat, const float* stat2, float* out,
                             int batch, size_t n_per, const NormParams& p, int* flag, vqa_stream_t stream) {
  if (!s0 || !out || !stat || (TWO && !s1)) return VQA_ERR_NULL;
  if ((KIND == kL1Fgm || KIND == kOptLinL1) && !stat2) return VQA_ERR_NULL;
  if ((p.mode & VQA_CHECK_RANGE) && !flag) return VQA_ERR_NULL;
  if (batch < 0 || batch > 65535) return VQA_ERR_SHAPE;
  if (!aligned4(s0) || !aligned4(out) || (s1 && !aligned4(s1))) return VQA_ERR_ALIGN;
  if (batch == 0 || n_per == 0) return VQA_OK;
  const bool vec = (n_per % 4 == 0) && aligned16(s0) && aligned16(out) && (!TWO || aligned16(s1));
  hipStream_t st = static_cast<hipStream_t>(stream);
  int gx = blocks_for(vec ? n_per / 4 : n_per, vec ? kBlock * kRedU : kBlock, grid_cap() / (batch < 8 ? batch : 8));
  dim3 grid(gx, batch);
  if (vec)
    per_sample_kernel<KIND, true, TWO><<<grid, kBlock, 0, st>>>(s0, s1, stat, stat2, out, n_per, p, flag);
  else
    per_sample_kernel<KIND, false, TWO><<<grid, kBlock, 0, st>>>(s0, s1, stat, stat2, out, n_per, p, flag);
  return launch_status();
}

}  // namespace vqa

using namespace vqa;

extern "C" {

size_t vqa_reduce_ws_bytes(int batch, size_t n_per_sample) {
  if (batch <= 0) return 0;
  return static_cast<size_t>(batch) * chunks_for(batch, n_per_sample) * 2 * sizeof(float);
}

int vqa_sumsq_per_sample(const float* t, const float* sub, float* out, int batch, size_t n_per_sample, float* ws,
                         vqa_stream_t stream) {
  clear_stale_error();
  if (!t || !out || !ws) return VQA_ERR_NULL;
  if (batch < 0 || batch > 65535) return VQA_ERR_SHAPE;
  if (!aligned4(t) || (sub && !aligned4(sub))) return VQA_ERR_ALIGN;
  if (batch == 0) return VQA_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int chunks = chunks_for(batch, n_per_sample);
  const bool vec = (n_per_sample % 4 == 0) && aligned16(t) && (!sub || aligned16(sub));
  dim3 grid(chunks, batch);
  if (vec && sub) sumsq_stage1<true, true><<<grid, kBlock, 0, st>>>(t, sub, ws, n_per_sample, chunks);
  else if (vec) sumsq_stage1<true, false><<<grid, kBlock, 0, st>>>(t, t, ws, n_per_sample, chunks);
  else if (sub) sumsq_stage1<false, true><<<grid, kBlock, 0, st>>>(t, sub, ws, n_per_sample, chunks);
  else sumsq_stage1<false, false><<<grid, kBlock, 0, st>>>(t, t, ws, n_per_sample, chunks);
  sum_stage2<<<batch, kBlock, 0, st>>>(ws, out, chunks);
  return launch_status();
}

int vqa_absmax_ties_per_sample(const float* g, float* amax, float* ties, int batch, size_t n_per_sample, float* ws,
                               vqa_stream_t stream) {
  clear_stale_error();
  if (!g || !amax || !ties || !ws) return VQA_ERR_NULL;
  if (batch < 0 || batch > 65535) return VQA_ERR_SHAPE;
  if (!aligned4(g)) return VQA_ERR_ALIGN;
  if (batch == 0) return VQA_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int chunks = chunks_for(batch, n_per_sample);
  dim3 grid(chunks, batch);
  if ((n_per_sample % 4 == 0) && aligned16(g))
    absmax_stage1<true><<<grid, kBlock, 0, st>>>(g, ws, n_per_sample, chunks, batch);
  else
    absmax_stage1<false><<<grid, kBlock, 0, st>>>(g, ws, n_per_sample, chunks, batch);
  absmax_stage2<<<batch, kBlock, 0, st>>>(ws, amax, ties, chunks, batch);
  return launch_status();
}

int vqa_l2_fgm(const float* x, const float* g, const float* sumsq_g, float* out, int batch, size_t n_per_sample,
               float eps_iter, float cmin, float cmax, unsigned mode, int* flag, vqa_stream_t stream) {
  clear_stale_error();
  NormParams p{eps_iter, cmin, cmax, mode};
  return launch_per_sample<kL2Fgm, true>(x, g, sumsq_g, nullptr, out, batch, n_per_sample, p, flag, stream);
}

int vqa_l2_project(const float* adv, const float* x0, const float* sumsq_eta, float* out, int batch,
                   size_t n_per_sample, float eps, float cmin, float cmax, unsigned mode, vqa_stream_t stream) {
  clear_stale_error();
  NormParams p{eps, cmin, cmax, mode & ~VQA_CHECK_RANGE};
  return launch_per_sample<kL2Project, true>(adv, x0, sumsq_eta, nullptr, out, batch, n_per_sample, p, nullptr,
                                             stream);
}

int vqa_l1_fgm(const float* x, const float* g, const float* amax, const float* ties, float* out, int batch,
               size_t n_per_sample, float eps_iter, float cmin, float cmax, unsigned mode, int* flag,
               vqa_stream_t stream) {
  clear_stale_error();
  NormParams p{eps_iter, cmin, cmax, mode};
  return launch_per_sample<kL1Fgm, true>(x, g, amax, ties, out, batch, n_per_sample, p, flag, stream);
}

int vqa_scale_per_sample(const float* t, const float* stat, const float* stat2, float* out, int batch,
                         size_t n_per_sample, float eps, int kind, int* flag, vqa_stream_t stream) {
  clear_stale_error();
  NormParams p{eps, 0.0f, 0.0f, 0u};
  switch (kind) {
    case 0: return launch_per_sample<kClipEtaL2, false>(t, nullptr, stat, nullptr, out, batch, n_per_sample, p, nullptr, stream);
    case 1: return launch_per_sample<kOptLinL2, false>(t, nullptr, stat, nullptr, out, batch, n_per_sample, p, flag, stream);
    case 2: return launch_per_sample<kOptLinL1, false>(t, nullptr, stat, stat2, out, batch, n_per_sample, p, flag, stream);
    default: return VQA_ERR_SHAPE;
  }
}

}  // extern "C"
