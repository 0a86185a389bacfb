// Elementwise glue of the frozen white box's pre-LN transformer blocks (gfx950 / MI355X): everything between two library
// GEMMs / attention calls of a layer that is not a contraction, fused so that the residual stream crosses HBM once per
// stage instead of once per eager op.
//
// The reference's block (VLMO_VQAttack/vlmo/modules/multiway_transformer.py:184-201; ALBEF's ViT block, vit.py) is
//     x1 = x  + g1 * attn(LN1(x));    x2 = x1 + g2 * mlp_{text|image}(LN2_{text|image}(x1))
// and its eager execution runs, per layer and direction, addcmul / LayerNorm / slice copies / cat / gradient-accumulation
// adds / LayerNorm backward as separate full passes (10.8 % of the attack's device time in round 3).  Here:
//
//   vqa_ln_fwd :  x_out = x + rscale * r  (optional prologue: the residual add of the PREVIOUS branch, r given whole or
//                 as the two modality buffers), then y = LN(x_out) with the text or the image expert's parameters by
//                 token position, written whole or split into the two contiguous per-modality buffers the expert GEMMs
//                 read.  Bytes per element: read x, r; write x_out, y = 16 B (eager: addcmul 12 + cat 8 + LN 8 +
//                 two slice copies 8 = 36 B).
//   vqa_ln_bwd :  dx = g_a + g_inj + LN'(dy)  -- LayerNorm's input gradient (frozen gamma / beta: dX only), the gradient
//                 arriving over the residual path and the loss kernel's gradient of this feature map added in the same
//                 pass -- plus, optionally, dr = rscale * dx written whole or split: the gradient of the branch that the
//                 forward prologue added.  Read dy, x, g_a (, g_inj); write dx (, dr) = 20-28 B (eager: LN backward 16 +
//                 accumulation adds 12-24 + addcmul backward 8 + slice-backward cat 8 = 44-56 B).
//   vqa_gelu_fwd / vqa_gelu_bwd : exact (erf) GELU and its derivative, 8 / 12 B per element.
//
// One wavefront owns one row of D <= 1024 floats in registers (NCH 16-byte pieces per lane); mean / variance and the
// two backward moments are DPP wave reductions (common.hpp); no LDS, no atomics, results bitwise reproducible.
#include "common.hpp"

namespace vqa {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kRowsPerBlock = kBlock / kWave;

struct SplitMap {        // row -> (segment, row inside the segment's contiguous buffer)
  int period, split;     // a batch element has `period` rows, the first `split` of them are segment 0 (text tokens)
};

struct SegRow {
  int seg;
  long row;
};

__device__ __forceinline__ SegRow seg_row(const SplitMap& m, int row) {
  const int b = row / m.period, s = row - b * m.period;
  if (s < m.split) return SegRow{0, static_cast<long>(b) * m.split + s};
  return SegRow{1, static_cast<long>(b) * (m.period - m.split) + (s - m.split)};
}

// NT: non-temporal load -- for the activation streams, which a launch reads exactly once (parameter vectors stay plain:
// every row re-reads them from the cache)
template <int NCH, bool NT = false>
__device__ __forceinline__ void load_vec(f32x4 (&v)[NCH], const float* __restrict__ p, int lane, int D) {
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int d = (k * kWave + lane) * 4;
    if (d < D)
      v[k] = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + d)) : *reinterpret_cast<const f32x4*>(p + d);
    else
      v[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  }
}

template <int NCH>
__device__ __forceinline__ void store_vec(const f32x4 (&v)[NCH], float* __restrict__ p, int lane, int D) {
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int d = (k * kWave + lane) * 4;
    if (d < D) *reinterpret_cast<f32x4*>(p + d) = v[k];
  }
}

struct LnFwdArgs {
  const float *x, *r0, *r1, *rscale;
  float* x_out;
  const float *gamma0, *beta0, *gamma1, *beta1;
  float *y0, *y1, *mean, *rstd;
  int rows, D;
  SplitMap map;
  float eps;
};

template <int NCH, bool NT>
__global__ __launch_bounds__(kBlock) void ln_fwd_kernel(LnFwdArgs A) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const float inv_d = 1.0f / static_cast<float>(A.D);
  for (int row = blockIdx.x * kRowsPerBlock + wave; row < A.rows; row += gridDim.x * kRowsPerBlock) {
    f32x4 x[NCH];
    load_vec<NCH, NT>(x, A.x + static_cast<long>(row) * A.D, lane, A.D);
    SegRow sr{0, row};
    if (A.map.period > 0) sr = seg_row(A.map, row);
    if (A.r0) {                                             // prologue: the residual add of the previous branch
      const float* rp = A.r1 ? (sr.seg ? A.r1 : A.r0) + sr.row * A.D : A.r0 + static_cast<long>(row) * A.D;
      f32x4 r[NCH];
      load_vec<NCH, NT>(r, rp, lane, A.D);
      if (A.rscale) {
        f32x4 s[NCH];
        load_vec<NCH>(s, A.rscale, lane, A.D);
#pragma unroll
        for (int k = 0; k < NCH; ++k) x[k] = x[k] + s[k] * r[k];
      } else {
#pragma unroll
        for (int k = 0; k < NCH; ++k) x[k] = x[k] + r[k];
      }
      store_vec<NCH>(x, A.x_out + static_cast<long>(row) * A.D, lane, A.D);
    }
    float s1 = 0.0f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) s1 += (x[k][0] + x[k][1]) + (x[k][2] + x[k][3]);
    const float mean = wave_sum(s1) * inv_d;
    float s2 = 0.0f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int d = (k * kWave + lane) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float c = d < A.D ? x[k][e] - mean : 0.0f;
        x[k][e] = c;
        s2 += c * c;
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(s2) * inv_d + A.eps);
    const bool second = A.gamma1 && sr.seg;                 // the image expert's LayerNorm for image tokens
    f32x4 g[NCH], b[NCH];
    load_vec<NCH>(g, second ? A.gamma1 : A.gamma0, lane, A.D);
    load_vec<NCH>(b, second ? A.beta1 : A.beta0, lane, A.D);
#pragma unroll
    for (int k = 0; k < NCH; ++k) x[k] = (x[k] * rstd) * g[k] + b[k];
    float* yp = A.y1 ? (sr.seg ? A.y1 : A.y0) + sr.row * A.D : A.y0 + static_cast<long>(row) * A.D;
    store_vec<NCH>(x, yp, lane, A.D);
    if (lane == 0) {
      A.mean[row] = mean;
      A.rstd[row] = rstd;
    }
  }
}

struct LnBwdArgs {
  const float *dy0, *dy1, *x, *mean, *rstd, *gamma0, *gamma1, *g_a, *g_inj, *rscale;
  float *dx, *dr0, *dr1;
  int rows, D;
  SplitMap map;
};

template <int NCH, bool NT>
__global__ __launch_bounds__(kBlock) void ln_bwd_kernel(LnBwdArgs A) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const float inv_d = 1.0f / static_cast<float>(A.D);
  for (int row = blockIdx.x * kRowsPerBlock + wave; row < A.rows; row += gridDim.x * kRowsPerBlock) {
    SegRow sr{0, row};
    if (A.map.period > 0) sr = seg_row(A.map, row);
    const long off = static_cast<long>(row) * A.D;
    f32x4 xh[NCH], g[NCH];
    load_vec<NCH, NT>(xh, A.x + off, lane, A.D);
    load_vec<NCH, NT>(g, A.dy1 ? (sr.seg ? A.dy1 : A.dy0) + sr.row * A.D : A.dy0 + off, lane, A.D);
    {
      f32x4 w[NCH];
      load_vec<NCH>(w, (A.gamma1 && sr.seg) ? A.gamma1 : A.gamma0, lane, A.D);
#pragma unroll
      for (int k = 0; k < NCH; ++k) g[k] = g[k] * w[k];
    }
    const float mean = A.mean[row], rstd = A.rstd[row];
    float c1 = 0.0f, c2 = 0.0f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int d = (k * kWave + lane) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float h = d < A.D ? (xh[k][e] - mean) * rstd : 0.0f;
        xh[k][e] = h;
        c1 += g[k][e];
        c2 += g[k][e] * h;
      }
    }
    c1 = wave_sum(c1) * inv_d;
    c2 = wave_sum(c2) * inv_d;
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) g[k][e] = rstd * ((g[k][e] - c1) - xh[k][e] * c2);
    if (A.g_a) {                                            // gradient arriving over the residual path
      load_vec<NCH, NT>(xh, A.g_a + off, lane, A.D);
#pragma unroll
      for (int k = 0; k < NCH; ++k) g[k] = g[k] + xh[k];
    }
    if (A.g_inj) {                                          // the loss kernel's gradient of this feature map
      load_vec<NCH, NT>(xh, A.g_inj + off, lane, A.D);
#pragma unroll
      for (int k = 0; k < NCH; ++k) g[k] = g[k] + xh[k];
    }
    store_vec<NCH>(g, A.dx + off, lane, A.D);
    if (A.dr0) {                                            // gradient of the branch the forward prologue added
      if (A.rscale) {
        load_vec<NCH>(xh, A.rscale, lane, A.D);
#pragma unroll
        for (int k = 0; k < NCH; ++k) g[k] = g[k] * xh[k];
      }
      store_vec<NCH>(g, A.dr1 ? (sr.seg ? A.dr1 : A.dr0) + sr.row * A.D : A.dr0 + off, lane, A.D);
    }
  }
}

// ---- exact GELU (torch.nn.functional.gelu, approximate='none') and its derivative -------------------------------
// The two kernels are bound by the vector ALU, not by HBM (113 M elements per image-expert call; the library erff costs
// ~35 VALU slots per element, the stream itself would take ~60 us): erf is evaluated here branch-free on PAIRS of
// elements with packed fp32 FMAs (v_pk_fma_f32: two lanes' worth of polynomial per slot).  Coefficients: two minimax
// polynomials, erf(a) = a + a * P(a^2) below 0.9277 and 1 - exp(Q(|a|)) above, each below 1 ulp of error in fp32.
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr float kSqrtHalf = 0.70710678118654752440f;
constexpr float kInvSqrt2Pi = 0.39894228040143267794f;      // M_2_SQRTPI * M_SQRT1_2 * 0.5
constexpr float kLog2e = 1.44269504088896340736f;

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk(float v) { return f32x2{v, v}; }

__device__ __forceinline__ f32x2 erf2(f32x2 a) {
  const f32x2 t = __builtin_elementwise_abs(a);
  const f32x2 s = a * a;
  // |a| > 0.9277: erf = 1 - exp(r(t))
  f32x2 r = pk_fma(pk(-1.72853470e-5f), t, pk(3.83197126e-4f));
  const f32x2 u = pk_fma(pk(-3.88396438e-3f), t, pk(2.42546219e-2f));
  r = pk_fma(r, s, u);
  r = pk_fma(r, t, pk(-1.06777877e-1f));
  r = pk_fma(r, t, pk(-6.34846687e-1f));
  r = pk_fma(r, t, pk(-1.28717512e-1f));
  r = pk_fma(r, t, -t);
  r = r * pk(kLog2e);
  f32x2 big = {1.0f - __builtin_amdgcn_exp2f(r[0]), 1.0f - __builtin_amdgcn_exp2f(r[1])};
  big = f32x2{__builtin_copysignf(big[0], a[0]), __builtin_copysignf(big[1], a[1])};
  // |a| <= 0.9277: erf = a + a * p(a^2)
  f32x2 p = pk_fma(pk(-5.96761703e-4f), s, pk(4.99119423e-3f));
  p = pk_fma(p, s, pk(-2.67681349e-2f));
  p = pk_fma(p, s, pk(1.12819925e-1f));
  p = pk_fma(p, s, pk(-3.76125336e-1f));
  p = pk_fma(p, s, pk(1.28379166e-1f));
  p = pk_fma(p, a, a);
  return f32x2{t[0] > 0.927734375f ? big[0] : p[0], t[1] > 0.927734375f ? big[1] : p[1]};
}

__device__ __forceinline__ f32x2 gelu2(f32x2 x) {            // 0.5 x (1 + erf(x / sqrt 2))
  const f32x2 e = erf2(x * pk(kSqrtHalf));
  return (pk(0.5f) * x) * (pk(1.0f) + e);
}

__device__ __forceinline__ f32x2 gelu_grad2(f32x2 x) {       // cdf + x pdf
  const f32x2 cdf = pk(0.5f) * (pk(1.0f) + erf2(x * pk(kSqrtHalf)));
  const f32x2 q = (x * x) * pk(-0.5f * kLog2e);
  const f32x2 pdf = f32x2{__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])} * pk(kInvSqrt2Pi);
  return pk_fma(x, pdf, cdf);
}

__device__ __forceinline__ float gelu_f(float x) { return gelu2(f32x2{x, x})[0]; }
__device__ __forceinline__ float gelu_grad_f(float x) { return gelu_grad2(f32x2{x, x})[0]; }

constexpr int kGeluUnroll = 4;

// NT bit0: non-temporal loads of h (and of da unless the update is in place), bit1: non-temporal stores (results larger
// than the 256 MB Infinity Cache cannot stay resident for the consuming GEMM anyway: the step kernel's rule, section 4)
template <bool BWD, int NT>
__global__ __launch_bounds__(kBlock) void gelu_kernel(const f32x4* __restrict__ h, const f32x4* da, f32x4* out,
                                                      size_t n4, bool in_place) {
  const size_t tile = static_cast<size_t>(kBlock) * kGeluUnroll;
  for (size_t base = static_cast<size_t>(blockIdx.x) * tile; base < n4; base += static_cast<size_t>(gridDim.x) * tile) {
    f32x4 vh[kGeluUnroll], vd[kGeluUnroll];
#pragma unroll
    for (int u = 0; u < kGeluUnroll; ++u) {
      const size_t i = base + static_cast<size_t>(u) * kBlock + threadIdx.x;
      if (i < n4) {
        vh[u] = (NT & 1) ? __builtin_nontemporal_load(&h[i]) : h[i];
        if (BWD) vd[u] = ((NT & 1) && !in_place) ? __builtin_nontemporal_load(&da[i]) : da[i];
      }
    }
#pragma unroll
    for (int u = 0; u < kGeluUnroll; ++u) {
      const size_t i = base + static_cast<size_t>(u) * kBlock + threadIdx.x;
      if (i < n4) {
        const f32x2 lo = {vh[u][0], vh[u][1]}, hi = {vh[u][2], vh[u][3]};
        f32x2 rl, rh;
        if (BWD) {
          rl = f32x2{vd[u][0], vd[u][1]} * gelu_grad2(lo);
          rh = f32x2{vd[u][2], vd[u][3]} * gelu_grad2(hi);
        } else {
          rl = gelu2(lo);
          rh = gelu2(hi);
        }
        const f32x4 res = {rl[0], rl[1], rh[0], rh[1]};
        if (NT & 2) __builtin_nontemporal_store(res, &out[i]);
        else out[i] = res;
      }
    }
  }
}

template <bool BWD>
__global__ __launch_bounds__(kBlock) void gelu_tail_kernel(const float* __restrict__ h, const float* da, float* out,
                                                           size_t begin, size_t n) {
  const size_t i = begin + static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i < n) out[i] = BWD ? da[i] * gelu_grad_f(h[i]) : gelu_f(h[i]);
}

VQA_KNOB g_block_nt = 3;    // option 10 (tuning build): bit0 nt loads of the activation streams, bit1 nt stores of large GELU results

template <bool BWD>
static int launch_gelu(const float* h, const float* da, float* out, size_t n, vqa_stream_t stream) {
  if (!h || !out || (BWD && !da)) return VQA_ERR_NULL;
  if (!aligned4(h) || !aligned4(out) || (da && !aligned4(da))) return VQA_ERR_ALIGN;
  if (n == 0) return VQA_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  size_t done = 0;
  if (aligned16(h) && aligned16(out) && (!BWD || aligned16(da)) && n >= 4) {
    const size_t n4 = n / 4;
    const int grid = blocks_for(n4, kBlock * kGeluUnroll);
    auto h4 = reinterpret_cast<const f32x4*>(h);
    auto d4 = reinterpret_cast<const f32x4*>(da);
    auto o4 = reinterpret_cast<f32x4*>(out);
    const bool in_place = static_cast<const void*>(da) == static_cast<const void*>(out);
    int nt = g_block_nt & 1;
    if ((g_block_nt & 2) && n * sizeof(float) > (256ull << 20)) nt |= 2;
    switch (nt) {
#ifdef VQA_TUNING
      case 0: gelu_kernel<BWD, 0><<<grid, kBlock, 0, st>>>(h4, d4, o4, n4, in_place); break;
      case 2: gelu_kernel<BWD, 2><<<grid, kBlock, 0, st>>>(h4, d4, o4, n4, in_place); break;
#endif
      case 1: gelu_kernel<BWD, 1><<<grid, kBlock, 0, st>>>(h4, d4, o4, n4, in_place); break;
      default: gelu_kernel<BWD, 3><<<grid, kBlock, 0, st>>>(h4, d4, o4, n4, in_place); break;
    }
    done = n4 * 4;
  }
  if (done < n)
    gelu_tail_kernel<BWD><<<static_cast<int>((n - done + kBlock - 1) / kBlock), kBlock, 0, st>>>(h, da, out, done, n);
  return launch_status();
}

static int check_rows(long rows, int D, long period, long split) {
  if (rows < 0 || rows > 0x7fffffffL || D <= 0 || D > 1024 || (D & 3)) return VQA_ERR_SHAPE;
  if (period < 0 || period > 0x7fffffffL) return VQA_ERR_SHAPE;
  if (period > 0 && (split < 0 || split > period || rows % period != 0)) return VQA_ERR_SHAPE;
  return VQA_OK;
}

static int row_grid(long rows) { return blocks_for(static_cast<size_t>(rows), kRowsPerBlock); }

}  // namespace vqa

using namespace vqa;

extern "C" {

int vqa_ln_fwd(const float* x, const float* r0, const float* r1, const float* rscale, float* x_out,
               const float* gamma0, const float* beta0, const float* gamma1, const float* beta1, float* y0, float* y1,
               float* mean, float* rstd, long rows, int D, long period, long split, float eps, vqa_stream_t stream) {
  clear_stale_error();
  if (!x || !gamma0 || !beta0 || !y0 || !mean || !rstd) return VQA_ERR_NULL;
  if ((r0 && !x_out) || (r1 && !r0) || (rscale && !r0) || (gamma1 && !beta1)) return VQA_ERR_NULL;
  const int rc = check_rows(rows, D, period, split);
  if (rc != VQA_OK) return rc;
  if ((r1 || y1 || gamma1) && period <= 0) return VQA_ERR_SHAPE;        // a split needs the token layout
  const void* ptrs[] = {x, r0, r1, rscale, x_out, gamma0, beta0, gamma1, beta1, y0, y1};
  for (const void* p : ptrs)
    if (p && !aligned16(p)) return VQA_ERR_ALIGN;
  if (rows == 0) return VQA_OK;
  LnFwdArgs A{x, r0, r1, rscale, x_out, gamma0, beta0, gamma1, beta1, y0, y1, mean, rstd, static_cast<int>(rows), D,
              SplitMap{static_cast<int>(period), static_cast<int>(split)}, eps};
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int grid = row_grid(rows);
#ifdef VQA_TUNING
  if (!(g_block_nt & 1)) {
    switch ((D + 255) / 256) {
      case 1: ln_fwd_kernel<1, false><<<grid, kBlock, 0, st>>>(A); break;
      case 2: ln_fwd_kernel<2, false><<<grid, kBlock, 0, st>>>(A); break;
      case 3: ln_fwd_kernel<3, false><<<grid, kBlock, 0, st>>>(A); break;
      default: ln_fwd_kernel<4, false><<<grid, kBlock, 0, st>>>(A); break;
    }
    return launch_status();
  }
#endif
  switch ((D + 255) / 256) {
    case 1: ln_fwd_kernel<1, true><<<grid, kBlock, 0, st>>>(A); break;
    case 2: ln_fwd_kernel<2, true><<<grid, kBlock, 0, st>>>(A); break;
    case 3: ln_fwd_kernel<3, true><<<grid, kBlock, 0, st>>>(A); break;
    default: ln_fwd_kernel<4, true><<<grid, kBlock, 0, st>>>(A); break;
  }
  return launch_status();
}

int vqa_ln_bwd(const float* dy0, const float* dy1, const float* x, const float* mean, const float* rstd,
               const float* gamma0, const float* gamma1, const float* g_a, const float* g_inj, const float* rscale,
               float* dx, float* dr0, float* dr1, long rows, int D, long period, long split, vqa_stream_t stream) {
  clear_stale_error();
  if (!dy0 || !x || !mean || !rstd || !gamma0 || !dx) return VQA_ERR_NULL;
  if ((dr1 && !dr0) || (rscale && !dr0)) return VQA_ERR_NULL;
  const int rc = check_rows(rows, D, period, split);
  if (rc != VQA_OK) return rc;
  if ((dy1 || dr1 || gamma1) && period <= 0) return VQA_ERR_SHAPE;
  const void* ptrs[] = {dy0, dy1, x, gamma0, gamma1, g_a, g_inj, rscale, dx, dr0, dr1};
  for (const void* p : ptrs)
    if (p && !aligned16(p)) return VQA_ERR_ALIGN;
  if (rows == 0) return VQA_OK;
  LnBwdArgs A{dy0, dy1, x, mean, rstd, gamma0, gamma1, g_a, g_inj, rscale, dx, dr0, dr1, static_cast<int>(rows), D,
              SplitMap{static_cast<int>(period), static_cast<int>(split)}};
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int grid = row_grid(rows);
#ifdef VQA_TUNING
  if (!(g_block_nt & 1)) {
    switch ((D + 255) / 256) {
      case 1: ln_bwd_kernel<1, false><<<grid, kBlock, 0, st>>>(A); break;
      case 2: ln_bwd_kernel<2, false><<<grid, kBlock, 0, st>>>(A); break;
      case 3: ln_bwd_kernel<3, false><<<grid, kBlock, 0, st>>>(A); break;
      default: ln_bwd_kernel<4, false><<<grid, kBlock, 0, st>>>(A); break;
    }
    return launch_status();
  }
#endif
  switch ((D + 255) / 256) {
    case 1: ln_bwd_kernel<1, true><<<grid, kBlock, 0, st>>>(A); break;
    case 2: ln_bwd_kernel<2, true><<<grid, kBlock, 0, st>>>(A); break;
    case 3: ln_bwd_kernel<3, true><<<grid, kBlock, 0, st>>>(A); break;
    default: ln_bwd_kernel<4, true><<<grid, kBlock, 0, st>>>(A); break;
  }
  return launch_status();
}

#ifdef VQA_TUNING
int vqa_block_set_option(int value) {    // reached through vqa_set_option(10, value)
  if (value < 0 || value > 3) return VQA_ERR_SHAPE;
  g_block_nt = value;
  return VQA_OK;
}
#endif

int vqa_gelu_fwd(const float* h, float* a, size_t n, vqa_stream_t stream) {
  clear_stale_error();
  return launch_gelu<false>(h, nullptr, a, n, stream);
}

int vqa_gelu_bwd(const float* h, const float* da, float* dh, size_t n, vqa_stream_t stream) {
  clear_stale_error();
  return launch_gelu<true>(h, da, dh, n, stream);
}

}  // extern "C"
