// Cross-modal feature-loss reduction: sum over rows of -cos(a_row, b_row), with the gradient w.r.t. `a`
// produced in the same pass (gfx950 / MI355X).
//
// One wavefront owns one row: D <= 2048 floats are held in registers as NCH 16-byte pieces per lane per
// operand (D = 768 -> 3, D = 1024 -> 4), so a and b are read from HBM exactly once and the gradient row is
// written once: 12*D bytes per row against 20*D for a separate forward and backward and far more for the
// reference's unfused chain (norm, clamp, div, mul, sum, neg, sum, sum + their autograd backward).
//
// What keeps the wave on the HBM roof instead of on the VALU:
//   * the row cursor (layer, outer, inner) is wave-uniform and lives in SGPRs; it advances by the grid stride with
//     three compare-and-carry steps on precomputed digits -- no 64-bit division per row;
//   * dot / |a|^2 / |b|^2 are reduced across the 64 lanes with DPP adds (common.hpp), three independent chains
//     interleaved -- no ds_bpermute, no LDS wait;
//   * two rows are in flight per wave: the loads of row k+1 are issued before row k is reduced;
//   * the grid is exactly the number of workgroups that are resident at once (occupancy x CUs), so the
//     grid-stride sweep has no part-empty tail wave of workgroups.
// LDS is used only for the 4-wave combine of the per-workgroup loss partial.  Row order per wave is fixed, the
// workgroup that arrives last folds the partials in index order, so the loss is bitwise reproducible for a given
// grid and no second launch is needed.
#include "common.hpp"

namespace vqa {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kLossMaxBlocks = 256 * 8;       // capacity of the partial buffer (workgroups)
constexpr int kWavesPerBlock = kBlock / kWave;

struct RowAddr {
  int rows0, rows1;                 // rows of one layer: (outer, inner)
  int n_layers;
  int step_i, step_o, step_l;       // digits of the grid stride (in rows) in the (layer, outer, inner) number system
  int mask_period;                  // row weights are indexed by (outer % mask_period, inner)
  long a0, a1, b0, b1, g0, g1;      // element strides of (outer, inner) for a, b and the gradient
};

// Up to kMaxLayers (a, b, grad) base pointers passed BY VALUE in the kernel arguments: one launch covers every
// per-layer feature map of a LayerFeatures list (13 for the base encoders, 25 for VLMo-large) without packing them.
constexpr int kMaxLayers = 32;
struct LayerTable {
  const float* a[kMaxLayers];
  const float* b[kMaxLayers];
  float* g[kMaxLayers];
};

struct Fold {           // in-kernel fold of the partials into the scalar loss (all NULL -> partials only)
  unsigned* counter;    // arrival counters (common.hpp), zero on entry; reset by the folding workgroup
  float* loss_out;
  int accumulate;
  float scale;
};

struct Cursor {         // wave-uniform (SGPR) position in the (layer, outer, inner) row space
  int layer, o, i;
};

__device__ __forceinline__ void advance(Cursor& c, const RowAddr& ra) {
  c.i += ra.step_i;
  if (c.i >= ra.rows1) { c.i -= ra.rows1; c.o += 1; }
  c.o += ra.step_o;
  if (c.o >= ra.rows0) { c.o -= ra.rows0; c.layer += 1; }
  c.layer += ra.step_l;
}

template <int NCH>
struct RowRegs {
  f32x4 a[NCH], b[NCH];
  float w;              // row weight (0 = padded token: nothing loaded)
  float* g;             // gradient row
};

template <int NCH, int NT>
__device__ __forceinline__ void load_row(RowRegs<NCH>& r, const Cursor& c, const LayerTable& tab, const RowAddr& ra,
                                         const uint8_t* __restrict__ row_mask, int lane, int D) {
  int om = c.o;
  if (ra.mask_period != ra.rows0) om = (ra.mask_period == 1) ? 0 : c.o % ra.mask_period;
  r.w = row_mask ? static_cast<float>(row_mask[static_cast<long>(om) * ra.rows1 + c.i]) : 1.0f;
  const float* __restrict__ pa = tab.a[c.layer] + c.o * ra.a0 + c.i * ra.a1;
  const float* __restrict__ pb = tab.b[c.layer] + c.o * ra.b0 + c.i * ra.b1;
  r.g = tab.g[c.layer] ? tab.g[c.layer] + c.o * ra.g0 + c.i * ra.g1 : nullptr;
  if (r.w != 0.0f) {    // wave-uniform: weight-0 rows (padded tokens) are never loaded
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int d = (k * kWave + lane) * 4;
      if (d < D) {
        r.a[k] = (NT & 1) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pa + d))
                          : *reinterpret_cast<const f32x4*>(pa + d);
        r.b[k] = (NT & 4) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pb + d))   // read once per step
                          : *reinterpret_cast<const f32x4*>(pb + d);
      } else {
        r.a[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        r.b[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      }
    }
  }
}

// -cos of one row (added to acc) and, with GRAD, d(-cos)/da written to the gradient row.
// (Round 3 A/B: with targets normalised once per attack the |b|^2 chain -- D FMAs, one wave reduction, one sqrt per row --
// can leave this function; measured 862-867 us against 853-854 us for the 13-map launch, i.e. no gain: the kernel waits
// for HBM, not for the vector ALU.  profiles/r03/kernel_roofline_cos_ce_b64.jsonl; not shipped.)
template <int NCH, bool GRAD, int NT>
__device__ __forceinline__ void reduce_row(const RowRegs<NCH>& r, float& acc, int lane, int D, float gscale,
                                           float cos_eps) {
  const bool live = r.w != 0.0f;
  float dot = 0.0f, na2 = 0.0f, nb2 = 0.0f;
  if (live) {
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dot += r.a[k][e] * r.b[k][e];
        na2 += r.a[k][e] * r.a[k][e];
        nb2 += r.b[k][e] * r.b[k][e];
      }
    }
    dot = wave_sum(dot);
    na2 = wave_sum(na2);
    nb2 = wave_sum(nb2);
  }
  const float na = sqrtf(na2), nb = sqrtf(nb2);
  const float dna = fmaxf(na, cos_eps), dnb = fmaxf(nb, cos_eps);
  const float inv = 1.0f / (dna * dnb);
  if (live) acc += -(r.w * (dot * inv));
  if (GRAD) {
    // d(-cos)/da = -( b/(dna*dnb) - [na > eps] * dot/(dna*dnb) * a/na^2 ), scaled by the upstream gscale
    const float gs = gscale * r.w;
    const float kb = -gs * inv;
    const float ka = (na > cos_eps) ? gs * dot * inv / (na * na) : 0.0f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int d = (k * kWave + lane) * 4;
      if (d < D) {
        f32x4 o;
        if (live) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = kb * r.b[k][e] + ka * r.a[k][e];
        } else {
          o = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        if (NT & 2) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(r.g + d));
        else *reinterpret_cast<f32x4*>(r.g + d) = o;
      }
    }
  }
}

// Sum of `count` floats in index order by one workgroup (fixed association: lane-strided, DPP, 4-wave combine).
__device__ __forceinline__ float block_sum_ordered(const float* p, int count, float* lds, bool coherent) {
  float acc = 0.0f;
  for (int i = threadIdx.x; i < count; i += kBlock)
    acc += coherent ? __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p[i];
  acc = wave_sum(acc);
  __syncthreads();                   // lds may still be read from the previous use
  if ((threadIdx.x & (kWave - 1)) == 0) lds[threadIdx.x / kWave] = acc;
  __syncthreads();
  float s = lds[0];
#pragma unroll
  for (int w = 1; w < kWavesPerBlock; ++w) s += lds[w];
  return s;
}

// Workgroup epilogue of both row kernels: 4-wave combine of the per-wave sums, publish the partial, and -- in the fold
// mode -- the workgroup that arrives last folds all partials in index order into the scalar loss.
__device__ __forceinline__ void finish_block(float acc, int lane, int wave, float* __restrict__ partial, const Fold& fold) {
  __shared__ float lds[kWavesPerBlock];
  __shared__ int lds_last;
  if (lane == 0) lds[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = lds[0];
#pragma unroll
    for (int w = 1; w < kWavesPerBlock; ++w) s += lds[w];
    if (fold.counter) {
      // publish write-through (agent scope), drain, then count this workgroup in: the last one to arrive folds
      __hip_atomic_store(partial + blockIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      lds_last = arrive_is_last(fold.counter) ? 1 : 0;
    } else {
      partial[blockIdx.x] = s;
      lds_last = 0;
    }
  }
  if (!fold.counter && blockIdx.x == 0)     // partials-only mode: unused slots read as zero by vqa_sum_partials
    for (int i = gridDim.x + threadIdx.x; i < kLossMaxBlocks; i += kBlock) partial[i] = 0.0f;
  __syncthreads();
  if (lds_last) {     // workgroup-uniform
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const float s = block_sum_ordered(partial, static_cast<int>(gridDim.x), lds, true) * fold.scale;
    if (threadIdx.x == 0) fold.loss_out[0] = fold.accumulate ? fold.loss_out[0] + s : s;
    arrive_reset(fold.counter);                                      // ready for the next launch
  }
}

template <int NCH, bool GRAD, bool PIPE2, int NT>
__global__ __launch_bounds__(kBlock) void neg_cos_rows_kernel(LayerTable tab, float* __restrict__ partial, Fold fold,
                                                              const uint8_t* __restrict__ row_mask, RowAddr ra,
                                                              int D, float gscale, float cos_eps) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) / kWave);
  const long per_layer = static_cast<long>(ra.rows0) * ra.rows1;
  const long total = per_layer * ra.n_layers;
  const long wstride = static_cast<long>(gridDim.x) * kWavesPerBlock;
  const long first = static_cast<long>(blockIdx.x) * kWavesPerBlock + wave;
  float acc = 0.0f;   // this wave's sum of -cos over its rows (same value in every lane)
  if (first < total) {
    long left = (total - first + wstride - 1) / wstride;      // rows this wave owns
    Cursor c;
    c.layer = static_cast<int>(first / per_layer);
    const long rem = first - c.layer * per_layer;
    c.o = static_cast<int>(rem / ra.rows1);
    c.i = static_cast<int>(rem - static_cast<long>(c.o) * ra.rows1);
    if (PIPE2) {
      RowRegs<NCH> r0, r1;
      load_row<NCH, NT>(r0, c, tab, ra, row_mask, lane, D);
      while (true) {                                          // two rows in flight, registers ping-pong r0 / r1
        if (left > 1) { advance(c, ra); load_row<NCH, NT>(r1, c, tab, ra, row_mask, lane, D); }
        reduce_row<NCH, GRAD, NT>(r0, acc, lane, D, gscale, cos_eps);
        if (--left == 0) break;
        if (left > 1) { advance(c, ra); load_row<NCH, NT>(r0, c, tab, ra, row_mask, lane, D); }
        reduce_row<NCH, GRAD, NT>(r1, acc, lane, D, gscale, cos_eps);
        if (--left == 0) break;
      }
    } else {
      RowRegs<NCH> r0;
      for (; left > 0; --left) {
        load_row<NCH, NT>(r0, c, tab, ra, row_mask, lane, D);
        reduce_row<NCH, GRAD, NT>(r0, acc, lane, D, gscale, cos_eps);
        if (left > 1) advance(c, ra);
      }
    }
  }
  finish_block(acc, lane, wave, partial, fold);
}

// ---------------------------------------------------------------------------------------------- whole-chunk rows
// The same arithmetic for rows of exactly NCH * 256 floats (768, 1024: every feature map of the bundled encoders), written
// WITHOUT a branch around any memory instruction.  The compiler counts a wave's outstanding loads and stores in one
// counter (vmcnt) and lets a wait name "all but the N youngest"; a load or store that only SOME path issues makes N
// unknown behind the join, and the wait it then writes is for everything.  In the general kernel above (per-lane `d < D`
// guards, weight-0 rows skipped, the next row loaded only if there is one) that turned "two rows in flight" into: the
// reduction of a row waits for the NEXT row's loads as well, and each 16-byte piece of the gradient waits for the
// previous piece's store to be acknowledged (tools/isa_loop_mix.py --waits).  Here every row issues the same
// instructions: weight-0 rows (padded tokens, < 1 % of the attack's maps) are loaded too and select zeros afterwards,
// the row after the last one re-reads the last one, and the mask byte is read through a pointer that is always valid.
// Results are bitwise those of the general kernel.  Measured (13 maps of 64 x 617 x 768, tools/kernel_roofline.py,
// alternating): 845.7 against 860.0 us without row weights (+1.7 %), 843 against 842 us with 30 of 617 rows padded
// (it reads the 4.9 % of dead rows the other skips): the launch is bound by the memory system either way, what the
// exact waits buy is small -- but they are exact.
template <int NCH, int NT>
__device__ __forceinline__ void load_row_full(RowRegs<NCH>& r, const Cursor& c, const LayerTable& tab, const RowAddr& ra,
                                              const uint8_t* __restrict__ row_mask, int lane) {
  int om = c.o;
  if (ra.mask_period != ra.rows0) om = (ra.mask_period == 1) ? 0 : c.o % ra.mask_period;
  const float* __restrict__ pa = tab.a[c.layer] + c.o * ra.a0 + c.i * ra.a1;
  const float* __restrict__ pb = tab.b[c.layer] + c.o * ra.b0 + c.i * ra.b1;
  // no mask: the byte comes from the row itself and is not used
  const uint8_t* mp = row_mask ? row_mask + static_cast<long>(om) * ra.rows1 + c.i : reinterpret_cast<const uint8_t*>(pa);
  const float wm = static_cast<float>(*mp);
  r.w = row_mask ? wm : 1.0f;
  r.g = tab.g[c.layer] ? tab.g[c.layer] + c.o * ra.g0 + c.i * ra.g1 : nullptr;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int d = (k * kWave + lane) * 4;
    r.a[k] = (NT & 1) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pa + d))
                      : *reinterpret_cast<const f32x4*>(pa + d);
    r.b[k] = (NT & 4) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pb + d))
                      : *reinterpret_cast<const f32x4*>(pb + d);
  }
}

template <int NCH, bool GRAD, int NT>
__device__ __forceinline__ void reduce_row_full(const RowRegs<NCH>& r, float& acc, int lane, float gscale, float cos_eps) {
  const bool live = r.w != 0.0f;
  float dot = 0.0f, na2 = 0.0f, nb2 = 0.0f;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      dot += r.a[k][e] * r.b[k][e];
      na2 += r.a[k][e] * r.a[k][e];
      nb2 += r.b[k][e] * r.b[k][e];
    }
  }
  dot = wave_sum(dot);
  na2 = wave_sum(na2);
  nb2 = wave_sum(nb2);
  if (!live) {                       // what the general kernel computes with for a row it never read
    dot = 0.0f;
    na2 = 0.0f;
    nb2 = 0.0f;
  }
  const float na = sqrtf(na2), nb = sqrtf(nb2);
  const float dna = fmaxf(na, cos_eps), dnb = fmaxf(nb, cos_eps);
  const float inv = 1.0f / (dna * dnb);
  const float term = -(r.w * (dot * inv));
  acc += live ? term : 0.0f;
  if (GRAD) {
    const float gs = gscale * r.w;
    const float kb = -gs * inv;
    const float ka = (na > cos_eps) ? gs * dot * inv / (na * na) : 0.0f;
    f32x4 o[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = kb * r.b[k][e] + ka * r.a[k][e];
        o[k][e] = live ? v : 0.0f;
      }
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) {      // the row's stores back to back, nothing waits between them
      const int d = (k * kWave + lane) * 4;
      if (NT & 2) __builtin_nontemporal_store(o[k], reinterpret_cast<f32x4*>(r.g + d));
      else *reinterpret_cast<f32x4*>(r.g + d) = o[k];
    }
  }
}

template <int NCH, bool GRAD, int NT>
__global__ __launch_bounds__(kBlock) void neg_cos_rows_full_kernel(LayerTable tab, float* __restrict__ partial, Fold fold,
                                                                   const uint8_t* __restrict__ row_mask, RowAddr ra,
                                                                   float gscale, float cos_eps) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) / kWave);
  const long per_layer = static_cast<long>(ra.rows0) * ra.rows1;
  const long total = per_layer * ra.n_layers;
  const long wstride = static_cast<long>(gridDim.x) * kWavesPerBlock;
  const long first = static_cast<long>(blockIdx.x) * kWavesPerBlock + wave;
  float acc = 0.0f;
  if (first < total) {
    long left = (total - first + wstride - 1) / wstride;      // rows this wave owns
    Cursor c;
    c.layer = static_cast<int>(first / per_layer);
    const long rem = first - c.layer * per_layer;
    c.o = static_cast<int>(rem / ra.rows1);
    c.i = static_cast<int>(rem - static_cast<long>(c.o) * ra.rows1);
    RowRegs<NCH> r0, r1;
    load_row_full<NCH, NT>(r0, c, tab, ra, row_mask, lane);
    while (true) {                                            // two rows in flight, registers ping-pong r0 / r1
      if (left > 1) advance(c, ra);                           // no next row: the cursor stays, the row is read again
      load_row_full<NCH, NT>(r1, c, tab, ra, row_mask, lane);
      reduce_row_full<NCH, GRAD, NT>(r0, acc, lane, gscale, cos_eps);
      if (--left == 0) break;
      if (left > 1) advance(c, ra);
      load_row_full<NCH, NT>(r0, c, tab, ra, row_mask, lane);
      reduce_row_full<NCH, GRAD, NT>(r1, acc, lane, gscale, cos_eps);
      if (--left == 0) break;
    }
  }
  finish_block(acc, lane, wave, partial, fold);
}

__global__ __launch_bounds__(kBlock) void sum_partials_kernel(const float* __restrict__ partial, int count,
                                                              float* __restrict__ dst, int accumulate,
                                                              float scale) {
  __shared__ float lds[kWavesPerBlock];
  const float s = block_sum_ordered(partial, count, lds, false) * scale;
  if (threadIdx.x == 0) dst[0] = accumulate ? dst[0] + s : s;
}

// launch shape (VQA_KNOB: compile-time constants in the shipped library, see common.hpp)
VQA_KNOB g_loss_blocks_per_cu = 0;   // option 6: 0 = exactly the resident workgroups, n > 0 = n per CU
VQA_KNOB g_loss_rows_in_flight = 2;  // option 7: 1 | 2
VQA_KNOB g_loss_nt = -1;             // option 8: bit0 nt loads of a, bit1 nt stores of grad, bit2 nt loads of b; -1 = by size
VQA_KNOB g_loss_full_rows = 1;       // option 12: rows of NCH * 256 floats take neg_cos_rows_full_kernel (0: the general one)

template <int NCH, bool GRAD, bool PIPE2, int NT>
static int launch_cos_inst(hipStream_t st, const LayerTable& tab, float* partial, const Fold& fold, const uint8_t* mask,
                           RowAddr ra, int D, float gscale, float eps) {
  // whole-chunk rows of token maps take the straight-line kernel; short rows-per-sample maps (text-only: most of a
  // padded question's rows can be weight 0, and that kernel reads them) keep the one that skips weight-0 rows
  const bool full = PIPE2 && D == NCH * 256 && g_loss_full_rows && (!mask || ra.rows1 > 128);
  static int occupancy_grid[2] = {0, 0};      // per instantiation and kernel: occupancy x CUs
  if (occupancy_grid[full] == 0)
    occupancy_grid[full] = full ? resident_blocks(neg_cos_rows_full_kernel<NCH, GRAD, NT>, kBlock, 6)
                                : resident_blocks(neg_cos_rows_kernel<NCH, GRAD, PIPE2, NT>, kBlock, 6);
  int resident = g_loss_blocks_per_cu > 0 ? g_loss_blocks_per_cu * cu_count() : occupancy_grid[full];
  if (resident > kLossMaxBlocks) resident = kLossMaxBlocks;
  const long total = static_cast<long>(ra.rows0) * ra.rows1 * ra.n_layers;
  long want = (total + kWavesPerBlock - 1) / kWavesPerBlock;
  if (want < 1) want = 1;
  const int grid = static_cast<int>(want < resident ? want : resident);
  if (total > 0) {              // digits of the grid stride; an empty launch still writes its zero partial / loss
    const long wstride = static_cast<long>(grid) * kWavesPerBlock;
    ra.step_i = static_cast<int>(wstride % ra.rows1);
    const long so = wstride / ra.rows1;
    ra.step_o = static_cast<int>(so % ra.rows0);
    ra.step_l = static_cast<int>(so / ra.rows0);
  }
  if (full)
    neg_cos_rows_full_kernel<NCH, GRAD, NT><<<grid, kBlock, 0, st>>>(tab, partial, fold, mask, ra, gscale, eps);
  else
    neg_cos_rows_kernel<NCH, GRAD, PIPE2, NT><<<grid, kBlock, 0, st>>>(tab, partial, fold, mask, ra, D, gscale, eps);
  return launch_status();
}

template <int NCH>
static int launch_cos(bool grad, hipStream_t st, const LayerTable& tab, float* partial, const Fold& fold,
                      const uint8_t* mask, const RowAddr& ra, int D, float gscale, float eps) {
#define VQA_COS_GO(G, P, N) launch_cos_inst<NCH, G, P, N>(st, tab, partial, fold, mask, ra, D, gscale, eps)
#ifdef VQA_TUNING
  if (!grad) return g_loss_rows_in_flight == 2 ? VQA_COS_GO(false, true, 4) : VQA_COS_GO(false, false, 4);
  if (g_loss_rows_in_flight != 2) return VQA_COS_GO(true, false, 4);
  switch (g_loss_nt) {          // A/B knob of the loss + gradient kernel (two rows in flight); -1 = the shipped rule
    case 0: return VQA_COS_GO(true, true, 0);
    case 4: return VQA_COS_GO(true, true, 4);
    case 5: return VQA_COS_GO(true, true, 5);
    case 6: return VQA_COS_GO(true, true, 6);
    case 7: return VQA_COS_GO(true, true, 7);
    default: break;
  }
#endif
  // Shipped: two rows in flight per wave; non-temporal loads of the targets, and -- once a launch moves more than the
  // 256 MB Infinity Cache can hold (the step kernel's rule) -- of the features and non-temporal gradient stores as well:
  // inside the attack (13 maps, 4.5 GB) 845 -> 825 us per launch, end to end unchanged
  // (profiles/r06/loss/loss_nt_insitu.jsonl); a small launch keeps its gradient in the cache for the backward.
  const long total_bytes = static_cast<long>(ra.rows0) * ra.rows1 * ra.n_layers * D * 12;
  if (grad && total_bytes > (256L << 20)) return VQA_COS_GO(true, true, 7);
  return grad ? VQA_COS_GO(true, true, 4) : VQA_COS_GO(false, true, 4);
#undef VQA_COS_GO
}

static int launch_cos_table(const LayerTable& tab, bool grad, float* partial, const Fold& fold,
                            const uint8_t* row_mask, const RowAddr& ra, int D, float gscale, float cos_eps,
                            hipStream_t st) {
  switch ((D + 255) / 256) {
    case 1: return launch_cos<1>(grad, st, tab, partial, fold, row_mask, ra, D, gscale, cos_eps);
    case 2: return launch_cos<2>(grad, st, tab, partial, fold, row_mask, ra, D, gscale, cos_eps);
    case 3: return launch_cos<3>(grad, st, tab, partial, fold, row_mask, ra, D, gscale, cos_eps);
    case 4: return launch_cos<4>(grad, st, tab, partial, fold, row_mask, ra, D, gscale, cos_eps);
    case 5: return launch_cos<5>(grad, st, tab, partial, fold, row_mask, ra, D, gscale, cos_eps);
    case 6: return launch_cos<6>(grad, st, tab, partial, fold, row_mask, ra, D, gscale, cos_eps);
    case 7: return launch_cos<7>(grad, st, tab, partial, fold, row_mask, ra, D, gscale, cos_eps);
    default: return launch_cos<8>(grad, st, tab, partial, fold, row_mask, ra, D, gscale, cos_eps);
  }
}

// shared argument checks of the two entry points; fills `ra` (grid-stride digits are set at launch)
static int make_row_addr(RowAddr& ra, long rows0, long rows1, int n_layers, int D, long a0, long a1, long b0, long b1,
                         long g0, long g1, bool has_grad, const uint8_t* row_mask, long mask_period) {
  if (rows0 < 0 || rows1 < 0 || D <= 0 || D > 2048 || (D & 3)) return VQA_ERR_SHAPE;
  if ((a0 | a1 | b0 | b1) & 3) return VQA_ERR_SHAPE;
  if (has_grad && ((g0 | g1) & 3)) return VQA_ERR_SHAPE;
  if (row_mask && mask_period <= 0) return VQA_ERR_SHAPE;
  if (rows0 > 0x3fffffffL || rows1 > 0x3fffffffL || mask_period > 0x3fffffffL) return VQA_ERR_SHAPE;
  if (rows0 * rows1 > 0x3fffffffL / (n_layers > 0 ? n_layers : 1)) return VQA_ERR_SHAPE;   // int row cursor
  ra = RowAddr{};
  ra.rows0 = static_cast<int>(rows1 == 0 ? 0 : rows0);
  ra.rows1 = static_cast<int>(rows1 > 0 ? rows1 : 1);
  ra.n_layers = n_layers;
  ra.mask_period = row_mask ? static_cast<int>(mask_period) : 1;
  ra.a0 = a0; ra.a1 = a1; ra.b0 = b0; ra.b1 = b1; ra.g0 = g0; ra.g1 = g1;
  return VQA_OK;
}

static Fold make_fold(float* partial, float* loss_out, int accumulate, float scale) {
  if (!loss_out) return Fold{nullptr, nullptr, 0, 0.0f};
  return Fold{reinterpret_cast<unsigned*>(partial + kLossMaxBlocks), loss_out, accumulate, scale};
}

}  // namespace vqa

using namespace vqa;

extern "C" {

int vqa_neg_cos_partials(void) { return kLossMaxBlocks + kArriveWords; }   // partials + the arrival counters

#ifdef VQA_TUNING
int vqa_loss_set_option(int which, int value) {     // reached through vqa_set_option(6 | 7 | 8, value)
  if (which == 6) {
    if (value < 0 || value > 8) return VQA_ERR_SHAPE;
    g_loss_blocks_per_cu = value;
    return VQA_OK;
  }
  if (which == 12) {
    g_loss_full_rows = value ? 1 : 0;
    return VQA_OK;
  }
  if (which == 8) {
    if (value != -1 && value != 0 && value != 4 && value != 5 && value != 6 && value != 7) return VQA_ERR_SHAPE;
    g_loss_nt = value;
    return VQA_OK;
  }
  if (value != 1 && value != 2) return VQA_ERR_SHAPE;
  g_loss_rows_in_flight = value;
  return VQA_OK;
}
#endif  // VQA_TUNING

int vqa_neg_cos_rows(const float* a, const float* b, float* ga, float* partial, const uint8_t* row_mask,
                     long mask_period, long rows0, long rows1, int D, long a_stride0, long a_stride1,
                     long b_stride0, long b_stride1, long g_stride0, long g_stride1, float gscale, float cos_eps,
                     float* loss_out, int accumulate, vqa_stream_t stream) {
  clear_stale_error();
  if (!a || !b || !partial) return VQA_ERR_NULL;
  RowAddr ra;
  const int rc = make_row_addr(ra, rows0, rows1, 1, D, a_stride0, a_stride1, b_stride0, b_stride1, g_stride0,
                               g_stride1, ga != nullptr, row_mask, mask_period);
  if (rc != VQA_OK) return rc;
  if (!aligned16(a) || !aligned16(b) || (ga && !aligned16(ga))) return VQA_ERR_ALIGN;
  LayerTable tab{};
  tab.a[0] = a;
  tab.b[0] = b;
  tab.g[0] = ga;
  return launch_cos_table(tab, ga != nullptr, partial, make_fold(partial, loss_out, accumulate, gscale), row_mask, ra,
                          D, gscale, cos_eps, static_cast<hipStream_t>(stream));
}

int vqa_neg_cos_max_layers(void) { return kMaxLayers; }

int vqa_neg_cos_rows_multi(const float* const* a, const float* const* b, float* const* ga, int n_layers,
                           float* partial, const uint8_t* row_mask, long mask_period, long rows0, long rows1, int D,
                           long a_stride0, long a_stride1, long b_stride0, long b_stride1, long g_stride0,
                           long g_stride1, float gscale, float cos_eps, float* loss_out, int accumulate,
                           vqa_stream_t stream) {
  clear_stale_error();
  if (!a || !b || !partial) return VQA_ERR_NULL;
  if (n_layers < 1 || n_layers > kMaxLayers) return VQA_ERR_SHAPE;
  RowAddr ra;
  const int rc = make_row_addr(ra, rows0, rows1, n_layers, D, a_stride0, a_stride1, b_stride0, b_stride1, g_stride0,
                               g_stride1, ga != nullptr, row_mask, mask_period);
  if (rc != VQA_OK) return rc;
  LayerTable tab{};
  for (int l = 0; l < n_layers; ++l) {
    if (!a[l] || !b[l] || (ga && !ga[l])) return VQA_ERR_NULL;
    if (!aligned16(a[l]) || !aligned16(b[l]) || (ga && !aligned16(ga[l]))) return VQA_ERR_ALIGN;
    tab.a[l] = a[l];
    tab.b[l] = b[l];
    tab.g[l] = ga ? ga[l] : nullptr;
  }
  return launch_cos_table(tab, ga != nullptr, partial, make_fold(partial, loss_out, accumulate, gscale), row_mask, ra,
                          D, gscale, cos_eps, static_cast<hipStream_t>(stream));
}

int vqa_sum_partials(const float* partial, int count, float* dst, int accumulate, float scale,
                     vqa_stream_t stream) {
  clear_stale_error();
  if (!partial || !dst) return VQA_ERR_NULL;
  if (count < 0) return VQA_ERR_SHAPE;
  sum_partials_kernel<<<1, kBlock, 0, static_cast<hipStream_t>(stream)>>>(partial, count, dst, accumulate, scale);
  return launch_status();
}

}  // extern "C"
