// L-infinity image-update kernels of the PGD loop (gfx950 / MI355X).
//
// All kernels here are pure HBM streams (arithmetic intensity < 0.5 flop/B): one 16-byte load per lane per
// stream, U tiles in flight per wave before the first use, 256-thread workgroups, a grid capped at
// 8 workgroups per CU that grid-strides over block-contiguous tiles.  No LDS, no MFMA: there is no reuse to
// stage and no contraction.  The arithmetic is the reference's op chain in its exact fp32 order (see
// include/vqattack_hip.h), compiled with -ffp-contract=off, so results are bit-identical to eager PyTorch.
#include "common.hpp"

namespace vqa {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- launch shape (VQA_KNOB: compile-time constants in the shipped library, see common.hpp)
// option 0: workgroups per CU in the grid.  The step kernel keeps 6 workgroups resident per CU (76 registers: all 12
// loads of a tile in flight): 12 = two whole rounds.  profiles/r06/step/step_ab.jsonl (tools/step_ab.py, alternating):
// batch 64 warm / after a cache flush 68.8 / 83.8 us at 12, 68.9 / 85.4 at 8 (previous build 69.6 / 87.0); batch 256
// 315.7 / 320.5 at 12, 318.4 / 321.7 at 8 (previous build 316.4 / 320.5) -- the memory system's ceiling either way.
VQA_KNOB g_opt_blocks_per_cu = 12;
// option 1: bit0 = nt loads of the second stream (the gradient: read exactly once), bit1 = nt stores of the result,
// bit2 = nt loads of the first and third stream (x, x0), bit3 = nt stores only when the result is larger than the
// 256 MB Infinity Cache (a smaller result is re-read from the cache by the white box's next forward; a larger one
// cannot stay resident anyway and the plain-store allocation only costs bandwidth: tools/stream_probe, +10 % at 1.8 GB)
VQA_KNOB g_opt_nontemporal = 1 | 4 | 8;
constexpr size_t kNtStoreBytes = 256ull << 20;
VQA_KNOB g_opt_unroll = 4;          // option 2: 16-byte tiles in flight per lane and stream (2, 4 or 8)
VQA_KNOB g_opt_chunked = 0;         // option 3: 0 = grid-stride tiles, 1 = one contiguous chunk per workgroup

struct StepParams {
  float eps_iter, eps, cmin, cmax;
  unsigned mode;
};

// ---- per-element bodies -----------------------------------------------------------------------
struct InitOp {   // out = clamp(x + clamp(eta, +-eps))
  static constexpr int kIn = 2;
  __device__ static float apply(const StepParams& p, float x, float eta, float, bool& bad) {
    if (p.mode & VQA_CHECK_RANGE) bad |= out_of_range(x, p.cmin, p.cmax);
    float v = x + clamp_torch(eta, -p.eps, p.eps);
    return (p.mode & VQA_CLIP) ? clamp_torch(v, p.cmin, p.cmax) : v;
  }
};
struct InitZeroOp {   // eta == 0: out = clamp(x + 0)
  static constexpr int kIn = 1;
  __device__ static float apply(const StepParams& p, float x, float, float, bool& bad) {
    if (p.mode & VQA_CHECK_RANGE) bad |= out_of_range(x, p.cmin, p.cmax);
    float v = x + 0.0f;
    return (p.mode & VQA_CLIP) ? clamp_torch(v, p.cmin, p.cmax) : v;
  }
};
struct FgmOp {   // out = clamp(x + eps_iter * sign(g))
  static constexpr int kIn = 2;
  __device__ static float apply(const StepParams& p, float x, float g, float, bool& bad) {
    if (p.mode & VQA_CHECK_RANGE) bad |= out_of_range(x, p.cmin, p.cmax);
    float v = x + p.eps_iter * sign_torch(g);
    return (p.mode & VQA_CLIP) ? clamp_torch(v, p.cmin, p.cmax) : v;
  }
};
struct StepOp {   // FGM update + projection on the eps-ball around x0
  static constexpr int kIn = 3;
  __device__ static float apply(const StepParams& p, float x, float g, float x0, bool& bad) {
    if (p.mode & VQA_CHECK_RANGE) bad |= out_of_range(x, p.cmin, p.cmax);
    float a = x + p.eps_iter * sign_torch(g);
    if (p.mode & VQA_CLIP) a = clamp_torch(a, p.cmin, p.cmax);
    float e = clamp_torch(a - x0, -p.eps, p.eps);
    float v = x0 + e;
    return (p.mode & VQA_CLIP) ? clamp_torch(v, p.cmin, p.cmax) : v;
  }
};
struct ProjectOp {   // out = clamp(x0 + clamp(adv - x0, +-eps)); stream 0 = adv, stream 1 = x0
  static constexpr int kIn = 2;
  __device__ static float apply(const StepParams& p, float adv, float x0, float, bool&) {
    float e = clamp_torch(adv - x0, -p.eps, p.eps);
    float v = x0 + e;
    return (p.mode & VQA_CLIP) ? clamp_torch(v, p.cmin, p.cmax) : v;
  }
};
struct ClipEtaOp {
  static constexpr int kIn = 1;
  __device__ static float apply(const StepParams& p, float eta, float, float, bool&) {
    return clamp_torch(eta, -p.eps, p.eps);
  }
};
struct ZeroClippedOp {   // zero_out_clipped_grads: stream 0 = grad, stream 1 = x
  static constexpr int kIn = 2;
  __device__ static float apply(const StepParams& p, float g, float x, float, bool&) {
    const float sg = sign_torch(g);
    const bool low = (x <= p.cmin) && (sg < 0.0f);
    const bool high = (x >= p.cmax) && (sg > 0.0f);
    return (low || high) ? 0.0f : g;
  }
};
struct SignScaleOp {
  static constexpr int kIn = 1;
  __device__ static float apply(const StepParams& p, float g, float, float, bool&) {
    return p.eps * sign_torch(g);
  }
};

// ---- 16-byte streaming kernel -----------------------------------------------------------------
// NT bit0: the second stream (the gradient, read exactly once) is loaded non-temporally;
// NT bit1: the result is stored non-temporally; NT bit2: the first and third stream are loaded non-temporally.
template <class Op, int U, int NT>
__global__ __launch_bounds__(kBlock) void stream4_kernel(const f32x4* s0,  // may alias out (in-place update)
                                                         const f32x4* __restrict__ s1,
                                                         const f32x4* __restrict__ s2,
                                                         f32x4* out, size_t n4,
                                                         StepParams p, int* __restrict__ flag, size_t chunk) {
  const size_t tile = static_cast<size_t>(kBlock) * U;
  // chunk == 0: tiles are dealt round-robin over the grid; chunk > 0: workgroup b owns [b*chunk, (b+1)*chunk)
  const size_t stride = chunk ? tile : static_cast<size_t>(gridDim.x) * tile;
  const size_t first = chunk ? static_cast<size_t>(blockIdx.x) * chunk : static_cast<size_t>(blockIdx.x) * tile;
  size_t last = chunk ? first + chunk : n4;
  if (last > n4) last = n4;
  bool bad = false;
  // Whole tiles first, in a loop of their own WITHOUT a branch around any load or store: the compiler counts a wave's
  // outstanding loads and stores in one counter and, behind a per-lane `i < last` guard, no longer knows how many are
  // younger than the one it needs -- the guarded form of this loop waited for the first of three loads before issuing
  // the fourth, and for every store to be acknowledged before computing the next 16 bytes (tools/isa_loop_mix.py --src
  // linf.hip --waits).  Here a tile's 3 U loads are issued back to back, consumed in order with exact waits, and its U
  // stores leave together.  Only the last tile of the buffer can be partial; it takes the guarded form below.
  size_t base = first;
  for (; base + tile <= last; base += stride) {
    f32x4 v0[U], v1[U], v2[U], r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + static_cast<size_t>(u) * kBlock + threadIdx.x;
      v0[u] = (NT & 4) ? __builtin_nontemporal_load(&s0[i]) : s0[i];
      if (Op::kIn > 1) v1[u] = (NT & 1) ? __builtin_nontemporal_load(&s1[i]) : s1[i];
      if (Op::kIn > 2) v2[u] = (NT & 4) ? __builtin_nontemporal_load(&s2[i]) : s2[i];
    }
    __builtin_amdgcn_sched_barrier(0);    // every load of the tile is in flight before the first one is waited for
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        r[u][k] = Op::apply(p, v0[u][k], Op::kIn > 1 ? v1[u][k] : 0.0f, Op::kIn > 2 ? v2[u][k] : 0.0f, bad);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + static_cast<size_t>(u) * kBlock + threadIdx.x;
      if (NT & 2)
        __builtin_nontemporal_store(r[u], &out[i]);
      else
        out[i] = r[u];
    }
  }
  if (base < last) {                      // the partial tile
    f32x4 v0[U], v1[U], v2[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + static_cast<size_t>(u) * kBlock + threadIdx.x;
      if (i < last) {
        v0[u] = (NT & 4) ? __builtin_nontemporal_load(&s0[i]) : s0[i];
        if (Op::kIn > 1) v1[u] = (NT & 1) ? __builtin_nontemporal_load(&s1[i]) : s1[i];
        if (Op::kIn > 2) v2[u] = (NT & 4) ? __builtin_nontemporal_load(&s2[i]) : s2[i];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + static_cast<size_t>(u) * kBlock + threadIdx.x;
      if (i < last) {
        f32x4 r;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          r[k] = Op::apply(p, v0[u][k], Op::kIn > 1 ? v1[u][k] : 0.0f, Op::kIn > 2 ? v2[u][k] : 0.0f, bad);
        if (NT & 2)
          __builtin_nontemporal_store(r, &out[i]);
        else
          out[i] = r;
      }
    }
  }
  if ((p.mode & VQA_CHECK_RANGE) && bad) atomicOr(flag, 1);
}

// scalar path: unaligned buffers and the <= 3 element tail
template <class Op>
__global__ __launch_bounds__(kBlock) void stream1_kernel(const float* s0,
                                                         const float* __restrict__ s1,
                                                         const float* __restrict__ s2,
                                                         float* out, size_t begin, size_t n,
                                                         StepParams p, int* __restrict__ flag) {
  bool bad = false;
  for (size_t i = begin + static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<size_t>(gridDim.x) * kBlock) {
    out[i] = Op::apply(p, s0[i], Op::kIn > 1 ? s1[i] : 0.0f, Op::kIn > 2 ? s2[i] : 0.0f, bad);
  }
  if ((p.mode & VQA_CHECK_RANGE) && bad) atomicOr(flag, 1);
}

template <class Op, int U>
static void launch_vec(int nt, int grid, hipStream_t st, const f32x4* a0, const f32x4* a1, const f32x4* a2, f32x4* o,
                       size_t n4, const StepParams& p, int* flag, size_t chunk) {
  // shipped hint set: nt loads on every read stream (1 | 4), nt stores when the result exceeds the Infinity Cache (| 2),
  // plain first stream for an in-place update (& ~4) -> {1, 3, 5, 7}
  switch (nt & 7) {
#ifdef VQA_TUNING
    case 0: stream4_kernel<Op, U, 0><<<grid, kBlock, 0, st>>>(a0, a1, a2, o, n4, p, flag, chunk); break;
    case 2: stream4_kernel<Op, U, 2><<<grid, kBlock, 0, st>>>(a0, a1, a2, o, n4, p, flag, chunk); break;
    case 4: stream4_kernel<Op, U, 4><<<grid, kBlock, 0, st>>>(a0, a1, a2, o, n4, p, flag, chunk); break;
    case 6: stream4_kernel<Op, U, 6><<<grid, kBlock, 0, st>>>(a0, a1, a2, o, n4, p, flag, chunk); break;
#endif
    case 1: stream4_kernel<Op, U, 1><<<grid, kBlock, 0, st>>>(a0, a1, a2, o, n4, p, flag, chunk); break;
    case 3: stream4_kernel<Op, U, 3><<<grid, kBlock, 0, st>>>(a0, a1, a2, o, n4, p, flag, chunk); break;
    case 5: stream4_kernel<Op, U, 5><<<grid, kBlock, 0, st>>>(a0, a1, a2, o, n4, p, flag, chunk); break;
    default: stream4_kernel<Op, U, 7><<<grid, kBlock, 0, st>>>(a0, a1, a2, o, n4, p, flag, chunk); break;
  }
}

// NTMASK restricts the non-temporal hints an op may use: bit0 only makes sense when the second stream is read once.
template <class Op, bool TUNABLE = false, int NTMASK = 7>
static int launch_stream(const float* s0, const float* s1, const float* s2, float* out, size_t n,
                         const StepParams& p, int* flag, vqa_stream_t stream) {
  if (!s0 || !out || (Op::kIn > 1 && !s1) || (Op::kIn > 2 && !s2)) return VQA_ERR_NULL;
  if ((p.mode & VQA_CHECK_RANGE) && !flag) return VQA_ERR_NULL;
  if (!aligned4(s0) || !aligned4(out) || (s1 && !aligned4(s1)) || (s2 && !aligned4(s2))) return VQA_ERR_ALIGN;
  if (n == 0) return VQA_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool vec = aligned16(s0) && aligned16(out) && (Op::kIn < 2 || aligned16(s1)) &&
                   (Op::kIn < 3 || aligned16(s2));
  size_t done = 0;
  if (vec && n >= 4) {
    const size_t n4 = n / 4;
    const int unroll = TUNABLE ? g_opt_unroll : 4;
    const int grid = blocks_for(n4, kBlock * unroll, cu_count() * g_opt_blocks_per_cu);
    size_t chunk = 0;
    if (g_opt_chunked) {   // equal contiguous chunks, rounded up to whole tiles
      const size_t tile = static_cast<size_t>(kBlock) * unroll;
      chunk = ((n4 + grid - 1) / grid + tile - 1) / tile * tile;
    }
    auto a0 = reinterpret_cast<const f32x4*>(s0);
    auto a1 = reinterpret_cast<const f32x4*>(s1);
    auto a2 = reinterpret_cast<const f32x4*>(s2);
    auto o = reinterpret_cast<f32x4*>(out);
    int nt = g_opt_nontemporal & 7;
    if ((g_opt_nontemporal & 8) && n * sizeof(float) > kNtStoreBytes) nt |= 2;
    nt &= NTMASK;
    if (s0 == out) nt &= ~4;        // in-place update: the first stream is about to be rewritten, keep it plain
#ifdef VQA_TUNING
    if (TUNABLE && unroll == 2) launch_vec<Op, 2>(nt, grid, st, a0, a1, a2, o, n4, p, flag, chunk);
    else if (TUNABLE && unroll == 8) launch_vec<Op, 8>(nt, grid, st, a0, a1, a2, o, n4, p, flag, chunk);
    else
#endif
    launch_vec<Op, 4>(nt, grid, st, a0, a1, a2, o, n4, p, flag, chunk);
    done = n4 * 4;
  }
  if (done < n) {
    const int grid = blocks_for(n - done, kBlock);
    stream1_kernel<Op><<<grid, kBlock, 0, st>>>(s0, s1, s2, out, done, n, p, flag);
  }
  return launch_status();
}

}  // namespace vqa

using namespace vqa;

extern "C" {

#ifdef VQA_TUNING
int vqa_ce_set_threads(int threads);   // ce.hip
int vqa_ce_set_variant(int variant);   // ce.hip
int vqa_loss_set_option(int which, int value);   // loss.hip
int vqa_attn_set_option(int value);              // attn.hip
int vqa_block_set_option(int value);             // block.hip

int vqa_set_option(int option, int value) {
  switch (option) {
    case 0:
      if (value < 1 || value > 64) return VQA_ERR_SHAPE;
      g_opt_blocks_per_cu = value;
      return VQA_OK;
    case 1:
      g_opt_nontemporal = value & 15;
      return VQA_OK;
    case 2:
      if (value != 2 && value != 4 && value != 8) return VQA_ERR_SHAPE;
      g_opt_unroll = value;
      return VQA_OK;
    case 3:
      g_opt_chunked = value ? 1 : 0;
      return VQA_OK;
    case 4:
      return vqa_ce_set_threads(value);
    case 5:
      return vqa_ce_set_variant(value);
    case 6:
    case 7:
    case 8:
      return vqa_loss_set_option(option, value);
    case 9:
      return vqa_attn_set_option(value);
    case 10:
      return vqa_block_set_option(value);
    case 12:
      return vqa_loss_set_option(option, value);
    default:
      return VQA_ERR_SHAPE;
  }
}
#endif  // VQA_TUNING

int vqa_linf_init(const float* x, const float* eta, float* out, size_t n, float eps, float cmin, float cmax,
                  unsigned mode, int* flag, vqa_stream_t stream) {
  clear_stale_error();
  StepParams p{0.0f, eps, cmin, cmax, mode};
  if (eta) return launch_stream<InitOp>(x, eta, nullptr, out, n, p, flag, stream);
  return launch_stream<InitZeroOp>(x, nullptr, nullptr, out, n, p, flag, stream);
}

int vqa_linf_fgm(const float* x, const float* g, float* out, size_t n, float eps_iter, float cmin, float cmax,
                 unsigned mode, int* flag, vqa_stream_t stream) {
  clear_stale_error();
  StepParams p{eps_iter, 0.0f, cmin, cmax, mode};
  return launch_stream<FgmOp>(x, g, nullptr, out, n, p, flag, stream);
}

int vqa_linf_step(const float* x, const float* g, const float* x0, float* out, size_t n, float eps_iter,
                  float eps, float cmin, float cmax, unsigned mode, int* flag, vqa_stream_t stream) {
  clear_stale_error();
  StepParams p{eps_iter, eps, cmin, cmax, mode};
  return launch_stream<StepOp, true>(x, g, x0, out, n, p, flag, stream);
}

int vqa_linf_project(const float* adv, const float* x0, float* out, size_t n, float eps, float cmin,
                     float cmax, unsigned mode, vqa_stream_t stream) {
  clear_stale_error();
  StepParams p{0.0f, eps, cmin, cmax, mode & ~VQA_CHECK_RANGE};
  return launch_stream<ProjectOp, false, 7>(adv, x0, nullptr, out, n, p, nullptr, stream);
}

int vqa_clip_eta_linf(const float* eta, float* out, size_t n, float eps, vqa_stream_t stream) {
  clear_stale_error();
  StepParams p{0.0f, eps, 0.0f, 0.0f, 0u};
  return launch_stream<ClipEtaOp>(eta, nullptr, nullptr, out, n, p, nullptr, stream);
}

int vqa_zero_out_clipped_grads(const float* grad, const float* x, float* out, size_t n, float cmin, float cmax,
                               vqa_stream_t stream) {
  clear_stale_error();
  StepParams p{0.0f, 0.0f, cmin, cmax, 0u};
  return launch_stream<ZeroClippedOp, false, 7>(grad, x, nullptr, out, n, p, nullptr, stream);
}

int vqa_optimize_linear_linf(const float* g, float* out, size_t n, float eps, vqa_stream_t stream) {
  clear_stale_error();
  StepParams p{0.0f, eps, 0.0f, 0.0f, 0u};
  return launch_stream<SignScaleOp>(g, nullptr, nullptr, out, n, p, nullptr, stream);
}

}  // extern "C"
