// Shared device helpers for the gfx950 kernels of the VQAttack PGD path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vqattack_hip.h"

namespace vqa {

constexpr int kWave = 64;          // CDNA4 wavefront
constexpr int kBlock = 256;        // 4 waves per workgroup, one per SIMD
constexpr int kBlocksPerCu = 8;    // default grid cap of a grid-stride launch: 8 workgroups per compute unit (grid_cap())

// Launch-shape constants of the kernels.  The shipped library runs ONE variant, fixed at compile time (chosen from the
// sweeps recorded in DESIGN.md); a build with -DVQA_TUNING (tools/ only: `python -m vqattack_amd.build --tuning`) turns
// the constants into process-wide knobs behind vqa_set_option() and also compiles the alternatives the sweeps compare.
#ifdef VQA_TUNING
#define VQA_KNOB static int
#else
#define VQA_KNOB [[maybe_unused]] constexpr int
#endif

// torch.sign: (g > 0) - (g < 0); sign(+-0) = 0 and sign(NaN) = 0.
__device__ __forceinline__ float sign_torch(float g) {
  return (g > 0.0f ? 1.0f : 0.0f) - (g < 0.0f ? 1.0f : 0.0f);
}

// torch.clamp(v, lo, hi) = min(max(v, lo), hi) with NaN propagated (fminf/fmaxf drop NaN).
__device__ __forceinline__ float clamp_torch(float v, float lo, float hi) {
  float r = fminf(fmaxf(v, lo), hi);
  return (v != v) ? v : r;
}

__device__ __forceinline__ bool out_of_range(float v, float lo, float hi) {
  return !(v >= lo) || !(v <= hi);   // true for NaN, like torch.all(ge) / torch.all(le) failing
}

// Cross-lane reductions on the DPP path (v_add_f32_dpp / v_max_f32_dpp: one VALU op per step, no LDS crossbar
// traffic and no lgkmcnt wait, unlike __shfl_xor which lowers to ds_bpermute_b32 on gfx9): two quad permutes, the
// two row mirrors, then row_bcast:15 / row_bcast:31 funnel the four 16-lane rows into lane 63, which is broadcast
// through an SGPR.  Every lane returns the same bits; the summation order is fixed, so results are reproducible.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                               CTRL, ROW_MASK, 0xF, false));
}
constexpr int kDppQuadSwap1 = 0xB1;     // quad_perm:[1,0,3,2]
constexpr int kDppQuadSwap2 = 0x4E;     // quad_perm:[2,3,0,1]
constexpr int kDppRowHalfMirror = 0x141;
constexpr int kDppRowMirror = 0x140;
constexpr int kDppRowBcast15 = 0x142;
constexpr int kDppRowBcast31 = 0x143;

__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_move<kDppQuadSwap1, 0xF>(0.0f, v);
  v += dpp_move<kDppQuadSwap2, 0xF>(0.0f, v);
  v += dpp_move<kDppRowHalfMirror, 0xF>(0.0f, v);
  v += dpp_move<kDppRowMirror, 0xF>(0.0f, v);
  v += dpp_move<kDppRowBcast15, 0xA>(0.0f, v);   // rows 1 and 3 += last lane of rows 0 and 2
  v += dpp_move<kDppRowBcast31, 0xC>(0.0f, v);   // rows 2 and 3 += lane 31
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_move<kDppQuadSwap1, 0xF>(v, v));
  v = fmaxf(v, dpp_move<kDppQuadSwap2, 0xF>(v, v));
  v = fmaxf(v, dpp_move<kDppRowHalfMirror, 0xF>(v, v));
  v = fmaxf(v, dpp_move<kDppRowMirror, 0xF>(v, v));
  v = fmaxf(v, dpp_move<kDppRowBcast15, 0xA>(v, v));   // rows outside the mask keep `old` = v: max(v, v)
  v = fmaxf(v, dpp_move<kDppRowBcast31, 0xC>(v, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ---- "last workgroup folds" arrival detection ------------------------------------------------------------------
// Every workgroup publishes its result with an agent-scope (write-through) store, drains it, and then counts itself in
// with RELAXED atomics -- no release fence, which would write back the whole L2 per workgroup.  Arrivals are counted
// in two levels (kArriveGroups group counters 64 bytes apart, then one top counter) so that at most ~grid/32 + 32
// same-address atomics serialise instead of `grid`.  Exactly one workgroup per launch gets `true`; it must issue an
// agent-scope acquire before reading what the others published, and call arrive_reset() when done.
// The counter block (kArriveWords unsigned) must be zero when the launch starts.
constexpr int kArriveGroups = 32;
constexpr int kArriveStride = 16;                                   // words between counters (64 bytes)
constexpr int kArriveWords = (kArriveGroups + 1) * kArriveStride;

__device__ __forceinline__ bool arrive_is_last(unsigned* counters) {    // call from ONE thread per workgroup
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // the published stores have left this CU
  const unsigned grid = gridDim.x;
  const unsigned g = blockIdx.x % kArriveGroups;
  const unsigned members = grid / kArriveGroups + (g < grid % kArriveGroups ? 1u : 0u);
  const unsigned prev = __hip_atomic_fetch_add(counters + g * kArriveStride, 1u, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
  if (prev + 1 != members) return false;
  const unsigned groups = grid < kArriveGroups ? grid : kArriveGroups;
  const unsigned top = __hip_atomic_fetch_add(counters + kArriveGroups * kArriveStride, 1u, __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
  return top + 1 == groups;
}

__device__ __forceinline__ void arrive_reset(unsigned* counters) {      // by the folding workgroup, any of its threads
  for (int i = threadIdx.x; i <= kArriveGroups; i += blockDim.x)
    __hip_atomic_store(counters + i * kArriveStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline bool aligned4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3u) == 0; }

// Compute units of the current device, queried once per device (MI355X: 256).
inline int cu_count() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n;
  }
  return cached[dev];
}

inline int grid_cap() { return cu_count() * kBlocksPerCu; }

inline int blocks_for(size_t work_items, int per_block, int cap = 0) {      // cap <= 0: grid_cap()
  if (cap <= 0) cap = grid_cap();
  size_t b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > static_cast<size_t>(cap)) b = cap;
  return static_cast<int>(b);
}

// Workgroups of `kernel` that are resident at once on the whole device (occupancy x CUs): the grid of a persistent /
// grid-stride kernel, so that no workgroup waits for a slot and the last wave of workgroups is not part-empty.
template <class K>
inline int resident_blocks(K kernel, int block_threads, int fallback_per_cu) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block_threads, 0) != hipSuccess || per_cu <= 0)
    per_cu = fallback_per_cu;
  return per_cu * cu_count();
}

// hipGetLastError() is sticky across calls: an entry point first drops whatever an unrelated earlier HIP call left
// behind, so that launch_status() reports THIS launch only.
inline void clear_stale_error() { (void)hipGetLastError(); }

inline int launch_status() { return static_cast<int>(hipGetLastError()); }

}  // namespace vqa
