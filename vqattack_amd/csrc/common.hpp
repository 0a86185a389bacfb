// Shared device helpers for the gfx950 kernels of the VQAttack PGD path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vqattack_hip.h"

namespace vqa {

constexpr int kWave = 64;          // CDNA4 wavefront
constexpr int kBlock = 256;        // 4 waves per workgroup, one per SIMD
constexpr int kMaxBlocks = 256 * 8;  // 256 CUs x 8 resident 256-thread workgroups

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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, kWave));
  return v;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline bool aligned4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3u) == 0; }

inline int blocks_for(size_t work_items, int per_block, int cap = kMaxBlocks) {
  size_t b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > static_cast<size_t>(cap)) b = cap;
  return static_cast<int>(b);
}

inline int launch_status() { return static_cast<int>(hipGetLastError()); }

}  // namespace vqa
