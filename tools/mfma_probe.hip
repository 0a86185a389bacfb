// Matrix-pipe probe for MI355X (gfx950), v_mfma_f32_32x32x2_f32: what does one SIMD sustain when 1..4 waves share it,
// on one dependent accumulator chain or two, with the A operand from registers or read from LDS the way attn.hip's
// S^T phase reads it, and with a barrier per 64 MFMAs?  Stand-alone (no torch):
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o tools/mfma_probe && tools/mfma_probe
// One JSON line per configuration: cycles per MFMA per SIMD from the kernel's own clock (s_memtime based
// wall_clock64 is 100 MHz, so the GPU clock is taken from hipDeviceProp) and from the HIP-event time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kBlock = 256;
constexpr int kIters = 256;          // outer iterations; 64 MFMAs each

// MODE 0: one chain, register operands.  1: two chains.  2: one chain, A from LDS (stride-65 rows).  3: mode 2 + a
// barrier per iteration.  4: two chains, A from LDS, barrier per iteration.
template <int MODE>
__global__ __launch_bounds__(kBlock, 2) void probe(float* out, float seed) {
  __shared__ float tile[32 * 65];
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  for (int i = threadIdx.x; i < 32 * 65; i += kBlock) tile[i] = seed * static_cast<float>(i & 7);
  __syncthreads();
  f32x16 a0 = {0}, a1 = {0};
  float b[32];
#pragma unroll
  for (int s = 0; s < 32; ++s) b[s] = seed * static_cast<float>(s + lane);
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        float a;
        if (MODE >= 2) a = tile[r * 65 + s + 32 * h];
        else a = b[(s + 1) & 31];
        if (MODE == 1 || MODE == 4) {
          if (s & 1) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[s], a1, 0, 0, 0);
          else a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[s], a0, 0, 0, 0);
        } else {
          a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[s], a0, 0, 0, 0);
        }
      }
    }
    if (MODE >= 3) __syncthreads();
  }
  float acc = 0.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += a0[i] + a1[i];
  if (acc == 123.456f) out[0] = acc;
}

// MODE 5..7: the skeleton of attn.hip's forward tile loop without the staging: S^T chain (A from a stride-65 LDS
// tile, B from registers), [6, 7: a softmax-like VALU block: max, 16 exp, sum, rescale of both accumulators,]
// P.V on two accumulators (A from a stride-72 LDS tile, B = the S^T accumulator registers), [7: no] barrier.
template <int MODE>
__global__ __launch_bounds__(kBlock, 2) void attn_like(float* out, float seed) {
  __shared__ float Ks[32 * 65];
  __shared__ float Vs[32 * 72];
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  for (int i = threadIdx.x; i < 32 * 65; i += kBlock) Ks[i] = seed * static_cast<float>(i & 7);
  for (int i = threadIdx.x; i < 32 * 72; i += kBlock) Vs[i] = seed * static_cast<float>(i & 3);
  __syncthreads();
  f32x16 o0 = {0}, o1 = {0};
  float qf[32];
#pragma unroll
  for (int s = 0; s < 32; ++s) qf[s] = seed * static_cast<float>(s + lane);
  float m = 0.0f, l = 0.0f;
  for (int it = 0; it < kIters; ++it) {
    f32x16 st = {0};
#pragma unroll
    for (int s = 0; s < 32; ++s) st = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[r * 65 + s + 32 * h], qf[s], st, 0, 0, 0);
    if (MODE >= 6) {
      float mx = st[0];
#pragma unroll
      for (int i = 1; i < 16; ++i) mx = fmaxf(mx, st[i]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m, mx);
      const float alpha = __expf(m - m_new);
      float rs = 0.0f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        st[i] = __expf(st[i] - m_new);
        rs += st[i];
      }
      l = l * alpha + rs;
      m = m_new;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        o0[i] *= alpha;
        o1[i] *= alpha;
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = (i & 3) + 8 * (i >> 2) + 4 * h;
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[key * 72 + r], st[i], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[key * 72 + 32 + r], st[i], o1, 0, 0, 0);
    }
    if (MODE != 7) __syncthreads();
  }
  float acc = l;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += o0[i] + o1[i];
  if (acc == 123.456f) out[0] = acc;
}

// MODE 8 / 9 (round 6): the same work as mode 6, SOFTWARE-PIPELINED across tiles -- the S^T chain of tile t + 1 is
// independent of the softmax of tile t, so its 32 MFMAs can be issued with tile t's vector block in their shadow (each
// dependent MFMA waits ~64 cycles for its predecessor; does the wave issue independent VALU work meanwhile, or does vector
// issue serialise with the matrix pipe as modes 6 / 7 suggested?).  8: source order S-chain, softmax, PV -- the compiler
// schedules; 9: the interleaving forced with sched_group_barrier (1 MFMA : 3 VALU).  Two accumulator sets ping-pong (no
// register copies).
template <int MODE>
__global__ __launch_bounds__(kBlock, 2) void attn_pipe(float* out, float seed) {
  __shared__ float Ks[32 * 65];
  __shared__ float Vs[32 * 72];
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  for (int i = threadIdx.x; i < 32 * 65; i += kBlock) Ks[i] = seed * static_cast<float>(i & 7);
  for (int i = threadIdx.x; i < 32 * 72; i += kBlock) Vs[i] = seed * static_cast<float>(i & 3);
  __syncthreads();
  f32x16 o0 = {0}, o1 = {0};
  float qf[32];
#pragma unroll
  for (int s = 0; s < 32; ++s) qf[s] = seed * static_cast<float>(s + lane);
  float m = 0.0f, l = 0.0f;
  auto tile = [&](f32x16& cur, f32x16& nxt) {
    nxt = f32x16{0};
#pragma unroll
    for (int s = 0; s < 32; ++s) nxt = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[r * 65 + s + 32 * h], qf[s], nxt, 0, 0, 0);
    float mx = cur[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, cur[i]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m, mx);
    const float alpha = __expf(m - m_new);
    float rs = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      cur[i] = __expf(cur[i] - m_new);
      rs += cur[i];
    }
    l = l * alpha + rs;
    m = m_new;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      o0[i] *= alpha;
      o1[i] *= alpha;
    }
    if (MODE == 9) {
#pragma unroll
      for (int g = 0; g < 32; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);       // three VALU instructions in its shadow
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = (i & 3) + 8 * (i >> 2) + 4 * h;
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[key * 72 + r], cur[i], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[key * 72 + 32 + r], cur[i], o1, 0, 0, 0);
    }
    __syncthreads();
  };
  f32x16 sa = {0}, sb = {0};
  for (int it = 0; it < kIters; it += 2) {
    tile(sa, sb);
    tile(sb, sa);
  }
  float acc = l;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += o0[i] + o1[i] + sa[i];
  if (acc == 123.456f) out[0] = acc;
}

template <int MODE>
static void launch(int grid, float* out) {
  if (MODE >= 8) attn_pipe<MODE><<<grid, kBlock>>>(out, 0.0f);
  else if (MODE >= 5) attn_like<MODE><<<grid, kBlock>>>(out, 0.0f);
  else probe<(MODE < 5 ? MODE : 0)><<<grid, kBlock>>>(out, 0.0f);
}

template <int MODE>
static void run(int waves_per_simd, int cus, double mhz, float* out) {
  const int grid = cus * waves_per_simd;        // 4 waves per workgroup = one per SIMD
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  launch<MODE>(grid, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  launch<MODE>(grid, out);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0.0f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double mfma_per_simd = static_cast<double>(waves_per_simd) * kIters * 64;
  const double cyc = ms * 1e-3 * mhz * 1e6 / mfma_per_simd;
  const double tflops = static_cast<double>(grid) * 4 * kIters * 64 * 4096.0 / (ms * 1e-3) / 1e12;
  printf("{\"mode\": %d, \"waves_per_simd\": %d, \"us\": %.1f, \"cycles_per_mfma_at_%.0fMHz\": %.1f, \"TFLOPs\": %.1f}\n",
         MODE, waves_per_simd, ms * 1e3, mhz, cyc, tflops);
  fflush(stdout);
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const double mhz = p.clockRate / 1e3;
  float* out;
  CK(hipMalloc(&out, 64));
  for (int w = 1; w <= 4; ++w) {
    if (w <= 2 || true) {
      run<0>(w, p.multiProcessorCount, mhz, out);
      run<1>(w, p.multiProcessorCount, mhz, out);
      run<2>(w, p.multiProcessorCount, mhz, out);
      run<3>(w, p.multiProcessorCount, mhz, out);
      run<4>(w, p.multiProcessorCount, mhz, out);
      run<5>(w, p.multiProcessorCount, mhz, out);
      run<6>(w, p.multiProcessorCount, mhz, out);
      run<7>(w, p.multiProcessorCount, mhz, out);
      run<8>(w, p.multiProcessorCount, mhz, out);
      run<9>(w, p.multiProcessorCount, mhz, out);
    }
  }
  return 0;
}
