// Streaming-rate probe for MI355X (gfx950): what do plain read / write / copy / 3-read-1-write streams reach at the
// footprints of the attack's kernels, and which launch shape gets there?  Stand-alone (no torch):
//   hipcc -O3 --offload-arch=gfx950 tools/stream_probe.hip -o tools/stream_probe && tools/stream_probe
// The same file builds as a shared library (-shared -fPIC -> tools/libstream_probe.so) whose one entry point,
// vqa_probe_stream(), is what bench.py reports as the platform's streaming rate next to the kernels' fractions.
// One JSON line per configuration: {"op", "MB" (bytes moved per launch), "grid", "unroll", "nt", "chunked", "us", "GBs"}.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kBlock = 256;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// OP: 0 = read (sum into a sink), 1 = write, 2 = copy (1r:1w), 3 = step-like (3r:1w), 4 = 2r:1w (loss-like)
template <int OP, int U, int NT>
__global__ __launch_bounds__(kBlock) void stream_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b,
                                                        const f32x4* __restrict__ c, f32x4* __restrict__ out,
                                                        size_t n4, size_t chunk, float* sink) {
  const size_t tile = static_cast<size_t>(kBlock) * U;
  const size_t stride = chunk ? tile : static_cast<size_t>(gridDim.x) * tile;
  const size_t first = chunk ? static_cast<size_t>(blockIdx.x) * chunk : static_cast<size_t>(blockIdx.x) * tile;
  size_t last = chunk ? first + chunk : n4;
  if (last > n4) last = n4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // Whole tiles in a loop without a branch around a load or store (all loads of a tile in flight, exact vmcnt waits, the
  // stores together) -- the structure csrc/linf.hip has since round 6; with per-lane guards in the loop the compiler's
  // waits serialise the loads and the stores of a tile, and a "plain stream" probe must not be handicapped that way.
  size_t base = first;
  for (; base + tile <= last; base += stride) {
    f32x4 va[U], vb[U], vc[U], r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + static_cast<size_t>(u) * kBlock + threadIdx.x;
      if (OP != 1) va[u] = (NT & 1) ? __builtin_nontemporal_load(&a[i]) : a[i];
      if (OP >= 3) vb[u] = (NT & 1) ? __builtin_nontemporal_load(&b[i]) : b[i];
      if (OP == 3) vc[u] = (NT & 1) ? __builtin_nontemporal_load(&c[i]) : c[i];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (OP == 0) acc += va[u];
      if (OP == 1) r[u] = f32x4{1.f, 2.f, 3.f, 4.f};
      if (OP == 2) r[u] = va[u];
      if (OP == 3) r[u] = va[u] + vb[u] * vc[u];
      if (OP == 4) r[u] = va[u] + vb[u];
    }
    if (OP != 0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t i = base + static_cast<size_t>(u) * kBlock + threadIdx.x;
        if (NT & 2) __builtin_nontemporal_store(r[u], &out[i]); else out[i] = r[u];
      }
    }
  }
  if (base < last) {                        // the partial tile
    f32x4 va[U], vb[U], vc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + static_cast<size_t>(u) * kBlock + threadIdx.x;
      if (i < last) {
        if (OP != 1) va[u] = (NT & 1) ? __builtin_nontemporal_load(&a[i]) : a[i];
        if (OP >= 3) vb[u] = (NT & 1) ? __builtin_nontemporal_load(&b[i]) : b[i];
        if (OP == 3) vc[u] = (NT & 1) ? __builtin_nontemporal_load(&c[i]) : c[i];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + static_cast<size_t>(u) * kBlock + threadIdx.x;
      if (i < last) {
        f32x4 r;
        if (OP == 0) { acc += va[u]; continue; }
        if (OP == 1) r = f32x4{1.f, 2.f, 3.f, 4.f};
        if (OP == 2) r = va[u];
        if (OP == 3) r = va[u] + vb[u] * vc[u];
        if (OP == 4) r = va[u] + vb[u];
        if (NT & 2) __builtin_nontemporal_store(r, &out[i]); else out[i] = r;
      }
    }
  }
  if (OP == 0 && acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

struct Bufs { f32x4 *a, *b, *c, *out; float* sink; };

template <int OP, int U, int NT>
static float run(const Bufs& B, size_t n4, int grid, bool chunked, int reps) {
  size_t chunk = 0;
  if (chunked) {
    const size_t tile = static_cast<size_t>(kBlock) * U;
    chunk = ((n4 + grid - 1) / grid + tile - 1) / tile * tile;
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) stream_kernel<OP, U, NT><<<grid, kBlock>>>(B.a, B.b, B.c, B.out, n4, chunk, B.sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) stream_kernel<OP, U, NT><<<grid, kBlock>>>(B.a, B.b, B.c, B.out, n4, chunk, B.sink);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms / reps;
}

template <int OP>
static void sweep(const char* name, const Bufs& B, size_t n4, int streams, int cus) {
  const double mb = static_cast<double>(n4) * 16 * streams / 1e6;
  const int reps = 12;
  struct Cfg { int per_cu, unroll, nt; bool chunked; };
  const std::vector<Cfg> cfgs = {{8, 4, 0, false}, {8, 4, 1, false}, {8, 4, 3, false}, {8, 4, 2, false}, {8, 4, 0, true},
                                 {8, 4, 3, true}, {4, 4, 0, false}, {16, 4, 0, false}, {32, 4, 0, false},
                                 {8, 2, 0, false}, {8, 8, 0, false}, {16, 2, 0, false}, {4, 8, 0, false},
                                 {16, 8, 3, false}, {6, 4, 0, false}};
  for (const Cfg& c : cfgs) {
    const int grid = cus * c.per_cu;
    float ms = 0.f;
#define GO(U, NT) ms = run<OP, U, NT>(B, n4, grid, c.chunked, reps)
    if (c.unroll == 2) { if (c.nt == 0) GO(2, 0); else if (c.nt == 1) GO(2, 1); else if (c.nt == 2) GO(2, 2); else GO(2, 3); }
    else if (c.unroll == 8) { if (c.nt == 0) GO(8, 0); else if (c.nt == 1) GO(8, 1); else if (c.nt == 2) GO(8, 2); else GO(8, 3); }
    else { if (c.nt == 0) GO(4, 0); else if (c.nt == 1) GO(4, 1); else if (c.nt == 2) GO(4, 2); else GO(4, 3); }
#undef GO
    printf("{\"op\": \"%s\", \"MB\": %.1f, \"grid_per_cu\": %d, \"unroll\": %d, \"nt\": %d, \"chunked\": %d, \"us\": %.1f, \"GBs\": %.0f}\n",
           name, mb, c.per_cu, c.unroll, c.nt, c.chunked ? 1 : 0, ms * 1e3, mb / ms);   // MB per ms = GB/s
    fflush(stdout);
  }
}

// One configuration, self-contained (allocates and frees its buffers): returns algorithmic GB/s (bytes per buffer x
// streams / mean launch time over `reps` back-to-back launches), or a negative HIP error code.
// op: 0 read, 1 write, 2 copy 1r:1w, 3 step-like 3r:1w, 4 loss-like 2r:1w.
extern "C" double vqa_probe_stream(int op, size_t bytes, int per_cu, int unroll, int nt, int chunked, int reps) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess) return -1.0;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (op < 0 || op > 4 || bytes < 16 || reps < 1 || per_cu < 1) return -2.0;
  Bufs B{};
  const int streams = op == 3 ? 4 : (op == 4 ? 3 : (op == 2 ? 2 : 1));
  bool ok = hipMalloc(&B.a, bytes) == hipSuccess && hipMalloc(&B.out, bytes) == hipSuccess &&
            hipMalloc(&B.sink, 64) == hipSuccess;
  if (ok && op >= 3) ok = hipMalloc(&B.b, bytes) == hipSuccess;
  if (ok && op == 3) ok = hipMalloc(&B.c, bytes) == hipSuccess;
  double gbs = -3.0;
  if (ok) {
    (void)hipMemset(B.a, 0, bytes); (void)hipMemset(B.out, 0, bytes);
    if (B.b) (void)hipMemset(B.b, 0, bytes);
    if (B.c) (void)hipMemset(B.c, 0, bytes);
    const size_t n4 = bytes / 16;
    const int grid = cus * per_cu;
    const bool ch = chunked != 0;
    float ms = 0.f;
#define GO3(OP, U) (nt == 0 ? run<OP, U, 0>(B, n4, grid, ch, reps) : nt == 1 ? run<OP, U, 1>(B, n4, grid, ch, reps) : \
                    nt == 2 ? run<OP, U, 2>(B, n4, grid, ch, reps) : run<OP, U, 3>(B, n4, grid, ch, reps))
#define GO2(OP) (unroll == 2 ? GO3(OP, 2) : unroll == 8 ? GO3(OP, 8) : GO3(OP, 4))
    switch (op) {
      case 0: ms = GO2(0); break;
      case 1: ms = GO2(1); break;
      case 2: ms = GO2(2); break;
      case 3: ms = GO2(3); break;
      default: ms = GO2(4); break;
    }
#undef GO2
#undef GO3
    gbs = static_cast<double>(bytes) * streams / 1e6 / ms;
  }
  if (B.a) (void)hipFree(B.a);
  if (B.b) (void)hipFree(B.b);
  if (B.c) (void)hipFree(B.c);
  if (B.out) (void)hipFree(B.out);
  if (B.sink) (void)hipFree(B.sink);
  return gbs;
}

int main(int argc, char** argv) {
  int dev = 0, cus = 256;
  CK(hipGetDevice(&dev));
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  // bytes per buffer: the image tensor of batch 64 and batch 256 (113 MB, 453 MB), plus 1.2 GB (the guide's sweep size)
  const size_t sizes[] = {113246208ul, 452984832ul, 1207959552ul};
  for (size_t bytes : sizes) {
    Bufs B;
    CK(hipMalloc(&B.a, bytes)); CK(hipMalloc(&B.b, bytes)); CK(hipMalloc(&B.c, bytes)); CK(hipMalloc(&B.out, bytes));
    CK(hipMalloc(&B.sink, 64));
    CK(hipMemset(B.a, 0, bytes)); CK(hipMemset(B.b, 0, bytes)); CK(hipMemset(B.c, 0, bytes)); CK(hipMemset(B.out, 0, bytes));
    const size_t n4 = bytes / 16;
    sweep<0>("read", B, n4, 1, cus);
    sweep<1>("write", B, n4, 1, cus);
    sweep<2>("copy_1r1w", B, n4, 2, cus);
    sweep<4>("loss_like_2r1w", B, n4, 3, cus);
    sweep<3>("step_like_3r1w", B, n4, 4, cus);
    CK(hipFree(B.a)); CK(hipFree(B.b)); CK(hipFree(B.c)); CK(hipFree(B.out)); CK(hipFree(B.sink));
  }
  return 0;
}
