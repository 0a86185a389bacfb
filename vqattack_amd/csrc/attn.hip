// fp32 attention for the frozen white boxes on the CDNA4 matrix cores (gfx950 / MI355X), head dimension 64.
//
// The attack's device time is dominated by the white box's forward / backward (SURVEY.md section 2.1 lists
// `Attention.forward`, vlmo/modules/multiway_transformer.py:88-118, as a callee of the hot path): fp32 attention is
// 29 % of it.  These kernels compute  softmax(scale * Q K^T + bias) V  and its gradients in exact fp32 on
// v_mfma_f32_32x32x2_f32 (one rounding per product, fp32 accumulation -- the arithmetic class of the reference's
// `q.float() @ k.float().transpose(-2, -1)`), flash-style: the S x S scores never reach HBM.
//
// MFMA 32x32x2 f32 lane maps (lane l, r = l & 31, h = l >> 5):  A[row r][k = h],  B[k = h][col r],
// C/D[row = (i & 3) + 8 (i >> 2) + 4 h][col r] for accumulator register i in [0, 16).
//
// Forward (one workgroup = 4 waves = 128 queries of one (batch, head); K / V tiles of 32 keys through LDS):
//   S^T = K . Q^T with the KEY on the accumulator rows and the QUERY on the lane, so that (a) a query's row maximum /
//   sum is 16 in-register values + one exchange with lane ^ 32, and (b) the accumulator registers ARE the B operand of
//   the second product  O^T += V^T . P^T  (k-pair of step i = keys r0(i), r0(i) + 4: no LDS round trip for P).
// Backward, deterministic (no float atomics): one kernel owns key blocks (dK, dV: 4 products per tile), one owns query
//   blocks (dQ: 3 products per tile); both recompute P from the forward's log-sum-exp; the row constants -LSE and
//   -delta (delta = rowsum(dO . O)) are loaded into the accumulators before the MFMA chains.
#include "common.hpp"

namespace vqa {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTile = 32;               // keys (or queries) per MFMA tile
constexpr int kKs = 65;                 // LDS row stride for "row on the lane" reads  (bank = (row + col) % 64)

struct AttnDims {
  int B, H, Sq, Sk;
  long q_sb, q_ss, q_sh;                // element strides of (batch, sequence, head); the head dimension is dense
  long k_sb, k_ss, k_sh;
  long v_sb, v_ss, v_sh;
  long o_sb, o_ss, o_sh;
  long bias_sb, bias_sh, bias_sr;       // additive bias (batch, head, query row); key stride 1; batch / head stride may be 0
  float scale;
};

// Tuning build only (tools/attn_stamps.py): where a tile of the loop spends its cycles.  s_memtime stamps bracket the
// segments of a tile; each wave adds its segment sums to d_attn_stamps[kernel * 10 + segment] (+ its tile count in slot
// 9) after the loop.  No stamp exists in the shipped library; a stamped build's run time is not a measurement (the
// fences forbid overlaps the real kernel has) -- its SHARES are.
#ifdef VQA_TUNING
__device__ int d_attn_ablate = 0;        // tuning build only, timing ablations (results are wrong): 1 = no score / dS stores
#define VQA_ABLATE(bit) (d_attn_ablate & (bit))
__device__ unsigned long long* d_attn_stamps = nullptr;
#define VQA_STAMP_DECL unsigned long long stamp_t = 0, stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const bool stamp_on = d_attn_stamps != nullptr;
#define VQA_STAMP_START()                                                                              \
  do {                                                                                                 \
    if (stamp_on) {                                                                                    \
      __builtin_amdgcn_sched_barrier(0);                                                               \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_t)::"memory");                  \
      __builtin_amdgcn_sched_barrier(0);                                                               \
    }                                                                                                  \
  } while (0)
#define VQA_STAMP(seg)                                                                                 \
  do {                                                                                                 \
    if (stamp_on) {                                                                                    \
      unsigned long long stamp_now;                                                                    \
      __builtin_amdgcn_sched_barrier(0);                                                               \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_now)::"memory");                \
      __builtin_amdgcn_sched_barrier(0);                                                               \
      stamp_sum[seg] += stamp_now - stamp_t;                                                           \
      stamp_t = stamp_now;                                                                             \
    }                                                                                                  \
  } while (0)
#define VQA_STAMP_FLUSH(kernel, tiles)                                                                 \
  do {                                                                                                 \
    if (stamp_on && (threadIdx.x & (kWave - 1)) == 0) {                                                \
      for (int i = 0; i < 8; ++i) atomicAdd(d_attn_stamps + (kernel) * 10 + i, stamp_sum[i]);          \
      atomicAdd(d_attn_stamps + (kernel) * 10 + 9, static_cast<unsigned long long>(tiles));            \
    }                                                                                                  \
  } while (0)
#else
#define VQA_ABLATE(bit) false
#define VQA_STAMP_DECL
#define VQA_STAMP_START() do {} while (0)
#define VQA_STAMP(seg) do {} while (0)
#define VQA_STAMP_FLUSH(kernel, tiles) do {} while (0)
#endif

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// A 32 x 64 tile (rows of a (seq, 64) matrix) moves in two steps, for software pipelining: the global loads of tile
// t + 1 are issued into registers before tile t is computed and are written to the OTHER LDS buffer after it, so their
// latency hides behind ~4000 MFMA cycles and a tile costs one barrier.  256 threads, two 16-byte loads each.
struct TileRegs {
  f32x4 v[2];
};

template <int STRIDE>
__device__ __forceinline__ void store_tile(float* lds, const TileRegs& t, float mul) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int idx = threadIdx.x * 2 + j;
    const int row = idx >> 4, c4 = (idx & 15) * 4;
    float* dst = lds + row * STRIDE + c4;
    if (STRIDE % 4 == 0) {
      *reinterpret_cast<f32x4*>(dst) = t.v[j] * mul;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[e] = t.v[j][e] * mul;
    }
  }
}

// Which (batch, head, block of 128 rows) a workgroup of the 1-D grid works on.  The blocks of one (batch, head) read the
// same K / V (or Q / dO) tiles, and every batch of a head reads the same bias slab: workgroups are dealt round-robin
// over the 8 XCDs, each with its own L2 (blockIdx % 8 labels the workgroups that share one), so the blocks of a pair
// are given ids with equal id % 8 and pairs are ordered head-major -- an XCD then fetches a pair's tiles once instead
// of once per block, and works on one or two bias slabs at a time.  Speed only; any placement is correct.
struct BlockCoord {
  int b, head, blk;
};
__device__ __forceinline__ BlockCoord block_coord(int n_blk, int B, int H) {
  const int n = blockIdx.x, pairs = B * H;
  const int grouped = (pairs & ~7) * n_blk;                 // the pairs that fill whole groups of 8
  int p, blk;
  if (n < grouped) {
    const int j = n >> 3;
    p = (n & 7) + 8 * (j / n_blk);
    blk = j % n_blk;
  } else {
    const int m = n - grouped;
    p = (pairs & ~7) + m / n_blk;
    blk = m % n_blk;
  }
  return BlockCoord{p % B, p / B, blk};
}

// ------------------------------------------------------------------------------------------------ loop split
// Small batches (the reference's own batch 1: 12 heads x 5 blocks = 60 workgroups on 256 CUs) leave the chip idle and a
// workgroup's time is its SERIAL chain over ~19 tiles.  SPLIT kernels cut that loop into `n` parts: workgroup (block,
// part) runs tiles [part * T / n, (part + 1) * T / n) and writes a PARTIAL result into `ws`; a small combine kernel
// reduces the parts in a fixed order (bitwise reproducible).  Forward partials: the unnormalised O^T accumulators plus
// (running maximum * log2 e, row sum) per query -- the flash-decoding reduction; backward partials: plain sums.
// Layouts (rows = B * H * S of the reduced quantity):  forward  ws[(row * n + part) * 64 + dim], then
// ml[(row * n + part) * 2 + {0, 1}] from float offset rows * n * 64;  sums  ws[(part * rows + row) * 64 + dim].
struct SplitArgs {
  int n;
  float* ws;
};

__host__ __device__ __forceinline__ long split_fwd_floats(long rows_q, int n) { return rows_q * n * 66; }
__host__ __device__ __forceinline__ long split_sum_floats(long rows, int n) { return rows * n * 64; }

// (b, h, s) row of 64 floats per 16 lanes: out = sum over parts, in part order.  blockIdx.y selects one of up to two
// (workspace, output) pairs of the same row count -- dk and dv are reduced by ONE launch.
struct SumTarget {
  const float* ws;
  float* out;
  long sb, ss, sh;
};
__global__ __launch_bounds__(kBlock) void attn_sum_parts_kernel(SumTarget t0, SumTarget t1, int n, long rows, int H, int S) {
  const long row = static_cast<long>(blockIdx.x) * (kBlock / 16) + threadIdx.x / 16;
  if (row >= rows) return;
  const SumTarget t = blockIdx.y == 0 ? t0 : t1;
  const int c4 = (threadIdx.x & 15) * 4;
  f32x4 acc = *reinterpret_cast<const f32x4*>(t.ws + row * 64 + c4);
  for (int p = 1; p < n; ++p) acc += *reinterpret_cast<const f32x4*>(t.ws + (p * rows + row) * 64 + c4);
  const int s = static_cast<int>(row % S), head = static_cast<int>((row / S) % H);
  const long b = row / (static_cast<long>(S) * H);
  *reinterpret_cast<f32x4*>(t.out + b * t.sb + head * t.sh + static_cast<long>(s) * t.ss + c4) = acc;
}

// ------------------------------------------------------------------------------------------------ forward
// On a CDNA4 SIMD vector instructions do not overlap the matrix pipe (tools/mfma_probe.hip: the tile loop's MFMAs alone
// run at 65 cycles each, and every VALU instruction between them adds its own issue time, whatever the occupancy), so
// the loop is written to need few of them per 64 MFMAs:
//   * the bias tile is the INITIAL accumulator of the S^T chain (prefetched one tile ahead), not 16 adds after it;
//   * exp(s - m) is v_exp_f32(fma(s, log2 e, -m log2 e)): one fma per element, no separate subtract / scale;
//   * the running maximum is updated lazily, only for rows whose new maximum exceeds it by more than kLazyMax (the
//     accumulators then hold values up to e^kLazyMax times larger, harmless in fp32): in steady state the 32-register
//     rescale of O and its exp are skipped by a wave-uniform branch; the decision is per query row, so a row's result
//     does not depend on its neighbours in the wave;
//   * the two lane halves keep partial row sums and exchange them once, after the loop (v_permlane32_swap);
//   * keys beyond Sk are masked in the last tile only; tile rows beyond the sequence are clamped, not zero-filled;
//   * V sits in LDS with dimensions d and d + 32 interleaved, so one ds_read_b64 with an immediate offset feeds both
//     P.V products of a key and no address arithmetic is left in the loop.
// The two optional workspaces -- scores (B, H, ceil128(Sq), ceil128(Sk)) written by the forward and dS^T
// (B, H, ceil128(Sk), ceil128(Sq)) written by the dK / dV kernel -- are BLOCKED: a (batch, head) slab is a grid of
// 32 x 32 tiles, [tile row][tile column][1024 floats], and a tile holds the producing wave's accumulator image as it
// sits in registers: the 16-byte chunk g (of 4) of lane l at float 256 g + 4 l.  A store instruction of the producer
// then writes 1 KB of consecutive bytes (8 cache lines) instead of 16-byte pieces of 32 lines, and the consumer -- which
// stages a 32 x 128 block through LDS anyway -- fetches 16 KB of consecutive bytes and puts every chunk where the
// row-major tile image wants it (chunk g of lane (r, h) = row r, columns 8 g + 4 h .. + 3 of its tile).  With the
// row-major form, taking the stores out of the two kernels (results wrong, timing only) saved 12 % of the forward and
// 10 % of the backward at the bench shape: the request rate of the scattered pieces, not their bytes.
// Both grids are padded to whole 128-row / 128-column groups, so every block a consumer touches exists; blocks no
// producer wave wrote are only read into lanes whose results are discarded.
__host__ __device__ __forceinline__ long sc_rows(int Sq) { return (Sq + 127L) / 128 * 128; }
__host__ __device__ __forceinline__ long sc_pitch(int Sk) { return (Sk + 127L) / 128 * 128; }
__host__ __device__ __forceinline__ long ds_rows(int Sk) { return (Sk + 127L) / 128 * 128; }
__host__ __device__ __forceinline__ long ds_pitch(int Sq) { return (Sq + 127L) / 128 * 128; }
constexpr int kBlockFloats = kTile * kTile;      // one 32 x 32 tile of a blocked workspace

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kLazyMax = 8.0f;
constexpr int kVi = 66;                 // LDS row stride of the interleaved V tile: conflict-free stores and b64 reads
// A 32 x 128 fp32 tile staged through LDS (the saved scores in the dK / dV kernel, dS^T in the dQ kernel): fetched with four
// 16-byte loads per thread (thread t: row t / 32 + 8 j, columns 4 (t % 32) ..), a tile ahead; read back one value per
// lane.  Row stride 136 floats: the two lane halves read rows 4 apart, 4 * 136 = 32 mod 64 banks -> conflict-free.
constexpr int kDsStride = 136;

struct DsRegs {
  f32x4 v[4];
};


typedef float f32x2 __attribute__((ext_vector_type(2)));

// Value of the same lane index in the other half of the wave, combined: max / sum over lanes l and l ^ 32.
__device__ __forceinline__ float halves_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float halves_sum(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// 32 x 64 tiles of a (rows, 64) operand -> registers, one tile after another; thread t fetches 32 bytes of row t / 8.
// Rows beyond the sequence are clamped to the last valid one (their products are masked or multiplied by P = 0), which
// only the LAST tile can need: the cursor keeps the thread's clamped address in that tile, computed once, and a plain
// pointer that moves by one tile per load -- a tile costs one 64-bit add and a select per operand.  (The form it
// replaces clamped the row and rebuilt the address per tile: a min, three 32-bit multiplies and a 64-bit multiply-add
// per operand, quarter-rate instructions, and vector instructions are paid in matrix-pipe time on this chip -- see the
// forward's notes; tools/isa_loop_mix.py lists what a loop is left with.)
struct TileCursor {
  const float* p;       // this thread's 32 bytes of the tile the cursor is at
  const float* last;    // the same of the sequence's last tile, row clamped
  long step;            // floats per tile (uniform)
  int last_tile;
};
__device__ __forceinline__ TileCursor tile_cursor(const float* __restrict__ base, long ss, int n_rows, int tile) {
  const int lrow = threadIdx.x >> 3, c8 = (threadIdx.x & 7) * 8;
  TileCursor c;
  c.last_tile = (n_rows - 1) / kTile;
  int row = c.last_tile * kTile + lrow;
  row = row < n_rows ? row : n_rows - 1;
  c.last = base + static_cast<long>(row) * ss + c8;
  c.p = base + (static_cast<long>(tile) * kTile + lrow) * ss + c8;
  c.step = kTile * ss;
  return c;
}
// loads tile `tile` -- the one the cursor is at -- and moves the cursor to the next
__device__ __forceinline__ TileRegs load_tile_next(TileCursor& c, int tile) {
  const float* p = tile == c.last_tile ? c.last : c.p;
  TileRegs t;
  t.v[0] = *reinterpret_cast<const f32x4*>(p);
  t.v[1] = *reinterpret_cast<const f32x4*>(p + 4);
  c.p += c.step;
  return t;
}

// registers -> LDS, dimensions d and d + 32 of a row next to each other
__device__ __forceinline__ void store_tile_interleaved(float* lds, const TileRegs& t, float mul) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int idx = threadIdx.x * 2 + j;
    const int row = idx >> 4, c4 = (idx & 15) * 4;
    float* dst = lds + row * kVi + 2 * (c4 & 31) + (c4 >> 5);
#pragma unroll
    for (int e = 0; e < 4; ++e) dst[2 * e] = t.v[j][e] * mul;
  }
}

// Bias of this lane's query row for the 32 keys of a tile, in the S^T accumulator layout (register 4 g + e = key
// k0 + 8 g + 4 h + e).  Rows are readable up to ceil32(Sk) floats (the caller's contract), so no clamping.
__device__ __forceinline__ f32x16 load_bias_tile(const float* __restrict__ row, int k0) {
  f32x16 b;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(row + k0 + 8 * g);
#pragma unroll
    for (int e = 0; e < 4; ++e) b[4 * g + e] = t[e];
  }
  return b;
}

template <bool HAS_BIAS, bool STORE_S, bool SPLIT = false>
__global__ __launch_bounds__(kBlock, 2) void attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                             const float* __restrict__ v,
                                                             const float* __restrict__ bias, float* __restrict__ o,
                                                             float* __restrict__ lse, float* __restrict__ scores,
                                                             AttnDims d, const int* __restrict__ key_hole,
                                                             SplitArgs sp) {
  __shared__ float Kbuf[2][kTile * kKs];
  __shared__ __attribute__((aligned(16))) float Vbuf[2][kTile * kVi];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int r = lane & 31, h = lane >> 5;
  const int nsp = SPLIT ? sp.n : 1;
  const BlockCoord bc = block_coord(((d.Sq + 127) / 128) * nsp, d.B, d.H);
  const int b = bc.b, head = bc.head;
  const int blk = SPLIT ? bc.blk / nsp : bc.blk, part = SPLIT ? bc.blk % nsp : 0;
  const int q0 = blk * 128 + wave * kTile;
  const bool active = q0 < d.Sq;                           // a wave past the last query only helps staging the tiles
  const int qi = q0 + r;                                   // this lane's query
  const int ql = qi < d.Sq ? qi : d.Sq - 1;                // clamped for loads
  const float* qp = q + b * d.q_sb + head * d.q_sh + static_cast<long>(ql) * d.q_ss + 32 * h;
  const float* kb = k + b * d.k_sb + head * d.k_sh;
  const float* vb = v + b * d.v_sb + head * d.v_sh;
  const float* bp = HAS_BIAS ? bias + b * d.bias_sb + head * d.bias_sh + static_cast<long>(ql) * d.bias_sr + 4 * h
                             : nullptr;
  // scores: this wave's row of tiles (query tile q0 / 32) in the blocked slab, at this lane's chunk
  float* scp = STORE_S ? scores + (static_cast<long>(b) * d.H + head) * sc_rows(d.Sq) * sc_pitch(d.Sk) +
                             (q0 / kTile) * sc_pitch(d.Sk) * kTile + 4 * lane
                       : nullptr;
  float qf[32];                                            // Q[query][32 h + s] * scale: the B operand of S^T = K . Q^T
#pragma unroll
  for (int s4 = 0; s4 < 8; ++s4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(qp + 4 * s4);
#pragma unroll
    for (int e = 0; e < 4; ++e) qf[4 * s4 + e] = t[e] * d.scale;
  }
  f32x16 o0 = {0}, o1 = {0};                               // O^T[dim (+32)][query]
  float m = -INFINITY, mc = 0.0f, l = 0.0f;                // running max, the same times log2 e (0 while -inf), row sum
  const int all_tiles = (d.Sk + kTile - 1) / kTile;
  const int t_lo = SPLIT ? part * all_tiles / nsp : 0;       // this workgroup's key tiles [t_lo, n_tiles)
  const int n_tiles = SPLIT ? (part + 1) * all_tiles / nsp : all_tiles;
  // keys [hole_lo, hole_hi) of this batch element are masked (-inf) for every head and query: the padded text tokens of
  // a question shorter than the batch's text length.  Wave-uniform (SGPRs); with it the additive bias can stay ONE
  // (1, H, S, S) slab shared by the batch instead of a per-sample (B, H, S, S) copy streamed from HBM by every call.
  const int hole_lo = key_hole ? key_hole[2 * b] : 0, hole_hi = key_hole ? key_hole[2 * b + 1] : 0;
  f32x16 bcur = {0};
  if (HAS_BIAS && active) bcur = load_bias_tile(bp, t_lo * kTile);
  TileCursor ck = tile_cursor(kb, d.k_ss, d.Sk, t_lo), cv = tile_cursor(vb, d.v_ss, d.Sk, t_lo);
  {
    const TileRegs tk = load_tile_next(ck, t_lo), tv = load_tile_next(cv, t_lo);
    store_tile<kKs>(Kbuf[t_lo & 1], tk, 1.0f);
    store_tile_interleaved(Vbuf[t_lo & 1], tv, 1.0f);
  }
  __syncthreads();
  VQA_STAMP_DECL
  VQA_STAMP_START();
  for (int kt = t_lo; kt < n_tiles; ++kt) {
    const int k0 = kt * kTile;
    const float* Ks = Kbuf[kt & 1] + r * kKs + 32 * h;
    const float* Vs = Vbuf[kt & 1] + 4 * h * kVi + 2 * r;
    const bool more = kt + 1 < n_tiles;
    TileRegs tk, tv;
    if (more) {                                  // in flight while this tile is computed
      tk = load_tile_next(ck, kt + 1);
      tv = load_tile_next(cv, kt + 1);
    }
    f32x16 sraw;                                 // the tile's raw scores (live waves of a STORE_S kernel only)
    if (active) {
      f32x16 bnext = {0};
      if (HAS_BIAS && more) bnext = load_bias_tile(bp, k0 + kTile);
      __builtin_amdgcn_sched_barrier(0);         // keep every load ahead of the MFMA chain that hides its latency
      VQA_STAMP(0);                              // segment 0: the prefetches' address work and issue
      f32x16 st = bcur;                          // S^T = bias + K . (scale Q)^T, key on the accumulator row
      if (k0 < hole_hi && k0 + kTile > hole_lo) {   // a tile that overlaps the hole (at most two of ~19): -inf goes into
#pragma unroll                                      // the initial accumulator, so the saved scores carry it too
        for (int i = 0; i < 16; ++i) {
          const int key = k0 + acc_row(i, h);
          st[i] = (key >= hole_lo && key < hole_hi) ? -INFINITY : st[i];
        }
      }
#pragma unroll
      for (int s = 0; s < 32; ++s) st = mfma(Ks[s], qf[s], st);
      VQA_STAMP(1);                              // segment 1: the S chain's issue (its last MFMA still runs)
      if (STORE_S) sraw = st;                    // stored at the end of the tile (see there)
      if (k0 + kTile > d.Sk) {                   // last, partial tile: keys beyond Sk never win the softmax
#pragma unroll
        for (int i = 0; i < 16; ++i) st[i] = k0 + acc_row(i, h) < d.Sk ? st[i] : -INFINITY;
      }
      float mx = st[0];
#pragma unroll
      for (int i = 1; i < 16; ++i) mx = fmaxf(mx, st[i]);
      mx = halves_max(mx);
      const bool grow = mx > m + kLazyMax;       // m = -inf: any finite score starts the row
      if (__builtin_amdgcn_ballot_w64(grow) != 0) {
        const float m_new = grow ? mx : m;
        const float mc_new = m_new * kLog2e;     // m_new is finite wherever it changed
        const float alpha = m == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(mc - mc_new);   // exactly 1 if unchanged
        m = m_new;
        mc = grow ? mc_new : mc;
        l *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          o0[i] *= alpha;
          o1[i] *= alpha;
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        st[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[i], kLog2e, -mc));
        l += st[i];
      }
      VQA_STAMP(2);                              // segment 2: score store, maximum, exponentials (waits for the chain)
#pragma unroll
      for (int i = 0; i < 16; ++i) {             // O^T += V^T . P^T, k-pair of step i = keys r0(i), r0(i) + 4
        const f32x2 vv = *reinterpret_cast<const f32x2*>(Vs + ((i & 3) + 8 * (i >> 2)) * kVi);
        o0 = mfma(vv[0], st[i], o0);
        o1 = mfma(vv[1], st[i], o1);
      }
      VQA_STAMP(3);                              // segment 3: the P . V chains' issue
      bcur = bnext;
    }
    if (more) {                                  // the other buffer was last read one tile ago (barrier in between)
      store_tile<kKs>(Kbuf[(kt + 1) & 1], tk, 1.0f);
      store_tile_interleaved(Vbuf[(kt + 1) & 1], tv, 1.0f);
    }
    VQA_STAMP(4);                                // segment 4: wait for the prefetched tile, publish it in LDS
    // The scores the backward's dK / dV kernel starts from (instead of Q . K^T) leave HERE, after the prefetched tile
    // has been taken out of its registers: the compiler counts loads and stores in one counter and, with stores
    // pending, waits for ALL of it before the first use of a load -- stored right after the chain, every tile ended
    // waiting for its own score stores to be acknowledged (12 % of the kernel, tools/attn_ab.py ABLATE=1).  Now the
    // next wait that covers them is a whole tile away.
    if (STORE_S && active && !VQA_ABLATE(1)) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(scp + static_cast<long>(kt) * kBlockFloats + 256 * g) =
            f32x4{sraw[4 * g], sraw[4 * g + 1], sraw[4 * g + 2], sraw[4 * g + 3]};
    }
    VQA_STAMP(5);                                // segment 5: the score stores' issue
    __syncthreads();
    VQA_STAMP(6);                                // segment 6: the barrier
  }
  if (active) VQA_STAMP_FLUSH(0, n_tiles - t_lo);
  if (SPLIT) {                                   // partial: unnormalised accumulators + (max * log2 e, row sum)
    if (qi < d.Sq) {
      l = halves_sum(l);
      const long row = (static_cast<long>(b) * d.H + head) * d.Sq + qi;
      float* wo = sp.ws + (row * nsp + part) * 64;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dim = 8 * g + 4 * h;
        *reinterpret_cast<f32x4*>(wo + dim) = f32x4{o0[4 * g], o0[4 * g + 1], o0[4 * g + 2], o0[4 * g + 3]};
        *reinterpret_cast<f32x4*>(wo + 32 + dim) = f32x4{o1[4 * g], o1[4 * g + 1], o1[4 * g + 2], o1[4 * g + 3]};
      }
      if (h == 0) {
        float* ml = sp.ws + static_cast<long>(d.B) * d.H * d.Sq * nsp * 64 + (row * nsp + part) * 2;
        ml[0] = m == -INFINITY ? -INFINITY : mc;
        ml[1] = l;
      }
    }
    return;
  }
  if (qi < d.Sq) {
    l = halves_sum(l);
    const float inv = 1.0f / l;
    float* op = o + b * d.o_sb + head * d.o_sh + static_cast<long>(qi) * d.o_ss;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int dim = 8 * g + 4 * h;
      f32x4 a = {o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv};
      f32x4 c = {o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv};
      *reinterpret_cast<f32x4*>(op + dim) = a;
      *reinterpret_cast<f32x4*>(op + 32 + dim) = c;
    }
    if (h == 0) lse[(static_cast<long>(b) * d.H + head) * d.Sq + qi] = mc * kLn2 + __logf(l);
  }
}


// Forward parts -> o, lse: weights 2^(mc_part - max mc), the flash-decoding reduction, parts in index order.
__global__ __launch_bounds__(kBlock) void attn_fwd_combine_kernel(const float* __restrict__ ws, int n,
                                                                  float* __restrict__ o, float* __restrict__ lse,
                                                                  AttnDims d) {
  const long rows = static_cast<long>(d.B) * d.H * d.Sq;
  const long row = static_cast<long>(blockIdx.x) * (kBlock / 16) + threadIdx.x / 16;
  if (row >= rows) return;
  const int c4 = (threadIdx.x & 15) * 4;
  const float* ml = ws + rows * n * 64 + row * n * 2;
  float top = -INFINITY;
  for (int p = 0; p < n; ++p) top = fmaxf(top, ml[2 * p]);
  float l = 0.0f;
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int p = 0; p < n; ++p) {
    const float w = ml[2 * p] == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(ml[2 * p] - top);
    l += w * ml[2 * p + 1];
    acc += *reinterpret_cast<const f32x4*>(ws + (row * n + p) * 64 + c4) * w;
  }
  const float inv = 1.0f / l;
  const int qi = static_cast<int>(row % d.Sq), head = static_cast<int>((row / d.Sq) % d.H);
  const long b = row / (static_cast<long>(d.Sq) * d.H);
  *reinterpret_cast<f32x4*>(o + b * d.o_sb + head * d.o_sh + static_cast<long>(qi) * d.o_ss + c4) = acc * inv;
  if ((threadIdx.x & 15) == 0) lse[row] = top * kLn2 + __logf(l);
}

// ------------------------------------------------------------------------------------------------ backward: dQ
// One workgroup = 4 waves = 128 queries of one (batch, head); loop over key tiles.  Query on the lane:
//   S^T = bias + K . (scale Q)^T (key rows in the accumulator),  P^T = exp(S^T - LSE),  dP^T = V . dO^T,
//   dS^T = P^T o (dP^T - delta),  dQ^T += K^T . dS^T  (sum over the key = the accumulator's row index: registers feed
//   the MFMA).  Also computes delta[b, h, q] = sum_d dO . O for its queries and stores it for the dK / dV kernel.
// Same instruction diet as the forward: bias tile = initial accumulator, exp through one fma, packed subtract /
// multiply for dS, masks in the last tile only.  K and V tiles are interleaved in LDS (dimensions d, d + 32 adjacent):
// the row reads of the first two products use immediate offsets 2 s, the column reads of the third are one ds_read_b64.
template <bool HAS_BIAS>
__global__ __launch_bounds__(kBlock, 3) void attn_bwd_dq_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                const float* __restrict__ v,
                                                                const float* __restrict__ bias,
                                                                const float* __restrict__ o,
                                                                const float* __restrict__ go,
                                                                const float* __restrict__ lse,
                                                                float* __restrict__ delta, float* __restrict__ dq,
                                                                AttnDims d, long dq_sb, long dq_ss, long dq_sh,
                                                                long go_sb, long go_ss, long go_sh) {
  __shared__ __attribute__((aligned(16))) float Kbuf[2][kTile * kVi];
  __shared__ __attribute__((aligned(16))) float Vbuf[2][kTile * kVi];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int r = lane & 31, h = lane >> 5;
  const BlockCoord bc = block_coord((d.Sq + 127) / 128, d.B, d.H);
  const int b = bc.b, head = bc.head;
  const int qi = bc.blk * 128 + wave * kTile + r;
  const bool active = bc.blk * 128 + wave * kTile < d.Sq;
  const int ql = qi < d.Sq ? qi : d.Sq - 1;
  const float* qp = q + b * d.q_sb + head * d.q_sh + static_cast<long>(ql) * d.q_ss + 32 * h;
  const float* gp = go + b * go_sb + head * go_sh + static_cast<long>(ql) * go_ss + 32 * h;
  const float* op = o + b * d.o_sb + head * d.o_sh + static_cast<long>(ql) * d.o_ss + 32 * h;
  const float* kb = k + b * d.k_sb + head * d.k_sh;
  const float* vb = v + b * d.v_sb + head * d.v_sh;
  const float* bp = HAS_BIAS ? bias + b * d.bias_sb + head * d.bias_sh + static_cast<long>(ql) * d.bias_sr + 4 * h
                             : nullptr;
  const long row = (static_cast<long>(b) * d.H + head) * d.Sq + ql;
  float qf[32], gf[32];
  float dl = 0.0f;
#pragma unroll
  for (int s4 = 0; s4 < 8; ++s4) {
    const f32x4 tq = *reinterpret_cast<const f32x4*>(qp + 4 * s4);
    const f32x4 tg = *reinterpret_cast<const f32x4*>(gp + 4 * s4);
    const f32x4 to = *reinterpret_cast<const f32x4*>(op + 4 * s4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      qf[4 * s4 + e] = tq[e] * d.scale;
      gf[4 * s4 + e] = tg[e];
      dl += tg[e] * to[e];
    }
  }
  dl = halves_sum(dl);
  if (h == 0 && qi < d.Sq) delta[row] = dl;
  const float nlc = -lse[row] * kLog2e;                    // P = exp2(S log2 e - LSE log2 e)
  f32x16 dq0 = {0}, dq1 = {0};
  const int n_tiles = (d.Sk + kTile - 1) / kTile;
  TileCursor ck = tile_cursor(kb, d.k_ss, d.Sk, 0), cv = tile_cursor(vb, d.v_ss, d.Sk, 0);
  {
    const TileRegs tk = load_tile_next(ck, 0), tv = load_tile_next(cv, 0);
    store_tile_interleaved(Kbuf[0], tk, 1.0f);
    store_tile_interleaved(Vbuf[0], tv, 1.0f);
  }
  __syncthreads();
  for (int kt = 0; kt < n_tiles; ++kt) {
    const int k0 = kt * kTile;
    const float* Kr = Kbuf[kt & 1] + r * kVi + h;          // row read: K[key r][32 h + s] at Kr[2 s]
    const float* Vr = Vbuf[kt & 1] + r * kVi + h;
    const float* Kc = Kbuf[kt & 1] + 4 * h * kVi + 2 * r;  // column read: K[key][r], K[key][r + 32]
    const bool more = kt + 1 < n_tiles;
    TileRegs tk, tv;
    if (more) {
      tk = load_tile_next(ck, kt + 1);
      tv = load_tile_next(cv, kt + 1);
    }
    if (active) {
      f32x16 st = {0}, dp = {0};
      if (HAS_BIAS) st = load_bias_tile(bp, k0);   // the initial accumulator of the S^T chain ...
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 32; ++s) dp = mfma(Vr[2 * s], gf[s], dp);   // ... which starts 32 MFMAs after its loads
#pragma unroll
      for (int s = 0; s < 32; ++s) st = mfma(Kr[2 * s], qf[s], st);
      if (k0 + kTile > d.Sk) {
#pragma unroll
        for (int i = 0; i < 16; ++i) st[i] = k0 + acc_row(i, h) < d.Sk ? st[i] : -INFINITY;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i)                 // dS^T = P^T o (dP^T - delta)
        st[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[i], kLog2e, nlc)) * (dp[i] - dl);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const f32x2 kk = *reinterpret_cast<const f32x2*>(Kc + ((i & 3) + 8 * (i >> 2)) * kVi);
        dq0 = mfma(kk[0], st[i], dq0);
        dq1 = mfma(kk[1], st[i], dq1);
      }
    }
    if (more) {
      store_tile_interleaved(Kbuf[(kt + 1) & 1], tk, 1.0f);
      store_tile_interleaved(Vbuf[(kt + 1) & 1], tv, 1.0f);
    }
    __syncthreads();
  }
  if (qi < d.Sq) {
    float* dp_ = dq + b * dq_sb + head * dq_sh + static_cast<long>(qi) * dq_ss;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int dim = 8 * g + 4 * h;
      f32x4 a = {dq0[4 * g] * d.scale, dq0[4 * g + 1] * d.scale, dq0[4 * g + 2] * d.scale, dq0[4 * g + 3] * d.scale};
      f32x4 c = {dq1[4 * g] * d.scale, dq1[4 * g + 1] * d.scale, dq1[4 * g + 2] * d.scale, dq1[4 * g + 3] * d.scale};
      *reinterpret_cast<f32x4*>(dp_ + dim) = a;
      *reinterpret_cast<f32x4*>(dp_ + 32 + dim) = c;
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward via dS
// With a workspace for dS^T (B, H, ceil128(Sk), ceil32(Sq)) the backward needs 5 products instead of 7: the dK / dV
// kernel stores the dS tile it computes anyway (transposed: its lanes are keys, so a lane writes 16-byte runs of
// queries into its own row), and dQ^T += K^T . dS^T becomes a kernel of one product per tile whose B operand is read
// from that workspace with fully coalesced dword loads (lane = query).  delta = rowsum(dO o O), which the dQ kernel of
// the recompute path produces on the way, comes from a small streaming pre-pass here.
// delta[b, h, q] = sum_d dO[b, q, h, d] * O[b, q, h, d]: 16 lanes per row, one 16-byte load of each operand per lane
__global__ __launch_bounds__(kBlock) void attn_delta_kernel(const float* __restrict__ o, const float* __restrict__ go,
                                                            float* __restrict__ delta, AttnDims d, long go_sb,
                                                            long go_ss, long go_sh) {
  const long row = static_cast<long>(blockIdx.x) * (kBlock / 16) + threadIdx.x / 16;      // (b, h, q) flattened
  const long n_rows = static_cast<long>(d.B) * d.H * d.Sq;
  const bool ok = row < n_rows;
  const long rr = ok ? row : n_rows - 1;
  const int qq = static_cast<int>(rr % d.Sq);
  const int head = static_cast<int>((rr / d.Sq) % d.H);
  const int b = static_cast<int>(rr / (static_cast<long>(d.Sq) * d.H));
  const int c4 = (threadIdx.x & 15) * 4;
  const f32x4 x = *reinterpret_cast<const f32x4*>(o + b * d.o_sb + head * d.o_sh + static_cast<long>(qq) * d.o_ss + c4);
  const f32x4 y = *reinterpret_cast<const f32x4*>(go + b * go_sb + head * go_sh + static_cast<long>(qq) * go_ss + c4);
  float t = x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
  t += __shfl_xor(t, 1, kWave);
  t += __shfl_xor(t, 2, kWave);
  t += __shfl_xor(t, 4, kWave);
  t += __shfl_xor(t, 8, kWave);
  if (ok && (threadIdx.x & 15) == 0) delta[row] = t;
}

// dQ^T += K^T . dS^T with the dS^T tile STAGED THROUGH LDS (round 4).  The direct form below feeds the MFMA's B operand
// with 16 dword loads per lane and tile (lane = query): 4-byte accesses reach roughly half the rate of 16-byte ones on
// this chip (MI355X_MICROARCH.md), and the kernel -- 1.3 GB of dS^T for 34 GFLOP -- sat at 3.1 TB/s, neither on the HBM
// nor on the matrix roof.  Here the workgroup fetches a tile's 32 key rows x 128 queries (16 KB, rows of 512 contiguous
// bytes) with four 16-byte loads per thread, prefetched a tile ahead into registers like the K tile, publishes it in LDS
// (row stride 136 floats: the two lane halves read rows 4 apart, 4 * 136 = 32 mod 64 banks -> conflict-free), and every
// lane reads its 16 B-operand values back with ds_read_b32.  Same products in the same order: results are bitwise those
// of the direct form.

template <bool SPLIT = false>
__global__ __launch_bounds__(kBlock, 2) void attn_bwd_dq_staged_kernel(const float* __restrict__ k,
                                                                       const float* __restrict__ ds,
                                                                       float* __restrict__ dq, AttnDims d, long dq_sb,
                                                                       long dq_ss, long dq_sh, SplitArgs sp) {
  __shared__ __attribute__((aligned(16))) float Kbuf[2][kTile * kVi];
  __shared__ __attribute__((aligned(16))) float Dbuf[2][kTile * kDsStride];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int r = lane & 31, h = lane >> 5;
  const int nsp = SPLIT ? sp.n : 1;
  const BlockCoord bc = block_coord(((d.Sq + 127) / 128) * nsp, d.B, d.H);
  const int b = bc.b, head = bc.head;
  const int blk = SPLIT ? bc.blk / nsp : bc.blk, part = SPLIT ? bc.blk % nsp : 0;
  const int Q0 = blk * 128;
  const int q0 = Q0 + wave * kTile;
  const int qi = q0 + r;
  const bool active = q0 < d.Sq;
  const float* kb = k + b * d.k_sb + head * d.k_sh;
  // staging role of this thread: 16-byte chunk t of each of the four 32 x 32 tiles (queries Q0 + 32 j ..) of a key tile,
  // consecutive in the blocked dS^T slab; chunk t = 64 g + 32 hh + rr holds dS^T[key rr][queries 8 g + 4 hh .. + 3]
  const long tile_row = ds_pitch(d.Sq) * kTile;              // floats per key tile of the slab
  const float* dsrc = ds + (static_cast<long>(b) * d.H + head) * ds_rows(d.Sk) * ds_pitch(d.Sq) +
                      static_cast<long>(Q0 / kTile) * kBlockFloats + 4 * threadIdx.x;
  const int st_g = threadIdx.x >> 6, st_hh = (threadIdx.x >> 5) & 1, st_rr = threadIdx.x & 31;
  const int st_off = st_rr * kDsStride + 8 * st_g + 4 * st_hh;
  auto load_ds = [&](int kt) {
    DsRegs t;
#pragma unroll
    for (int j = 0; j < 4; ++j)                  // read exactly once
      t.v[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dsrc + kt * tile_row + j * kBlockFloats));
    return t;
  };
  auto store_ds = [&](float* buf, const DsRegs& t) {
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(buf + st_off + kTile * j) = t.v[j];
  };
  f32x16 dq0 = {0}, dq1 = {0};
  const int all_tiles = (d.Sk + kTile - 1) / kTile;
  const int t_lo = SPLIT ? part * all_tiles / nsp : 0;
  const int n_tiles = SPLIT ? (part + 1) * all_tiles / nsp : all_tiles;
  TileCursor ck = tile_cursor(kb, d.k_ss, d.Sk, t_lo);
  {
    const TileRegs tk = load_tile_next(ck, t_lo);
    const DsRegs td = load_ds(t_lo);
    store_tile_interleaved(Kbuf[t_lo & 1], tk, 1.0f);
    store_ds(Dbuf[t_lo & 1], td);
  }
  __syncthreads();
  for (int kt = t_lo; kt < n_tiles; ++kt) {
    const bool more = kt + 1 < n_tiles;
    const float* Kc = Kbuf[kt & 1] + 4 * h * kVi + 2 * r;
    const float* Dc = Dbuf[kt & 1] + 4 * h * kDsStride + wave * kTile + r;
    TileRegs tk;
    DsRegs td;
    if (more) {                                  // in flight while this tile's 32 MFMAs run
      tk = load_tile_next(ck, kt + 1);
      td = load_ds(kt + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (active) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2);
        const f32x2 kk = *reinterpret_cast<const f32x2*>(Kc + row * kVi);
        const float dsv = Dc[row * kDsStride];
        dq0 = mfma(kk[0], dsv, dq0);
        dq1 = mfma(kk[1], dsv, dq1);
      }
    }
    if (more) {                                  // the other buffers were last read one tile ago, a barrier in between
      store_tile_interleaved(Kbuf[(kt + 1) & 1], tk, 1.0f);
      store_ds(Dbuf[(kt + 1) & 1], td);
    }
    __syncthreads();
  }
  if (qi < d.Sq) {
    const long rows = static_cast<long>(d.B) * d.H * d.Sq;
    float* dp_ = SPLIT ? sp.ws + (part * rows + (static_cast<long>(b) * d.H + head) * d.Sq + qi) * 64
                       : dq + b * dq_sb + head * dq_sh + static_cast<long>(qi) * dq_ss;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int dim = 8 * g + 4 * h;
      f32x4 a = {dq0[4 * g] * d.scale, dq0[4 * g + 1] * d.scale, dq0[4 * g + 2] * d.scale, dq0[4 * g + 3] * d.scale};
      f32x4 c = {dq1[4 * g] * d.scale, dq1[4 * g + 1] * d.scale, dq1[4 * g + 2] * d.scale, dq1[4 * g + 3] * d.scale};
      *reinterpret_cast<f32x4*>(dp_ + dim) = a;
      *reinterpret_cast<f32x4*>(dp_ + 32 + dim) = c;
    }
  }
}


// ------------------------------------------------------------------------------------------------ backward: dK, dV
// One workgroup = 4 waves = 128 keys of one (batch, head); loop over query tiles of 32.  Key on the lane:
//   S = bias + (scale Q) . K^T (query rows in the accumulator),  P = exp(S - LSE),  dP = dO . V^T,
//   dS = P o (dP - delta),  dV^T += dO^T . P,  dK^T += (scale Q)^T . dS  (both sum over the query = the accumulators'
//   row index).  The per-query constants -LSE log2 e and -delta ride in LDS next to the tiles, four consecutive
//   accumulator rows per 16-byte read; -delta is the INITIAL accumulator of the dP chain (dS = P o dP is then one
//   multiply per element); the bias of this lane's key for the tile's query rows is 16 dword loads from a uniform row
//   base + a per-lane 32-bit offset, prefetched one tile ahead into the initial accumulator of the S chain.
template <bool HAS_BIAS, bool STORE_DS, bool FROM_SCORES, bool SPLIT = false>
__global__ __launch_bounds__(kBlock, 2) void attn_bwd_dkv_kernel(const float* __restrict__ q,
                                                                 const float* __restrict__ k,
                                                                 const float* __restrict__ v,
                                                                 const float* __restrict__ bias,
                                                                 const float* __restrict__ go,
                                                                 const float* __restrict__ lse,
                                                                 const float* __restrict__ delta,
                                                                 float* __restrict__ dk, float* __restrict__ dv,
                                                                 AttnDims d, long dk_sb, long dk_ss, long dk_sh,
                                                                 long dv_sb, long dv_ss, long dv_sh, long go_sb,
                                                                 long go_ss, long go_sh, float* __restrict__ ds,
                                                                 const float* __restrict__ scores, SplitArgs sp) {
  __shared__ __attribute__((aligned(16))) float Qbuf[2][kTile * kVi];
  __shared__ __attribute__((aligned(16))) float Gbuf[2][kTile * kVi];
  __shared__ __attribute__((aligned(16))) float Lbuf[2][kTile], Dbuf[2][kTile];
  // FROM_SCORES: the saved-score tile (32 queries x this block's 128 keys) goes through LDS, fetched a tile ahead with
  // 16-byte loads -- round 3 read it with 16 dword loads per lane issued right before the dP chain, whose 32 MFMAs
  // (~0.85 us) do not cover an HBM access under load
  __shared__ __attribute__((aligned(16))) float Sbuf[FROM_SCORES ? 2 : 1][FROM_SCORES ? kTile * kDsStride : 4];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int r = lane & 31, h = lane >> 5;
  const int nsp = SPLIT ? sp.n : 1;
  const BlockCoord bc0 = block_coord(((d.Sk + 127) / 128) * nsp, d.B, d.H);
  const BlockCoord bc{bc0.b, bc0.head, SPLIT ? bc0.blk / nsp : bc0.blk};
  const int part = SPLIT ? bc0.blk % nsp : 0;
  const int b = bc.b, head = bc.head;
  const int ki = bc.blk * 128 + wave * kTile + r;                     // this lane's key
  const bool active = bc.blk * 128 + wave * kTile < d.Sk;
  const bool all_live = bc.blk * 128 + wave * kTile + kTile <= d.Sk;    // wave-uniform
  const int kl = ki < d.Sk ? ki : d.Sk - 1;
  const float* kp = k + b * d.k_sb + head * d.k_sh + static_cast<long>(kl) * d.k_ss + 32 * h;
  const float* vp = v + b * d.v_sb + head * d.v_sh + static_cast<long>(kl) * d.v_ss + 32 * h;
  const float* qb = q + b * d.q_sb + head * d.q_sh;
  const float* gb = go + b * go_sb + head * go_sh;
  // The S chain's initial accumulator is the bias tile, a (query row, key) matrix of one (batch, head) -- unless the
  // forward saved its scores (bias included): then the scores replace the chain (they come through LDS, see below).
  constexpr bool kInit = HAS_BIAS && !FROM_SCORES;
  const long init_sr = d.bias_sr;
  const float* bslab = kInit ? bias + b * d.bias_sb + head * d.bias_sh : nullptr;      // uniform
  const unsigned bvoff = kInit ? static_cast<unsigned>(kl + 4 * h * static_cast<int>(init_sr)) : 0u;
  const long rows = (static_cast<long>(b) * d.H + head) * d.Sq;
  // dS^T: this wave's row of tiles (key tile 4 blk + wave) in the blocked slab, at this lane's chunk
  float* dsp = STORE_DS ? ds + (static_cast<long>(b) * d.H + head) * ds_rows(d.Sk) * ds_pitch(d.Sq) +
                              (4 * bc.blk + wave) * ds_pitch(d.Sq) * kTile + 4 * lane
                        : nullptr;
  float kf[32], vf[32];
#pragma unroll
  for (int s4 = 0; s4 < 8; ++s4) {
    const f32x4 tk = *reinterpret_cast<const f32x4*>(kp + 4 * s4);
    const f32x4 tv = *reinterpret_cast<const f32x4*>(vp + 4 * s4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      kf[4 * s4 + e] = tk[e];
      vf[4 * s4 + e] = tv[e];
    }
  }
  f32x16 dk0 = {0}, dk1 = {0}, dv0 = {0}, dv1 = {0};
  const int all_tiles = (d.Sq + kTile - 1) / kTile;
  const int t_lo = SPLIT ? part * all_tiles / nsp : 0;       // this workgroup's query tiles [t_lo, n_tiles)
  const int n_tiles = SPLIT ? (part + 1) * all_tiles / nsp : all_tiles;
  // per-query row constants of a tile: wave 0 fetches (lse, delta) of query q0 + lane % 32 a tile ahead, RAW and from a
  // clamped row -- the values are first touched when they are published in LDS as (-lse log2 e, -delta), a tile later,
  // where queries beyond Sq get (-inf, 0): P = exp2(-inf) = 0.  No branch on the lane and no default value around the
  // loads: scaling at the load made wave 0 wait for them -- and, vmcnt being in order, for the Q / dO / score tiles issued
  // before them -- at the top of every tile, and the other three waves for wave 0 at the barrier (tools/attn_stamps.py).
  const bool rc_wave = __builtin_amdgcn_readfirstlane(wave) == 0;
  auto row_consts = [&](int q0, float& rl, float& rd) {
    int qq = q0 + r;
    qq = qq < d.Sq ? qq : d.Sq - 1;
    rl = lse[rows + qq];
    rd = delta[rows + qq];
  };
  auto publish_consts = [&](int bi, int q0, float rl, float rd) {
    if (threadIdx.x < kTile) {
      const bool ok = q0 + static_cast<int>(threadIdx.x) < d.Sq;
      Lbuf[bi][threadIdx.x] = ok ? -rl * kLog2e : -INFINITY;
      Dbuf[bi][threadIdx.x] = ok ? -rd : 0.0f;
    }
  };
  // bias[query q0 + acc_row(i, h)][this key]: a full tile reads from uniform row bases; the last, partial query tile
  // clamps the row per lane (its P is 0 whatever is read)
  auto load_bias = [&](int q0) {
    f32x16 t;
    if (q0 + kTile <= d.Sq) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        t[i] = (bslab + static_cast<long>(q0 + (i & 3) + 8 * (i >> 2)) * init_sr)[bvoff];
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int qq = q0 + acc_row(i, h);
        t[i] = bslab[static_cast<long>(qq < d.Sq ? qq : d.Sq - 1) * init_sr + kl];
      }
    }
    return t;
  };
  // staging role of this thread for the score tile: 16-byte chunk t of each of the four 32 x 32 tiles (keys K0 + 32 j ..)
  // of a query tile, consecutive in the blocked slab; chunk t = 64 g + 32 hh + rr holds S[query rr][keys 8 g + 4 hh .. + 3].
  // Rows beyond Sq: the forward that saved the scores wrote the whole tile of every live wave -- finite scores of the
  // clamped last query (or -inf) -- and their P is exp2(s log2 e - inf) = 0.
  const long sc_tile_row = sc_pitch(d.Sk) * kTile;           // floats per query tile of the slab
  const float* sptr = FROM_SCORES ? scores + (static_cast<long>(b) * d.H + head) * sc_rows(d.Sq) * sc_pitch(d.Sk) +
                                        t_lo * sc_tile_row + static_cast<long>(4 * bc.blk) * kBlockFloats + 4 * threadIdx.x
                                  : nullptr;
  const int st_off = (threadIdx.x & 31) * kDsStride + 8 * (threadIdx.x >> 6) + 4 * ((threadIdx.x >> 5) & 1);
  auto load_scores_next = [&]() {
    DsRegs t;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      t.v[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(sptr + j * kBlockFloats));
    sptr += sc_tile_row;
    return t;
  };
  auto store_scores = [&](float* buf, const DsRegs& t) {
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(buf + st_off + kTile * j) = t.v[j];
  };
  f32x16 bcur = {0};
  if (kInit && active) bcur = load_bias(t_lo * kTile);
  // Q in LDS: scaled where the S chain reads it; with saved scores it only feeds dK^T += Q^T . dS, and the scale is
  // applied to the 32 dK accumulators once, after the loop, instead of to 8 tile values per thread and tile
  const float q_mul = FROM_SCORES ? 1.0f : d.scale, dk_mul = FROM_SCORES ? d.scale : 1.0f;
  TileCursor cq = tile_cursor(qb, d.q_ss, d.Sq, t_lo), cg = tile_cursor(gb, go_ss, d.Sq, t_lo);
  float rl = 0.0f, rd = 0.0f;                      // wave 0's prefetched row constants
  {
    const int qs = t_lo * kTile, bi = t_lo & 1;
    const TileRegs tq = load_tile_next(cq, t_lo), tg = load_tile_next(cg, t_lo);
    if (rc_wave) row_consts(qs, rl, rd);
    if (FROM_SCORES) store_scores(Sbuf[FROM_SCORES ? bi : 0], load_scores_next());
    store_tile_interleaved(Qbuf[bi], tq, q_mul);
    store_tile_interleaved(Gbuf[bi], tg, 1.0f);
    publish_consts(bi, qs, rl, rd);
  }
  __syncthreads();
  VQA_STAMP_DECL
  VQA_STAMP_START();
  for (int qt = t_lo; qt < n_tiles; ++qt) {
    const int q0 = qt * kTile;
    const float* Qr = Qbuf[qt & 1] + r * kVi + h;          // row read: Q[query r][32 h + s] at Qr[2 s]
    const float* Gr = Gbuf[qt & 1] + r * kVi + h;
    const float* Qc = Qbuf[qt & 1] + 4 * h * kVi + 2 * r;  // column read: Q[query][r], Q[query][r + 32]
    const float* Gc = Gbuf[qt & 1] + 4 * h * kVi + 2 * r;
    const float* Ls = Lbuf[qt & 1] + 4 * h;
    const float* Ds = Dbuf[qt & 1] + 4 * h;
    const bool more = qt + 1 < n_tiles;
    TileRegs tq, tg;
    DsRegs ts;
    if (more) {
      tq = load_tile_next(cq, qt + 1);
      tg = load_tile_next(cg, qt + 1);
      if (FROM_SCORES) ts = load_scores_next();
      if (rc_wave) row_consts(q0 + kTile, rl, rd);
    }
    f32x16 dp;                                     // dP, then dS (live waves only)
    if (active) {
      f32x16 bnext = {0};
      if (kInit && more) bnext = load_bias(q0 + kTile);
      __builtin_amdgcn_sched_barrier(0);
      VQA_STAMP(0);                                // segment 0: the prefetches' address work and issue
      // dP's chain starts from -delta of its query rows (the accumulator's rows), so dS = P o dP needs no subtraction
      f32x16 st = bcur;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 nd = *reinterpret_cast<const f32x4*>(Ds + 8 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) dp[4 * g + e] = nd[e];
      }
      if (FROM_SCORES) {                           // P first: its LDS reads and exponentials need no matrix result
        const float* Sc = Sbuf[qt & 1] + 4 * h * kDsStride + wave * kTile + r;   // S[query rowi][this lane's key]
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 nl = *reinterpret_cast<const f32x4*>(Ls + 8 * g);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int i = 4 * g + e;
            st[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(Sc[((i & 3) + 8 * (i >> 2)) * kDsStride], kLog2e, nl[e]));
          }
        }
        VQA_STAMP(1);                              // segment 1: LDS reads of -delta, S, -LSE; 16 exponentials
#pragma unroll
        for (int s = 0; s < 32; ++s) dp = mfma(Gr[2 * s], vf[s], dp);
        VQA_STAMP(2);                              // segment 2: the dP chain's issue
      } else {
#pragma unroll
        for (int s = 0; s < 32; ++s) {
          st = mfma(Qr[2 * s], kf[s], st);
          dp = mfma(Gr[2 * s], vf[s], dp);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 nl = *reinterpret_cast<const f32x4*>(Ls + 8 * g);
#pragma unroll
          for (int e = 0; e < 4; ++e) st[4 * g + e] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[4 * g + e], kLog2e, nl[e]));   // P
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) dp[i] *= st[i];                                     // dS
      VQA_STAMP(3);                                // segment 3: dS (waits for the chain's last MFMA)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int off = ((i & 3) + 8 * (i >> 2)) * kVi;
        const f32x2 gg = *reinterpret_cast<const f32x2*>(Gc + off);
        const f32x2 qq = *reinterpret_cast<const f32x2*>(Qc + off);
        dv0 = mfma(gg[0], st[i], dv0);
        dv1 = mfma(gg[1], st[i], dv1);
        dk0 = mfma(qq[0], dp[i], dk0);
        dk1 = mfma(qq[1], dp[i], dk1);
      }
      VQA_STAMP(4);                                // segment 4: the dV / dK chains' issue
      if (!FROM_SCORES) bcur = bnext;
    }
    if (more) {
      store_tile_interleaved(Qbuf[(qt + 1) & 1], tq, q_mul);
      store_tile_interleaved(Gbuf[(qt + 1) & 1], tg, 1.0f);
      if (FROM_SCORES) store_scores(Sbuf[(qt + 1) & 1], ts);
      asm volatile("" : "+v"(rl), "+v"(rd));      // the raw values are first needed HERE (the compiler would scale at the load)
      publish_consts((qt + 1) & 1, q0 + kTile, rl, rd);
    }
    VQA_STAMP(5);                                  // segment 5: wait for the prefetched tiles, publish them in LDS
    // dS^T leaves here, after the prefetched tiles have been taken out of their registers -- see the forward's score
    // stores: with stores pending the compiler's wait for a load is a wait for everything
    if (STORE_DS && active && !VQA_ABLATE(1)) {    // four consecutive queries per 16-byte store
      if (all_live) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(dsp + static_cast<long>(qt) * kBlockFloats + 256 * g) =
              f32x4{dp[4 * g], dp[4 * g + 1], dp[4 * g + 2], dp[4 * g + 3]};
      } else {                                     // the wave that straddles Sk: zeros in the rows of keys beyond it
        const bool live = ki < d.Sk;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(dsp + static_cast<long>(qt) * kBlockFloats + 256 * g) =
              live ? f32x4{dp[4 * g], dp[4 * g + 1], dp[4 * g + 2], dp[4 * g + 3]} : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      }
    }
    VQA_STAMP(6);                                  // segment 6: the dS^T stores' issue
    __syncthreads();
    VQA_STAMP(7);                                  // segment 7: the barrier
  }
  if (active) VQA_STAMP_FLUSH(1, n_tiles - t_lo);
  if (ki < d.Sk) {
    const long rows = static_cast<long>(d.B) * d.H * d.Sk;
    const long row = (static_cast<long>(b) * d.H + head) * d.Sk + ki;
    float* pk = SPLIT ? sp.ws + (part * rows + row) * 64 : dk + b * dk_sb + head * dk_sh + static_cast<long>(ki) * dk_ss;
    float* pv = SPLIT ? sp.ws + ((nsp + part) * rows + row) * 64
                      : dv + b * dv_sb + head * dv_sh + static_cast<long>(ki) * dv_ss;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int dim = 8 * g + 4 * h;
      *reinterpret_cast<f32x4*>(pk + dim) =
          f32x4{dk0[4 * g] * dk_mul, dk0[4 * g + 1] * dk_mul, dk0[4 * g + 2] * dk_mul, dk0[4 * g + 3] * dk_mul};
      *reinterpret_cast<f32x4*>(pk + 32 + dim) =
          f32x4{dk1[4 * g] * dk_mul, dk1[4 * g + 1] * dk_mul, dk1[4 * g + 2] * dk_mul, dk1[4 * g + 3] * dk_mul};
      *reinterpret_cast<f32x4*>(pv + dim) = f32x4{dv0[4 * g], dv0[4 * g + 1], dv0[4 * g + 2], dv0[4 * g + 3]};
      *reinterpret_cast<f32x4*>(pv + 32 + dim) = f32x4{dv1[4 * g], dv1[4 * g + 1], dv1[4 * g + 2], dv1[4 * g + 3]};
    }
  }
}

}  // namespace vqa

using namespace vqa;

extern "C" {

#ifdef VQA_TUNING
int vqa_attn_set_option(int value) {     // vqa_set_option(9, value): round 3's A/B of the dQ kernel, gone with the blocked dS^T
  return VQA_ERR_SHAPE;
}
int vqa_attn_set_ablation(int bits) {
  return hipMemcpyToSymbol(HIP_SYMBOL(vqa::d_attn_ablate), &bits, sizeof(bits)) == hipSuccess ? VQA_OK : VQA_ERR_NULL;
}
// 20 zeroed 64-bit words on the device (forward: 0..9, dK / dV kernel: 10..19), or NULL to switch the stamps off
int vqa_attn_set_stamps(unsigned long long* buffer) {
  return hipMemcpyToSymbol(HIP_SYMBOL(vqa::d_attn_stamps), &buffer, sizeof(buffer)) == hipSuccess ? VQA_OK : VQA_ERR_NULL;
}
#endif

static int check_attn(const float* q, const float* k, const float* v, const AttnDims& d) {
  if (!q || !k || !v) return VQA_ERR_NULL;
  if (d.B < 0 || d.H <= 0 || d.Sq <= 0 || d.Sk <= 0 || d.H > 65535 || d.B > 65535) return VQA_ERR_SHAPE;
  const long blocks = ((d.Sq > d.Sk ? d.Sq : d.Sk) + 127L) / 128;
  if (blocks * d.H * d.B > 2147483647L) return VQA_ERR_SHAPE;      // 1-D grid of (block, head, batch)
  if ((d.q_sb | d.q_ss | d.q_sh | d.k_sb | d.k_ss | d.k_sh | d.v_sb | d.v_ss | d.v_sh | d.o_sb | d.o_ss | d.o_sh) & 3)
    return VQA_ERR_SHAPE;
  if (!aligned16(q) || !aligned16(k) || !aligned16(v)) return VQA_ERR_ALIGN;
  return VQA_OK;
}

/* strides: 12 longs = {q_sb, q_ss, q_sh, k_sb, k_ss, k_sh, v_sb, v_ss, v_sh, o_sb, o_ss, o_sh}; bias_strides: 3 longs */
long vqa_attn_scores_floats(int B, int H, int Sq, int Sk) {
  if (B < 0 || H <= 0 || Sq <= 0 || Sk <= 0) return 0;
  return static_cast<long>(B) * H * sc_rows(Sq) * sc_pitch(Sk);
}

long vqa_attn_split_ws_floats(int B, int H, int Sq, int Sk, int nsplit) {
  if (B < 0 || H <= 0 || Sq <= 0 || Sk <= 0 || nsplit < 1) return 0;
  if (nsplit == 1) return 0;
  const long rq = static_cast<long>(B) * H * Sq, rk = static_cast<long>(B) * H * Sk;
  const long fwd = split_fwd_floats(rq, nsplit), dkv = 2 * split_sum_floats(rk, nsplit), dq = split_sum_floats(rq, nsplit);
  return fwd > dkv ? (fwd > dq ? fwd : dq) : (dkv > dq ? dkv : dq);
}

static int check_split(int nsplit, const float* split_ws, int tiles, long wgs) {
  if (nsplit < 1) return VQA_ERR_SHAPE;
  if (nsplit == 1) return VQA_OK;
  if (!split_ws) return VQA_ERR_NULL;
  if (!aligned16(split_ws)) return VQA_ERR_ALIGN;
  if (nsplit > tiles || wgs * nsplit > 2147483647L) return VQA_ERR_SHAPE;      // every part owns at least one tile
  return VQA_OK;
}

int vqa_attn_fwd(const float* q, const float* k, const float* v, const float* bias, float* o, float* lse, float* scores,
                 int B, int H, int Sq, int Sk, const long* strides, const long* bias_strides, float scale,
                 const int* key_hole, int nsplit, float* split_ws, vqa_stream_t stream) {
  clear_stale_error();
  if (!strides || !o || !lse || (bias && !bias_strides)) return VQA_ERR_NULL;
  AttnDims d{B, H, Sq, Sk, strides[0], strides[1], strides[2], strides[3], strides[4], strides[5], strides[6],
             strides[7], strides[8], strides[9], strides[10], strides[11], 0, 0, 0, scale};
  if (bias) {
    d.bias_sb = bias_strides[0];
    d.bias_sh = bias_strides[1];
    d.bias_sr = bias_strides[2];
    if (((d.bias_sb | d.bias_sh | d.bias_sr) & 3) || !aligned16(bias)) return VQA_ERR_ALIGN;
  }
  int rc = check_attn(q, k, v, d);
  if (rc != VQA_OK) return rc;
  if (!aligned16(o) || !aligned16(scores)) return VQA_ERR_ALIGN;
  const long wgs = ((Sq + 127L) / 128) * H * B;
  rc = check_split(nsplit, split_ws, (Sk + kTile - 1) / kTile, wgs);
  if (rc != VQA_OK) return rc;
  if (B == 0) return VQA_OK;
  const dim3 grid(static_cast<unsigned>(wgs * nsplit));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const SplitArgs sp{nsplit, split_ws};
  if (nsplit > 1) {
    if (bias && scores) attn_fwd_kernel<true, true, true><<<grid, kBlock, 0, st>>>(q, k, v, bias, o, lse, scores, d, key_hole, sp);
    else if (bias) attn_fwd_kernel<true, false, true><<<grid, kBlock, 0, st>>>(q, k, v, bias, o, lse, scores, d, key_hole, sp);
    else if (scores) attn_fwd_kernel<false, true, true><<<grid, kBlock, 0, st>>>(q, k, v, bias, o, lse, scores, d, key_hole, sp);
    else attn_fwd_kernel<false, false, true><<<grid, kBlock, 0, st>>>(q, k, v, bias, o, lse, scores, d, key_hole, sp);
    const long rows = static_cast<long>(B) * H * Sq;
    attn_fwd_combine_kernel<<<dim3(static_cast<unsigned>((rows + kBlock / 16 - 1) / (kBlock / 16))), kBlock, 0, st>>>(
        split_ws, nsplit, o, lse, d);
    return launch_status();
  }
  if (bias && scores) attn_fwd_kernel<true, true><<<grid, kBlock, 0, st>>>(q, k, v, bias, o, lse, scores, d, key_hole, sp);
  else if (bias) attn_fwd_kernel<true, false><<<grid, kBlock, 0, st>>>(q, k, v, bias, o, lse, scores, d, key_hole, sp);
  else if (scores) attn_fwd_kernel<false, true><<<grid, kBlock, 0, st>>>(q, k, v, bias, o, lse, scores, d, key_hole, sp);
  else attn_fwd_kernel<false, false><<<grid, kBlock, 0, st>>>(q, k, v, bias, o, lse, scores, d, key_hole, sp);
  return launch_status();
}

long vqa_attn_bwd_ws_floats(int B, int H, int Sq, int Sk) {
  if (B < 0 || H <= 0 || Sq <= 0 || Sk <= 0) return 0;
  return static_cast<long>(B) * H * ds_rows(Sk) * ds_pitch(Sq);
}

/* grad_strides: 12 longs = {go_sb, go_ss, go_sh, dq_sb, dq_ss, dq_sh, dk_sb, dk_ss, dk_sh, dv_sb, dv_ss, dv_sh} */
int vqa_attn_bwd(const float* q, const float* k, const float* v, const float* bias, const float* o, const float* go,
                 const float* lse, const float* scores, float* delta, float* dq, float* dk, float* dv, float* ds_ws, int B,
                 int H, int Sq, int Sk, const long* strides, const long* bias_strides, const long* grad_strides,
                 float scale, int nsplit, float* split_ws, vqa_stream_t stream) {
  clear_stale_error();
  if (!strides || !grad_strides || !o || !go || !lse || !delta || !dq || !dk || !dv || (bias && !bias_strides))
    return VQA_ERR_NULL;
  AttnDims d{B, H, Sq, Sk, strides[0], strides[1], strides[2], strides[3], strides[4], strides[5], strides[6],
             strides[7], strides[8], strides[9], strides[10], strides[11], 0, 0, 0, scale};
  if (bias) {
    d.bias_sb = bias_strides[0];
    d.bias_sh = bias_strides[1];
    d.bias_sr = bias_strides[2];
    if (((d.bias_sb | d.bias_sh | d.bias_sr) & 3) || !aligned16(bias)) return VQA_ERR_ALIGN;
    // the dK / dV kernel addresses a (batch, head) slab of the bias with 32-bit lane offsets
    if (d.bias_sr < 0 || (static_cast<long>(Sq) + 4) * d.bias_sr + Sk >= 2147483647L) return VQA_ERR_SHAPE;
  }
  const int rc = check_attn(q, k, v, d);
  if (rc != VQA_OK) return rc;
  long any = 0;
  for (int i = 0; i < 12; ++i) any |= grad_strides[i];
  if (any & 3) return VQA_ERR_SHAPE;
  if (!aligned16(o) || !aligned16(go) || !aligned16(dq) || !aligned16(dk) || !aligned16(dv) || !aligned16(ds_ws) ||
      !aligned16(scores))
    return VQA_ERR_ALIGN;
  // the dS path addresses a (batch, head) slab of either workspace with 32-bit lane offsets
  if (ds_ws && (ds_rows(Sk) * ds_pitch(Sq) >= 2147483647L || sc_rows(Sq) * sc_pitch(Sk) >= 2147483647L))
    return VQA_ERR_SHAPE;
  {
    const int tq = (Sq + kTile - 1) / kTile, tk = (Sk + kTile - 1) / kTile;
    const long big = (((Sq > Sk ? Sq : Sk) + 127L) / 128) * H * B;
    const int rs = check_split(nsplit, split_ws, tq < tk ? tq : tk, big);
    if (rs != VQA_OK) return rs;
    if (nsplit > 1 && !(ds_ws && scores)) return VQA_ERR_SHAPE;      // the split forms exist for the saved-scores backward
  }
  if (B == 0) return VQA_OK;
  const long* g = grad_strides;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const SplitArgs sp{nsplit, split_ws};
  const dim3 gq(static_cast<unsigned>(((Sq + 127) / 128) * H * B * nsplit)),
      gk(static_cast<unsigned>(((Sk + 127) / 128) * H * B * nsplit));
  if (ds_ws && nsplit > 1) {                     // small batches: the loops of both kernels cut into nsplit parts
    const long rq = static_cast<long>(B) * H * Sq, rk = static_cast<long>(B) * H * Sk;
    const int per = kBlock / 16;
    attn_delta_kernel<<<dim3(static_cast<unsigned>((rq + per - 1) / per)), kBlock, 0, st>>>(o, go, delta, d, g[0], g[1], g[2]);
    if (bias)
      attn_bwd_dkv_kernel<true, true, true, true><<<gk, kBlock, 0, st>>>(q, k, v, bias, go, lse, delta, dk, dv, d, g[6], g[7],
                                                                         g[8], g[9], g[10], g[11], g[0], g[1], g[2], ds_ws,
                                                                         scores, sp);
    else
      attn_bwd_dkv_kernel<false, true, true, true><<<gk, kBlock, 0, st>>>(q, k, v, bias, go, lse, delta, dk, dv, d, g[6],
                                                                          g[7], g[8], g[9], g[10], g[11], g[0], g[1], g[2],
                                                                          ds_ws, scores, sp);
    const dim3 gsk(static_cast<unsigned>((rk + per - 1) / per), 2), gsq(static_cast<unsigned>((rq + per - 1) / per), 1);
    const SumTarget tk{split_ws, dk, g[6], g[7], g[8]}, tv{split_ws + split_sum_floats(rk, nsplit), dv, g[9], g[10], g[11]};
    attn_sum_parts_kernel<<<gsk, kBlock, 0, st>>>(tk, tv, nsplit, rk, H, Sk);                       // dk and dv in one launch
    attn_bwd_dq_staged_kernel<true><<<gq, kBlock, 0, st>>>(k, ds_ws, dq, d, g[3], g[4], g[5], sp);   // after the sums: same ws
    const SumTarget tq{split_ws, dq, g[3], g[4], g[5]};
    attn_sum_parts_kernel<<<gsq, kBlock, 0, st>>>(tq, tq, nsplit, rq, H, Sq);
    return launch_status();
  }
  if (ds_ws) {                                   // 5 products: delta pre-pass, dK / dV (+ dS^T store), dQ from dS^T
    const long n_rows = static_cast<long>(B) * H * Sq;
    const long blocks = (n_rows + kBlock / 16 - 1) / (kBlock / 16);
    if (blocks > 2147483647L) return VQA_ERR_SHAPE;
    attn_delta_kernel<<<dim3(static_cast<unsigned>(blocks)), kBlock, 0, st>>>(o, go, delta, d, g[0], g[1], g[2]);
    if (scores && bias)                          // saved scores already include the bias
      attn_bwd_dkv_kernel<true, true, true><<<gk, kBlock, 0, st>>>(q, k, v, bias, go, lse, delta, dk, dv, d, g[6], g[7],
                                                                   g[8], g[9], g[10], g[11], g[0], g[1], g[2], ds_ws,
                                                                   scores, sp);
    else if (scores)
      attn_bwd_dkv_kernel<false, true, true><<<gk, kBlock, 0, st>>>(q, k, v, bias, go, lse, delta, dk, dv, d, g[6], g[7],
                                                                    g[8], g[9], g[10], g[11], g[0], g[1], g[2], ds_ws,
                                                                    scores, sp);
    else if (bias)
      attn_bwd_dkv_kernel<true, true, false><<<gk, kBlock, 0, st>>>(q, k, v, bias, go, lse, delta, dk, dv, d, g[6], g[7],
                                                                    g[8], g[9], g[10], g[11], g[0], g[1], g[2], ds_ws,
                                                                    nullptr, sp);
    else
      attn_bwd_dkv_kernel<false, true, false><<<gk, kBlock, 0, st>>>(q, k, v, bias, go, lse, delta, dk, dv, d, g[6],
                                                                     g[7], g[8], g[9], g[10], g[11], g[0], g[1], g[2],
                                                                     ds_ws, nullptr, sp);
    attn_bwd_dq_staged_kernel<false><<<gq, kBlock, 0, st>>>(k, ds_ws, dq, d, g[3], g[4], g[5], sp);
  } else if (bias) {                             // no workspace: 7 products, both kernels recompute the scores
    attn_bwd_dq_kernel<true><<<gq, kBlock, 0, st>>>(q, k, v, bias, o, go, lse, delta, dq, d, g[3], g[4], g[5], g[0], g[1],
                                                     g[2]);
    attn_bwd_dkv_kernel<true, false, false><<<gk, kBlock, 0, st>>>(q, k, v, bias, go, lse, delta, dk, dv, d, g[6], g[7],
                                                                   g[8], g[9], g[10], g[11], g[0], g[1], g[2], nullptr,
                                                                   nullptr, sp);
  } else {
    attn_bwd_dq_kernel<false><<<gq, kBlock, 0, st>>>(q, k, v, bias, o, go, lse, delta, dq, d, g[3], g[4], g[5], g[0],
                                                      g[1], g[2]);
    attn_bwd_dkv_kernel<false, false, false><<<gk, kBlock, 0, st>>>(q, k, v, bias, go, lse, delta, dk, dv, d, g[6], g[7],
                                                                    g[8], g[9], g[10], g[11], g[0], g[1], g[2], nullptr,
                                                                    nullptr, sp);
  }
  return launch_status();
}

}  // extern "C"
