// Text side of the joint attack: gather of text-embedding gradient rows and candidate-substitution scoring
// (gfx950 / MI355X).  Both are row-granular HBM/L2 gathers: one wavefront per row, 16-byte lanes, the row held
// in registers (D <= 2048), wave-shuffle reductions -- no LDS, no atomics.
#include "common.hpp"

namespace vqa {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kMaxCh = 8;   // 8 x 256 floats = D <= 2048

// dst[b, k, :] = src[b, idx[k], :]
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(const float* __restrict__ src,
                                                             const int64_t* __restrict__ idx,
                                                             float* __restrict__ dst, int B, int L, int K, int D,
                                                             int vec) {
  const int lane = threadIdx.x & (kWave - 1);
  const long rows = static_cast<long>(B) * K;
  for (long r = static_cast<long>(blockIdx.x) * kWavesPerBlock + threadIdx.x / kWave; r < rows;
       r += static_cast<long>(gridDim.x) * kWavesPerBlock) {
    const long b = r / K, k = r - b * K;
    long t = idx[k];
    if (t < 0) t += L;                       // python-style negative index
    if (t < 0 || t >= L) continue;           // host validates; never read out of bounds
    const float* s = src + (b * L + t) * D;
    float* d = dst + r * D;
    if (vec) {
      for (int c = lane * 4; c < D; c += kWave * 4)
        *reinterpret_cast<f32x4*>(d + c) = *reinterpret_cast<const f32x4*>(s + c);
    } else {
      for (int c = lane; c < D; c += kWave) d[c] = s[c];
    }
  }
}

// One wave per candidate {sample, position, grad row, vocab id}.
template <int NCH>
__global__ __launch_bounds__(kBlock) void cand_dir_sim_kernel(
    const float* __restrict__ word, const float* __restrict__ pos, const float* __restrict__ type,
    const float* __restrict__ gamma, const float* __restrict__ beta, float ln_eps,
    const float* __restrict__ e_ori, const float* __restrict__ grad, const int32_t* __restrict__ cand,
    float* __restrict__ out, int n_cand, int L, int K, int D) {
  const int lane = threadIdx.x & (kWave - 1);
  for (int c = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave; c < n_cand; c += gridDim.x * kWavesPerBlock) {
    const int s = cand[4 * c + 0], p = cand[4 * c + 1], k = cand[4 * c + 2], v = cand[4 * c + 3];
    const float* pw = word + static_cast<long>(v) * D;
    const float* pp = pos + static_cast<long>(p) * D;
    const float* po = e_ori + (static_cast<long>(s) * L + p) * D;
    const float* pg = grad + (static_cast<long>(s) * K + k) * D;
    f32x4 e[NCH];
    float sum = 0.0f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int d = (ch * kWave + lane) * 4;
      if (d < D) {
        f32x4 w = *reinterpret_cast<const f32x4*>(pw + d);
        f32x4 q = *reinterpret_cast<const f32x4*>(pp + d);
        f32x4 t = *reinterpret_cast<const f32x4*>(type + d);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          e[ch][j] = (w[j] + t[j]) + q[j];   // BertEmbeddings: inputs_embeds + token_type, then + position
          sum += e[ch][j];
        }
      } else {
        e[ch] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      }
    }
    const float mean = wave_sum(sum) / static_cast<float>(D);
    float var = 0.0f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int d = (ch * kWave + lane) * 4;
      if (d < D) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float c0 = e[ch][j] - mean;
          var += c0 * c0;
        }
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(var) / static_cast<float>(D) + ln_eps);
    float dd = 0.0f, gg = 0.0f, dg = 0.0f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int d = (ch * kWave + lane) * 4;
      if (d < D) {
        f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + d);
        f32x4 be = *reinterpret_cast<const f32x4*>(beta + d);
        f32x4 eo = *reinterpret_cast<const f32x4*>(po + d);
        f32x4 gr = *reinterpret_cast<const f32x4*>(pg + d);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float ln = (e[ch][j] - mean) * rstd * ga[j] + be[j];
          const float dir = ln - eo[j];
          dd += dir * dir;
          gg += gr[j] * gr[j];
          dg += dir * gr[j];
        }
      }
    }
    dd = wave_sum(dd);
    gg = wave_sum(gg);
    dg = wave_sum(dg);
    // F.normalize(p=2, eps=1e-12) on both vectors, then CosineSimilarity(eps=1e-6) of the unit vectors
    const float nd = fmaxf(sqrtf(dd), 1e-12f), ng = fmaxf(sqrtf(gg), 1e-12f);
    const float und = sqrtf(dd) / nd, ung = sqrtf(gg) / ng;       // norms of the normalised vectors (1 or 0)
    const float cosv = (dg / (nd * ng)) / (fmaxf(und, 1e-6f) * fmaxf(ung, 1e-6f));
    if (lane == 0) out[c] = cosv;
  }
}

// BERT text embeddings for explicit (row, position, token) triples: dst[row[i]] = LN(word[tok[i]] + type[0] + pos[position[i]]).
// With rows = 0..B*L-1 this embeds a whole question batch in one launch (instead of gather + add + add + LayerNorm);
// with a short list it overwrites only the rows of substituted words (the joint attack's masked-token substitution).
template <int NCH>
__global__ __launch_bounds__(kBlock) void embed_tokens_kernel(const float* __restrict__ word,
                                                              const float* __restrict__ pos,
                                                              const float* __restrict__ type,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float ln_eps,
                                                              const int32_t* __restrict__ triples, int n,
                                                              float* __restrict__ dst, int D) {
  const int lane = threadIdx.x & (kWave - 1);
  for (int i = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave; i < n; i += gridDim.x * kWavesPerBlock) {
    const long row = triples[3 * i + 0];
    const float* pw = word + static_cast<long>(triples[3 * i + 2]) * D;
    const float* pp = pos + static_cast<long>(triples[3 * i + 1]) * D;
    f32x4 e[NCH];
    float sum = 0.0f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int d = (ch * kWave + lane) * 4;
      if (d < D) {
        f32x4 w = *reinterpret_cast<const f32x4*>(pw + d);
        f32x4 q = *reinterpret_cast<const f32x4*>(pp + d);
        f32x4 t = *reinterpret_cast<const f32x4*>(type + d);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          e[ch][j] = (w[j] + t[j]) + q[j];
          sum += e[ch][j];
        }
      } else {
        e[ch] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      }
    }
    const float mean = wave_sum(sum) / static_cast<float>(D);
    float var = 0.0f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int d = (ch * kWave + lane) * 4;
      if (d < D) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float c0 = e[ch][j] - mean;
          var += c0 * c0;
        }
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(var) / static_cast<float>(D) + ln_eps);
    float* out = dst + row * D;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int d = (ch * kWave + lane) * 4;
      if (d < D) {
        f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + d);
        f32x4 be = *reinterpret_cast<const f32x4*>(beta + d);
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = (e[ch][j] - mean) * rstd * ga[j] + be[j];
        *reinterpret_cast<f32x4*>(out + d) = r;
      }
    }
  }
}


// Greedy acceptance of update_adv_text for a whole batch, one wavefront per sample, with the bag-of-embeddings
// sentence similarity (the stand-in for the reference's TF-Hub sentence encoder):
//     sim(ids) = <m(ori), m(ids)> / (|m(ori)| |m(ids)| + 1e-12),   m(ids) = mean over non-pad tokens of table[id].
// The sample's candidates are visited in descending dir_sim order; a candidate whose word is still free is accepted
// iff the similarity of the question with that word replaced exceeds the threshold, which then rises to that
// similarity (ALBEF_attack/adv_attack.py:299-323).  The current ids live one per lane (L <= 64), the sentence vectors
// as NE = E/64 floats per lane; every trial re-sums the sentence's rows of the table in token order (L loads per lane,
// independent, L2-resident), reductions are DPP.  Sequential by nature within a sample, parallel over samples.
template <int NE>
__global__ __launch_bounds__(kWave) void greedy_accept_kernel(const int32_t* __restrict__ cand,
                                                              const int32_t* __restrict__ order,
                                                              const int32_t* __restrict__ seg, int L,
                                                              const int64_t* __restrict__ ori_ids,
                                                              int64_t* __restrict__ cur_ids, int32_t* __restrict__ new_id,
                                                              int32_t* __restrict__ acc_rank,
                                                              const float* __restrict__ table, int V, int E,
                                                              float threshold) {
  const int lane = threadIdx.x;
  const int s = blockIdx.x;
  const long base = static_cast<long>(s) * L;
  // lane t keeps token t of the original and of the current question (0 = padding beyond L)
  long ori_tok = lane < L ? ori_ids[base + lane] : 0;
  long cur_tok = lane < L ? cur_ids[base + lane] : 0;
  if (ori_tok < 0 || ori_tok >= V) ori_tok = 0;      // out-of-vocabulary ids count as padding (host validates)
  if (cur_tok < 0 || cur_tok >= V) cur_tok = 0;
  int accepted = -1, my_rank = -1, n_acc = 0;         // lane t: id accepted at position t and its acceptance order
  float ov[NE], tv[NE];
  // m(ori)
  int n = 0;
#pragma unroll
  for (int e = 0; e < NE; ++e) ov[e] = 0.0f;
  for (int t = 0; t < L; ++t) {
    const long id = __shfl(ori_tok, t, kWave);
    if (id != 0) {
      ++n;
#pragma unroll
      for (int e = 0; e < NE; ++e)
        if (e * kWave + lane < E) ov[e] += table[id * E + e * kWave + lane];
    }
  }
  float oo = 0.0f;
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    ov[e] = ov[e] / static_cast<float>(n);
    oo += ov[e] * ov[e];
  }
  const float no = sqrtf(wave_sum(oo));
  float thr = threshold;
  unsigned long long taken = 0ull;                   // bit p: word at position p already substituted (wave-uniform)
  const int k1 = seg[s + 1];
  for (int k = seg[s]; k < k1; ++k) {
    const int c = order[k];
    const int p = cand[4 * c + 1], v = cand[4 * c + 3];
    if (p < 0 || p >= L || v < 0 || v >= V) continue;
    if ((taken >> p) & 1ull) continue;
    n = 0;
#pragma unroll
    for (int e = 0; e < NE; ++e) tv[e] = 0.0f;
    for (int t = 0; t < L; ++t) {
      long id = __shfl(cur_tok, t, kWave);
      if (t == p) id = v;
      if (id != 0) {
        ++n;
#pragma unroll
        for (int e = 0; e < NE; ++e)
          if (e * kWave + lane < E) tv[e] += table[id * E + e * kWave + lane];
      }
    }
    float dot = 0.0f, tt = 0.0f;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      tv[e] = tv[e] / static_cast<float>(n);
      dot += ov[e] * tv[e];
      tt += tv[e] * tv[e];
    }
    dot = wave_sum(dot);
    const float sim = dot / (no * sqrtf(wave_sum(tt)) + 1e-12f);
    if (sim > thr) {                                  // wave-uniform
      thr = sim;
      taken |= 1ull << p;
      if (lane == p) {
        cur_tok = v;
        accepted = v;
        my_rank = n_acc;
      }
      ++n_acc;
    }
  }
  if (lane < L) {
    cur_ids[base + lane] = cur_tok;
    new_id[base + lane] = accepted;
    if (acc_rank) acc_rank[base + lane] = my_rank;
  }
}

}  // namespace vqa

using namespace vqa;

extern "C" {

int vqa_gather_rows(const float* src, const int64_t* idx, float* dst, int B, int L, int K, int D,
                    vqa_stream_t stream) {
  clear_stale_error();
  if (!src || !idx || !dst) return VQA_ERR_NULL;
  if (B < 0 || L <= 0 || K < 0 || D <= 0) return VQA_ERR_SHAPE;
  if (!aligned4(src) || !aligned4(dst)) return VQA_ERR_ALIGN;
  if (B == 0 || K == 0) return VQA_OK;
  const int vec = (D % 4 == 0) && aligned16(src) && aligned16(dst);
  const int grid = blocks_for(static_cast<size_t>(B) * K, kWavesPerBlock);
  gather_rows_kernel<<<grid, kBlock, 0, static_cast<hipStream_t>(stream)>>>(src, idx, dst, B, L, K, D, vec);
  return launch_status();
}

int vqa_cand_dir_sim(const float* word, const float* pos, const float* type, const float* gamma, const float* beta,
                     float ln_eps, const float* e_ori, const float* grad, const int32_t* cand, float* out,
                     int n_cand, int L, int K, int D, vqa_stream_t stream) {
  clear_stale_error();
  if (!word || !pos || !type || !gamma || !beta || !e_ori || !grad || !cand || !out) return VQA_ERR_NULL;
  if (n_cand < 0 || L <= 0 || K <= 0 || D <= 0 || D > 256 * kMaxCh || (D & 3)) return VQA_ERR_SHAPE;
  if (!aligned16(word) || !aligned16(pos) || !aligned16(type) || !aligned16(gamma) || !aligned16(beta) ||
      !aligned16(e_ori) || !aligned16(grad))
    return VQA_ERR_ALIGN;
  if (n_cand == 0) return VQA_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int grid = blocks_for(static_cast<size_t>(n_cand), kWavesPerBlock);
#define VQA_CAND(N) \
  cand_dir_sim_kernel<N><<<grid, kBlock, 0, st>>>(word, pos, type, gamma, beta, ln_eps, e_ori, grad, cand, out, \
                                                  n_cand, L, K, D)
  switch ((D + 255) / 256) {
    case 1: VQA_CAND(1); break;
    case 2: VQA_CAND(2); break;
    case 3: VQA_CAND(3); break;
    case 4: VQA_CAND(4); break;
    case 5: VQA_CAND(5); break;
    case 6: VQA_CAND(6); break;
    case 7: VQA_CAND(7); break;
    default: VQA_CAND(8); break;
  }
#undef VQA_CAND
  return launch_status();
}

int vqa_embed_tokens(const float* word, const float* pos, const float* type, const float* gamma, const float* beta,
                     float ln_eps, const int32_t* triples, int n, float* dst, int D, vqa_stream_t stream) {
  clear_stale_error();
  if (!word || !pos || !type || !gamma || !beta || !triples || !dst) return VQA_ERR_NULL;
  if (n < 0 || D <= 0 || D > 256 * kMaxCh || (D & 3)) return VQA_ERR_SHAPE;
  if (!aligned16(word) || !aligned16(pos) || !aligned16(type) || !aligned16(gamma) || !aligned16(beta) ||
      !aligned16(dst))
    return VQA_ERR_ALIGN;
  if (n == 0) return VQA_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int grid = blocks_for(static_cast<size_t>(n), kWavesPerBlock);
#define VQA_EMB(N) \
  embed_tokens_kernel<N><<<grid, kBlock, 0, st>>>(word, pos, type, gamma, beta, ln_eps, triples, n, dst, D)
  switch ((D + 255) / 256) {
    case 1: VQA_EMB(1); break;
    case 2: VQA_EMB(2); break;
    case 3: VQA_EMB(3); break;
    case 4: VQA_EMB(4); break;
    case 5: VQA_EMB(5); break;
    case 6: VQA_EMB(6); break;
    case 7: VQA_EMB(7); break;
    default: VQA_EMB(8); break;
  }
#undef VQA_EMB
  return launch_status();
}

int vqa_greedy_accept(const int32_t* cand, const int32_t* order, const int32_t* seg, int B, int L,
                      const int64_t* ori_ids, int64_t* cur_ids, int32_t* new_id, int32_t* acc_rank, const float* table,
                      int V, int E, float threshold, vqa_stream_t stream) {
  clear_stale_error();
  if (!cand || !order || !seg || !ori_ids || !cur_ids || !new_id || !table) return VQA_ERR_NULL;
  if (B < 0 || L <= 0 || L > kWave || V <= 0 || E <= 0 || E > kWave * 8) return VQA_ERR_SHAPE;
  if (!aligned4(table)) return VQA_ERR_ALIGN;
  if (B == 0) return VQA_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
#define VQA_ACC(N) \
  greedy_accept_kernel<N><<<B, kWave, 0, st>>>(cand, order, seg, L, ori_ids, cur_ids, new_id, acc_rank, table, V, E, \
                                               threshold)
  switch ((E + kWave - 1) / kWave) {
    case 1: VQA_ACC(1); break;
    case 2: VQA_ACC(2); break;
    case 3: VQA_ACC(3); break;
    case 4: VQA_ACC(4); break;
    default: VQA_ACC(8); break;
  }
#undef VQA_ACC
  return launch_status();
}

int vqa_abi_version(void) { return 4; }

const char* vqa_error_string(int code) {
  switch (code) {
    case VQA_OK: return "ok";
    case VQA_ERR_NULL: return "a required pointer is NULL";
    case VQA_ERR_SHAPE: return "a count/stride/option argument is out of the supported range";
    case VQA_ERR_ALIGN: return "a pointer is not aligned as required";
    default: return code > 0 ? hipGetErrorString(static_cast<hipError_t>(code)) : "unknown vqattack error";
  }
}

}  // extern "C"
