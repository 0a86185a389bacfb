/* TEST INFRASTRUCTURE ONLY -- plain C restatement of the reference's per-iteration op chain for norm = inf.
 *
 * Follows, element by element and in the reference's order (fp32, one rounding per op, no FMA contraction):
 *   s  = torch.sign(g)                     utils.py:86            (sign(+-0) = 0, sign(NaN) = 0)
 *   p  = eps_iter * s                      utils.py:127
 *   a  = clamp(x + p, cmin, cmax)          fast_gradient_method.py:152-160
 *   e  = clamp(a - x0, -eps, eps)          projected_gradient_descent.py:146-147 + utils.py:21
 *   x' = clamp(x0 + e, cmin, cmax)         projected_gradient_descent.py:149-151
 * (paths relative to ALBEF_VQAttack/cleverhans/cleverhans/torch).  It is a second, PyTorch-independent statement of
 * the arithmetic the HIP kernel vqa_linf_step must reproduce bit for bit; tests/test_oracle_c.py checks it against the
 * golden-pinned torch oracle.  Build: `make -C oracle` (gcc -O2 -ffp-contract=off). */
#include <math.h>
#include <stddef.h>

static float sign_torch(float g) { return (float)(g > 0.0f) - (float)(g < 0.0f); }

/* torch.clamp: min(max(v, lo), hi), NaN propagates */
static float clamp_torch(float v, float lo, float hi) {
  if (v != v) return v;
  float r = v < lo ? lo : v;
  return r > hi ? hi : r;
}

void oracle_linf_step(const float* x, const float* g, const float* x0, float* out, size_t n, float eps_iter, float eps,
                      float cmin, float cmax, int clip) {
  for (size_t i = 0; i < n; ++i) {
    float a = x[i] + eps_iter * sign_torch(g[i]);
    if (clip) a = clamp_torch(a, cmin, cmax);
    float e = clamp_torch(a - x0[i], -eps, eps);
    float v = x0[i] + e;
    out[i] = clip ? clamp_torch(v, cmin, cmax) : v;
  }
}

/* 1 when every element of x lies in [cmin, cmax] (NaN fails), like the reference's torch.all(ge)/torch.all(le) flags */
int oracle_range_ok(const float* x, size_t n, float cmin, float cmax) {
  for (size_t i = 0; i < n; ++i)
    if (!(x[i] >= cmin) || !(x[i] <= cmax)) return 0;
  return 1;
}
