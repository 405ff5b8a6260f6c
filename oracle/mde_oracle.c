/*
 * mde_oracle.c -- CPU restatement of the reference's average-distortion forward/backward.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker the parity tests, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg compare the HIP path against; nothing under pymde_amd/ may
 * import, link or call it.
 *
 * It follows the reference algorithm line by line, in the reference's EDGE order with a
 * scatter-add backward (no plan, no symmetrised CSR -- deliberately not the HIP design):
 *
 *   forward   [ref: pymde/average_distortion.py:63-92]
 *     diff_k = X[i_k] - X[j_k];  d_k = sqrt(sum diff_k^2)
 *     E = (1/p) sum_k f_k(d_k)
 *     g_k = (f'_k(d_k) / p) / d_k;  NaN -> 1, Inf -> 1            (:81-88)
 *   backward  [ref: pymde/average_distortion.py:94-106]
 *     grad[i_k] += g_k diff_k;  grad[j_k] -= g_k diff_k;  grad *= grad_output
 *
 * f_k and f'_k restate pymde/functions/penalties.py:112-400 and losses.py:61-239 (the
 * derivative is what torch autograd yields for those expressions, including its conventions
 * sign(0) = 0, max/min ties split 1/2, pow(x, 0) having zero gradient).  Per-edge arithmetic
 * is float32 with libm (powf/log1pf/expm1f/...), sums are accumulated in double.
 *
 * Parity pin: tests/test_oracle.py checks this file against the golden vectors generated from
 * the reference itself (tests/golden/make_golden.py) and against the reference's own
 * known-answer tests (pymde/test_optim.py:75-118).
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -shared).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* function kinds: same numbering as include/mde_hip.h */
enum {
  F_NONE = 0, F_LINEAR = 1, F_QUADRATIC = 2, F_CUBIC = 3, F_POWER = 4, F_HUBER = 5, F_LOGISTIC = 6,
  F_SIGMOID = 7, F_HINGE = 8, F_LOG1P = 9, F_LOG = 10, F_INVPOWER = 11, F_LOGRATIO = 12,
  F_DEADZONE_QUADRATIC = 13, F_DEADZONE_CUBIC = 14, F_CLIPPED_QUADRATIC = 15,
  L_QUADRATIC = 32, L_WEIGHTED_QUADRATIC = 33, L_HUBER = 34, L_CUBIC = 35, L_POWER = 36,
  L_WEIGHTED_POWER = 37, L_ABSOLUTE = 38, L_LOGISTIC = 39, L_FRACTIONAL = 40,
  L_SOFT_FRACTIONAL = 41, L_CLIPPED_QUADRATIC = 42, L_LOG1P = 43
};

typedef struct {
  int32_t kind, kind_neg;
  const float* a0; /* weights / deviations, EDGE order */
  const float* a1; /* weights of the weighted losses, or NULL */
  int32_t a0_scalar, a1_scalar;
  float s0, s1, s2; /* scalars of kind     */
  float n0, n1, n2; /* scalars of kind_neg */
} oracle_func;

static float sgnf(float x) { return (x > 0.f) - (x < 0.f); }

/* torch.pow(x, e) backward: e * x^(e-1), defined as 0 when e == 0 */
static float dpowf(float x, float e) { return e == 0.f ? 0.f : e * powf(x, e - 1.f); }

/* f(d) and f'(d) of one kind.  a0 = weight or deviation, a1 = second per-edge parameter. */
static void eval_kind(int kind, float d, float a0, float a1, float s0, float s1, float s2, float* f,
                      float* fp) {
  (void)s2;
  switch (kind) {
    case F_LINEAR: /* penalties.py:112-120  w d */
      *f = a0 * d;
      *fp = a0;
      break;
    case F_QUADRATIC: /* :123-131  w d^2 */
      *f = a0 * d * d;
      *fp = a0 * 2.f * d;
      break;
    case F_CUBIC: /* :163-171 */
      *f = a0 * d * d * d;
      *fp = a0 * 3.f * d * d;
      break;
    case F_POWER: /* :191-202 */
      *f = a0 * powf(d, s0);
      *fp = a0 * dpowf(d, s0);
      break;
    case F_HUBER: /* :205-243, strict < at :230 */
      if (d < s0) {
        *f = a0 * 0.5f * d * d;
        *fp = a0 * d;
      } else {
        *f = a0 * s0 * (d - 0.5f * s0);
        *fp = a0 * s0;
      }
      break;
    case F_LOGISTIC: { /* :246-266  w logsumexp(0, alpha (d - threshold)) */
      const float z = s1 * (d - s0);
      const float m = z > 0.f ? z : 0.f;
      *f = a0 * (m + log1pf(expf(-fabsf(z))));
      *fp = a0 * s1 / (1.f + expf(-z));
      break;
    }
    case F_SIGMOID: { /* :269-283 */
      const float sg = 1.f / (1.f + expf(-s1 * (d - s0)));
      *f = a0 * sg;
      *fp = a0 * s1 * sg * (1.f - sg);
      break;
    }
    case F_HINGE: { /* :286-307  max(0, w (d - (threshold - sign(w) sigma))) */
      const float v = a0 * (d - (s0 - sgnf(a0) * s1));
      *f = v > 0.f ? v : 0.f;
      *fp = v > 0.f ? a0 : (v == 0.f ? 0.5f * a0 : 0.f);
      break;
    }
    case F_LOG1P: { /* :310-321  w log1p(d^e) */
      const float pe = powf(d, s0);
      *f = a0 * log1pf(pe);
      *fp = a0 * dpowf(d, s0) / (1.f + pe);
      break;
    }
    case F_LOG: { /* :324-337  w log(-expm1(-d^e)) */
      const float u = powf(d, s0);
      const float A = -expm1f(-u);
      *f = a0 * logf(A);
      /* d/du log(-expm1(-u)) = exp(-u) / (-expm1(-u)) */
      *fp = a0 * (expf(-u) / A) * dpowf(d, s0);
      break;
    }
    case F_INVPOWER: { /* :340-353  |w| / d^e */
      const float pe = powf(d, s0);
      *f = fabsf(a0) / pe;
      *fp = -fabsf(a0) * dpowf(d, s0) / (pe * pe);
      break;
    }
    case F_LOGRATIO: { /* :356-369  w log(d^e / (1 + d^e)) */
      const float pe = powf(d, s0);
      *f = a0 * logf(pe / (1.f + pe));
      /* d/dpe log(pe/(1+pe)) = 1/(pe (1+pe)) */
      *fp = a0 * dpowf(d, s0) / (pe * (1.f + pe));
      break;
    }
    case F_DEADZONE_QUADRATIC: /* :134-148 */
      *f = d < s0 ? 0.f : a0 * d * d;
      *fp = d < s0 ? 0.f : a0 * 2.f * d;
      break;
    case F_DEADZONE_CUBIC: /* :174-188 */
      *f = d < s0 ? 0.f : a0 * d * d * d;
      *fp = d < s0 ? 0.f : a0 * 3.f * d * d;
      break;
    case F_CLIPPED_QUADRATIC: { /* :151-160 */
      const float c = (s0 + 1.f) * (s0 + 1.f), q = d * d;
      *f = a0 * (q < c ? q : c);
      *fp = q < c ? a0 * 2.f * d : (q == c ? a0 * d : 0.f);
      break;
    }
    case L_QUADRATIC: /* losses.py:61-69 */
      *f = (a0 - d) * (a0 - d);
      *fp = -2.f * (a0 - d);
      break;
    case L_WEIGHTED_QUADRATIC: /* :72-87 */
      *f = a1 * (a0 - d) * (a0 - d);
      *fp = -2.f * a1 * (a0 - d);
      break;
    case L_CLIPPED_QUADRATIC: { /* :90-98 */
      const float r = a0 - d, q = r * r, c = (s0 + 1.f) * (s0 + 1.f);
      *f = q < c ? q : c;
      *fp = q < c ? -2.f * r : (q == c ? -r : 0.f);
      break;
    }
    case L_HUBER: { /* :101-125, strict < at :122 */
      const float r = fabsf(a0 - d), sg = sgnf(d - a0);
      if (r < s0) {
        *f = r * r;
        *fp = 2.f * r * sg;
      } else {
        *f = s0 * (2.f * r - s0);
        *fp = 2.f * s0 * sg;
      }
      break;
    }
    case L_CUBIC: { /* :128-136 */
      const float r = fabsf(a0 - d);
      *f = r * r * r;
      *fp = 3.f * r * r * sgnf(d - a0);
      break;
    }
    case L_POWER: { /* :139-148 */
      const float r = fabsf(a0 - d);
      *f = powf(r, s0);
      *fp = dpowf(r, s0) * sgnf(d - a0);
      break;
    }
    case L_WEIGHTED_POWER: { /* :151-163 */
      const float r = fabsf(a0 - d);
      *f = a1 * powf(r, s0);
      *fp = a1 * dpowf(r, s0) * sgnf(d - a0);
      break;
    }
    case L_ABSOLUTE: /* :166-174 */
      *f = fabsf(a0 - d);
      *fp = sgnf(d - a0);
      break;
    case L_LOGISTIC: { /* :177-186  log(1 + exp(|delta - d|)) */
      const float r = fabsf(a0 - d);
      *f = logf(1.f + expf(r));
      *fp = (expf(r) / (1.f + expf(r))) * sgnf(d - a0);
      break;
    }
    case L_FRACTIONAL: { /* :189-200  max(delta/d, d/delta) - 1 */
      const float q1 = a0 / d, q2 = d / a0;
      const float g1 = -a0 / (d * d), g2 = 1.f / a0;
      *f = (q1 > q2 ? q1 : q2) - 1.f;
      *fp = q1 > q2 ? g1 : (q1 < q2 ? g2 : 0.5f * (g1 + g2));
      break;
    }
    case L_SOFT_FRACTIONAL: { /* :203-229 */
      const float q1 = s0 * a0 / d, q2 = s0 * d / a0;
      const float m = q1 > q2 ? q1 : q2;
      const float e1 = expf(q1 - m), e2 = expf(q2 - m);
      const float lse = m + logf(e1 + e2);
      *f = (1.f / s0) * (lse - (logf(2.f) + s0));
      const float w1 = e1 / (e1 + e2), w2 = e2 / (e1 + e2);
      *fp = (1.f / s0) * (w1 * (-s0 * a0 / (d * d)) + w2 * (s0 / a0));
      break;
    }
    case L_LOG1P: { /* :232-239  log(1 + (d - delta)^e); libm powf has torch.pow's semantics
                       (negative base: finite for an integer exponent, NaN otherwise) */
      const float r = d - a0, pe = powf(r, s0);
      *f = logf(1.f + pe);
      *fp = dpowf(r, s0) / (1.f + pe);
      break;
    }
    default:
      *f = 0.f;
      *fp = 0.f;
  }
}

static void eval_func(const oracle_func* F, int64_t k, float d, float* f, float* fp) {
  const float a0 = F->a0_scalar ? F->a0[0] : F->a0[k];
  const float a1 = F->a1 ? (F->a1_scalar ? F->a1[0] : F->a1[k]) : 0.f;
  /* PushAndPull: attractive where w >= 0 (penalties.py:390), repulsive elsewhere */
  if (F->kind_neg != F_NONE && !(a0 >= 0.f))
    eval_kind(F->kind_neg, d, a0, a1, F->n0, F->n1, F->n2, f, fp);
  else
    eval_kind(F->kind, d, a0, a1, F->s0, F->s1, F->s2, f, fp);
}

/* distances d_k = ||X[i_k] - X[j_k]||_2   [ref: problem.py:246-283, average_distortion.py:38-45] */
void oracle_distances(int64_t n, int64_t p, const int64_t* edges, const float* X, int32_t dim,
                      float* dist) {
  (void)n;
#pragma omp parallel for schedule(static)
  for (int64_t k = 0; k < p; ++k) {
    const float* xi = X + edges[2 * k] * dim;
    const float* xj = X + edges[2 * k + 1] * dim;
    float ss = 0.f;
    for (int c = 0; c < dim; ++c) {
      const float df = xi[c] - xj[c];
      ss += df * df;
    }
    dist[k] = sqrtf(ss);
  }
}

/* per-edge distortions f_k(dist_k)   [ref: problem.py:285-308] */
void oracle_distortions(int64_t p, const float* dist, const oracle_func* F, float* out) {
#pragma omp parallel for schedule(static)
  for (int64_t k = 0; k < p; ++k) {
    float f, fp;
    eval_func(F, k, dist[k], &f, &fp);
    out[k] = f;
  }
}

/* E(X) and dE/dX.  grad may be NULL (forward only).  Returns E. */
double oracle_average_distortion(int64_t n, int64_t p, const int64_t* edges, const float* X,
                                 int32_t dim, const oracle_func* F, float grad_output, float* grad) {
  double total = 0.0;
  int nthreads = 1;
#ifdef _OPENMP
  nthreads = omp_get_max_threads();
#endif
  double* acc = NULL; /* per-thread gradient accumulators (scatter-add without atomics) */
  if (grad) acc = (double*)calloc((size_t)nthreads * (size_t)n * (size_t)dim, sizeof(double));
#pragma omp parallel reduction(+ : total)
  {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    double* my = acc ? acc + (size_t)tid * (size_t)n * (size_t)dim : NULL;
    float diff[4096];
#pragma omp for schedule(static)
    for (int64_t k = 0; k < p; ++k) {
      const int64_t i = edges[2 * k], j = edges[2 * k + 1];
      const float* xi = X + i * dim;
      const float* xj = X + j * dim;
      float ss = 0.f;
      for (int c = 0; c < dim; ++c) {
        diff[c] = xi[c] - xj[c];
        ss += diff[c] * diff[c];
      }
      const float d = sqrtf(ss);
      float f, fp;
      eval_func(F, k, d, &f, &fp);
      total += (double)f;
      if (my) {
        float g = (fp / (float)p) / d;
        if (isnan(g)) g = 1.0f; /* average_distortion.py:85-86 */
        if (isinf(g)) g = 1.0f; /* :87-88 */
        for (int c = 0; c < dim; ++c) {
          const double v = (double)(g * diff[c]);
          my[i * dim + c] += v;
          my[j * dim + c] -= v;
        }
      }
    }
  }
  if (grad) {
    const int64_t N = n * (int64_t)dim;
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < N; ++q) {
      double s = 0.0;
      for (int t = 0; t < nthreads; ++t) s += acc[(size_t)t * (size_t)N + q];
      grad[q] = (float)(s * (double)grad_output);
    }
    free(acc);
  }
  return p > 0 ? total / (double)p : 0.0;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void oracle_set_num_threads(int t) {
#ifdef _OPENMP
  omp_set_num_threads(t);
#else
  (void)t;
#endif
}
