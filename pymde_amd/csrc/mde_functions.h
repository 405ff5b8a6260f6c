// mde_functions.h -- device-side distortion functions f_k(d) and f'_k(d)/d.
//
// Each kind restates one class of the reference's penalty / loss surface
// [ref: pymde/functions/penalties.py:112-400, pymde/functions/losses.py:61-239]; the
// derivative is the closed form of what torch autograd produces there.  Everything is
// fp32, built from the gfx950 transcendental instructions (v_sqrt/v_rcp/v_log/v_exp, 1 ulp)
// with cancellation-safe forms for log1p / expm1.
//
//   eval<KIND, ECLS>(ss, a0, a1, S, f, gd)
//     ss  = squared distance, a0/a1 per-edge parameters, S scalars
//     f   = f_k(d)                    (d = sqrt(ss))
//     gd  = f'_k(d) / d               (may be NaN/Inf at d = 0; the caller applies the
//                                      reference's NaN->1, Inf->1 rule, average_distortion.py:81-88)
//   ECLS is a compile-time hint for the exponent: 0 = read S.s0 at run time,
//   1 -> e=1, 2 -> e=1.5, 3 -> e=2, 4 -> e=3, 5 -> e=0.5.
#pragma once
#include "mde_common.h"

struct MdeScalars {
  float s0, s1, s2;
};

#define MDE_DEV __device__ __forceinline__

MDE_DEV float mde_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
MDE_DEV float mde_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
MDE_DEV float mde_log2(float x) { return __builtin_amdgcn_logf(x); }
MDE_DEV float mde_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
MDE_DEV float mde_log(float x) { return mde_log2(x) * 0.6931471805599453f; }
MDE_DEV float mde_exp(float x) { return mde_exp2(x * 1.4426950408889634f); }

// log(1+u) for u > -0.5: log(t) + (u - (t - 1))/t with t = fl(1+u).  The second term restores the
// rounding error of 1+u; measured on gfx950 (tools/mathprobe.hip) the maximum relative error over
// u in [1e-7, 1e4] is 1.9e-7, identical to a series/compensated hybrid.
MDE_DEV float mde_log1p(float u) {
  const float t = 1.0f + u;
  const float c = u - (t - 1.0f);  // rounding error of 1+u
  return fmaf(c, mde_rcp(t), mde_log(t));
}
// 1 - exp(-u) for u >= 0 and exp(-u)
MDE_DEV float mde_one_minus_expneg(float u, float& em) {
  em = mde_exp(-u);
  const float direct = 1.0f - em;
  const float ser = u * fmaf(u, fmaf(u, fmaf(u, fmaf(u, fmaf(u, -1.0f / 720.0f, 1.0f / 120.0f), -1.0f / 24.0f),
                                               1.0f / 6.0f),
                                     -0.5f),
                             1.0f);
  return u < 0.0625f ? ser : direct;
}
// numerically stable log(1 + exp(z)) and sigmoid(z)
MDE_DEV float mde_softplus(float z) { return fmaxf(z, 0.0f) + mde_log1p(mde_exp(-fabsf(z))); }
MDE_DEV float mde_sigmoid(float z) { return mde_rcp(1.0f + mde_exp(-z)); }
MDE_DEV float mde_sign(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }

// x^e for x >= 0.  e = 0 -> 1 (torch.pow convention).
template <int ECLS>
MDE_DEV float mde_pow(float x, float e) {
  if constexpr (ECLS == 1) return x;
  if constexpr (ECLS == 2) return x * mde_sqrt(x);
  if constexpr (ECLS == 3) return x * x;
  if constexpr (ECLS == 4) return x * x * x;
  if constexpr (ECLS == 5) return mde_sqrt(x);
  // run-time exponent; the comparisons are wave-uniform (scalar branches)
  if (e == 1.0f) return x;
  if (e == 2.0f) return x * x;
  if (e == 1.5f) return x * mde_sqrt(x);
  if (e == 3.0f) return x * x * x;
  if (e == 0.5f) return mde_sqrt(x);
  if (e == 0.0f) return 1.0f;
  return mde_exp2(e * mde_log2(x));
}
template <int ECLS>
MDE_DEV float mde_expo(float e) {
  if constexpr (ECLS == 1) return 1.0f;
  if constexpr (ECLS == 2) return 1.5f;
  if constexpr (ECLS == 3) return 2.0f;
  if constexpr (ECLS == 4) return 3.0f;
  if constexpr (ECLS == 5) return 0.5f;
  return e;
}

template <int KIND, int ECLS>
MDE_DEV void mde_eval(float ss, float a0, float a1, const MdeScalars& S, float& f, float& gd) {
  // ---------------------------------------------------------------- penalties (a0 = w)
  if constexpr (KIND == MDE_F_LINEAR) {  // penalties.py:112-120
    const float d = mde_sqrt(ss);
    f = a0 * d;
    gd = a0 * mde_rcp(d);
  } else if constexpr (KIND == MDE_F_QUADRATIC) {  // penalties.py:123-131
    f = a0 * ss;
    gd = 2.0f * a0;
  } else if constexpr (KIND == MDE_F_CUBIC) {  // penalties.py:163-171
    const float d = mde_sqrt(ss);
    f = a0 * ss * d;
    gd = 3.0f * a0 * d;
  } else if constexpr (KIND == MDE_F_POWER) {  // penalties.py:191-202
    const float d = mde_sqrt(ss);
    const float e = mde_expo<ECLS>(S.s0);
    const float pe = mde_pow<ECLS>(d, e);
    f = a0 * pe;
    gd = a0 * e * pe * mde_rcp(ss);
  } else if constexpr (KIND == MDE_F_HUBER) {  // penalties.py:205-243 (strict <)
    const float d = mde_sqrt(ss);
    const float t = S.s0;
    const bool lt = d < t;
    f = lt ? a0 * 0.5f * ss : a0 * t * (d - 0.5f * t);
    gd = lt ? a0 : a0 * t * mde_rcp(d);
  } else if constexpr (KIND == MDE_F_LOGISTIC) {  // penalties.py:246-266
    const float d = mde_sqrt(ss);
    const float z = S.s1 * (d - S.s0);
    f = a0 * mde_softplus(z);
    gd = a0 * S.s1 * mde_sigmoid(z) * mde_rcp(d);
  } else if constexpr (KIND == MDE_F_SIGMOID) {  // penalties.py:269-283
    const float d = mde_sqrt(ss);
    const float sg = mde_sigmoid(S.s1 * (d - S.s0));
    f = a0 * sg;
    gd = a0 * S.s1 * sg * (1.0f - sg) * mde_rcp(d);
  } else if constexpr (KIND == MDE_F_HINGE) {  // penalties.py:286-307
    const float d = mde_sqrt(ss);
    const float v = a0 * (d - (S.s0 - mde_sign(a0) * S.s1));
    f = fmaxf(0.0f, v);
    const float sub = (v > 0.0f) ? 1.0f : ((v == 0.0f) ? 0.5f : 0.0f);  // torch.max tie -> 1/2
    gd = sub * a0 * mde_rcp(d);
  } else if constexpr (KIND == MDE_F_LOG1P && ECLS == 2) {  // penalties.py:310-321, exponent 1.5
    // d^1.5 = d sqrt(d) and f'/d = 1.5 w / (sqrt(d) (1 + d^1.5)): two square roots and ONE
    // reciprocal, r = 1 / (sqrt(d) t + tiny).  The tiny term (far below one ulp of the product for
    // every d >= 1e-44) keeps r finite at d = 0, where the reference's NaN/Inf -> 1 rule only ever
    // multiplies x_v - x_u = 0: gd needs no fix-up here.  log1p(u) = log t + (u - (t - 1)) / t with
    // 1 / t = sqrt(d) r.
    const float d = mde_sqrt(ss);
    const float sd = mde_sqrt(d);
    const float pe = d * sd;
    const float t = 1.0f + pe;
    const float r = mde_rcp(fmaf(sd, t, 1.0e-30f));
    f = a0 * fmaf((pe - (t - 1.0f)) * sd, r, mde_log(t));
    gd = (1.5f * a0) * r;
  } else if constexpr (KIND == MDE_F_LOG1P) {  // penalties.py:310-321
    const float d = mde_sqrt(ss);
    const float e = mde_expo<ECLS>(S.s0);
    const float pe = mde_pow<ECLS>(d, e);
    f = a0 * mde_log1p(pe);
    // f' / d = w e d^(e-2) / (1 + d^e)
    gd = a0 * e * pe * mde_rcp(ss * (1.0f + pe));
  } else if constexpr (KIND == MDE_F_LOG) {  // penalties.py:324-337
    const float d = mde_sqrt(ss);
    const float e = mde_expo<ECLS>(S.s0);
    const float u = mde_pow<ECLS>(d, e);
    float em;
    const float A = mde_one_minus_expneg(u, em);  // -expm1(-u)
    f = a0 * ((u > 1.0f) ? mde_log1p(-em) : mde_log(A));
    // d/du log(1-exp(-u)) = exp(-u)/(1-exp(-u)) = 1/expm1(u)
    gd = a0 * e * u * em * mde_rcp(A * ss);
  } else if constexpr (KIND == MDE_F_INVPOWER) {  // penalties.py:340-353 (uses |w|)
    const float d = mde_sqrt(ss);
    const float e = mde_expo<ECLS>(S.s0);
    const float pe = mde_pow<ECLS>(d, e);
    const float aw = fabsf(a0);
    const float ip = mde_rcp(pe);
    f = aw * ip;
    gd = -aw * e * ip * mde_rcp(ss);
  } else if constexpr (KIND == MDE_F_LOGRATIO) {  // penalties.py:356-369
    const float d = mde_sqrt(ss);
    const float e = mde_expo<ECLS>(S.s0);
    const float pe = mde_pow<ECLS>(d, e);
    f = -a0 * mde_log1p(mde_rcp(pe));  // log(pe/(1+pe)) = -log1p(1/pe)
    gd = a0 * e * mde_rcp(ss * (1.0f + pe));
  } else if constexpr (KIND == MDE_F_DEADZONE_QUADRATIC) {  // penalties.py:134-148
    const float d = mde_sqrt(ss);
    const bool lt = d < S.s0;
    f = lt ? 0.0f * a0 : a0 * ss;
    gd = lt ? 0.0f * a0 : 2.0f * a0;
  } else if constexpr (KIND == MDE_F_DEADZONE_CUBIC) {  // penalties.py:174-188
    const float d = mde_sqrt(ss);
    const bool lt = d < S.s0;
    f = lt ? 0.0f * a0 : a0 * ss * d;
    gd = lt ? 0.0f * a0 : 3.0f * a0 * d;
  } else if constexpr (KIND == MDE_F_CLIPPED_QUADRATIC) {  // penalties.py:151-160
    const float c = (S.s0 + 1.0f) * (S.s0 + 1.0f);
    f = a0 * fminf(ss, c);
    gd = (ss < c) ? 2.0f * a0 : ((ss == c) ? a0 : 0.0f);
    // ---------------------------------------------------------------- losses (a0 = delta)
  } else if constexpr (KIND == MDE_F_L_QUADRATIC) {  // losses.py:61-69
    const float d = mde_sqrt(ss);
    const float r = a0 - d;
    f = r * r;
    gd = -2.0f * r * mde_rcp(d);
  } else if constexpr (KIND == MDE_F_L_WEIGHTED_QUADRATIC) {  // losses.py:72-87
    const float d = mde_sqrt(ss);
    const float r = a0 - d;
    f = a1 * r * r;
    gd = -2.0f * a1 * r * mde_rcp(d);
  } else if constexpr (KIND == MDE_F_L_CLIPPED_QUADRATIC) {  // losses.py:90-98
    const float d = mde_sqrt(ss);
    const float r = a0 - d;
    const float c = (S.s0 + 1.0f) * (S.s0 + 1.0f);
    const float r2 = r * r;
    f = fminf(r2, c);
    gd = ((r2 < c) ? 1.0f : ((r2 == c) ? 0.5f : 0.0f)) * -2.0f * r * mde_rcp(d);
  } else if constexpr (KIND == MDE_F_L_HUBER) {  // losses.py:101-125 (strict <)
    const float d = mde_sqrt(ss);
    const float r = fabsf(a0 - d);
    const float sg = mde_sign(d - a0);
    const float t = S.s0;
    const bool lt = r < t;
    f = lt ? r * r : t * (2.0f * r - t);
    gd = (lt ? 2.0f * r : 2.0f * t) * sg * mde_rcp(d);
  } else if constexpr (KIND == MDE_F_L_CUBIC) {  // losses.py:128-136
    const float d = mde_sqrt(ss);
    const float r = fabsf(a0 - d);
    f = r * r * r;
    gd = 3.0f * r * r * mde_sign(d - a0) * mde_rcp(d);
  } else if constexpr (KIND == MDE_F_L_POWER) {  // losses.py:139-148
    const float d = mde_sqrt(ss);
    const float r = fabsf(a0 - d);
    const float e = mde_expo<ECLS>(S.s0);
    const float pe = mde_pow<ECLS>(r, e);
    f = pe;
    // e r^(e-1) sign(d - delta); at r == 0 autograd gives e*0^(e-1)*sign(0) = 0 for e >= 1
    const float rp = (r > 0.0f) ? pe * mde_rcp(r) : ((e >= 1.0f) ? 0.0f : __builtin_nanf(""));
    gd = e * rp * mde_sign(d - a0) * mde_rcp(d);
  } else if constexpr (KIND == MDE_F_L_WEIGHTED_POWER) {  // losses.py:151-163
    const float d = mde_sqrt(ss);
    const float r = fabsf(a0 - d);
    const float e = mde_expo<ECLS>(S.s0);
    const float pe = mde_pow<ECLS>(r, e);
    f = a1 * pe;
    const float rp = (r > 0.0f) ? pe * mde_rcp(r) : ((e >= 1.0f) ? 0.0f : __builtin_nanf(""));
    gd = a1 * e * rp * mde_sign(d - a0) * mde_rcp(d);
  } else if constexpr (KIND == MDE_F_L_ABSOLUTE) {  // losses.py:166-174
    const float d = mde_sqrt(ss);
    f = fabsf(a0 - d);
    gd = mde_sign(d - a0) * mde_rcp(d);
  } else if constexpr (KIND == MDE_F_L_LOGISTIC) {  // losses.py:177-186
    const float d = mde_sqrt(ss);
    const float r = fabsf(a0 - d);
    f = mde_softplus(r);
    gd = mde_sigmoid(r) * mde_sign(d - a0) * mde_rcp(d);
  } else if constexpr (KIND == MDE_F_L_FRACTIONAL) {  // losses.py:189-200
    const float d = mde_sqrt(ss);
    const float q1 = a0 * mde_rcp(d), q2 = d * mde_rcp(a0);
    f = fmaxf(q1, q2) - 1.0f;
    const float g1 = -a0 * mde_rcp(ss), g2 = mde_rcp(a0);
    const float fp = (q1 > q2) ? g1 : ((q1 < q2) ? g2 : 0.5f * (g1 + g2));
    gd = fp * mde_rcp(d);
  } else if constexpr (KIND == MDE_F_L_SOFT_FRACTIONAL) {  // losses.py:203-229
    const float d = mde_sqrt(ss);
    const float gm = S.s0;
    const float q1 = gm * a0 * mde_rcp(d), q2 = gm * d * mde_rcp(a0);
    const float mx = fmaxf(q1, q2), mn = fminf(q1, q2);
    const float ex = mde_exp(mn - mx);
    f = mde_rcp(gm) * (mx + mde_log1p(ex) - (0.6931471805599453f + gm));
    const float wmx = mde_rcp(1.0f + ex), wmn = ex * wmx;  // softmax weights
    const float w1 = (q1 >= q2) ? wmx : wmn, w2 = (q1 >= q2) ? wmn : wmx;
    const float fp = w1 * (-a0 * mde_rcp(ss)) + w2 * mde_rcp(a0);
    gd = fp * mde_rcp(d);
  } else if constexpr (KIND == MDE_F_L_LOG1P) {  // losses.py:232-239  log(1 + (d - delta)^e)
    const float d = mde_sqrt(ss);
    const float r = d - a0;
    const float e = S.s0;
    // torch.pow: a negative base is fine for an integer exponent, NaN otherwise
    const bool e_int = (e == floorf(e));
    const bool e_odd = e_int && (fmodf(fabsf(e), 2.0f) == 1.0f);
    const float ar = fabsf(r);
    float pw = mde_pow<0>(ar, e), pm1 = mde_pow<0>(ar, e - 1.0f);  // |r|^e, |r|^(e-1)
    if (r < 0.0f) {
      pw = e_int ? (e_odd ? -pw : pw) : __builtin_nanf("");
      pm1 = e_int ? (e_odd ? pm1 : -pm1) : __builtin_nanf("");
    }
    const float t = 1.0f + pw;
    f = mde_log(t);
    gd = e * pm1 * mde_rcp(t) * mde_rcp(d);
  } else {
    f = 0.0f;
    gd = 0.0f;
  }
}

// run-time dispatch over every kind (used by the universal kernels and by PushAndPull
// combinations that have no dedicated instantiation)
MDE_DEV void mde_eval_rt(int kind, float ss, float a0, float a1, const MdeScalars& S, float& f,
                         float& gd) {
  switch (kind) {
#define MDE_CASE(K)                        \
  case K:                                  \
    mde_eval<K, 0>(ss, a0, a1, S, f, gd);  \
    break;
    MDE_CASE(MDE_F_LINEAR)
    MDE_CASE(MDE_F_QUADRATIC)
    MDE_CASE(MDE_F_CUBIC)
    MDE_CASE(MDE_F_POWER)
    MDE_CASE(MDE_F_HUBER)
    MDE_CASE(MDE_F_LOGISTIC)
    MDE_CASE(MDE_F_SIGMOID)
    MDE_CASE(MDE_F_HINGE)
    MDE_CASE(MDE_F_LOG1P)
    MDE_CASE(MDE_F_LOG)
    MDE_CASE(MDE_F_INVPOWER)
    MDE_CASE(MDE_F_LOGRATIO)
    MDE_CASE(MDE_F_DEADZONE_QUADRATIC)
    MDE_CASE(MDE_F_DEADZONE_CUBIC)
    MDE_CASE(MDE_F_CLIPPED_QUADRATIC)
    MDE_CASE(MDE_F_L_QUADRATIC)
    MDE_CASE(MDE_F_L_WEIGHTED_QUADRATIC)
    MDE_CASE(MDE_F_L_HUBER)
    MDE_CASE(MDE_F_L_CUBIC)
    MDE_CASE(MDE_F_L_POWER)
    MDE_CASE(MDE_F_L_WEIGHTED_POWER)
    MDE_CASE(MDE_F_L_ABSOLUTE)
    MDE_CASE(MDE_F_L_LOGISTIC)
    MDE_CASE(MDE_F_L_FRACTIONAL)
    MDE_CASE(MDE_F_L_SOFT_FRACTIONAL)
    MDE_CASE(MDE_F_L_CLIPPED_QUADRATIC)
    MDE_CASE(MDE_F_L_LOG1P)
#undef MDE_CASE
    default:
      f = 0.0f;
      gd = 0.0f;
  }
}

static inline bool mde_kind_valid(int kind) {
  return (kind >= MDE_F_LINEAR && kind <= MDE_F_CLIPPED_QUADRATIC) ||
         (kind >= MDE_F_L_QUADRATIC && kind <= MDE_F_L_LOG1P);
}

// ---------------------------------------------------------------- functors used by the kernels
// A functor carries the scalars and evaluates one edge.  `Fn::eval(ss, a0, a1, f, gd)`.
struct MdeFuncArgs {
  int kind, kind_neg;
  MdeScalars S, N;
};

// universal: every kind, PushAndPull by sign of the weight (penalties.py:390: w >= 0 attractive)
// kFiniteG: f'/d is finite for every finite parameter and distance (no NaN/Inf fix-up needed when
// the caller knows the parameters are finite)
// kRingFused: the LDS-ring kernel spells the evaluation out itself (mde_ring.hip: Log1p with exponent
// 1.5, parameters pre-multiplied by kParamScale) instead of calling eval()
struct FnRuntime {
  static constexpr bool kFiniteG = false;
  static constexpr bool kRingFused = false;
  static constexpr float kParamScale = 1.0f;
  MdeFuncArgs A;
  MDE_DEV void eval(float ss, float a0, float a1, float& f, float& gd) const {
    if (A.kind_neg != MDE_F_NONE && a0 < 0.0f)
      mde_eval_rt(A.kind_neg, ss, a0, a1, A.N, f, gd);
    else
      mde_eval_rt(A.kind, ss, a0, a1, A.S, f, gd);
  }
};
// one kind, exponent class fixed at compile time
template <int KIND, int ECLS>
struct FnSingle {
  static constexpr bool kFiniteG = (KIND == MDE_F_LOG1P && ECLS == 2) || KIND == MDE_F_QUADRATIC;
  static constexpr bool kRingFused = (KIND == MDE_F_LOG1P && ECLS == 2);
  static constexpr float kParamScale = kRingFused ? 1.5f : 1.0f;
  MdeFuncArgs A;
  MDE_DEV void eval(float ss, float a0, float a1, float& f, float& gd) const {
    mde_eval<KIND, ECLS>(ss, a0, a1, A.S, f, gd);
  }
};
// PushAndPull with both branches fixed at compile time
template <int KA, int EA, int KR, int ER>
struct FnPushPull {
  static constexpr bool kMerged =
#ifndef MDE_PUSHPULL_DIVERGENT
      (KA == MDE_F_LOG1P && EA == 2 && KR == MDE_F_LOG && ER == 1);
#else
      false;
#endif
  // (the merged pair keeps f'/d finite at d = 0 the way Log1p does: a term far below one ulp of the
  // reciprocal's argument on the attractive side, a clamp at 2e-38 (the smallest normal numbers) on the repulsive one (exact for every
  // d >= 2e-13); at d = 0 the reference's NaN / Inf -> 1 rule multiplies
  // x_v - x_u = 0 and so does this 0)
  static constexpr bool kFiniteG = kMerged;
  static constexpr bool kRingFused = false;
  static constexpr float kParamScale = 1.0f;
  MdeFuncArgs A;
  MDE_DEV void eval(float ss, float a0, float a1, float& f, float& gd) const {
    if constexpr (kMerged) {
      // preserve_neighbors' pair (Log1p with exponent 1.5 pulls, Log with exponent 1 pushes) in ONE
      // instruction stream: a wave of a random graph holds both signs, so the divergent if / else below
      // issues both branches -- 6 quarter-rate instructions per entry for the gradient, 10 with the loss
      // term.  Here the branches share sqrt(ss) and meet in ONE reciprocal and ONE logarithm whose
      // ARGUMENTS are selected per lane: 4 and 5.  Attractive lanes compute exactly what
      // mde_eval<LOG1P, 2> does; repulsive lanes what mde_eval<LOG, 1> does except that the 1 / (1 - em)
      // of its log1p correction (|correction| <= 3e-8, d > 1 only) is the series 1 + em.
      // [ref: penalties.py:390 -- zero weight is attractive]  The per-lane choices are BIT selects on the sign
      // of the weight (v_bfi_b32 with an arithmetic-shift mask) instead of v_cmp + v_cndmask (round 5: config 4b
      // 0.2433 -> 0.2402 ms; the shorter series of 1 - exp(-d) took it from 0.2515 to 0.2433).
      // (-0.0 counts as repulsive here: its terms are -0 x finite = 0 either way)
      const uint32_t rep = (uint32_t)(__float_as_int(a0) >> 31);  // all ones for a repulsive (negative) weight
      // (inline asm: written as (rep_v & m) | (att_v & ~m) the compiler turns it back into v_cmp + v_cndmask)
      auto bfi = [&](uint32_t m, float set_v, float clr_v) __attribute__((always_inline)) {
        float out;
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(out) : "v"(m), "v"(set_v), "v"(clr_v));
        return out;
      };
      auto sel = [&](float att_v, float rep_v) __attribute__((always_inline)) { return bfi(rep, rep_v, att_v); };
      const float d = mde_sqrt(ss);
      const float sd = mde_sqrt(d);
      const float pe = d * sd;
      const float t = 1.0f + pe;
      // -expm1(-d): 1 - exp(-d), below d = 1/16 the series d (1 - d/2 + d^2/6 - d^3/24) (next term d^4/120: 1.3e-7)
      const float em = mde_exp(-d);
      const float ser = d * fmaf(d, fmaf(d, fmaf(d, -1.0f / 24.0f, 1.0f / 6.0f), -0.5f), 1.0f);
      // (a third bit select here -- mask from the sign of d - 1/16 -- measured no faster than v_cmp + v_cndmask: 0.2418 vs 0.2402 ms)
      const float om = d < 0.0625f ? ser : 1.0f - em;
      // (the repulsive argument (1 - exp(-d)) d^2 ~ d^3 is CLAMPED from below, not biased: a bias of 1e-36 put
      // f'/d off by a factor of two at d = 1e-12; the clamp is exact down to d ~ 2e-13 -- below that d^3 leaves
      // the normal range -- and keeps r finite at d = 0, where the term multiplies x_v - x_u = 0)
      const float r = mde_rcp(sel(fmaf(sd, t, 1.0e-30f), fmaxf(om * ss, 2.0e-38f)));
      gd = a0 * (sel(1.5f, d * em) * r);
      const float t2 = 1.0f - em;
      const float corr = sel((pe - (t - 1.0f)) * sd * r, d > 1.0f ? (-em - (t2 - 1.0f)) * (1.0f + em) : 0.0f);
      f = a0 * (mde_log(sel(t, om)) + corr);
      return;
    }
    // other pairs: a divergent if / else (both branches + a select measured SLOWER on the ring kernel:
    // config 4b 0.295 vs 0.267 ms, round 4)
    if (a0 >= 0.0f)  // [ref: penalties.py:390 -- zero weight is attractive]
      mde_eval<KA, EA>(ss, a0, a1, A.S, f, gd);
    else
      mde_eval<KR, ER>(ss, a0, a1, A.N, f, gd);
  }
};

MDE_DEV float mde_fix_g(float g) {
  // average_distortion.py:85-88: NaN -> 1.0, Inf -> 1.0
  return (fabsf(g) <= 3.402823466e+38f) ? g : 1.0f;
}
// the same rule for a g that is still to be multiplied by 1/p: `one` = p
MDE_DEV float mde_fix_g_to(float g, float one) {
  return (fabsf(g) <= 3.402823466e+38f) ? g : one;
}

static inline int mde_exp_class(float e) {
  if (e == 1.0f) return 1;
  if (e == 1.5f) return 2;
  if (e == 2.0f) return 3;
  if (e == 3.0f) return 4;
  if (e == 0.5f) return 5;
  return 0;
}
