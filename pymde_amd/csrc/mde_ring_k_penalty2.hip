// mde_ring_k_penalty2.hip -- LDS-ring kernels of the remaining public penalties with compile-time functors
// (round 5; the run-time functor costs 2.5x on this instruction-bound kernel)
// [ref: pymde/functions/penalties.py: Power 191-202, Logistic 246-266, Sigmoid 269-283, Hinge 286-307,
// InvPower 340-353, LogRatio 356-369].
#include "mde_ring_kernel.h"

int mde_ring_launch_penalty2(const RingArgs& A, const mde_func* f, int* nblocks) {
#ifdef MDE_RING_MINIMAL
  return 0;
#else
  if ((A.d != 2 && A.d != 3) || f->kind_neg != MDE_F_NONE) return 0;
  const MdeFuncArgs a = ring_func_args(f);
  const int ea = mde_exp_class(f->s0);
  switch (f->kind) {
    // w = 0 gives f = 0 whatever the distance (softplus, sigmoid and the hinge are finite): LIN
    case MDE_F_LOGISTIC: MDE_RING23(FnSingle<MDE_F_LOGISTIC COMMA 0>, true);
    case MDE_F_SIGMOID: MDE_RING23(FnSingle<MDE_F_SIGMOID COMMA 0>, true);
    case MDE_F_HINGE: MDE_RING23(FnSingle<MDE_F_HINGE COMMA 0>, true);
    case MDE_F_POWER:  // (a negative exponent: 0 x inf on a padding lane that sits on its own column -- masked)
      if (ea == 2) MDE_RING23(FnSingle<MDE_F_POWER COMMA 2>, false);
      if (ea == 5) MDE_RING23(FnSingle<MDE_F_POWER COMMA 5>, false);
      MDE_RING23(FnSingle<MDE_F_POWER COMMA 0>, false);  // run-time exponent, compile-time kind
    case MDE_F_INVPOWER:
      if (ea == 1) MDE_RING23(FnSingle<MDE_F_INVPOWER COMMA 1>, false);
      MDE_RING23(FnSingle<MDE_F_INVPOWER COMMA 0>, false);
    case MDE_F_LOGRATIO:
      if (ea == 3) MDE_RING23(FnSingle<MDE_F_LOGRATIO COMMA 3>, false);
      MDE_RING23(FnSingle<MDE_F_LOGRATIO COMMA 0>, false);
    default: return 0;
  }
#endif
}
