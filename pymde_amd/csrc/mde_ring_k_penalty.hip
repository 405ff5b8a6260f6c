// mde_ring_k_penalty.hip -- LDS-ring kernels of the other penalties with compile-time functors
// [ref: pymde/functions/penalties.py: Linear 112-120, Quadratic 123-131, Cubic 163-171, Huber 205-243,
// Log 324-337].
#include "mde_ring_kernel.h"

int mde_ring_launch_penalty(const RingArgs& A, const mde_func* f, int* nblocks) {
#ifdef MDE_RING_MINIMAL
  return 0;
#else
  if ((A.d != 2 && A.d != 3) || f->kind_neg != MDE_F_NONE) return 0;
  const MdeFuncArgs a = ring_func_args(f);
  const int ea = mde_exp_class(f->s0);
  switch (f->kind) {
    case MDE_F_QUADRATIC: MDE_RING23(FnSingle<MDE_F_QUADRATIC COMMA 0>, true);
    case MDE_F_LINEAR: MDE_RING23(FnSingle<MDE_F_LINEAR COMMA 0>, true);
    case MDE_F_CUBIC: MDE_RING23(FnSingle<MDE_F_CUBIC COMMA 0>, true);
    case MDE_F_HUBER: MDE_RING23(FnSingle<MDE_F_HUBER COMMA 0>, true);
    case MDE_F_LOG:  // w log(-expm1(-0)) = 0 x -inf on a padding lane: masked (LIN = false)
      if (ea == 1) MDE_RING23(FnSingle<MDE_F_LOG COMMA 1>, false);
      if (ea == 3) MDE_RING23(FnSingle<MDE_F_LOG COMMA 3>, false);
      return 0;
    default: return 0;
  }
#endif
}
