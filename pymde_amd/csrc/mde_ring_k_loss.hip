// mde_ring_k_loss.hip -- LDS-ring kernels of the losses pymde.preserve_distances uses
// [ref: pymde/functions/losses.py: Quadratic 61-69, WeightedQuadratic 72-87, Huber 101-125, Absolute
// 166-174].  The per-edge parameter is a target deviation, not a weight: padding lanes are masked
// (LIN = false).
#include "mde_ring_kernel.h"

int mde_ring_launch_loss(const RingArgs& A, const mde_func* f, int* nblocks) {
#ifdef MDE_RING_MINIMAL
  return 0;
#else
  if ((A.d != 2 && A.d != 3) || f->kind_neg != MDE_F_NONE) return 0;
  const MdeFuncArgs a = ring_func_args(f);
  switch (f->kind) {
    case MDE_F_L_QUADRATIC: MDE_RING23(FnSingle<MDE_F_L_QUADRATIC COMMA 0>, false);
    case MDE_F_L_WEIGHTED_QUADRATIC: MDE_RING23(FnSingle<MDE_F_L_WEIGHTED_QUADRATIC COMMA 0>, false);
    case MDE_F_L_ABSOLUTE: MDE_RING23(FnSingle<MDE_F_L_ABSOLUTE COMMA 0>, false);
    case MDE_F_L_HUBER: MDE_RING23(FnSingle<MDE_F_L_HUBER COMMA 0>, false);
    default: return 0;
  }
#endif
}
