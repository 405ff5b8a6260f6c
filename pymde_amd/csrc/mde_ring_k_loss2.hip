// mde_ring_k_loss2.hip -- LDS-ring kernels of the remaining public losses with compile-time functors (round 5)
// [ref: pymde/functions/losses.py: Cubic 128-136, Power 139-148, Logistic 177-186, Fractional 189-200,
// SoftFractional 203-229].  The per-edge parameter is a target deviation: padding lanes are masked (LIN = false).
#include "mde_ring_kernel.h"

int mde_ring_launch_loss2(const RingArgs& A, const mde_func* f, int* nblocks) {
#ifdef MDE_RING_MINIMAL
  return 0;
#else
  if ((A.d != 2 && A.d != 3) || f->kind_neg != MDE_F_NONE) return 0;
  const MdeFuncArgs a = ring_func_args(f);
  switch (f->kind) {
    case MDE_F_L_CUBIC: MDE_RING23(FnSingle<MDE_F_L_CUBIC COMMA 0>, false);
    case MDE_F_L_POWER: MDE_RING23(FnSingle<MDE_F_L_POWER COMMA 0>, false);
    case MDE_F_L_LOGISTIC: MDE_RING23(FnSingle<MDE_F_L_LOGISTIC COMMA 0>, false);
    case MDE_F_L_FRACTIONAL: MDE_RING23(FnSingle<MDE_F_L_FRACTIONAL COMMA 0>, false);
    case MDE_F_L_SOFT_FRACTIONAL: MDE_RING23(FnSingle<MDE_F_L_SOFT_FRACTIONAL COMMA 0>, false);
    default: return 0;
  }
#endif
}
