// mde_ring_k_pushpull.hip -- LDS-ring kernels of PushAndPull(Log1p, Log | LogRatio), what
// pymde.preserve_neighbors builds when it adds dissimilar pairs [ref: pymde/functions/penalties.py:66-109,
// pymde/recipes.py:395-400].
#include "mde_ring_kernel.h"

int mde_ring_launch_pushpull(const RingArgs& A, const mde_func* f, int* nblocks) {
#ifdef MDE_RING_MINIMAL
  return 0;
#else
  if ((A.d != 2 && A.d != 3) || f->kind != MDE_F_LOG1P || mde_exp_class(f->s0) != 2) return 0;
  const MdeFuncArgs a = ring_func_args(f);
  const int en = mde_exp_class(f->n0);
  if (f->kind_neg == MDE_F_LOG && en == 1) MDE_RING23(FnPushPull<MDE_F_LOG1P COMMA 2 COMMA MDE_F_LOG COMMA 1>, true);
  if (f->kind_neg == MDE_F_LOGRATIO && en == 3)
    MDE_RING23(FnPushPull<MDE_F_LOG1P COMMA 2 COMMA MDE_F_LOGRATIO COMMA 3>, true);
  return 0;
#endif
}
