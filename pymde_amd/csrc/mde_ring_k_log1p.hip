// mde_ring_k_log1p.hip -- LDS-ring kernels of the Log1p penalty, the attractive penalty of
// pymde.preserve_neighbors' default [ref: pymde/functions/penalties.py:310-321] and the headline workload.
#include "mde_ring_kernel.h"

int mde_ring_launch_log1p(const RingArgs& A, const mde_func* f, int* nblocks) {
  if ((A.d != 2 && A.d != 3) || f->kind_neg != MDE_F_NONE || f->kind != MDE_F_LOG1P) return 0;
  const MdeFuncArgs a = ring_func_args(f);
  // LIN: a padding lane's parameter 0 gives f = 0 whatever its distance is
  switch (mde_exp_class(f->s0)) {
    case 2: MDE_RING23(FnSingle<MDE_F_LOG1P COMMA 2>, true);
#ifndef MDE_RING_MINIMAL  // (design experiments: only the headline instantiation, everything else on the CSR kernels)
    case 1: MDE_RING23(FnSingle<MDE_F_LOG1P COMMA 1>, true);
    case 3: MDE_RING23(FnSingle<MDE_F_LOG1P COMMA 3>, true);
#endif
    default: return 0;
  }
}
