// mde_ring_k_runtime.hip -- LDS-ring kernels with the run-time functor: every distortion function the
// library knows, at d = 1 .. 4 (~2.5x the time per evaluation of a compile-time functor at config 4).
#include "mde_ring_kernel.h"

int mde_ring_launch_runtime(const RingArgs& A, const mde_func* f, int* nblocks) {
#ifdef MDE_RING_MINIMAL
  return 0;
#else
  FnRuntime fn{ring_func_args(f)};
  int rc;
  switch (A.d) {
    case 1: rc = launch_ring<1, FnRuntime, false>(A, fn, nblocks); break;
    case 2: rc = launch_ring<2, FnRuntime, false>(A, fn, nblocks); break;
    case 3: rc = launch_ring<3, FnRuntime, false>(A, fn, nblocks); break;
    case 4: rc = launch_ring<4, FnRuntime, false>(A, fn, nblocks); break;
    default: return 0;
  }
  return rc == MDE_OK ? 1 : rc;
#endif
}
