// streamprobe.hip -- how fast can W waves per CU read a read-once stream from HBM, as a function of
// W, of the number of 1 KB loads each wave keeps in flight, and of the layout (every wave its own
// contiguous piece, or block b of every wave side by side)?  Design probe for the consumer side of
// the LDS-ring kernel (pymde_amd/csrc/mde_ring.hip), which reads 2 KB + 64 B per four iterations
// per wave with 10 waves per CU.
//   hipcc --offload-arch=gfx950 -O3 -o tools/streamprobe tools/streamprobe.hip && ./tools/streamprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int P, bool INTERLEAVED, int LDSKB>
__global__ void k_read(const u4* __restrict__ src, size_t blocks_per_wave, unsigned* out) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const size_t gw = (size_t)blockIdx.x * nw + wave, total_w = (size_t)gridDim.x * nw;
  u4 q[P];
  auto at = [&](size_t b) { return INTERLEAVED ? (b * total_w + gw) * 64 + lane : (gw * blocks_per_wave + b) * 64 + lane; };
#pragma unroll
  for (int u = 0; u < P; ++u) q[u] = src[at(u)];
  unsigned acc = 0;
  for (size_t b = 0; b < blocks_per_wave; b += P) {
#pragma unroll
    for (int u = 0; u < P; ++u) {
      acc ^= q[u].x ^ q[u].y ^ q[u].z ^ q[u].w;
      const size_t nb = b + u + P;
      q[u] = src[at(nb < blocks_per_wave ? nb : blocks_per_wave - 1)];
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
  if (LDSKB && threadIdx.x == 0) lds[0] = 1;
}

template <int P, bool I>
static void run(const u4* src, size_t bytes, int grid, int block, int ldskb, unsigned* out) {
  const size_t waves = (size_t)grid * (block / 64);
  size_t bpw = bytes / 1024 / waves;
  bpw -= bpw % P;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(a);
    if (ldskb)
      hipLaunchKernelGGL((k_read<P, I, 1>), dim3(grid), dim3(block), ldskb * 1024, 0, src, bpw, out);
    else
      hipLaunchKernelGGL((k_read<P, I, 0>), dim3(grid), dim3(block), 0, 0, src, bpw, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  const double gb = (double)bpw * waves * 1024 / 1e9;
  printf("grid %5d x %4d threads (%2d waves/WG, LDS %3d KB)  %d x 1 KB in flight per wave  %-12s  %.3f ms  %.2f TB/s\n", grid,
         block, block / 64, ldskb, P, I ? "block-major" : "wave-major", best, gb / best);
}

int main() {
  const size_t bytes = 460ull << 20;
  u4* src;
  unsigned* out;
  hipMalloc(&src, bytes + (1 << 20));
  hipMalloc(&out, 64);
  hipMemset(src, 1, bytes + (1 << 20));
  // the ring kernel's shape: 253 workgroups, one per CU (160 KB of LDS), 10 reading waves
  run<3, false>(src, bytes, 253, 640, 150, out);
  run<3, true>(src, bytes, 253, 640, 150, out);
  run<6, false>(src, bytes, 253, 640, 150, out);
  run<12, false>(src, bytes, 253, 640, 150, out);
  run<3, false>(src, bytes, 253, 1024, 150, out);
  run<6, false>(src, bytes, 253, 1024, 150, out);
  run<6, true>(src, bytes, 253, 1024, 150, out);
  run<6, false>(src, bytes, 256, 1024, 150, out);
  // without the LDS allocation: more workgroups per CU
  run<3, false>(src, bytes, 506, 640, 0, out);
  run<3, false>(src, bytes, 1024, 512, 0, out);
  run<3, true>(src, bytes, 1024, 512, 0, out);
  run<6, true>(src, bytes, 2048, 256, 0, out);
  run<2, true>(src, bytes, 4096, 256, 0, out);
  return 0;
}
