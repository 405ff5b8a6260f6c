// dmaprobe.hip -- what slows the LDS-DMA ring fill when other waves of the workgroup are busy?
// (gfx950 design probe, round 3; not part of the product)
//   hipcc --offload-arch=gfx950 -O3 -o tools/dmaprobe tools/dmaprobe.hip && ./tools/dmaprobe
// One 1024-thread workgroup per CU.  P producer waves stream an L2-resident 8 MB table into a ring of
// 12 x 8 KB LDS slots with global_load_lds_dwordx4 (free running, DEPTH chunks in flight per producer,
// producers alternate whole chunks).  The other waves run ONE kind of instruction in a loop until the
// producers are done; the table gives the producers' time (shader clocks, slowest producer of any
// workgroup), clocks per 1 KiB piece per CU, and what the other waves got done meanwhile.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define BS 1024
#define RING_OFF 65536

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// MODE of the non-producer waves: 0 idle, 1 ds_read_b64 linear, 2 ds_read_b64 random, 3 ds_write_b64
// linear, 4 VALU fma, 5 global_load_dwordx4 stream, 6 ds_read_b32 poll + s_sleep, 7 transcendental,
// 8 the ring kernel's mix (3 reads + write + ~14 VALU + 3 transcendentals per trip)
template <int P, int DEPTH, int MODE>
__global__ __launch_bounds__(BS) void k_dma(const char* __restrict__ X, const f4* __restrict__ stream, int nchunks,
                                            int nother, unsigned long long* __restrict__ res, float* out) {
  __shared__ __attribute__((aligned(16))) char L[163840 - 256];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < RING_OFF / 4; i += BS) reinterpret_cast<float*>(L)[i] = 1.0f;
  int* done = reinterpret_cast<int*>(L + 65536 - 64);
  __syncthreads();
  if (tid == 0) *done = 0;
  __syncthreads();
  float accum = 0.f;
  if (wave >= 16 - P) {
    const int p = wave - (16 - P);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int j = p; j < nchunks; j += P) {
      const char* src = X + (size_t)j * 8192 + lane * 16;
      const uint32_t dst = RING_OFF + (uint32_t)(j % 12) * 8192u;
#pragma unroll
      for (int k = 0; k < 8; ++k) glds16(src + k * 1024, dst + k * 1024);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (DEPTH - 1)) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) {
      atomicMax(&res[0], t1 - t0);
      atomicAdd(done, 1);
    }
  } else if (MODE != 0 && wave < nother) {
    unsigned long long trips = 0;
    const unsigned addr = (unsigned)(lane * 8 + wave * 2048);
    unsigned h = tid * 2654435761u;
    const unsigned raddr = ((h >> 8) % 12288u) * 8u + RING_OFF;
    f2 v = {1.0f, 2.0f}, acc2 = {0.f, 0.f};
    float a0 = 1.0f + lane, a1 = 2.0f, a2 = 3.0f, a3 = 4.0f;
    size_t sidx = (size_t)blockIdx.x * 16384 + tid;
    volatile int* vd = done;
    for (;;) {
      if (MODE == 1 || MODE == 2) {
        f2 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("ds_read_b64 %0, %1" : "=v"(t[k]) : "v"(MODE == 1 ? addr : raddr));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 8; ++k) acc2 += t[k];
      } else if (MODE == 3) {
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else if (MODE == 4) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          a0 = fmaf(a0, 1.0000001f, 1e-9f); a1 = fmaf(a1, 1.0000001f, 1e-9f);
          a2 = fmaf(a2, 1.0000001f, 1e-9f); a3 = fmaf(a3, 1.0000001f, 1e-9f);
        }
      } else if (MODE == 5) {
        f4 t0 = stream[sidx], t1 = stream[sidx + 1024], t2 = stream[sidx + 2048], t3 = stream[sidx + 3072];
        sidx = (sidx + 4096) & ((1u << 24) - 1);
        a0 += t0.x + t1.y + t2.z + t3.w;
      } else if (MODE == 6) {
        int f;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(f) : "v"(65536u - 128u) : "memory");
        a0 += (float)f;
        __builtin_amdgcn_s_sleep(1);
      } else if (MODE == 7) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          a0 = __builtin_amdgcn_sqrtf(a0 + 1.0f); a1 = __builtin_amdgcn_rcpf(a1 + 1.0f);
        }
      } else if (MODE == 8) {
        f2 xv, xu, ac;
        asm volatile("ds_read_b64 %0, %1" : "=v"(xv) : "v"(addr));
        asm volatile("ds_read_b64 %0, %1" : "=v"(xu) : "v"(raddr));
        asm volatile("ds_read_b64 %0, %1 offset:32768\n\ts_waitcnt lgkmcnt(0)" : "=v"(ac) : "v"(addr));
        const float dx = xv.x - xu.x + a0, dy = xv.y - xu.y;
        const float ss = fmaf(dx, dx, dy * dy);
        const float d = __builtin_amdgcn_sqrtf(ss), sd = __builtin_amdgcn_sqrtf(d);
        const float t = fmaf(d, sd, 1.0f);
        const float r = __builtin_amdgcn_rcpf(fmaf(sd, t, 1e-30f));
        ac.x = fmaf(dx, r, ac.x); ac.y = fmaf(dy, r, ac.y);
        asm volatile("ds_write_b64 %0, %1 offset:32768" ::"v"(addr), "v"(ac) : "memory");
        a0 = a0 * 0.999f;
      }
      ++trips;
      if (*vd >= P) break;
    }
    accum = a0 + a1 + a2 + a3 + acc2.x + acc2.y;
    if (lane == 0) atomicAdd(&res[1], trips);
  }
  __syncthreads();
  float t = accum;
  for (int i = tid; i < 8192; i += BS) t += reinterpret_cast<float*>(L + RING_OFF)[i];
  if (t == 12345.678f) out[0] = t;
}

// VGPR-staged producers: NL x 1 KiB global loads in flight per producer wave, each landed piece is
// written to its ring slot with ds_write_b128 (the in-flight bytes need no ring slots)
template <int P, int NL, int MODE>
__global__ __launch_bounds__(BS) void k_vstage(const char* __restrict__ X, int nchunks, int nother,
                                               unsigned long long* __restrict__ res, float* out) {
  __shared__ __attribute__((aligned(16))) char L[163840 - 256];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < RING_OFF / 4; i += BS) reinterpret_cast<float*>(L)[i] = 1.0f;
  int* done = reinterpret_cast<int*>(L + 65536 - 64);
  __syncthreads();
  if (tid == 0) *done = 0;
  __syncthreads();
  float accum = 0.f;
  if (wave >= 16 - P) {
    const int p = wave - (16 - P);
    const unsigned long long t0 = __builtin_readcyclecounter();
    // pieces p, p + P, ... of the table (piece q -> ring byte (q * 1024) % 98304)
    const int npieces = nchunks * 8;
    f4 buf[NL];
    int q = p;
#pragma unroll
    for (int k = 0; k < NL; ++k) buf[k] = *reinterpret_cast<const f4*>(X + (size_t)min(q + k * P, npieces - 1) * 1024 + lane * 16);
    for (; q < npieces; q += P * NL) {
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        const int qq = q + k * P;
        const f4 v = buf[k];
        // refill this register with the piece NL turns ahead, then store the landed one
        buf[k] = *reinterpret_cast<const f4*>(X + (size_t)min(qq + P * NL, npieces - 1) * 1024 + lane * 16);
        if (qq < npieces) *reinterpret_cast<f4*>(L + RING_OFF + (size_t)((qq * 1024) % 98304) + lane * 16) = v;
      }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) {
      atomicMax(&res[0], t1 - t0);
      atomicAdd(done, 1);
    }
  } else if (MODE != 0 && wave < nother) {
    unsigned long long trips = 0;
    const unsigned addr = (unsigned)(lane * 8 + wave * 2048);
    unsigned h = tid * 2654435761u;
    const unsigned raddr = ((h >> 8) % 12288u) * 8u + RING_OFF;
    float a0 = 1.0f + lane;
    volatile int* vd = done;
    for (;;) {
      f2 xv, xu, ac;
      asm volatile("ds_read_b64 %0, %1" : "=v"(xv) : "v"(addr));
      asm volatile("ds_read_b64 %0, %1" : "=v"(xu) : "v"(raddr));
      asm volatile("ds_read_b64 %0, %1 offset:32768\n\ts_waitcnt lgkmcnt(0)" : "=v"(ac) : "v"(addr));
      const float dx = xv.x - xu.x + a0, dy = xv.y - xu.y;
      const float ss = fmaf(dx, dx, dy * dy);
      const float d = __builtin_amdgcn_sqrtf(ss), sd = __builtin_amdgcn_sqrtf(d);
      const float t = fmaf(d, sd, 1.0f);
      const float r = __builtin_amdgcn_rcpf(fmaf(sd, t, 1e-30f));
      ac.x = fmaf(dx, r, ac.x); ac.y = fmaf(dy, r, ac.y);
      asm volatile("ds_write_b64 %0, %1 offset:32768" ::"v"(addr), "v"(ac) : "memory");
      a0 = a0 * 0.999f;
      ++trips;
      if (*vd >= P) break;
    }
    accum = a0;
    if (lane == 0) atomicAdd(&res[1], trips);
  }
  __syncthreads();
  float t = accum;
  for (int i = tid; i < 8192; i += BS) t += reinterpret_cast<float*>(L + RING_OFF)[i];
  if (t == 12345.678f) out[0] = t;
}

template <int P, int NL, int MODE>
static void runv(const char* name, const char* X, int nother) {
  unsigned long long* res; float* out;
  CK(hipMalloc(&res, 64)); CK(hipMalloc(&out, 64));
  const int nchunks = (8 << 20) / 8192;
  hipLaunchKernelGGL((k_vstage<P, NL, MODE>), dim3(256), dim3(BS), 0, 0, X, nchunks, nother, res, out);
  CK(hipDeviceSynchronize());
  CK(hipMemset(res, 0, 64));
  hipLaunchKernelGGL((k_vstage<P, NL, MODE>), dim3(256), dim3(BS), 0, 0, X, nchunks, nother, res, out);
  CK(hipDeviceSynchronize());
  unsigned long long h[2];
  CK(hipMemcpy(h, res, 16, hipMemcpyDeviceToHost));
  const double clk = (double)h[0];
  printf("VGPR-staged P=%d, %2d loads in flight per wave %-22s fill %7.0f clk = %5.1f clk/piece/CU | other waves: %2d x %7.0f trips, %6.1f clk per trip per CU\n",
         P, NL, name, clk, clk / 8192.0, nother, nother ? (double)h[1] / 256.0 / nother : 0.0, h[1] ? clk / ((double)h[1] / 256.0) : 0.0);
  CK(hipFree(res)); CK(hipFree(out));
}

template <int P, int DEPTH, int MODE>
static void run(const char* name, const char* X, const f4* stream, int nother, int per_trip) {
  unsigned long long* res; float* out;
  CK(hipMalloc(&res, 64)); CK(hipMalloc(&out, 64));
  const int nchunks = (8 << 20) / 8192;
  hipLaunchKernelGGL((k_dma<P, DEPTH, MODE>), dim3(256), dim3(BS), 0, 0, X, stream, nchunks, nother, res, out);
  CK(hipDeviceSynchronize());
  CK(hipMemset(res, 0, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_dma<P, DEPTH, MODE>), dim3(256), dim3(BS), 0, 0, X, stream, nchunks, nother, res, out);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[2];
  CK(hipMemcpy(h, res, 16, hipMemcpyDeviceToHost));
  const double clk = (double)h[0];
  printf("P=%d depth=%d %-44s fill %7.0f clk = %5.1f clk/piece/CU (wall %.3f ms) | other waves: %2d x %7.0f trips, %6.1f clk per %s per CU\n",
         P, DEPTH, name, clk, clk / 8192.0, ms, nother, nother ? (double)h[1] / 256.0 / nother : 0.0,
         h[1] ? clk / ((double)h[1] / 256.0 * per_trip) : 0.0, "wave-instruction");
  CK(hipFree(res)); CK(hipFree(out));
}

int main() {
  char* X; CK(hipMalloc(&X, 8 << 20)); CK(hipMemset(X, 0, 8 << 20));
  f4* stream; CK(hipMalloc(&stream, (size_t)(1u << 24) * 16 + (4096 * 16))); CK(hipMemset(stream, 0, (size_t)(1u << 24) * 16));
  run<2, 2, 0>("producers only", X, stream, 0, 1);
  run<2, 3, 0>("producers only", X, stream, 0, 1);
  run<4, 2, 0>("producers only", X, stream, 0, 1);
  run<2, 2, 1>("+ ds_read_b64 linear x8 per trip", X, stream, 12, 8);
  run<2, 2, 2>("+ ds_read_b64 random x8 per trip", X, stream, 12, 8);
  run<2, 2, 3>("+ ds_write_b64 linear x8 per trip", X, stream, 12, 8);
  run<2, 2, 4>("+ v_fma_f32 x64 per trip", X, stream, 12, 64);
  run<2, 2, 7>("+ v_sqrt/v_rcp x16 per trip", X, stream, 12, 16);
  run<2, 2, 5>("+ global_load_dwordx4 x4 per trip", X, stream, 12, 4);
  run<2, 2, 6>("+ ds_read_b32 poll + s_sleep 1", X, stream, 12, 1);
  run<2, 2, 8>("+ ring-kernel mix (per trip)", X, stream, 12, 1);
  run<2, 3, 8>("+ ring-kernel mix (per trip)", X, stream, 12, 1);
  run<4, 2, 8>("+ ring-kernel mix (per trip)", X, stream, 12, 1);
  run<2, 2, 8>("+ ring-kernel mix, 6 waves", X, stream, 6, 1);
  run<2, 2, 1>("+ ds_read_b64 linear, 4 waves", X, stream, 4, 8);
  runv<2, 16, 0>("producers only", X, 0);
  runv<4, 8, 0>("producers only", X, 0);
  runv<4, 16, 0>("producers only", X, 0);
  runv<4, 24, 0>("producers only", X, 0);
  runv<4, 8, 8>("+ ring-kernel mix", X, 12);
  runv<4, 16, 8>("+ ring-kernel mix", X, 12);
  runv<4, 24, 8>("+ ring-kernel mix", X, 12);
  runv<2, 24, 8>("+ ring-kernel mix", X, 12);
  return 0;
}
