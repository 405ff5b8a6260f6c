// valuprobe.hip -- issue rates of the VALU / transcendental / SDWA / LDS forms the ring kernel's
// inner loop is built from (gfx950 design probe, round 3; not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -o tools/valuprobe tools/valuprobe.hip && ./tools/valuprobe
// Every pattern runs REPS x 64 instructions per wave (8 independent dependency chains), with 1, 2, 3
// or 4 waves per SIMD on every CU; the table gives shader clocks per wave-instruction per SIMD
// (s_memtime around the loop, slowest wave of the first workgroups) and the wall-clock figure.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int REPS = 2000;

#define BODY8(INS)                                                                   \
  asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)              \
               INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)              \
               INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)              \
               INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)              \
               INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)              \
               INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)              \
               INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)              \
               INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)              \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
               : "v"(b), "v"(c), "s"(sm))

typedef float f2 __attribute__((ext_vector_type(2)));

// operands: %0..%7 chains (f2: register pairs), %8 = b (pair), %9 = c (pair), %10 sgpr
#define I_FMA(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n\t"
#define I_PKFMA(k) "v_pk_fma_f32 %" #k ", %" #k ", %8, %9\n\t"
#define I_PKMUL(k) "v_pk_mul_f32 %" #k ", %" #k ", %8\n\t"
#define I_PKADD(k) "v_pk_add_f32 %" #k ", %" #k ", %8\n\t"
#define I_ADD(k) "v_add_f32 %" #k ", %" #k ", %8\n\t"
#define I_SQRT(k) "v_sqrt_f32 %" #k ", %" #k "\n\t"
#define I_RSQ(k) "v_rsq_f32 %" #k ", %" #k "\n\t"
#define I_RCP(k) "v_rcp_f32 %" #k ", %" #k "\n\t"
#define I_LOG(k) "v_log_f32 %" #k ", %" #k "\n\t"
#define I_EXP(k) "v_exp_f32 %" #k ", %" #k "\n\t"
#define I_SHR(k) "v_lshrrev_b32 %" #k ", 1, %" #k "\n\t"
#define I_SDWA(k) "v_and_b32_sdwa %" #k ", %10, %" #k " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
#define I_MIN(k) "v_min_f32 %" #k ", %" #k ", %8\n\t"
#define I_CNDMASK(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n\t"
#define I_BFE(k) "v_bfe_u32 %" #k ", %" #k ", 1, 30\n\t"
#define I_MULLEG(k) "v_mul_legacy_f32 %" #k ", %" #k ", %8\n\t"
// mixes: 3 plain + 1 transcendental, interleaved
#define I_MIX(k) "v_sqrt_f32 %" #k ", %" #k "\n\tv_fma_f32 %" #k ", %" #k ", %8, %9\n\t"

template <int PAT, class T>
__global__ __launch_bounds__(1024) void k_probe(T* out, unsigned long long* cyc, T b, T c, unsigned sm, int reps) {
  T a[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = b * (float)(threadIdx.x + k + 1);
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
    if constexpr (PAT == 0) BODY8(I_FMA);
    if constexpr (PAT == 1) BODY8(I_PKFMA);
    if constexpr (PAT == 2) BODY8(I_PKMUL);
    if constexpr (PAT == 3) BODY8(I_PKADD);
    if constexpr (PAT == 4) BODY8(I_ADD);
    if constexpr (PAT == 5) BODY8(I_SQRT);
    if constexpr (PAT == 6) BODY8(I_RSQ);
    if constexpr (PAT == 7) BODY8(I_RCP);
    if constexpr (PAT == 8) BODY8(I_LOG);
    if constexpr (PAT == 9) BODY8(I_EXP);
    if constexpr (PAT == 10) BODY8(I_SHR);
    if constexpr (PAT == 11) BODY8(I_SDWA);
    if constexpr (PAT == 12) BODY8(I_MIN);
    if constexpr (PAT == 13) BODY8(I_CNDMASK);
    if constexpr (PAT == 14) BODY8(I_BFE);
    if constexpr (PAT == 15) BODY8(I_MULLEG);
    if constexpr (PAT == 16) BODY8(I_MIX);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  T s = a[0];
#pragma unroll
  for (int k = 1; k < 8; ++k) s += a[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int PAT, class T>
static void run(const char* name, int ninstr_per_body) {
  T* out; unsigned long long* cyc;
  CK(hipMalloc(&out, sizeof(T) * 256 * 1024));
  CK(hipMalloc(&cyc, 8 * 256 * 16));
  printf("%-34s", name);
  for (int wps = 1; wps <= 4; ++wps) {
    const int bs = 256 * wps;
    T b, c;
    if constexpr (sizeof(T) == 8) { b = T{1.0000001f, 0.9999999f}; c = T{1e-9f, 1e-9f}; } else { b = 1.0000001f; c = 1e-9f; }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_probe<PAT, T>), dim3(256), dim3(bs), 0, 0, out, cyc, b, c, 0xfff8u, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_probe<PAT, T>), dim3(256), dim3(bs), 0, 0, out, cyc, b, c, 0xfff8u, REPS);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(256 * 4 * wps);
    CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long mx = 0;
    for (auto v : h) mx = v > mx ? v : mx;
    const double n = (double)REPS * ninstr_per_body;  // wave-instructions per wave
    // clocks per wave-instruction per SIMD = elapsed / (n * waves on the SIMD)
    printf(" | %dw: %5.2f clk (wall %5.2f)", wps, (double)mx / (n * wps), ms * 1e-3 * 2.4e9 / (n * wps));
  }
  printf("\n");
  CK(hipFree(out)); CK(hipFree(cyc));
}

// ---------------------------------------------------------------- LDS stores / reads with chosen conflicts
// every wave issues `reps` x 16 accesses; lane l of a wave touches slot perm[l] (8-byte slots)
template <int KIND>
__global__ __launch_bounds__(1024) void k_lds(const int* perm, unsigned long long* cyc, float* out, int reps) {
  __shared__ __attribute__((aligned(16))) char L[65536];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<float*>(L)[i] = (float)i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const unsigned addr = (unsigned)perm[lane] * 8u + (threadIdx.x >> 6) * 2048u;
  f2 v = {1.0f, 2.0f};
  f2 acc = {0.f, 0.f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
    if constexpr (KIND == 0) {
      f2 t[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(t[k]) : "v"(addr), "n"(0));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 16; ++k) acc += t[k];
    } else if constexpr (KIND == 1) {
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
      // read - add - write chain as the kernel does it
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        f2 t;
        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(addr) : "memory");
        t += v;
        asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(t) : "memory");
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (acc.x == 123.f) out[0] = acc.y;
  if (lane == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

static void lds_case(const char* name, const std::vector<int>& perm) {
  int* dperm; unsigned long long* cyc; float* out;
  CK(hipMalloc(&dperm, 256)); CK(hipMalloc(&cyc, 8 * 256 * 16)); CK(hipMalloc(&out, 64));
  CK(hipMemcpy(dperm, perm.data(), 256, hipMemcpyHostToDevice));
  printf("%-44s", name);
  const int reps = 400;
  for (int kind = 0; kind < 3; ++kind) {
    for (int waves : {4, 16}) {
      if (kind == 0) hipLaunchKernelGGL(k_lds<0>, dim3(256), dim3(64 * waves), 0, 0, dperm, cyc, out, reps);
      if (kind == 1) hipLaunchKernelGGL(k_lds<1>, dim3(256), dim3(64 * waves), 0, 0, dperm, cyc, out, reps);
      if (kind == 2) hipLaunchKernelGGL(k_lds<2>, dim3(256), dim3(64 * waves), 0, 0, dperm, cyc, out, reps);
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> h(256 * waves);
      CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
      unsigned long long mx = 0;
      for (auto v : h) mx = v > mx ? v : mx;
      const double per = (double)mx / ((double)reps * 16 * waves);  // clocks per wave-access per CU
      printf(" | %s %2dw %5.2f", kind == 0 ? "rd" : kind == 1 ? "wr" : "rmw", waves, per);
    }
  }
  printf("\n");
  CK(hipFree(dperm)); CK(hipFree(cyc)); CK(hipFree(out));
}

int main() {
  printf("clocks per wave-instruction per SIMD (s_memtime; wall = event time x 2.4 GHz), 1..4 waves per SIMD, 256 CUs busy\n");
  run<0, float>("v_fma_f32", 64);
  run<1, f2>("v_pk_fma_f32", 64);
  run<2, f2>("v_pk_mul_f32", 64);
  run<3, f2>("v_pk_add_f32", 64);
  run<4, float>("v_add_f32", 64);
  run<5, float>("v_sqrt_f32", 64);
  run<6, float>("v_rsq_f32", 64);
  run<7, float>("v_rcp_f32", 64);
  run<8, float>("v_log_f32", 64);
  run<9, float>("v_exp_f32", 64);
  run<10, float>("v_lshrrev_b32", 64);
  run<11, float>("v_and_b32_sdwa (WORD_1, sgpr mask)", 64);
  run<12, float>("v_min_f32", 64);
  run<13, float>("v_cndmask_b32", 64);
  run<14, float>("v_bfe_u32", 64);
  run<15, float>("v_mul_legacy_f32", 64);
  run<16, float>("v_sqrt_f32 + v_fma_f32 pairs (per instr)", 128);
  printf("\nLDS b64 accesses, clocks per wave-access per CU (4 and 16 waves per CU issuing); lane -> 8-byte slot patterns\n");
  std::vector<int> p(64);
  for (int l = 0; l < 64; ++l) p[l] = l;
  lds_case("linear (conflict-free)", p);
  for (int l = 0; l < 64; ++l) p[l] = (l & 31) + 32 * (l >> 5) * 5;           // halves on the same 32 classes
  lds_case("each class once per 32-lane half", p);
  for (int l = 0; l < 64; ++l) p[l] = (l & 15) + 32 * (l >> 4);                // classes 0..15 in every 16-lane quarter
  lds_case("16 classes, one per 16-lane quarter", p);
  for (int l = 0; l < 64; ++l) p[l] = ((l & 15) >> 1) * 2 + 32 * ((l >> 4) * 2 + (l & 1));  // 8 even classes twice per quarter
  lds_case("8 classes twice per quarter", p);
  for (int l = 0; l < 64; ++l) p[l] = (l & 31) / 2 + 32 * ((l >> 5) * 2 + (l & 1));  // 2 per class per half
  lds_case("16 classes twice per half", p);
  for (int l = 0; l < 64; ++l) p[l] = (l & 31) / 3 + 32 * ((l >> 5) * 3 + (l % 3));  // 3 per class per half
  lds_case("11 classes, three per half", p);
  srand(1);
  for (int l = 0; l < 64; ++l) p[l] = rand() % 4096;
  lds_case("random slots in 32 KB", p);
  return 0;
}
