// mde_ring_kernel.h -- the LDS-ring kernel template and its launcher (included by mde_ring_k_*.hip only).
#pragma once
#include <algorithm>
#include <cstring>
#include <type_traits>
#include <vector>

#include "mde_ring.h"
#define COMMA ,

// compile-time functors for d = 2 and 3, the dimensions embeddings are drawn in (the run-time functor is
// ~6x the code per entry, and this kernel is bound by the instructions it issues); d = 1 and 4 take the
// run-time functor (mde_ring_k_runtime.hip)
#define MDE_RING23(FN, LIN)                                    \
  do {                                                         \
    FN fn{a};                                                  \
    int rc;                                                    \
    if (A.d == 2)                                              \
      rc = launch_ring<2, FN, LIN>(A, fn, nblocks);            \
    else                                                       \
      rc = launch_ring<3, FN, LIN>(A, fn, nblocks);            \
    return rc == MDE_OK ? 1 : rc;                              \
  } while (0)

// ---------------------------------------------------------------- the kernel
typedef uint32_t ring_u4 __attribute__((ext_vector_type(4)));
typedef float ring_f4 __attribute__((ext_vector_type(4)));

template <int D>
struct RingVec;
template <>
struct RingVec<1> {
  typedef float T;
};
template <>
struct RingVec<2> {
  typedef float2 T;
};
template <>
struct RingVec<4> {
  typedef float4 T;
};
// D floats at an LDS byte address (aligned to the vector size for D = 1, 2, 4)
template <int D>
__device__ __forceinline__ void ring_ld(const char* p, float (&v)[D]) {
  if constexpr (D == 3) {
    const float* q = reinterpret_cast<const float*>(p);
    v[0] = q[0];
    v[1] = q[1];
    v[2] = q[2];
  } else {
    const typename RingVec<D>::T t = *reinterpret_cast<const typename RingVec<D>::T*>(p);
    const float* q = reinterpret_cast<const float*>(&t);
#pragma unroll
    for (int c = 0; c < D; ++c) v[c] = q[c];
  }
}
// The same for the OPERANDS of an entry (x_v, x_u), as volatile LDS loads: the consumer publishes "my
// slots are free" with a volatile LDS store right after ISSUING these reads (release_to), and a wave's LDS
// instructions execute in the order they are issued -- so the compiler must keep every one of them in front
// of that store, which it owes only to volatile accesses.  (Round 4: at d = 3 the three-word read was
// split and two words of x_u were issued BEHIND the store; a producer refilled the slot in between a few
// times per 10^8 entries -- found by the full-size d = 3 test, not reproducible below ~10^7 entries.)
template <int D>
__device__ __forceinline__ void ring_ld_operand(char* lds_base, uint32_t addr, float (&v)[D]) {
  typedef __attribute__((address_space(3))) const volatile float* lds_vf;
  typedef float ring_f2v __attribute__((ext_vector_type(2)));
  typedef float ring_f4v __attribute__((ext_vector_type(4)));
  if constexpr (D == 2) {
    const ring_f2v t = *(__attribute__((address_space(3))) const volatile ring_f2v*)(lds_base + addr);
    v[0] = t.x;
    v[1] = t.y;
  } else if constexpr (D == 4) {
    const ring_f4v t = *(__attribute__((address_space(3))) const volatile ring_f4v*)(lds_base + addr);
    v[0] = t.x;
    v[1] = t.y;
    v[2] = t.z;
    v[3] = t.w;
  } else {
    lds_vf q = (lds_vf)(lds_base + addr);
#pragma unroll
    for (int c = 0; c < D; ++c) v[c] = q[c];
  }
}
template <int D>
__device__ __forceinline__ void ring_st(char* p, const float (&v)[D]) {
  if constexpr (D == 3) {
    float* q = reinterpret_cast<float*>(p);
    q[0] = v[0];
    q[1] = v[1];
    q[2] = v[2];
  } else {
    typename RingVec<D>::T t;
    float* q = reinterpret_cast<float*>(&t);
#pragma unroll
    for (int c = 0; c < D; ++c) q[c] = v[c];
    *reinterpret_cast<typename RingVec<D>::T*>(p) = t;
  }
}

// control words: explicit LDS instructions on absolute addresses (a `volatile` generic pointer
// would turn into flat loads / stores and drag vmcnt(0) waits into the stream pipeline)
__device__ __forceinline__ void ring_ctrl_store(uint32_t addr, int v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// the same as a compiler-counted LDS instruction (its lgkmcnt bookkeeping stays exact)
__device__ __forceinline__ void ring_ctrl_store_counted(char* lds_base, uint32_t addr, int v) {
  typedef __attribute__((address_space(3))) volatile int* lds_vint;
  *(lds_vint)(lds_base + addr) = v;
}
__device__ __forceinline__ int ring_ctrl_load(uint32_t addr) {
  int v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
  return v;
}

#if MDE_RING_ABLATE
// timing probes (MDE_RING_DBG & 512): per workgroup, shader clocks summed over its waves
//   [0] consumer loop, [1] of it inside the chunk poll, [2] poll trips, [3] producer loop,
//   [4] of it blocked on a slot, [6] slot polls
static __device__ unsigned long long g_ring_probe[8][1024];
// hang diagnosis: a poll that spins longer than MDE_RING_SPINMAX trips leaves a record and gives up
static __device__ int g_ring_diag[64][8];
static __device__ int g_ring_ndiag;
#ifndef MDE_RING_SPINMAX
#define MDE_RING_SPINMAX 100000
#endif
#define RING_CLK() __builtin_readcyclecounter()
#endif
// CB: the first parameter comes from a codebook -- `packed` is the stream with value indices in its
// 3 low bits (mde_plan_expand_codebook), a0 the 8-entry value table; no parameter stream is read.
// LIN: f(0) = 0 whatever the lane's parameter is once that parameter is 0 (the functors
// instantiated below): padding lanes carry parameter 0, sit on a dummy row and need no masking.
// The loss term of an edge is added on ONE of its two entries (the one whose row is the smaller
// vertex; the header says, per block of four iterations, none / all / test per lane), so half of the
// blocks skip the loss arithmetic altogether; the gradient is owner-computes as before.
// PS, the form of the first parameter: 0 = an fp32 array next to the packed words (or one scalar), 1 = codebook
// (value index in the packed word's spare bits, <= 7 / 3 values), 2 = byte index (one byte per entry beside the
// packed words -- 5 B per half-edge --, a 256-entry value table in the 1 KB behind the ring; round 5: the hop
// counts of a distance-preserving problem on a graph are tens of distinct integers, which streamed 8 B until now)
// MDE_RING_FWORDS (round 6): producers publish INDEPENDENTLY -- F[p] = the next chunk of producer p that has not landed --
// and a consumer takes the minimum of the four words (one ds_read_b128, prefetched a pair ahead like round 5's single
// word).  Round 5 published in chunk order through ONE word: a producer that has stored chunk j waits until
// LANDED == j before it writes j + 1 -- a serial chain of two LDS round trips per chunk (~0.14 us), which is what the
// "staged bytes" bound of the large-table regimes turned out to be.
#ifndef MDE_RING_FWORDS
#define MDE_RING_FWORDS 0
#endif
#ifdef MDE_RING_EVAL2
constexpr bool defined_MDE_RING_EVAL2 = true;
#else
constexpr bool defined_MDE_RING_EVAL2 = false;
#endif
template <int D, class Fn, bool HAS_GRAD, int PS, bool LIN>
__global__ __launch_bounds__(MDE_RING_BS) void k_fused_ring(
    int nloc, int row_lo, int n, int R, int Q, int NC, int ring_off, int S, const int32_t* __restrict__ wave_iter,
    const uint32_t* __restrict__ hdr, const uint32_t* __restrict__ packed, const uint32_t* __restrict__ bidx,
    const int32_t* __restrict__ slot_row, const float* __restrict__ a0, const float* __restrict__ a1, int a0_scalar, int a1_scalar,
    const float* __restrict__ X,
    float* __restrict__ grad, float* __restrict__ partial, double* __restrict__ loss_partials, Fn fn,
    float fix_value, float grad_scale, float* __restrict__ loss_out, double loss_scale, int fold, int dbg_arg) {
#if MDE_RING_ABLATE
#ifndef MDE_RING_ABLATE_MASK
#define MDE_RING_ABLATE_MASK (~0)  // (a narrower mask lets hipcc fold the other probes away)
#endif
  const int dbg = dbg_arg & (MDE_RING_ABLATE_MASK);  // timing probes: 1 consumers never wait, 8 producers never wait, 64 / 128 a role skips its loop, 512 clocks
#else
  constexpr int dbg = 0;
#endif
  constexpr int BS = MDE_RING_BS, NCW = MDE_RING_NCW, NPROD = MDE_RING_NPROD;
  constexpr bool CB = PS == 1, BX = PS == 2;
  constexpr int GR_OFF = ring_gr_off(D), CTRL_PROG = MDE_RING_CTRL_PROG(D), CTRL_F = MDE_RING_CTRL_F(D), CTRL_CB = MDE_RING_CTRL_CB(D);
  constexpr int C = ring_chunk_cols(D), CBYTES = ring_chunk_bytes(D), PIECES = CBYTES / 1024;
  static_assert(GR_OFF + (ring_row_cap(D) + 32) * 4 * D <= 65536 + GR_OFF && GR_OFF < 65536 && CTRL_CB + 32 < 65536,
                "region bases must fit the 16-bit offset of the LDS instructions");
  static_assert(NPROD <= 8 && NCW <= 16, "control-word layout");
  // statically sized: the LDS addresses unpacked from the stream are absolute
  __shared__ __attribute__((aligned(16))) char L[MDE_RING_LDS_BYTES];
  float* XR = reinterpret_cast<float*>(L);           // x_v of the block's rows
  float* GR = reinterpret_cast<float*>(L + GR_OFF);  // gradient accumulators (same slots)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keep it scalar
  // block b = rb * Q + qg: row block rb, column group qg owns chunks [j_lo, j_hi)
  const int rb = blockIdx.x / Q, qg = blockIdx.x % Q;
  const int j_lo = (int)(((int64_t)qg * NC + Q - 1) / Q), j_hi = (int)(((int64_t)(qg + 1) * NC + Q - 1) / Q);
  const int r0 = rb * R;
  const int nr = min(R, nloc - r0);
  const uint32_t dummy_row = (uint32_t)R * 4u * (uint32_t)D;  // the padding lanes' row slots start here
  const float a0s = (a0_scalar && PS == 0) ? a0[0] : 1.0f;
  const uint32_t tab_off = (uint32_t)ring_off + (uint32_t)S * (uint32_t)CBYTES;  // (BX: the value table behind the ring)
  const float a1s = (a1 && a1_scalar) ? a1[0] : 0.0f;
  const bool a1_arr = a1 && !a1_scalar;
  // ---- prologue: accumulators, x_v, control words
  {
    float4* z = reinterpret_cast<float4*>(L + GR_OFF);
    const int nz = ((R + 32) * 4 * D + 15) / 16;
    for (int i = tid; i < nz; i += BS) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* Xrow = X + (size_t)(row_lo + r0) * D;
    if (slot_row) {
      // permuted row blocks (round 6): slot s holds local row slot_row[rb * R + s] (-1: nobody; such a slot has no
      // entries, its x_v is never read)
      const int32_t* sr = slot_row + (size_t)rb * R;
      for (int sl = tid; sl < R; sl += BS) {
        const int r = sr[sl];
        float xv[D];
#pragma unroll
        for (int c = 0; c < D; ++c) xv[c] = r >= 0 ? X[(size_t)(row_lo + r) * D + c] : 0.0f;
#pragma unroll
        for (int c = 0; c < D; ++c) XR[sl * D + c] = xv[c];
      }
    } else if ((reinterpret_cast<uintptr_t>(Xrow) & 15) == 0) {
      // 16-byte loads, eight in flight per thread
      const ring_f4* X4 = reinterpret_cast<const ring_f4*>(Xrow);
      ring_f4* XR4 = reinterpret_cast<ring_f4*>(L);
      const int n4 = (nr * D) >> 2;
      for (int i0 = 0; i0 < n4; i0 += 8 * BS) {
        ring_f4 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = X4[min(i0 + tid + k * BS, max(n4 - 1, 0))];
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (i0 + tid + k * BS < n4) XR4[i0 + tid + k * BS] = t[k];
      }
      for (int i = (n4 << 2) + tid; i < nr * D; i += BS) XR[i] = Xrow[i];
    } else {
      for (int i = tid; i < nr * D; i += BS) XR[i] = Xrow[i];
    }
    if (tid < 32 * D) XR[R * D + tid] = 0.0f;  // the dummy rows
    int* prog = reinterpret_cast<int*>(L + CTRL_PROG);
    int* F = reinterpret_cast<int*>(L + CTRL_F);
    if (tid < 16) prog[tid] = (tid < NCW) ? j_lo : MDE_RING_DONE;
#if MDE_RING_FWORDS
    static_assert(NPROD == 4 && MDE_RING_UNIT == 1, "MDE_RING_FWORDS: four producers, one chunk per step");
    if (tid >= 16 && tid < 16 + NPROD) F[tid - 16] = (j_lo + (tid - 16) < j_hi) ? j_lo + (tid - 16) : MDE_RING_DONE;
#else
    if (tid == 16) F[0] = j_lo;  // LANDED: every chunk below this is in its ring slot
#endif
    if (CB && tid >= 32 && tid < 32 + MDE_RING_CB_VALUES)
      reinterpret_cast<float*>(L + CTRL_CB)[tid - 32] = a0[tid - 32] * Fn::kParamScale;
    if (BX && tid >= 64 && tid < 64 + MDE_RING_BX_VALUES)
      reinterpret_cast<float*>(L + tab_off)[tid - 64] = a0[tid - 64] * Fn::kParamScale;
  }
  __syncthreads();
  float loss = 0.0f, loss2 = 0.0f;  // (the fused Log1p path keeps the log2 terms and the corrections apart)

  if (wave >= NCW) {
    if (MDE_RING_PRODPRIO) __builtin_amdgcn_s_setprio(MDE_RING_PRODPRIO);
    // ---------------- producer p: chunks j_lo + p, j_lo + p + NPROD, ...  staged through VGPRs:
    // DEPTH chunks are in flight as plain 16-byte global loads (compiler-counted), the oldest one is
    // written into its ring slot (ds_write_b128) once every consumer is past the chunk that slot
    // held, and published.  The chunks in flight need no ring slot (round 3's LDS-DMA parked every
    // chunk in flight in a slot: with row blocks twice as tall the ring is too small for that).
    const int p = wave - NCW;
    const char* Xb = reinterpret_cast<const char*>(X);
    const size_t nbytes = (size_t)n * D * 4;
    const size_t last16 = nbytes - 16;  // (X is 16-byte aligned, n * D * 4 >= 16: checked by the launcher)
    constexpr int DEPTH = MDE_RING_DEPTH, UNIT = MDE_RING_UNIT;
    // a producer step moves UNIT consecutive chunks: one slot poll and one publish for all of them
    ring_f4 buf[DEPTH][UNIT * PIECES];
    // The chunk loads are INLINE ASM and their waits are placed by hand (round 5).  Written as plain loads, the
    // loop-carried buffers made hipcc wait with vmcnt(3..0) in front of a chunk's stores -- for ALL loads in
    // flight, the other buffers' too: the wave overlapped half a chunk period with its loads instead of
    // DEPTH - 1 periods (and DEPTH 3 / 4 measured nothing).  A producer wave issues no other vector-memory
    // instruction inside its loop, so the count is simply: the (DEPTH - 1) x UNIT x PIECES loads issued after
    // this buffer's may stay in flight.  The wait takes the buffer's registers as read-write operands, so
    // every later use of them depends on it.
    auto ld16 = [&](ring_f4& dst, const ring_f4* src) __attribute__((always_inline)) {
#if MDE_RING_ASMLOAD
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src) : "memory");
#else
      dst = *src;
#endif
    };
    auto fetch = [&](int k, int j0u) __attribute__((always_inline)) {
#pragma unroll
      for (int c = 0; c < UNIT; ++c) {
        const int j = j0u + c;
        if (j < NC - 1) {
          // a chunk that lies wholly inside the table: one base address, the pieces at immediate offsets
          // (the producers share their SIMDs' issue slots with the consumers: every instruction here counts)
          const ring_f4* src = reinterpret_cast<const ring_f4*>(Xb + (size_t)j * CBYTES + (size_t)lane * 16);
#pragma unroll
          for (int i = 0; i < PIECES; ++i) ld16(buf[k][c * PIECES + i], src + i * 64);
        } else {
          // the table's last chunk, or a prefetch past the end of the table (clamped, never written)
          const size_t off0 = (size_t)min(j, NC - 1) * CBYTES + (size_t)lane * 16;
#pragma unroll
          for (int i = 0; i < PIECES; ++i) {
            const size_t off = off0 + (size_t)i * 1024;
            ld16(buf[k][c * PIECES + i], reinterpret_cast<const ring_f4*>(Xb + (off < last16 ? off : last16)));
          }
        }
      }
    };
    auto wait_buf = [&](int k) __attribute__((always_inline)) {
#if MDE_RING_ASMLOAD
      static_assert(UNIT * PIECES == 4 || UNIT * PIECES == 2 || UNIT * PIECES == 3 || UNIT * PIECES == 8 || UNIT * PIECES == 6, "operand list of the wait");
      constexpr int NEWER = (DEPTH - 1) * UNIT * PIECES;
      if constexpr (UNIT * PIECES == 4)
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(buf[k][0]), "+v"(buf[k][1]), "+v"(buf[k][2]), "+v"(buf[k][3]) : "n"(NEWER) : "memory");
      else if constexpr (UNIT * PIECES == 3)
        asm volatile("s_waitcnt vmcnt(%3)" : "+v"(buf[k][0]), "+v"(buf[k][1]), "+v"(buf[k][2]) : "n"(NEWER) : "memory");
      else if constexpr (UNIT * PIECES == 2)
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(buf[k][0]), "+v"(buf[k][1]) : "n"(NEWER) : "memory");
      else if constexpr (UNIT * PIECES == 6)
        asm volatile("s_waitcnt vmcnt(%6)" : "+v"(buf[k][0]), "+v"(buf[k][1]), "+v"(buf[k][2]), "+v"(buf[k][3]), "+v"(buf[k][4]),
                     "+v"(buf[k][5]) : "n"(NEWER) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(%8)" : "+v"(buf[k][0]), "+v"(buf[k][1]), "+v"(buf[k][2]), "+v"(buf[k][3]), "+v"(buf[k][4]),
                     "+v"(buf[k][5]), "+v"(buf[k][6]), "+v"(buf[k][7]) : "n"(NEWER) : "memory");
#endif
    };
    // (see the stores below) the straddling 16-byte piece of the table's last chunk, for the lane that owns it
    ring_f4 tail_v = {0.0f, 0.0f, 0.0f, 0.0f};
    int tail_rel = -1;  // its byte offset inside the chunk
    if ((nbytes & 15) != 0 && j_hi == NC) {
      const size_t off_str = nbytes & ~(size_t)15;
      const size_t rel = off_str - (size_t)(NC - 1) * CBYTES;
      if (off_str >= (size_t)(NC - 1) * CBYTES && lane == (int)((rel & 1023) >> 4)) {
        tail_rel = (int)rel;
        const float* tp = reinterpret_cast<const float*>(Xb + off_str);
        const int nt = (int)((nbytes - off_str) >> 2);  // 1..3 floats
        tail_v[0] = tp[0];
        if (nt > 1) tail_v[1] = tp[1];
        if (nt > 2) tail_v[2] = tp[2];
      }
    }
    int minprog = j_lo, landed = j_lo;
    const uint32_t poll_addr = lane < 16 ? CTRL_PROG + 4u * (uint32_t)lane : (uint32_t)CTRL_F;
#if MDE_RING_ABLATE
    unsigned long long pr_t0 = RING_CLK(), pr_blocked = 0, pr_polls = 0, pr_chain = 0, pr_data = 0, pr_write = 0;
#endif
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) fetch(k, j_lo + UNIT * (p + k * NPROD));
    int slot = (j_lo + UNIT * p) % S;
    for (int j0 = j_lo + UNIT * p; j0 < j_hi && !(dbg & 128); j0 += UNIT * DEPTH * NPROD) {
#pragma unroll
      for (int k = 0; k < DEPTH; ++k) {
        const int j = j0 + k * UNIT * NPROD;
        if (j < j_hi) {
          const int jl = min(j + UNIT, j_hi) - 1;  // the step's last chunk
          // slot jl % S still holds chunk jl - S until every consumer is past it (the slots of the step's
          // earlier chunks were released before)
          while (jl - S >= minprog && !(dbg & 8)) {
#if MDE_RING_ABLATE
            const unsigned long long tb0 = RING_CLK();
            ++pr_polls;
#endif
            // one LDS read (lane & 15 = consumer, the words past NCW hold DONE), the minimum over each row of
            // 16 lanes with DPP shifts (3-4 VALU instead of one v_readlane + s_min per consumer), one v_readlane
            // (lanes 16.. read LANDED with the same instruction: the publish below finds it without a poll of its own)
            int v = ring_ctrl_load(poll_addr);
            // (v_min_i32_dpp: lanes without a source in their row keep their value; two wait states between a VALU
            // write and a DPP read of the same register)
            asm volatile("s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1"
                         : "+v"(v));
            if (NCW > 8) asm volatile("v_min_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(v));
            minprog = __builtin_amdgcn_readlane(v, NCW <= 8 ? 7 : 15);
            landed = __builtin_amdgcn_readlane(v, 16);
            if (jl - S >= minprog) __builtin_amdgcn_s_sleep(MDE_RING_PSLEEP);
#if MDE_RING_ABLATE
            pr_blocked += RING_CLK() - tb0;
            if (pr_polls > MDE_RING_SPINMAX) {
              if (lane == 0) {
                const int kk = atomicAdd(&g_ring_ndiag, 1);
                if (kk < 64) {
                  int* r = g_ring_diag[kk];
                  r[0] = 2; r[1] = blockIdx.x; r[2] = p; r[3] = j; r[4] = minprog; r[5] = S; r[6] = j_lo; r[7] = j_hi;
                }
              }
              pr_polls = 0;
              minprog = jl;  // (give up: overwrite the slot)
            }
#endif
          }
#if MDE_RING_ABLATE
          // (probe: how long the chunk's own loads still take once its slot is free -- the newer loads of the
          // wave's other buffers may stay in flight)
          unsigned long long td0 = 0;
          if (dbg & 512) {
            td0 = RING_CLK();
            __builtin_amdgcn_s_waitcnt(0x0F70 | (((DEPTH - 1) * UNIT * PIECES) & 15) | ((((DEPTH - 1) * UNIT * PIECES) >> 4) << 14));
            pr_data += RING_CLK() - td0;
            td0 = RING_CLK();
          }
#endif
          wait_buf(k);
#pragma unroll
          for (int c = 0; c < UNIT; ++c) {
            const int jc = j + c;
            if (UNIT > 1 && jc >= j_hi) break;
            int sc = slot + c;
            if (UNIT > 1 && sc >= S) sc -= S;
            char* dst = L + (uint32_t)__builtin_amdgcn_readfirstlane(ring_off + sc * CBYTES) + lane * 16;
#pragma unroll
            for (int i = 0; i < PIECES; ++i) *reinterpret_cast<ring_f4*>(dst + i * 1024) = buf[k][c * PIECES + i];
            // the table's last chunk when the table is not a multiple of 16 bytes: the one lane whose 16 bytes
            // straddle the end loaded the LAST 16 bytes instead -- it stores the piece it prepared before the loop
            // over them (round 5: the shift used to sit in the loop as a second copy of the four stores, and the
            // join of the two copies made hipcc wait for ALL loads in flight -- vmcnt(0) -- before every chunk)
            if (jc == NC - 1 && tail_rel >= 0) *reinterpret_cast<ring_f4*>(dst - lane * 16 + tail_rel) = tail_v;
          }
          // (the LDS executes a wave's accesses in order: whoever sees the new LANDED sees the chunks -- provided
          // their stores are ISSUED in front of the publish: float stores against an int store, which
          // type-based alias analysis would let the compiler swap)
          asm volatile("" ::: "memory");
          // publish IN CHUNK ORDER (round 5): LANDED == j says every chunk below j is in its slot, so the consumers
          // read ONE word (a v_readfirstlane) where they took a minimum over the producers' four.  The chunk below
          // belongs to the producer next door, which got its slot earlier: the wait is short.
          // (LANDED == j, once seen, holds until this wave publishes: the value read with the slot poll will do)
#if MDE_RING_FWORDS
          // my next chunk that has not landed (past the end of my chunks: DONE) -- nobody to wait for
          (void)landed;
          ring_ctrl_store_counted(L, CTRL_F + 4u * (uint32_t)p, (j + NPROD < j_hi) ? j + NPROD : MDE_RING_DONE);
#else
          while (landed != j && !(dbg & 8)) {
            landed = __builtin_amdgcn_readfirstlane(ring_ctrl_load((uint32_t)CTRL_F + 0u * (uint32_t)lane));
#if MDE_RING_ABLATE
            ++pr_chain;
#endif
            if (landed != j) __builtin_amdgcn_s_sleep(0);
          }
          ring_ctrl_store_counted(L, CTRL_F, jl + 1);
#endif
#if MDE_RING_ABLATE
          if (dbg & 512) pr_write += RING_CLK() - td0;
#endif
          fetch(k, j + UNIT * DEPTH * NPROD);
          slot += UNIT * NPROD;
          while (slot >= S) slot -= S;
        }
      }
    }
#if MDE_RING_ABLATE
    if ((dbg & 512) && lane == 0 && blockIdx.x < 1024) {
      atomicAdd(&g_ring_probe[3][blockIdx.x], RING_CLK() - pr_t0);
      atomicAdd(&g_ring_probe[4][blockIdx.x], pr_blocked);
      atomicAdd(&g_ring_probe[6][blockIdx.x], pr_polls);
      atomicAdd(&g_ring_probe[5][blockIdx.x], pr_data);
      atomicAdd(&g_ring_probe[7][blockIdx.x], pr_write);
    }
#endif
  } else {
    // ---------------- consumer: my contiguous stream of wave iterations, 4 per block
    if (MDE_RING_CONSPRIO) __builtin_amdgcn_s_setprio(MDE_RING_CONSPRIO);
    const int ib = __builtin_amdgcn_readfirstlane(wave_iter[blockIdx.x * NCW + wave]);
    const int NB = (__builtin_amdgcn_readfirstlane(wave_iter[blockIdx.x * NCW + wave + 1]) - ib) >> 2;
    if (NB > 0 && !(dbg & 64)) {
      const bool a0_arr = !a0_scalar && PS == 0;
      const uint32_t* bp = BX ? bidx + (size_t)(ib >> 2) * 64 + lane : nullptr;
      const ring_u4* sp = reinterpret_cast<const ring_u4*>(packed) + (size_t)(ib >> 2) * 64 + lane;
      const ring_f4* ap = reinterpret_cast<const ring_f4*>(a0_arr ? a0 : reinterpret_cast<const float*>(packed)) +
                          (size_t)(ib >> 2) * 64 + lane;
      const int lastb = NB - 1;
      // PFB blocks (4 iterations each) of packed words, parameters and headers in flight.  All
      // stream loads are issued from ONE place (the refill after a block is consumed),
      // unconditional and clamped, never predicated: on every path the same loads are in flight
      // when a block is consumed, so the compiler's vmcnt counts are exact and nothing waits for a
      // load just issued.
      // The 16 header words of a block come with ONE vector load: lane l holds word l & 15, i.e. every
      // row of 16 lanes holds the whole header, and a word reaches all lanes with a DPP row broadcast
      // (one VALU instruction, no round trip through the scalar unit).  Round 3 fetched the headers
      // with s_load_dwordx16: scalar loads share lgkmcnt with the LDS instructions and return out of
      // order, so the first LDS wait after a refill drained the header load just issued.
      constexpr int PFB = MDE_RING_PFB;
      ring_u4 pq[PFB] = {};
      uint32_t hv[PFB] = {};
      ring_f4 wq[PFB] = {};
      uint32_t bq[PFB] = {};  // (BX: the value indices of a block's four iterations, one byte each)
      const uint32_t* hvp = hdr + (size_t)ib * MDE_RING_HW + (lane & 15);
      auto load_block = [&](int u, int b) __attribute__((always_inline)) {
        const int bc = min(b, lastb);
#if defined(MDE_RING_STREAM_L2)  // (design probe, wrong results: the packed words of 8 blocks over and over -- an L2-resident stream)
        pq[u] = sp[(size_t)(bc & 7) * 64];
#elif defined(MDE_RING_STREAM_NT)
        pq[u] = __builtin_nontemporal_load(&sp[(size_t)bc * 64]);
#else
        pq[u] = sp[(size_t)bc * 64];
#endif
        hv[u] = hvp[(size_t)bc * 16];
        if (PS == 0) wq[u] = ap[(size_t)bc * 64];
        if (BX) bq[u] = bp[(size_t)bc * 64];
      };
#if MDE_RING_ABLATE
      unsigned long long cs_t0 = RING_CLK(), cs_poll = 0, cs_trips = 0;
#endif

      // the LDS operands of an entry: x_v, x_u, the parameter
      struct Pre {
        float xr[D], xc[D], p0;
#ifdef MDE_RING_KEEPROW
        uint32_t row;  // (round 6 probe: the row address kept from the operand read to the accumulator update)
#endif
      };
      // packed word -> LDS byte addresses (ring_pack_word)
      // (MDE_RING_PROBE_ROWLIN / _COLLIN: timing probes with wrong results -- the row / column side of every entry at
      // a conflict-free address, same instruction count: what the LDS bank conflicts of that side cost)
#ifdef MDE_RING_PROBE_ROWLIN
      auto row_of = [&](uint32_t w) __attribute__((always_inline)) { return ((w & 0x7f00u) | ((uint32_t)lane << 3)) & 0xfff8u; };
#else
      auto row_of = [&](uint32_t w) __attribute__((always_inline)) { return D == 2 ? (w & 0xfff8u) : (w & 0xfffcu); };
#endif
#ifdef MDE_RING_PROBE_COLLIN
      auto col_of = [&](uint32_t w) __attribute__((always_inline)) {
        return (((w >> 13) & 0x3f000u) | ((uint32_t)lane << 3)) & 0x3fff8u;
      };
#else
      auto col_of = [&](uint32_t w) __attribute__((always_inline)) {
        return D == 2 ? ((w >> 13) & 0x3fff8u) : ((w >> 14) & 0x3fffcu);
      };
#endif
      auto issue_x = [&](uint32_t w, float p0, uint32_t bi4) __attribute__((always_inline)) {
        Pre r;
        // (codebook index: the bits the row address leaves free -- 3 at d = 2, 2 at d = 3; byte index: bi4 = 4 x index)
        if constexpr (BX)
          r.p0 = *reinterpret_cast<const float*>(L + tab_off + bi4);
        else
          r.p0 = CB ? *reinterpret_cast<const float*>(L + CTRL_CB + ((w & (D == 2 ? 7u : 3u)) << 2)) : p0 * Fn::kParamScale;
#ifdef MDE_RING_KEEPROW
        r.row = row_of(w);
        ring_ld_operand<D>(L, r.row, r.xr);
#else
        ring_ld_operand<D>(L, row_of(w), r.xr);
#endif
        ring_ld_operand<D>(L, col_of(w), r.xc);
        return r;
      };
      // 4 x the value index of iteration q of a block (byte q of the block's index word)
      auto bx4 = [&](uint32_t bw, int q) __attribute__((always_inline)) { return BX ? ((bw >> (8 * q)) & 0xffu) << 2 : 0u; };
      // this entry adds the loss term iff its row is the smaller vertex (blocks around the diagonal)
      auto counts_here = [&](uint32_t w, uint32_t hm) __attribute__((always_inline)) {
        const uint32_t off = col_of(w) - (uint32_t)ring_off;
        const uint32_t slot = off / (uint32_t)CBYTES, cidx = (off - slot * (uint32_t)CBYTES) / (4u * (uint32_t)D);
        const uint32_t ms = hm % (uint32_t)S;
        const uint32_t j = hm + (slot >= ms ? slot - ms : slot + (uint32_t)S - ms);
        const uint32_t u = j * (uint32_t)C + cidx;
        const uint32_t vv = (uint32_t)(row_lo + r0) + row_of(w) / (4u * (uint32_t)D);
        return vv < u;
      };
      // evaluate the entry and add its gradient term to the row's accumulator (read earlier).  LC is
      // the loss class of the BLOCK (compile time: no branch per iteration): 0 no entry adds its loss
      // term here, 1 every entry does, 2 test per lane.
      auto finish = [&](auto lc_tag, uint32_t w, const Pre& x, float (&acc)[D], float p1, uint32_t hm)
          __attribute__((always_inline)) {
        constexpr int LC = decltype(lc_tag)::value;
#ifdef MDE_RING_KEEPROW
        const uint32_t rowaddr = x.row;
#else
        const uint32_t rowaddr = row_of(w);
#endif
        float v[D], ss = (defined_MDE_RING_EVAL2 && Fn::kRingFused) ? 1.0e-30f : 0.0f;
#pragma unroll
        for (int c = 0; c < D; ++c) {
          v[c] = x.xr[c] - x.xc[c];
          ss = fmaf(v[c], v[c], ss);
        }
        float gd;
        if constexpr (Fn::kRingFused) {
          // Log1p, exponent 1.5 (mde_functions.h, MDE_F_LOG1P): x.p0 = 1.5 w.  Two square roots and
          // one reciprocal; r stays finite at d = 0, where v = 0 anyway.
          const float d = mde_sqrt(ss), sd = mde_sqrt(d);
          const float t = fmaf(d, sd, 1.0f);
          // (round 6 probe MDE_RING_EVAL2: sd t = sd + d sd sd = sd + ss, so the reciprocal's argument needs neither t
          // nor a constant of its own once 1e-30 sits in ss -- one VALU instruction less where no loss term is added)
          const float r = defined_MDE_RING_EVAL2 ? mde_rcp(sd + ss) : mde_rcp(fmaf(sd, t, 1.0e-30f));
          gd = x.p0 * r;
          if constexpr (LC != 0) {
            // w log1p(u) = w ln2 log2(t) + (w / t) (u - (t - 1)), 1 / t = sqrt(d) r: the two sums are
            // kept apart and combined (with 1 / 1.5) once per wave
            const float c = d * sd - (t - 1.0f);
            float wl = x.p0, gc = gd;
            if (!LIN) {
              // (one scalar parameter for every lane: the padding lanes carry it too)
              const bool real = rowaddr < dummy_row;
              wl = real ? wl : 0.0f;
              gc = real ? gc : 0.0f;
            }
            if constexpr (LC == 2) {
              const bool here = counts_here(w, hm);
              wl = here ? wl : 0.0f;
              gc = here ? gc : 0.0f;
            }
            loss = fmaf(wl, mde_log2(t), loss);
            loss2 = fmaf(gc, sd * c, loss2);
          }
        } else {
          float f;  // (dead code when LC == 0)
          fn.eval(ss, x.p0, p1, f, gd);
          if constexpr (LC != 0) {
            if (!LIN) f = (rowaddr < dummy_row) ? f : 0.0f;
            if constexpr (LC == 2) f = counts_here(w, hm) ? f : 0.0f;
            loss += f;
          }
        }
        if (!HAS_GRAD) return;
        // (codebook values are finite -- mde_plan_expand_codebook refuses others)
        const float g = (Fn::kFiniteG && PS != 0) ? gd : mde_fix_g_to(gd, fix_value);
#pragma unroll
        for (int c = 0; c < D; ++c) acc[c] = fmaf(v[c], g, acc[c]);
        ring_st<D>(L + GR_OFF + rowaddr, acc);
      };
      // Hand-shake, once per PAIR of iterations (q = 0 or 2 of block u; header words at lanes 4 q ..).
      // A wave needs its chunks resident only at the moment it ISSUES the LDS reads of a pair (the LDS
      // executes in order: whatever lands later lands behind them): wait_pair() waits until the newest
      // chunk of the pair has landed, then the reads go out, then release_to() publishes the oldest
      // chunk of the NEXT pair -- between two hand-shakes the wave holds nothing, so the slots it is
      // done with are free a whole pair earlier than with "publish, wait, read".  The layout anchors
      // the chunk window of a pair at its first iteration (k_ring_schedule), so one test covers both.
      const uint32_t prog_addr = CTRL_PROG + 4u * (uint32_t)wave;
      int ready = j_lo;  // chunks below this have landed
      // The producers' F words are read one hand-shake AHEAD (round 5): a poll issued when the wave needs its
      // answer costs a whole LDS round trip behind everything the wave has in flight (ring_ctrl_load drains
      // the LDS queue), and with a 9-slot ring the cached `ready` covers hardly more than one pair -- nearly
      // every pair polled at least once, ~280 clocks each, a third of the loop.  The prefetched words are a pair
      // old, i.e. merely conservative (F only grows); when they are not enough the wave polls as before.
      typedef __attribute__((address_space(3))) const volatile int* lds_cvint;
      const uint32_t f_addr = (uint32_t)CTRL_F + 0u * (uint32_t)lane;  // (a VGPR for the inline-asm poll)
#if MDE_RING_FWORDS
      typedef int ring_i4 __attribute__((ext_vector_type(4)));
      typedef __attribute__((address_space(3))) const volatile ring_i4* lds_cvint4;
      auto min4 = [](const ring_i4& v) __attribute__((always_inline)) { return min(min(v.x, v.y), min(v.z, v.w)); };
      ring_i4 fl_pre = *(lds_cvint4)(L + CTRL_F);
#else
      int fl_pre = *(lds_cvint)(L + CTRL_F);
#endif
      auto wait_pair = [&](int u, int q) __attribute__((always_inline)) {
        const int need = __builtin_amdgcn_readlane((int)hv[u], 4 * q + 1);
        if (__builtin_expect(need >= ready && !(dbg & 1), 0)) {
#if MDE_RING_FWORDS
          ready = __builtin_amdgcn_readfirstlane(min4(fl_pre));
#else
          ready = __builtin_amdgcn_readfirstlane(fl_pre);
#endif
          if (__builtin_expect(need >= ready, 0)) {
#if MDE_RING_ABLATE
            const unsigned long long tp0 = RING_CLK();
#endif
            for (;;) {
#if MDE_RING_FWORDS
              {
                ring_i4 fv;
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(fv) : "v"(f_addr) : "memory");
                ready = __builtin_amdgcn_readfirstlane(min4(fv));
              }
#else
              ready = __builtin_amdgcn_readfirstlane(ring_ctrl_load(f_addr));
#endif
#if MDE_RING_ABLATE
              ++cs_trips;
              if (cs_trips > MDE_RING_SPINMAX) {
                if (lane == 0) {
                  const int k = atomicAdd(&g_ring_ndiag, 1);
                  if (k < 64) {
                    int* r = g_ring_diag[k];
                    r[0] = 1; r[1] = blockIdx.x; r[2] = wave; r[3] = need; r[4] = ready; r[5] = 0; r[6] = u * 4 + q; r[7] = NB;
                  }
                }
                cs_trips = 0;
                ready = need + 1;
                break;
              }
#endif
              if (need < ready) break;
              __builtin_amdgcn_s_sleep(MDE_RING_CSLEEP);
            }
#if MDE_RING_ABLATE
            cs_poll += RING_CLK() - tp0;
#endif
          }
          asm volatile("" ::: "memory");
        }
      };
      // (issued right behind a pair's operand reads and its release)
#if MDE_RING_FWORDS
      auto prefetch_landed = [&]() __attribute__((always_inline)) { fl_pre = *(lds_cvint4)(L + CTRL_F); };
#else
      auto prefetch_landed = [&]() __attribute__((always_inline)) { fl_pre = *(lds_cvint)(L + CTRL_F); };
#endif
      // (m never decreases along a stream; past its end the headers are copies of the last block and
      // the value published is merely too old)
      // (the lane that holds the header word stores it: no v_readlane / v_mov round trip through the scalar unit)
      // EVERY lane stores its header word: the lane that holds word 4 q (the pair's m) to prog[wave], the others
      // into the spare half of the control block (lanes l and l + 32 share a word: different 32-lane passes) --
      // no branch around a one-lane store, no address or data to move per pair
      const uint32_t rel_addr[2] = {lane == 0 ? 4u * (uint32_t)wave : 128u + 4u * (uint32_t)(lane & 31),
                                    lane == 8 ? 4u * (uint32_t)wave : 128u + 4u * (uint32_t)(lane & 31)};
      auto release_to = [&](int u, int q) __attribute__((always_inline)) {
        ring_ctrl_store_counted(L, CTRL_PROG + rel_addr[q >> 1], (int)hv[u]);
      };

      // Software pipeline over PAIRS of iterations.  In the region of pair p the wave (after the
      // hand-shake for the chunks of pair p + 1) issues the LDS reads of the operands of pair p + 1
      // (x_v, x_u, parameter), then evaluates the two iterations of pair p (operands read one pair
      // ago): accumulator write of k and, right behind it, the accumulator read of k + 1 (the LDS
      // executes a wave's accesses in order, so a row shared by consecutive iterations sees the
      // update).  Every LDS access of the loop is a compiler-counted instruction (the control-word
      // store included: round 3 issued it from inline asm, which put hipcc's lgkmcnt counts off by
      // one), and the region of a pair is ONE basic block: hipcc's scheduler interleaves the two
      // evaluations and places the waits.
      // What bounds this loop is instruction issue, not LDS or HBM (round 4, tools/r4_run.sh:
      // the consumers alone take 0.15 ms with a test + branch per iteration and 0.10 ms as
      // straight-line code; LDS conflicts and the accumulator chain cost nothing): hence one
      // hand-shake branch per pair, one loss-class branch per block of four (three copies of the
      // block body), and no header word through an SGPR.
      Pre xa, xb;  // operands of the pair being evaluated
      float acc[D];
      auto block_body = [&](auto lc_tag, int u, int b) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int q = 2 * h;
          // the pair after this one (block and slot); past the end of the stream it is a copy of
          // the last block: resident chunks, harmless reads, nothing new published
          const int un = h == 0 ? u : (u + 1) % PFB, qn = (q + 2) & 3;
          wait_pair(un, qn);
          const Pre xna = issue_x(pq[un][qn], (a0_scalar || PS != 0) ? a0s : wq[un][qn], bx4(bq[un], qn));
          const Pre xnb = issue_x(pq[un][qn + 1], (a0_scalar || PS != 0) ? a0s : wq[un][qn + 1], bx4(bq[un], qn + 1));
          release_to((u + 1) % PFB, q);  // (the pair after that one: next block, same slot)
          prefetch_landed();
          const float p1a = a1_arr ? a1[((size_t)((ib >> 2) + b) * 64 + lane) * 4 + q] : a1s;
          const float p1b = a1_arr ? a1[((size_t)((ib >> 2) + b) * 64 + lane) * 4 + q + 1] : a1s;
          const bool lc2 = decltype(lc_tag)::value == 2;
          const uint32_t hma = lc2 ? (uint32_t)__builtin_amdgcn_readlane((int)hv[u], 4 * q) : 0u;
          const uint32_t hmb = lc2 ? (uint32_t)__builtin_amdgcn_readlane((int)hv[u], 4 * q + 4) : 0u;
          finish(lc_tag, pq[u][q], xa, acc, p1a, hma);
          if (HAS_GRAD) ring_ld<D>(L + GR_OFF + row_of(pq[u][q + 1]), acc);
          finish(lc_tag, pq[u][q + 1], xb, acc, p1b, hmb);
          if (HAS_GRAD) ring_ld<D>(L + GR_OFF + row_of(pq[un][qn]), acc);
          xa = xna;
          xb = xnb;
        }
      };
#pragma unroll
      for (int u = 0; u < PFB; ++u) load_block(u, u);
      release_to(0, 0);  // (a sparse stream may begin chunks after j_lo: the producers must know before this wave waits)
      wait_pair(0, 0);
      xa = issue_x(pq[0][0], (a0_scalar || PS != 0) ? a0s : wq[0][0], bx4(bq[0], 0));
      xb = issue_x(pq[0][1], (a0_scalar || PS != 0) ? a0s : wq[0][1], bx4(bq[0], 1));
      release_to(0, 2);
      prefetch_landed();
      if (HAS_GRAD) ring_ld<D>(L + GR_OFF + row_of(pq[0][0]), acc);
      for (int base = 0; base < NB; base += PFB) {
#pragma unroll
        for (int u = 0; u < PFB; ++u) {
          const int b = base + u;
          if (b < NB) {
            // loss class of the block: header word 3 of its first iteration
            const int bcls = __builtin_amdgcn_readlane((int)hv[u], 3) & 3;
            if (bcls == 0)
              block_body(std::integral_constant<int, 0>(), u, b);
            else if (bcls == 1)
              block_body(std::integral_constant<int, 1>(), u, b);
            else
              block_body(std::integral_constant<int, 2>(), u, b);
          }
          load_block(u, b + PFB);
        }
      }
#if MDE_RING_ABLATE
      if ((dbg & 512) && lane == 0 && blockIdx.x < 1024) {
        atomicAdd(&g_ring_probe[0][blockIdx.x], RING_CLK() - cs_t0);
        atomicAdd(&g_ring_probe[1][blockIdx.x], cs_poll);
        atomicAdd(&g_ring_probe[2][blockIdx.x], cs_trips);
      }
#endif
    }
    if constexpr (Fn::kRingFused) loss = fmaf(loss, 0.6931471805599453f, loss2) * (1.0f / Fn::kParamScale);
    ring_ctrl_store_counted(L, CTRL_PROG + 4u * (uint32_t)wave, MDE_RING_DONE);
    // (the two roles are laid out one after the other: leave no counted load pending here, or
    // hipcc carries the stream prefetches into the producer code as waits -- see above)
    __builtin_amdgcn_s_waitcnt(0x0F70);
  }
  __syncthreads();
  if (HAS_GRAD && slot_row) {
    // permuted row blocks (round 6): the same three endings as below, row by row through slot_row -- slot sl of
    // this block is local row sr[sl] (-1: nobody)
    const int32_t* sr = slot_row + (size_t)rb * R;
    if (fold) {
      unsigned int* sync = reinterpret_cast<unsigned int*>(partial + (size_t)Q * nloc * D) + 2 * rb;
      int* tk = reinterpret_cast<int*>(L + 512);
      if (tid == 0) *tk = (int)__hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      const bool second = *tk != 0;
      float* mine = partial + (size_t)qg * nloc * D;
      const float* other = partial + (size_t)(1 - qg) * nloc * D;
      if (!second) {
        for (int sl = tid; sl < R; sl += BS) {
          const int r = sr[sl];
          if (r < 0) continue;
#pragma unroll
          for (int c = 0; c < D; ++c)
            __hip_atomic_store(mine + (size_t)r * D + c, GR[sl * D + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the rows have completed
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        if (tid == 0) {
          while (__hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(2);
          __hip_atomic_store(&sync[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&sync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        for (int sl = tid; sl < R; sl += BS) {
          const int r = sr[sl];
          if (r < 0) continue;
#pragma unroll
          for (int c = 0; c < D; ++c) {
            const float o = __hip_atomic_load(other + (size_t)r * D + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float m = GR[sl * D + c];
            grad[(size_t)(row_lo + r) * D + c] = (qg == 0 ? m + o : o + m) * grad_scale;  // (group order)
          }
        }
      }
      __syncthreads();
    } else {
      float* dst = (Q == 1) ? grad + (size_t)row_lo * D : partial + (size_t)qg * nloc * D;
      const float sc = (Q == 1) ? grad_scale : 1.0f;
      for (int sl = tid; sl < R; sl += BS) {
        const int r = sr[sl];
        if (r < 0) continue;
#pragma unroll
        for (int c = 0; c < D; ++c) dst[(size_t)r * D + c] = GR[sl * D + c] * sc;
      }
    }
  } else if (HAS_GRAD && fold) {
    // Q == 2, one launch (round 5): the two column groups of a row block meet at a ticket.  The group that
    // arrives FIRST leaves its unscaled rows in `partial` (relaxed device-scope stores: write-through, the
    // XCDs' L2s are not coherent) and raises a flag once they have completed; the SECOND adds them to the
    // rows it still holds in LDS and writes the final gradient rows.  a + b = b + a to the last bit, so
    // the result does not depend on who arrives first and equals k_ring_combine's (partial[0] + partial[1])
    // * scale.  With 16 groups (8-way shards) an in-launch reducer was slower than the second launch both
    // times it was built; with two there is one 63 KB partial to read, and the second launch
    // cost 6.5 us + a launch gap per evaluation.
    unsigned int* sync = reinterpret_cast<unsigned int*>(partial + (size_t)Q * nloc * D) + 2 * rb;
    int* tk = reinterpret_cast<int*>(L + 512);  // (the x_v region is free now)
    if (tid == 0) *tk = (int)__hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const bool second = *tk != 0;
    float* mine = partial + ((size_t)qg * nloc + r0) * D;
    const float* other = partial + ((size_t)(1 - qg) * nloc + r0) * D;
    float* grow = grad + (size_t)(row_lo + r0) * D;
    if (!second) {
      if constexpr (D == 2) {
        for (int i = tid; i < nr; i += BS)
          __hip_atomic_store(reinterpret_cast<double*>(mine) + i, reinterpret_cast<const double*>(GR)[i], __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
      } else {
        for (int i = tid; i < nr * D; i += BS) __hip_atomic_store(mine + i, GR[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the rows have completed
      __syncthreads();
      if (tid == 0) __hip_atomic_store(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (tid == 0) {
        while (__hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(2);
        // (both words back to zero for the next launch)
        __hip_atomic_store(&sync[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      if constexpr (D == 2) {
        for (int i = tid; i < nr; i += BS) {
          const double o = __hip_atomic_load(reinterpret_cast<const double*>(other) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const float2 ov = *reinterpret_cast<const float2*>(&o);
          const float2 mv = reinterpret_cast<const float2*>(GR)[i];
          // (group order, as k_ring_combine adds them)
          float2 r;
          r.x = (qg == 0 ? mv.x + ov.x : ov.x + mv.x) * grad_scale;
          r.y = (qg == 0 ? mv.y + ov.y : ov.y + mv.y) * grad_scale;
          reinterpret_cast<float2*>(grow)[i] = r;
        }
      } else {
        for (int i = tid; i < nr * D; i += BS) {
          const float o = __hip_atomic_load(other + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          grow[i] = (qg == 0 ? GR[i] + o : o + GR[i]) * grad_scale;
        }
      }
    }
    __syncthreads();
  } else if (HAS_GRAD) {
    // Q == 1: the rows are final.  Q > 1: unscaled per-group partials, summed by k_ring_combine
    float* grow = (Q == 1) ? grad + (size_t)(row_lo + r0) * D : partial + ((size_t)qg * nloc + r0) * D;
    const float sc = (Q == 1) ? grad_scale : 1.0f;
    for (int i = tid; i < nr * D; i += BS) grow[i] = GR[i] * sc;
  }
  // block-wide loss partial (the x_v region is free now), then the loss itself: the last
  // workgroup to arrive adds the partials of all of them in a fixed order (no second launch).  The
  // partials travel as relaxed device-scope atomic stores / loads (mde_common.h: a device-scope fence
  // here writes the whole L2 back -- it cost 5 us of the 230 at config 4 and 8 of the 64 us of an
  // 8-way shard's evaluation).
  double* red = reinterpret_cast<double*>(L);
  unsigned int* ticket = reinterpret_cast<unsigned int*>(loss_partials + MDE_MAX_PARTIALS);
  const double v = mde_wave_sum((double)loss);
  if (lane == 0) red[wave] = v;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    for (int i = 0; i < NCW; ++i) s += red[i];
    mde_st_partial(loss_partials + blockIdx.x, s);
  }
  // (mde_last_block's logic with the flag inside L: this kernel's LDS is full)
  int* last_flag = reinterpret_cast<int*>(L + 256);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the partial store has completed
  __syncthreads();
  if (tid == 0) {
    const unsigned int k = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *last_flag = (k == gridDim.x - 1u) ? 1 : 0;
    if (*last_flag) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!*last_flag) return;
  double t = 0.0;
  for (int i = tid; i < (int)gridDim.x; i += BS) t += mde_ld_partial(loss_partials + i);
  t = mde_wave_sum(t);
  __syncthreads();
  if (lane == 0) red[wave] = t;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    for (int i = 0; i < BS / 64; ++i) s += red[i];
    *loss_out = (float)(s * loss_scale);
    loss_partials[MDE_PARTIALS_LOSS_D] = s * loss_scale;  // (the same in double: mde_plan_loss_double)
  }
}

// grad[row] = scale * sum_q partial[q][row]  (q ascending).  V = float4 / float: one element per
// thread, the first eight group loads in flight together
template <class V>
__global__ __launch_bounds__(MDE_BLOCK) void k_ring_combine(int64_t m, int Q, const V* __restrict__ partial,
                                                            float scale, V* __restrict__ out) {
  constexpr int W = sizeof(V) / sizeof(float);
  const int64_t i = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x;
  if (i >= m) return;
  V a[8];
#pragma unroll
  for (int q = 0; q < 8; ++q)
    if (q < Q) a[q] = partial[(size_t)q * m + i];
  float s[W];
#pragma unroll
  for (int k = 0; k < W; ++k) s[k] = reinterpret_cast<const float*>(&a[0])[k];
#pragma unroll
  for (int q = 1; q < 8; ++q)
    if (q < Q) {
#pragma unroll
      for (int k = 0; k < W; ++k) s[k] += reinterpret_cast<const float*>(&a[q])[k];
    }
  for (int q = 8; q < Q; ++q) {
    const V t = partial[(size_t)q * m + i];
#pragma unroll
    for (int k = 0; k < W; ++k) s[k] += reinterpret_cast<const float*>(&t)[k];
  }
  V r;
#pragma unroll
  for (int k = 0; k < W; ++k) reinterpret_cast<float*>(&r)[k] = s[k] * scale;
  out[i] = r;
}


template <int D, class Fn, bool LIN>
static int launch_ring(const RingArgs& A, const Fn& fn, int* nblocks) {
  const mde_ring_layout& L = A.plan->ring;
  const bool cb = A.a0_scalar == 2, bx = A.a0_scalar == 3;
  if ((cb || bx) && D != 2 && D != 3) {
    mde_set_error("codebook / byte-index parameter streams exist for d = 2 and d = 3 only");
    return MDE_E_INVALID;
  }
  if (bx && MDE_RING_LDS_BYTES - (L.ring_off + L.slots * ring_chunk_bytes(D)) < 4 * MDE_RING_BX_VALUES) {
    mde_set_error("byte-index parameter stream: no room for the value table behind the ring");
    return MDE_E_INVALID;
  }
  if (reinterpret_cast<uintptr_t>(A.X) & 15) {
    mde_set_error("the LDS-ring kernel needs a 16-byte aligned embedding matrix");
    return MDE_E_INVALID;
  }
  if constexpr (LIN) {
    // one scalar parameter for every edge: padding lanes would carry it too -- mask them instead
    if (A.a0_scalar == 1) return launch_ring<D, Fn, false>(A, fn, nblocks);
  }
  auto kern = A.grad ? k_fused_ring<D, Fn, true, 0, LIN> : k_fused_ring<D, Fn, false, 0, LIN>;
  if constexpr (D == 2 || D == 3) {
    if (cb) kern = A.grad ? k_fused_ring<D, Fn, true, 1, LIN> : k_fused_ring<D, Fn, false, 1, LIN>;
    if (bx) kern = A.grad ? k_fused_ring<D, Fn, true, 2, LIN> : k_fused_ring<D, Fn, false, 2, LIN>;
  }
  // codebook form: a0 = [H packed words | 8 values]; byte-index form: a0 = [H index bytes | 256 values]
  const uint32_t* stream = cb ? reinterpret_cast<const uint32_t*>(A.a0) : L.packed;
  const uint32_t* bidx = bx ? reinterpret_cast<const uint32_t*>(A.a0) : nullptr;
  const float* a0 = cb ? A.a0 + L.H : (bx ? A.a0 + L.H / 4 : A.a0);
  const int Q = L.col_groups;
  *nblocks = L.n_row_blocks * Q;
  // the accumulators hold sum f'/d (x_v - x_u); 1/p is applied with the output scale (the
  // NaN/Inf -> 1 rule of the reference then reads "-> p")
  const float out_scale = A.grad_scale * A.inv_p;
  const float fix_value = A.inv_p > 0.0f ? 1.0f / A.inv_p : 1.0f;
#if MDE_RING_ABLATE
  const int dbg = getenv("MDE_RING_DBG") ? atoi(getenv("MDE_RING_DBG")) : 0;
#else
  const int dbg = 0;
#endif
  // two column groups per row block: their rows are added inside the launch (MDE_RING_FOLD=0: by k_ring_combine)
  static const int fold_env = getenv("MDE_RING_FOLD") ? atoi(getenv("MDE_RING_FOLD")) : 1;
  const int fold = (Q == 2 && A.grad && fold_env) ? 1 : 0;
  // (every edge adds its loss term once here, not once per endpoint: twice the caller's scale)
  hipLaunchKernelGGL(kern, dim3(L.n_row_blocks * Q), dim3(MDE_RING_BS), 0, A.st,
                     (int)(A.plan->row_hi - A.plan->row_lo), (int)A.plan->row_lo, (int)A.plan->n,
                     L.rows_per_block, Q, L.n_chunks, L.ring_off, L.slots, L.wave_iter, L.hdr, stream, bidx, L.slot_row, a0, A.a1,
                     A.a0_scalar, A.a1_scalar, A.X, A.grad, L.partial, A.plan->partials, fn, fix_value, out_scale,
                     A.loss_out, (L.count_all ? 1.0 : 2.0) * A.loss_scale, fold, dbg);
  MDE_LAUNCH_CHECK();
#if MDE_RING_ABLATE
  {
    static int dl = 0;
    if (++dl == 3) {
      int nd = 0;
      int hd[64][8];
      (void)hipStreamSynchronize(A.st);
      (void)hipMemcpyFromSymbol(&nd, HIP_SYMBOL(g_ring_ndiag), sizeof(int));
      (void)hipMemcpyFromSymbol(hd, HIP_SYMBOL(g_ring_diag), sizeof(hd));
      fprintf(stderr, "[mde ring diag] %d give-ups in the first launches\n", nd);
      for (int k = 0; k < nd && k < 24; ++k)
        fprintf(stderr, hd[k][0] == 1 ? "[mde ring diag] consumer wg %d wave %d: need %d ready %d m %d slot %d NB %d\n"
                                      : "[mde ring diag] producer wg %d p %d: j %d minprog %d infl %d oldest %d j_hi %d\n",
                hd[k][1], hd[k][2], hd[k][3], hd[k][4], hd[k][5], hd[k][6], hd[k][7]);
    }
  }
  if (dbg & 512) {
    static int launches = 0;
    if (++launches == 8) {
      std::vector<unsigned long long> h(8 * 1024);
      (void)hipStreamSynchronize(A.st);
      (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_ring_probe), sizeof(unsigned long long) * 8 * 1024);
      const int nb = std::min(1024, L.n_row_blocks * Q);
      double s[8] = {0};
      for (int k = 0; k < 8; ++k)
        for (int i = 0; i < nb; ++i) s[k] += (double)h[k * 1024 + i];
      const double L8 = 8.0 * nb;  // launches x workgroups
      fprintf(stderr, "[mde ring probe] per consumer wave and launch: loop %.0f clk, in chunk polls %.0f clk (%.1f%%), %.0f poll trips | "
              "per producer wave: loop %.0f clk, blocked on a slot %.0f (%.1f%%), waiting for the chunk's loads %.0f (%.1f%%), %.0f slot polls, LDS stores + ordered publish %.0f clk\n",
              s[0] / L8 / MDE_RING_NCW, s[1] / L8 / MDE_RING_NCW, 100.0 * s[1] / s[0], s[2] / L8 / MDE_RING_NCW,
              s[3] / L8 / MDE_RING_NPROD, s[4] / L8 / MDE_RING_NPROD, 100.0 * s[4] / s[3], s[5] / L8 / MDE_RING_NPROD,
              100.0 * s[5] / s[3], s[6] / L8 / MDE_RING_NPROD, s[7] / L8 / MDE_RING_NPROD);
      // per workgroup: the consumers' loop and the part of it that is NOT polling (is there a tail? do the
      // workgroups that add all the loss terms -- low row blocks, upper column group -- run longer?)
      std::vector<double> loop(nb), work(nb);
      for (int i = 0; i < nb; ++i) {
        loop[i] = (double)h[i] / 8.0 / MDE_RING_NCW;
        work[i] = ((double)h[i] - (double)h[1024 + i]) / 8.0 / MDE_RING_NCW;
      }
      auto quant = [&](std::vector<double> v, const char* what) {
        std::sort(v.begin(), v.end());
        double m = 0;
        for (double x : v) m += x;
        fprintf(stderr, "[mde ring probe] per workgroup, %s: min %.0f p10 %.0f median %.0f mean %.0f p90 %.0f max %.0f clk\n", what, v[0],
                v[v.size() / 10], v[v.size() / 2], m / v.size(), v[v.size() * 9 / 10], v.back());
      };
      quant(loop, "consumer loop");
      quant(work, "consumer loop minus polls");
      for (int q8 = 0; q8 < 8; ++q8) {
        double a = 0, b2 = 0;
        int c = 0;
        for (int i = q8 * nb / 8; i < (q8 + 1) * nb / 8; ++i, ++c) {
          a += loop[i];
          b2 += work[i];
        }
        fprintf(stderr, "[mde ring probe] workgroups %d..%d: loop %.0f, minus polls %.0f\n", q8 * nb / 8, (q8 + 1) * nb / 8 - 1, a / c, b2 / c);
      }
    }
  }
#endif
  // Q column groups per row block (sharded plans): the per-group partials are added by a second, 6 us
  // launch.  Folding that into the ring kernel (the last group of a row block to arrive adds the Q
  // partials) was built twice -- round 2 with a fence, round 3 with write-through stores and relaxed
  // loads -- and measured slower both times (8-way shard: 77 vs 57 us per evaluation): 8 x 32 KB of
  // partials per row block are an order of magnitude more than an in-launch reducer reads for free.
  if (Q > 1 && A.grad && !fold) {
    const int64_t nlocD = (A.plan->row_hi - A.plan->row_lo) * (int64_t)D;
    float* out = A.grad + (size_t)A.plan->row_lo * D;
    if (nlocD % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
      const int64_t m = nlocD / 4;
      hipLaunchKernelGGL(k_ring_combine<float4>, dim3((unsigned)((m + MDE_BLOCK - 1) / MDE_BLOCK)), dim3(MDE_BLOCK),
                         0, A.st, m, Q, reinterpret_cast<const float4*>(L.partial), out_scale,
                         reinterpret_cast<float4*>(out));
    } else {
      hipLaunchKernelGGL(k_ring_combine<float>, dim3((unsigned)((nlocD + MDE_BLOCK - 1) / MDE_BLOCK)),
                         dim3(MDE_BLOCK), 0, A.st, nlocD, Q, L.partial, out_scale, out);
    }
    MDE_LAUNCH_CHECK();
  }
  return MDE_OK;
}
