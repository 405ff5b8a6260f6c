// mde_ring.h -- geometry of the LDS-ring kernel (layout 1), shared by the layout builder and the
// dispatcher (mde_ring.hip), the kernel template (mde_ring_kernel.h) and the translation units that
// instantiate it (mde_ring_k_*.hip).
#pragma once
#include "mde_common.h"
#include "mde_functions.h"
#include "mde_plan.h"

// LDS map (bytes) of the kernel, per embedding dimension d:
//   [0, XCAP)              x_v of the block's rows (R rows + 32 dummy rows for padding lanes)
//   [XCAP, XCAP + 256)     control words: prog[16], F[8], parameter codebook[8]
//   [XCAP + 256, ...)      gradient accumulators, same slots as x_v (GR_OFF = XCAP + 256)
//   [ring_off, 160 KB)     the chunk ring: S slots of CBYTES bytes (ring_off and S depend on R)
// XCAP (= the tallest row block) is a compile-time constant per d chosen so that GR_OFF + row
// address stays below 2^16: the accumulator and control accesses carry their region base as the
// 16-bit immediate offset of the LDS instruction.  The packed words hold ABSOLUTE LDS addresses.
// Round 4: row blocks twice as tall as round 3's (the staging volume -- every workgroup streams
// the whole table past its rows -- is what the kernel's duration followed), which leaves ~36 KB of
// ring at d = 2; the chunks in flight therefore live in the producers' VGPRs, not in ring slots.
#define MDE_RING_LDS_BYTES 163840
// (round 5: 7816 / 5200 rows at d = 2 / 3 instead of 7872 / 5216 -- 1M rows are still 128 blocks at d = 2, evenly
// filled now (127 x 7872 left a last block of 256 rows), and the 1 KB this frees behind the ring holds the
// 256-entry value table of the byte-index parameter stream)
__host__ __device__ constexpr int ring_row_cap(int d) { return d == 1 ? 12288 : (d == 2 ? 7816 : (d == 3 ? 5200 : 3936)); }
__host__ __device__ constexpr int ring_ctrl_off(int d) { return (ring_row_cap(d) + 32) * 4 * d; }
__host__ __device__ constexpr int ring_gr_off(int d) { return ring_ctrl_off(d) + 256; }
#define MDE_RING_CTRL_PROG(d) (ring_ctrl_off(d))        // int prog[16]: oldest chunk consumer w still reads
#define MDE_RING_CTRL_F(d) (ring_ctrl_off(d) + 64)      // int F[NPROD]: next chunk of producer p that has not landed
#define MDE_RING_CTRL_CB(d) (ring_ctrl_off(d) + 96)     // float[8]: parameter codebook
#ifndef MDE_RING_NCW
#define MDE_RING_NCW 8             // consumer waves (two per SIMD; with the 4 producers every SIMD holds 3 waves)
#endif
#ifndef MDE_RING_CSLEEP
#define MDE_RING_CSLEEP 1          // s_sleep argument of a consumer waiting for a chunk (x 64 clocks)
#endif
#ifndef MDE_RING_PSLEEP
#define MDE_RING_PSLEEP 1          // s_sleep argument of a producer waiting for a slot
#endif
#ifndef MDE_RING_PRODPRIO
#define MDE_RING_PRODPRIO 2        // (round 4: 0.208 ms at priority 3 against 0.192 at 0 -- a producer poll was a dozen VALU instructions then and took the consumers' issue slots.  Round 5, with the one-word hand-shake: how fast a producer REACTS to a released slot is on the critical path -- 0.195 / 0.184 / 0.181 / 0.183 ms at priority 0 / 1 / 2 / 3, and 0.216 when a blocked producer sleeps 12 x 64 clocks instead of 1)
#endif
#ifndef MDE_RING_CONSPRIO
#define MDE_RING_CONSPRIO 0
#endif
#ifndef MDE_RING_NPROD
#define MDE_RING_NPROD 4           // producer waves
#endif
#define MDE_RING_BS (64 * (MDE_RING_NCW + MDE_RING_NPROD))
#ifndef MDE_RING_DEPTH
#define MDE_RING_DEPTH 4           // chunks in flight (in VGPRs) per producer wave (round 5, with hand-placed waits: 2 / 3 / 4 measure 0.213 / 0.207 / 0.201 ms on the fp32 stream, 0.180 all three on the codebook stream)
#endif
#ifndef MDE_RING_UNIT
#define MDE_RING_UNIT 1            // consecutive chunks a producer step moves (one slot poll, one publish)
#endif
#ifndef MDE_RING_ASMLOAD
#define MDE_RING_ASMLOAD 1         // the producers' chunk loads as inline asm with hand-placed vmcnt waits (0: plain loads)
#endif
#ifndef MDE_RING_PFB
#define MDE_RING_PFB 3             // stream blocks (4 iterations each) in flight per consumer wave
#endif
#define MDE_RING_CB_VALUES 8
#define MDE_RING_BX_VALUES 256      // entries of the byte-index stream's value table (entry 0 = the padding lanes' weight 0)
#ifndef MDE_RING_ABLATE
#define MDE_RING_ABLATE 0
#endif
#define MDE_RING_DONE 0x7fffffff

// chunk geometry per embedding dimension: CBYTES bytes (PIECES x 1 KiB) per chunk
#ifndef MDE_RING_C2
#define MDE_RING_C2 512
#endif
#ifndef MDE_RING_C3
#define MDE_RING_C3 256
#endif
__host__ __device__ constexpr int ring_chunk_cols(int d) { return d == 1 ? 1024 : (d == 2 ? MDE_RING_C2 : (d == 3 ? MDE_RING_C3 : 256)); }
__host__ __device__ constexpr int ring_chunk_bytes(int d) { return ring_chunk_cols(d) * 4 * d; }
// ring placement for row blocks of R rows
__host__ __device__ constexpr int ring_off_for(int d, int R) { return (ring_gr_off(d) + (R + 32) * 4 * d + 255) / 256 * 256; }
__host__ __device__ constexpr int ring_slots_for(int d, int R) {
  return (MDE_RING_LDS_BYTES - ring_off_for(d, R)) / ring_chunk_bytes(d) > 32 ? 32
                                                                               : (MDE_RING_LDS_BYTES - ring_off_for(d, R)) / ring_chunk_bytes(d);
}
// a PAIR of iterations may reference chunks m .. m + span; the ring also needs slack for the
// spread between the consumer waves (a slot is free once ALL of them are past its chunk)
__host__ __device__ constexpr int ring_max_span(int S) {
#ifdef MDE_RING_SPAN
  return MDE_RING_SPAN;
#endif
  // (config 4 at 9 slots: windows of 4 / 5 / 6 chunks measure 0.188 / 0.190 / 0.194 ms, 3 costs padding: 0.201)
  return S - 5 > 6 ? 6 : (S - 5 < 2 ? 2 : S - 5);
}

// header word of a wave iteration: [15:0] m = lowest chunk referenced, [20:16] span (highest = m +
// span), [26:21] DPP fold rounds (longest run of equal rows - 1), [27] the iteration has padding
#define MDE_RING_HDR(m, span, rounds, pad) ((uint32_t)(m) | ((uint32_t)(span) << 16) | ((uint32_t)(rounds) << 21) | ((uint32_t)(pad) << 27))

// Header of a wave iteration, four words (scalar registers in the kernel: no field extraction):
//   [0] m    = oldest chunk this or a later iteration of the stream still reads
//   [1] need = newest chunk this iteration reads
//   [2] whose loss terms are added here -- every edge sits in two streams (once per endpoint) and
//       its loss term is added by the entry whose row is the smaller vertex: 0 = no entry of the
//       iteration, 1 = every entry, 2 = mixed (the kernel tests v < u per lane; only the
//       iterations around the diagonal)
//   [3] bits 1:0 (first iteration of a block of four only): loss class of the BLOCK -- 0 / 1 when every
//       non-empty iteration of the block has that class, else 2 (the kernel branches once per block);
//       bit 2: the iteration has padding lanes; bits 15:8: its number of entries
#define MDE_RING_HW 4
#define MDE_RING_H3_PAD 4u

// Packed word of an entry (absolute LDS byte addresses, fixed when the layout is built):
//   d = 2:  [2:0] parameter codebook index (0 when the parameters stream as fp32)
//           [15:3] x_v address / 8 (the accumulator sits GR_OFF above it)   [30:16] x_u address / 8
//   else:   [15:2] x_v address / 4, [1:0] codebook index (d = 3 only)       [31:16] x_u address / 4
__host__ __device__ inline uint32_t ring_pack_word(int d, uint32_t rowaddr, uint32_t coladdr) {
  return d == 2 ? (rowaddr | ((coladdr >> 3) << 16)) : (rowaddr | ((coladdr >> 2) << 16));
}

// ---------------------------------------------------------------- launching
struct RingArgs {
  mde_plan* plan;
  const float* X;
  int d;
  const float *a0, *a1;
  int a0_scalar, a1_scalar;
  float* grad;
  float inv_p, grad_scale;
  hipStream_t st;
  float* loss_out;
  double loss_scale;
};

inline MdeFuncArgs ring_func_args(const mde_func* f) {
  MdeFuncArgs a;
  a.kind = f->kind;
  a.kind_neg = f->kind_neg;
  a.S = {f->s0, f->s1, f->s2};
  a.N = {f->n0, f->n1, f->n2};
  return a;
}

// The instantiations of k_fused_ring live in one translation unit per family of distortion functions:
// HIP loads a unit's code object on the first launch of one of its kernels, and a process uses one
// family -- with everything in one unit the first launch paid ~22 ms for 15 MB of code it never ran
// (round 4: that was the first kernel of the LAYOUT BUILD, which shared the unit).  Each returns 1 when
// it launched, 0 when the function is not one of its own (d outside 2..3 for the compile-time
// families), < 0 on error.
int mde_ring_launch_log1p(const RingArgs& A, const mde_func* f, int* nblocks);     // mde_ring_k_log1p.hip
int mde_ring_launch_pushpull(const RingArgs& A, const mde_func* f, int* nblocks);  // mde_ring_k_pushpull.hip
int mde_ring_launch_penalty(const RingArgs& A, const mde_func* f, int* nblocks);   // mde_ring_k_penalty.hip
int mde_ring_launch_loss(const RingArgs& A, const mde_func* f, int* nblocks);      // mde_ring_k_loss.hip
int mde_ring_launch_penalty2(const RingArgs& A, const mde_func* f, int* nblocks);  // mde_ring_k_penalty2.hip
int mde_ring_launch_loss2(const RingArgs& A, const mde_func* f, int* nblocks);     // mde_ring_k_loss2.hip
int mde_ring_launch_runtime(const RingArgs& A, const mde_func* f, int* nblocks);   // mde_ring_k_runtime.hip
