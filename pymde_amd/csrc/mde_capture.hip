// mde_capture.hip -- HIP-graph capture of a launch sequence.
//
// One L-BFGS iteration of a mid-size problem is ~25 short kernels (direction update, trial point,
// retraction, fused distortion, tangent projection, statistics) and two read-backs; enqueued one
// by one from the host they cost more host time than GPU time (0.18 ms per iteration at n = 70k
// against ~0.1 ms of kernels).  The solver therefore records the sequence of the usual iteration
// (previous step accepted at t = 1, first trial at t = 1) once per buffer parity with stream
// capture -- the very same entry points enqueue the very same kernels -- and replays the
// instantiated graph with one launch.  [ref: the loop being accelerated is pymde/optim.py:100-175
// + pymde/lbfgs.py:390-590; the reference has no counterpart of this file.]
#include "mde_common.h"

struct mde_capture {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
};

extern "C" int mde_capture_begin(void* stream) {
  hipStream_t st = mde_stream(stream);
  if (!st) {
    mde_set_error("mde_capture_begin: the legacy default stream cannot be captured; use a created stream");
    return MDE_E_INVALID;
  }
  MDE_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
  return MDE_OK;
}

extern "C" int mde_capture_end(void* stream, mde_capture** out) {
  if (!out) return MDE_E_INVALID;
  *out = nullptr;
  hipStream_t st = mde_stream(stream);
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture(st, &graph);
  if (e != hipSuccess || !graph) {
    if (graph) (void)hipGraphDestroy(graph);
    return mde_hip_fail(e != hipSuccess ? e : hipErrorUnknown, "hipStreamEndCapture", __FILE__, __LINE__);
  }
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(graph);
    return mde_hip_fail(e, "hipGraphInstantiate", __FILE__, __LINE__);
  }
  mde_capture* c = new mde_capture();
  c->graph = graph;
  c->exec = exec;
  *out = c;
  return MDE_OK;
}

// leave capture mode without keeping anything (error paths)
extern "C" int mde_capture_abort(void* stream) {
  hipStream_t st = mde_stream(stream);
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &status) == hipSuccess && status != hipStreamCaptureStatusNone) {
    hipGraph_t graph = nullptr;
    (void)hipStreamEndCapture(st, &graph);
    if (graph) (void)hipGraphDestroy(graph);
  }
  (void)hipGetLastError();
  return MDE_OK;
}

extern "C" int mde_capture_launch(mde_capture* c, void* stream) {
  if (!c || !c->exec) return MDE_E_INVALID;
  MDE_HIP(hipGraphLaunch(c->exec, mde_stream(stream)));
  return MDE_OK;
}

extern "C" int mde_capture_destroy(mde_capture* c) {
  if (!c) return MDE_OK;
  if (c->exec) (void)hipGraphExecDestroy(c->exec);
  if (c->graph) (void)hipGraphDestroy(c->graph);
  delete c;
  return MDE_OK;
}

// device -> pinned host copy on the stream (a graph node when the stream is being captured)
extern "C" int mde_copy_to_host(void* dst_host, const void* src_dev, int64_t bytes, void* stream) {
  if (!dst_host || !src_dev || bytes < 0) return MDE_E_INVALID;
  if (bytes == 0) return MDE_OK;
  MDE_HIP(hipMemcpyAsync(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost, mde_stream(stream)));
  return MDE_OK;
}
