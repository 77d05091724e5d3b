// Launch plans and error plumbing of libddx_hip.
//
// A plan is a recorded list of kernel launches (closures over plain descriptors).  It is replayed with ONE FFI call,
// either eagerly or as a hipGraph captured from that replay -- the MI355X stand-in for the reference's
// torch.compile(fullgraph=True) of UNet.forward (reference src/modules/module.py:145-149): shapes are static, so the
// whole forward is one graph launch with no per-kernel host work.
#include <cstdlib>
#include <string>
#include <vector>

#include "common.hpp"

struct ddx_op_meta { const char* tag; double flops, bytes; };

// Two lanes: ops recorded between ddx_plan_fork() and ddx_plan_main() run on a side stream that first waits for
// everything recorded before the fork; ddx_plan_join() makes the main stream wait for the side stream.  Under graph capture
// this produces parallel branches (independent small kernels of one block overlap instead of queueing).
enum { DDX_LANE_MAIN = 0, DDX_LANE_SIDE = 1, DDX_MARK_FORK = 2, DDX_MARK_JOIN = 3 };

struct ddx_plan {
  std::vector<ddx::LaunchFn> ops;
  std::vector<ddx_op_meta> meta;
  std::vector<int> kind;  // per op: lane of a launch, or a fork / join marker (ops[i] is empty for markers)
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool recording = false;
  int lane = DDX_LANE_MAIN;
  bool forked = false;
};

namespace ddx {

static thread_local ddx_plan* g_recording = nullptr;
static thread_local std::string g_last_error;

int set_error(int code, const char* msg) {
  g_last_error = msg ? msg : "";
  return code;
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return DDX_ERR_LAUNCH;
  }
  return DDX_OK;
}

int dispatch(LaunchFn&& fn, ddx_stream stream, const char* tag, double flops, double bytes) {
  if (g_recording) {
    g_recording->ops.emplace_back(std::move(fn));
    g_recording->meta.push_back(ddx_op_meta{tag, flops, bytes});
    g_recording->kind.push_back(g_recording->lane);
    return DDX_OK;
  }
  return fn(reinterpret_cast<hipStream_t>(stream));
}

}  // namespace ddx

extern "C" const char* ddx_version(void) { return "libddx_hip 0.1 (gfx950)"; }

// sizeof / offset of the last field of the header's descriptor structs as compiled (bindings check their mirrors against these)
#include <cstddef>
extern "C" int64_t ddx_abi_sizeof(int32_t which) {
  switch (which) {
    case 0: return sizeof(ddx_wprep_desc);
    case 1: return sizeof(ddx_conv_desc);
    case 2: return sizeof(ddx_dgrad_act_desc);
    case 3: return sizeof(ddx_wgrad_desc);
    case 4: return sizeof(ddx_linear_bwd_job);
    case 5: return sizeof(ddx_wpath_job);
    case 6: return sizeof(ddx_linear_job);
    case 7: return sizeof(ddx_melstft_desc);
    case 8: return sizeof(ddx_msmel_desc);
    case 9: return sizeof(ddx_bgemm_desc);
    case 10: return sizeof(ddx_mss_desc);
    case 11: return sizeof(ddx_optim_job);
    case 12: return sizeof(ddx_optim_job_ex);
    case 13: return sizeof(ddx_conv_pair_desc);
    default: return -1;
  }
}
extern "C" int64_t ddx_abi_offsetof_tail(int32_t which) {
  switch (which) {
    case 0: return offsetof(ddx_wprep_desc, rows_total);
    case 1: return offsetof(ddx_conv_desc, out_head_eps);
    case 2: return offsetof(ddx_dgrad_act_desc, scale1);
    case 3: return offsetof(ddx_wgrad_desc, accumulate);
    case 4: return offsetof(ddx_linear_bwd_job, groups);
    case 5: return offsetof(ddx_wpath_job, reserved);
    case 6: return offsetof(ddx_linear_job, normalize);
    case 7: return offsetof(ddx_melstft_desc, scale);
    case 8: return offsetof(ddx_msmel_desc, offset);
    case 9: return offsetof(ddx_bgemm_desc, alpha);
    case 10: return offsetof(ddx_mss_desc, stats);
    case 11: return offsetof(ddx_optim_job, n);
    case 12: return offsetof(ddx_optim_job_ex, reserved);
    case 13: return offsetof(ddx_conv_pair_desc, out2_scale);
    default: return -1;
  }
}
extern "C" const char* ddx_last_error(void) { return ddx::g_last_error.c_str(); }

extern "C" ddx_plan* ddx_plan_begin(void) {
  if (ddx::g_recording) { ddx::set_error(DDX_ERR_ARG, "plan_begin: already recording"); return nullptr; }
  ddx_plan* p = new ddx_plan();
  p->recording = true;
  ddx::g_recording = p;
  return p;
}

static void plan_mark(ddx_plan* p, int kind) {
  p->ops.emplace_back(ddx::LaunchFn());
  p->meta.push_back(ddx_op_meta{kind == DDX_MARK_FORK ? "fork" : "join", 0.0, 0.0});
  p->kind.push_back(kind);
}

extern "C" int ddx_plan_fork(void) {
  ddx_plan* p = ddx::g_recording;
  if (!p) return DDX_OK;  // not recording: everything runs in issue order on the caller's stream
  if (p->forked) return ddx::set_error(DDX_ERR_ARG, "plan_fork: already forked (join first)");
  plan_mark(p, DDX_MARK_FORK);
  p->forked = true;
  p->lane = DDX_LANE_SIDE;
  return DDX_OK;
}

extern "C" int ddx_plan_main(void) {
  if (ddx::g_recording) ddx::g_recording->lane = DDX_LANE_MAIN;
  return DDX_OK;
}

extern "C" int ddx_plan_join(void) {
  ddx_plan* p = ddx::g_recording;
  if (!p) return DDX_OK;
  if (!p->forked) return DDX_OK;
  plan_mark(p, DDX_MARK_JOIN);
  p->forked = false;
  p->lane = DDX_LANE_MAIN;
  return DDX_OK;
}

// replay every op in order; launches of the side lane go to the plan's side stream, markers become event edges
// Append the launches of a finished plan to the plan being recorded (a sampler step = [glue, UNet forward, glue, UNet forward, glue]
// as ONE plan / hipGraph).  The closures are copied: `src` may be replayed or destroyed independently afterwards, the buffers its
// launches point into must outlive both.
extern "C" int ddx_plan_include(const ddx_plan* src) {
  ddx_plan* p = ddx::g_recording;
  if (!p) return ddx::set_error(DDX_ERR_ARG, "plan_include: no plan is being recorded");
  if (!src || src->recording || src == p) return ddx::set_error(DDX_ERR_ARG, "plan_include: bad source plan");
  for (size_t i = 0; i < src->ops.size(); ++i) {
    if (src->kind[i] == DDX_MARK_FORK || src->kind[i] == DDX_MARK_JOIN) {
      if (p->forked && src->kind[i] == DDX_MARK_FORK) return ddx::set_error(DDX_ERR_ARG, "plan_include: fork inside a fork");
      p->forked = src->kind[i] == DDX_MARK_FORK;
    }
    p->ops.push_back(src->ops[i]);
    p->meta.push_back(src->meta[i]);
    p->kind.push_back(src->kind[i]);
  }
  return DDX_OK;
}

static int plan_replay(ddx_plan* p, hipStream_t s, bool lanes) {
  if (lanes && !p->side) {
    bool any = false;
    for (int k : p->kind) any = any || k == DDX_MARK_FORK;
    if (any) {
      if (hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking) != hipSuccess ||
          hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming) != hipSuccess)
        return ddx::set_error(DDX_ERR_LAUNCH, "plan: side stream / events");
    }
  }
  for (size_t i = 0; i < p->ops.size(); ++i) {
    const int k = p->kind[i];
    if (k == DDX_MARK_FORK) {
      if (lanes && (hipEventRecord(p->ev_fork, s) != hipSuccess || hipStreamWaitEvent(p->side, p->ev_fork, 0) != hipSuccess))
        return ddx::set_error(DDX_ERR_LAUNCH, "plan: fork");
      continue;
    }
    if (k == DDX_MARK_JOIN) {
      if (lanes && (hipEventRecord(p->ev_join, p->side) != hipSuccess || hipStreamWaitEvent(s, p->ev_join, 0) != hipSuccess))
        return ddx::set_error(DDX_ERR_LAUNCH, "plan: join");
      continue;
    }
    const int rc = p->ops[i]((lanes && k == DDX_LANE_SIDE) ? p->side : s);
    if (rc != DDX_OK) return rc;
  }
  return DDX_OK;
}

extern "C" int ddx_plan_end(ddx_plan* p) {
  if (!p || ddx::g_recording != p) return ddx::set_error(DDX_ERR_ARG, "plan_end: not the recording plan");
  if (p->forked) { plan_mark(p, DDX_MARK_JOIN); p->forked = false; p->lane = DDX_LANE_MAIN; }
  p->recording = false;
  ddx::g_recording = nullptr;
  return DDX_OK;
}

extern "C" int ddx_plan_num_ops(const ddx_plan* p) { return p ? (int)p->ops.size() : -1; }

extern "C" int ddx_plan_run(ddx_plan* p, ddx_stream stream) {
  if (!p || p->recording) return ddx::set_error(DDX_ERR_ARG, "plan_run: bad plan");
  return plan_replay(p, reinterpret_cast<hipStream_t>(stream), true);
}

extern "C" int ddx_plan_graph_build(ddx_plan* p, ddx_stream stream) {
  if (!p || p->recording) return ddx::set_error(DDX_ERR_ARG, "plan_graph_build: bad plan");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (p->exec) { (void)hipGraphExecDestroy(p->exec); p->exec = nullptr; }
  if (p->graph) { (void)hipGraphDestroy(p->graph); p->graph = nullptr; }
  if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess)
    return ddx::set_error(DDX_ERR_LAUNCH, "plan_graph_build: hipStreamBeginCapture");
  const int rc = plan_replay(p, s, true);
  hipGraph_t g = nullptr;
  const hipError_t e = hipStreamEndCapture(s, &g);
  if (rc != DDX_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
  if (e != hipSuccess || !g) return ddx::set_error(DDX_ERR_LAUNCH, "plan_graph_build: hipStreamEndCapture");
  hipGraphExec_t ex = nullptr;
  if (hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) != hipSuccess) {
    (void)hipGraphDestroy(g);
    return ddx::set_error(DDX_ERR_LAUNCH, "plan_graph_build: hipGraphInstantiate");
  }
  p->graph = g;
  p->exec = ex;
  return DDX_OK;
}

extern "C" int ddx_plan_graph_launch(ddx_plan* p, ddx_stream stream) {
  if (!p || !p->exec) return ddx::set_error(DDX_ERR_ARG, "plan_graph_launch: graph not built");
  if (hipGraphLaunch(p->exec, reinterpret_cast<hipStream_t>(stream)) != hipSuccess)
    return ddx::set_error(DDX_ERR_LAUNCH, "plan_graph_launch: hipGraphLaunch");
  return DDX_OK;
}

extern "C" int ddx_plan_op_info(const ddx_plan* p, int i, const char** tag, double* flops, double* bytes) {
  if (!p || i < 0 || i >= (int)p->ops.size()) return ddx::set_error(DDX_ERR_ARG, "plan_op_info: bad index");
  if (tag) *tag = p->meta[i].tag;
  if (flops) *flops = p->meta[i].flops;
  if (bytes) *bytes = p->meta[i].bytes;
  return DDX_OK;
}

// Eager replay with a hipEvent pair around every op, on the launch stream; ms_out[i] = mean milliseconds of op i.
extern "C" int ddx_plan_profile(ddx_plan* p, ddx_stream stream, int reps, float* ms_out) {
  if (!p || p->recording || reps <= 0 || !ms_out) return ddx::set_error(DDX_ERR_ARG, "plan_profile: bad args");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t n = p->ops.size();
  std::vector<hipEvent_t> ev(n + 1);
  for (auto& e : ev)
    if (hipEventCreate(&e) != hipSuccess) return ddx::set_error(DDX_ERR_LAUNCH, "plan_profile: hipEventCreate");
  for (size_t i = 0; i < n; ++i) ms_out[i] = 0.f;
  int rc = DDX_OK;
  for (int r = 0; r < reps && rc == DDX_OK; ++r) {
    (void)hipEventRecord(ev[0], s);
    for (size_t i = 0; i < n; ++i) {  // lanes serialised: per-op durations, not the overlapped schedule
      if (p->ops[i]) rc = p->ops[i](s);
      if (rc != DDX_OK) break;
      (void)hipEventRecord(ev[i + 1], s);
    }
    if (rc != DDX_OK) break;
    if (hipStreamSynchronize(s) != hipSuccess) { rc = ddx::set_error(DDX_ERR_LAUNCH, "plan_profile: sync"); break; }
    for (size_t i = 0; i < n; ++i) {
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
      ms_out[i] += ms / (float)reps;
    }
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  return rc;
}

extern "C" void ddx_plan_destroy(ddx_plan* p) {
  if (!p) return;
  if (ddx::g_recording == p) ddx::g_recording = nullptr;
  if (p->exec) (void)hipGraphExecDestroy(p->exec);
  if (p->graph) (void)hipGraphDestroy(p->graph);
  if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
  if (p->ev_join) (void)hipEventDestroy(p->ev_join);
  if (p->side) (void)hipStreamDestroy(p->side);
  delete p;
}
