// gmk_api.hip -- version / error strings.
#include "gmk_common.h"
#include "plan.h"

extern "C" int gmk_version(void) { return 1; }

extern "C" const char* gmk_strerror(int code) {
  switch (code) {
    case GMK_OK: return "ok";
    case GMK_E_SHAPE: return "gmk: unsupported or inconsistent shape";
    case GMK_E_ALIGN: return "gmk: pointer is not 16-byte aligned";
    case GMK_E_UNSUPPORTED: return "gmk: unsupported option";
    case GMK_E_WORKSPACE: return "gmk: workspace too small";
    case GMK_E_NULL: return "gmk: required pointer is NULL";
    case GMK_E_STATE: return "gmk: call out of order (plan recording already active / not active / still recording)";
    case GMK_E_TOO_LARGE: return "gmk: input exceeds the stated maximum of this allocation-free entry point";
    default: break;
  }
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "gmk: unknown error";
}

// Host plumbing of the two-stream backward: `waiter` waits for everything enqueued so far on `on`.  One event record +
// one stream wait from C (the same pair through torch.cuda.Event costs ~15 us of Python per call, 13 calls per train
// step).  Events come from a per-device ring created on first use; re-recording an event does not disturb a wait that
// was enqueued on its previous record.
#include <mutex>
#define GMK_EVENT_RING 64
#define GMK_MAX_DEVICES 16
// next event of the fork ring of the device that owns `on` (an event must be recorded on a stream of its own device;
// the NULL stream belongs to the current device)
static int ring_event(hipStream_t on, hipEvent_t* out) {
  static hipEvent_t ring[GMK_MAX_DEVICES][GMK_EVENT_RING];
  static bool made[GMK_MAX_DEVICES];
  static unsigned next[GMK_MAX_DEVICES];
  static std::mutex mu;
  int dev = 0;
  hipError_t e = on ? hipStreamGetDevice(on, &dev) : hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  GMK_REQUIRE(dev >= 0 && dev < GMK_MAX_DEVICES, GMK_E_UNSUPPORTED);
  std::lock_guard<std::mutex> lock(mu);
  if (!made[dev]) {
    int cur = 0;
    e = hipGetDevice(&cur);
    if (e != hipSuccess) return (int)e;
    if (cur != dev && (e = hipSetDevice(dev)) != hipSuccess) return (int)e;   // events are created on the current device
    // No SYSTEM-scope fence on these events: they only ever order streams of ONE device against each other, and every
    // kernel dispatch carries its own agent-scope acquire / release.  The default record writes the L2s back and
    // invalidates them for the host's and other devices' sake, ~25 times per train step: 0.972 -> 0.944 ms per step
    // (GMK_EVENT_SCOPE=system restores it for an A/B; hipEventReleaseToDevice alone: no gain).  Whatever leaves the
    // device afterwards -- a host read, an RCCL collective -- goes through torch's own events and copies, which fence.
    unsigned flags = hipEventDisableTiming | hipEventDisableSystemFence;
    if (const char* sc = getenv("GMK_EVENT_SCOPE")) {
      if (sc[0] == 's') flags = hipEventDisableTiming;
      else if (sc[0] == 'd') flags = hipEventDisableTiming | hipEventReleaseToDevice;
    }
    for (int i = 0; i < GMK_EVENT_RING; ++i) {
      e = hipEventCreateWithFlags(&ring[dev][i], flags);
      if (e != hipSuccess) return (int)e;
    }
    made[dev] = true;
    if (cur != dev && (e = hipSetDevice(cur)) != hipSuccess) return (int)e;
  }
  *out = ring[dev][next[dev]++ % GMK_EVENT_RING];
  return GMK_OK;
}

extern "C" int gmk_stream_wait_stream(gmk_stream_t waiter, gmk_stream_t on) {
  GMK_PLAN_HOOK(gmk_stream_wait_stream, waiter, on);
  // (Binding the event to the forking kernel itself -- hipExtLaunchKernelGGL's stop event, so that no marker follows the
  // kernel -- was measured in round 3: 1.004 -> 1.001 ms per step; not worth a second launch path.)
  hipEvent_t ev;
  const int rc = ring_event((hipStream_t)on, &ev);
  if (rc != GMK_OK) return rc;
  hipError_t e = hipEventRecord(ev, (hipStream_t)on);
  if (e != hipSuccess) return (int)e;
  e = hipStreamWaitEvent((hipStream_t)waiter, ev, 0);
  return e == hipSuccess ? GMK_OK : (int)e;
}

// hipFuncSetAttribute costs ~2 us of host time; the launchers of kernels with more than 64 KiB of dynamic LDS used to call
// it on every launch.  One table entry per (kernel, device): the largest limit set so far.
#include <map>
hipError_t gmk_ensure_dyn_lds(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, int> done;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  int& have = done[std::make_pair(kernel, dev)];
  if (have >= bytes) return hipSuccess;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) have = bytes;
  return e;
}
