// plan.hip -- C-side launch plans: the train step's ~60 kernel launches and ~25 stream fork/join edges recorded once and
// replayed from ONE C call per step (per gradient bucket under data parallelism), instead of ~13 us of Python + ctypes per
// launch.  Host code only; see plan.h for how a call is recorded and include/gmk.h for the contract.
#include "gmk_common.h"
#include "plan.h"
#include <vector>

struct gmk_plan {
  std::vector<std::function<int(const gmk_plan_detail::Subst&)>> calls;
  std::vector<int> seg_end;  // seg_end[k] = index one past the last call of segment k
};

static thread_local gmk_plan* t_rec = nullptr;   // plan being recorded on this host thread
static thread_local bool t_replaying = false;    // replayed calls are not recorded again

namespace gmk_plan_detail {
bool recording() { return t_rec != nullptr && !t_replaying; }
void push(std::function<int(const Subst&)> call) { t_rec->calls.push_back(std::move(call)); }
}  // namespace gmk_plan_detail

extern "C" int gmk_plan_begin(gmk_plan** plan) {
  GMK_REQUIRE(plan, GMK_E_NULL);
  GMK_REQUIRE(t_rec == nullptr, GMK_E_STATE);
  *plan = t_rec = new gmk_plan();
  return GMK_OK;
}

extern "C" int gmk_plan_mark(void) {
  GMK_REQUIRE(t_rec != nullptr, GMK_E_STATE);
  t_rec->seg_end.push_back((int)t_rec->calls.size());
  return (int)t_rec->seg_end.size() - 1;
}

extern "C" int gmk_plan_end(void) {
  GMK_REQUIRE(t_rec != nullptr, GMK_E_STATE);
  if (t_rec->seg_end.empty() || t_rec->seg_end.back() != (int)t_rec->calls.size())
    t_rec->seg_end.push_back((int)t_rec->calls.size());
  t_rec = nullptr;
  return GMK_OK;
}

extern "C" int gmk_plan_segments(const gmk_plan* plan) { return plan ? (int)plan->seg_end.size() : GMK_E_NULL; }
extern "C" int gmk_plan_calls(const gmk_plan* plan) { return plan ? (int)plan->calls.size() : GMK_E_NULL; }

extern "C" void gmk_plan_destroy(gmk_plan* plan) {
  if (plan == t_rec) t_rec = nullptr;
  delete plan;
}

extern "C" int gmk_train_step_enqueue(const gmk_plan* plan, int seg_begin, int seg_end, int nsub, const void* const* from,
                                      const void* const* to, const size_t* bytes) {
  GMK_REQUIRE(plan, GMK_E_NULL);
  GMK_REQUIRE(plan != t_rec, GMK_E_STATE);   // still recording
  GMK_REQUIRE(nsub >= 0 && (nsub == 0 || (from && to && bytes)), GMK_E_NULL);
  const int nseg = (int)plan->seg_end.size();
  GMK_REQUIRE(seg_begin >= 0 && seg_begin <= seg_end && seg_end <= nseg, GMK_E_SHAPE);
  for (int i = 0; i < nsub; ++i) GMK_REQUIRE(from[i] && to[i], GMK_E_NULL);
  if (seg_begin == seg_end) return GMK_OK;
  const gmk_plan_detail::Subst s{nsub, (const char* const*)from, (const char* const*)to, bytes};
  const int a = seg_begin == 0 ? 0 : plan->seg_end[seg_begin - 1], b = plan->seg_end[seg_end - 1];
  struct Guard { bool prev; Guard() : prev(t_replaying) { t_replaying = true; } ~Guard() { t_replaying = prev; } } guard;
  for (int i = a; i < b; ++i) {
    const int rc = plan->calls[i](s);
    if (rc != GMK_OK) return rc;
  }
  return GMK_OK;
}
