// plan.h -- recording hook of the launch plans (plan.hip): every launching entry point of include/gmk.h starts with
// GMK_PLAN_HOOK(its own name, its arguments...).  While a plan is being recorded on the calling host thread the call is
// appended to it -- function pointer + arguments by value -- and then runs as usual; gmk_train_step_enqueue() later calls
// the same entry points again with the same arguments (device pointers inside the caller's substitution ranges moved to
// their new addresses), so a replayed step takes exactly the code path of a direct call: same argument checks, same
// kernel selection, same launches.
#pragma once
#include <stddef.h>
#include <functional>

namespace gmk_plan_detail {

struct Subst {
  int n;
  const char* const* from;
  const char* const* to;
  const size_t* bytes;
  const void* map(const void* p) const {
    const char* c = (const char*)p;
    for (int i = 0; i < n; ++i)
      if (c >= from[i] && c < from[i] + bytes[i]) return to[i] + (c - from[i]);
    return p;
  }
};

bool recording();
void push(std::function<int(const Subst&)> call);

template <class T> inline T sub(const Subst&, T v) { return v; }
template <class T> inline T* sub(const Subst& s, T* v) { return (T*)s.map((const void*)v); }

template <class... A, class... B>
inline void record(int (*fn)(A...), B... args) {
  static_assert(sizeof...(A) == sizeof...(B), "hook must pass every argument of the entry point");
  push([=](const Subst& s) { return fn(sub(s, static_cast<A>(args))...); });
}

}  // namespace gmk_plan_detail

#define GMK_PLAN_HOOK(fn, ...)                                                        \
  do {                                                                                \
    if (gmk_plan_detail::recording()) gmk_plan_detail::record(&fn, __VA_ARGS__);      \
  } while (0)
