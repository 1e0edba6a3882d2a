// Environment knobs of the launchers (no HIP dependency: problems.h / igemm16.h are also compiled by the host-side index-math tests).
#pragma once
#include <stdio.h>
#include <stdlib.h>

namespace vp {

// A/B knobs that the launchers consult per launch: a getenv() is a linear scan of the environment, so each knob is looked up ONCE
// per process and call site -- unless VP_ENV_DYNAMIC=1, which the in-process A/B tools (tools/ab_env.py, ab_build.py,
// ab_multi.py) set before they load the library so that they can flip knobs between launches.
struct EnvCache { bool set; char val[1024]; };
inline EnvCache env_read(const char* name) {
  EnvCache c;
  c.set = false; c.val[0] = 0;
  if (const char* e = getenv(name)) { c.set = true; snprintf(c.val, sizeof(c.val), "%s", e); }
  return c;
}
inline bool env_dynamic() {
  static const bool d = [] { const char* e = getenv("VP_ENV_DYNAMIC"); return e && atoi(e) != 0; }();
  return d;
}
#define VP_GETENV(name)                                                                  \
  ([]() -> const char* {                                                                 \
    static const ::vp::EnvCache vp_env_c = ::vp::env_read(name);                         \
    return ::vp::env_dynamic() ? getenv(name) : (vp_env_c.set ? vp_env_c.val : nullptr); \
  }())

}  // namespace vp
