// Shared by the engine translation units (engine*.hip): includes, the HIP error macro, small helpers.
#pragma once
#include "engine.hpp"

#include <algorithm>
#include <limits>
#include <mutex>
#include <thread>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/feasthip.h"
#include "smalleig.hpp"

namespace feasthip {

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return hip_fail(e_, #x); } while (0)

static inline cz to_cz(zc v) { return cz_make(v.real(), v.imag()); }

}  // namespace feasthip
