// Host runtime and C ABI (include/hipfact.h) of the hipfact KKT backend.
//
// One handle = one backend instance = one HIP device + one stream, like one
// SleqpFact object in the reference (fact/fact.c:21-45); no process-global
// state.  All numerics run on the device; there is no CPU fallback: without a
// usable GPU hipfact_create fails with HIPFACT_EDEVICE.
//
// One translation unit (the kernels are compiled together with their launchers), split by role:
//   runtime_types.inc   buffers, plan state, the handle          abi_core.inc         create .. solution (SleqpFact)
//   runtime_plan.inc    upload of a plan, work items             abi_working_set.inc  superset plans, assemble_kkt
//   runtime_queue.inc   factor / solve queues, graphs, cache     abi_krylov.inc       products, Steihaug CG, GLTR
//   vtable_superset.inc row dictionary (plain vtable)            abi_options.inc      options, info, debug copies
//   dense_cols.inc      dense Jacobian columns                   krylov_device.inc    device-controlled CG
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/hipfact.h"
#include "device_types.h"
#include "plan.h"
#include "tridiag_tr.h"

// single translation unit: the kernels are compiled together with their launcher
#include "kernels.hip"
#define DENSE_COLS_KERNELS
#include "dense_cols.inc"
#undef DENSE_COLS_KERNELS
#define KRYLOV_DEVICE_KERNELS
#include "krylov_device.inc"
#undef KRYLOV_DEVICE_KERNELS

#include "runtime_types.inc"
#include "runtime_plan.inc"
#include "runtime_queue.inc"

extern "C" {
#include "abi_core.inc"
#include "abi_working_set.inc"
#include "abi_krylov.inc"
#include "abi_options.inc"
}  // extern "C"

#ifdef HIPFACT_TRACE
// in-kernel timeline of the dataflow launch (scripts/timeline.py; never part of the product build)
extern "C" int hipfact_debug_trace(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hipfact::g_trace), sizeof(long long) * hipfact::TRACE_WGS * 8);
}
extern "C" int hipfact_debug_trace_owner(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hipfact::g_own), sizeof(long long) * hipfact::TRACE_WGS * 8);
}
extern "C" int hipfact_debug_trace_pivot(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hipfact::g_piv), sizeof(long long) * hipfact::TRACE_WGS * 24);
}
#endif
