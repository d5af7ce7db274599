// Host runtime and C ABI (include/hipfact.h) of the hipfact KKT backend.
//
// One handle = one backend instance = one HIP device + one stream, like one
// SleqpFact object in the reference (fact/fact.c:21-45); no process-global
// state.  All numerics run on the device; there is no CPU fallback: without a
// usable GPU hipfact_create fails with HIPFACT_EDEVICE.
//
// The host translation unit (the kernels are compiled separately: kernels_factor.hip / kernels_solve.hip, declared in the
// generated kernels_decl.h), split by role:
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
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/hipfact.h"
#include "device_types.h"
#include "plan.h"
#include "tridiag_tr.h"

// the kernels are translation units of their own (kernels_factor.hip, kernels_solve.hip); this one sees their declarations
#include "kernel_types.h"
#include "kernels_decl.h"

#include "runtime_types.inc"
#include "runtime_plan.inc"
#include "runtime_queue.inc"

extern "C" {
#include "abi_core.inc"
#include "abi_working_set.inc"
#include "abi_krylov.inc"
#include "abi_options.inc"
}  // extern "C"
