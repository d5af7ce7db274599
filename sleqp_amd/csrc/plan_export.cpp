// Host-only C ABI around the symbolic plan (include/hipfact.h, "host-only
// symbolic plan").  No HIP call is made here, so the analysis can be exercised
// on a machine without a GPU.
#include <cstring>
#include <new>

#include "../../include/hipfact.h"
#include "plan.h"

struct hipfact_plan {
  hipfact::Plan plan;
};

extern "C" {

int hipfact_plan_create(int N, const int* colptr, const int* rowidx, const double* vals, hipfact_plan** out) {
  if (!out) return HIPFACT_EINVAL;
  *out = nullptr;
  hipfact_plan* p = new (std::nothrow) hipfact_plan();
  if (!p) return HIPFACT_ENOMEM;
  hipfact::PlanParams prm;
  *out = p;
  try {
    if (!hipfact::build_plan_bounds(N, colptr, rowidx, vals, prm, p->plan)) return HIPFACT_EINVAL;
  } catch (const std::bad_alloc&) {
    p->plan.error = "out of memory";
    return HIPFACT_ENOMEM;
  } catch (...) {
    p->plan.error = "internal error";
    return HIPFACT_EINTERNAL;
  }
  return HIPFACT_OK;
}

void hipfact_plan_free(hipfact_plan** plan) {
  if (plan && *plan) {
    delete *plan;
    *plan = nullptr;
  }
}

const char* hipfact_plan_error(const hipfact_plan* plan) { return plan ? plan->plan.error.c_str() : "null plan"; }

int hipfact_plan_array(const hipfact_plan* plan, const char* name, const void** data, int64_t* len, int* elem_size) {
  if (!plan || !name || !data || !len || !elem_size) return HIPFACT_EINVAL;
  const hipfact::Plan& P = plan->plan;
#define ARR(field)                                  \
  if (!strcmp(name, #field)) {                      \
    *data = P.field.data();                         \
    *len = (int64_t)P.field.size();                 \
    *elem_size = (int)sizeof(P.field[0]);           \
    return HIPFACT_OK;                              \
  }
  ARR(Kp) ARR(Ki) ARR(perm) ARR(iperm) ARR(Mp) ARR(Mi) ARR(Mtarget) ARR(prod_ptr) ARR(prod_a) ARR(prod_b) ARR(src)
  ARR(sn_c0) ARR(sn_r) ARR(sn_rowptr) ARR(sn_rows) ARR(sn_parent) ARR(sn_level) ARR(sn_Loff) ARR(sn_Uoff)
  ARR(sn_uoff) ARR(child_ptr) ARR(child_idx) ARR(rel_ptr) ARR(rel) ARR(level_ptr) ARR(level_sn) ARR(Ar_ptr)
  ARR(Ar_col) ARR(Ar_src) ARR(Kc_y) ARR(dense_cols) ARR(late_cols)
  ARR(bnd_row) ARR(bnd_col) ARR(row_ext) ARR(ent_ext) ARR(cut_ptr) ARR(cut_row) ARR(cut_ent)
#undef ARR
  return HIPFACT_EINVAL;
}

int hipfact_plan_scalar(const hipfact_plan* plan, const char* name, double* value) {
  if (!plan || !name || !value) return HIPFACT_EINVAL;
  const hipfact::Plan& P = plan->plan;
#define SC(field)               \
  if (!strcmp(name, #field)) {  \
    *value = (double)P.field;   \
    return HIPFACT_OK;          \
  }
  SC(N) SC(n) SC(m) SC(my) SC(n_late) SC(n_late_rows) SC(saddle) SC(nnzK) SC(nsuper) SC(nlevels) SC(L_size) SC(U_size) SC(u_size) SC(nnzL)
  SC(nnzL_true) SC(flops) SC(flops_dense) SC(nprod) SC(max_r) SC(max_w) SC(max_u) SC(t_order) SC(t_symbolic)
  SC(t_total) SC(N_ext) SC(n_bounds)
#undef SC
  return HIPFACT_EINVAL;
}

}  // extern "C"
