// Plain-old-data descriptors shared by the host runtime and the gfx950 kernels.
#pragma once
#include <cstdint>

namespace hipfact {

// One front (supernode) of the multifrontal LDL^T.
//   panel  : L arena + Loff, r x w column-major (ld = r).  After factorisation the
//            top w x w block holds inv(L11) strictly below the diagonal and the
//            pivots d on the diagonal; rows w..r hold L21.
//   update : U arena + Uoff, u x u column-major (ld = u), lower triangle valid.
//   uvec   : solve workspace + uoff, u doubles (update vector passed to the parent).
struct SnDesc {
  long long Loff;
  long long Uoff;
  long long uoff;
  long long rowoff;  // into rows[]: r sorted pivot-order row indices, own columns first
  long long reloff;  // into rel[]: u positions inside the parent's front
  int c0;            // first column (pivot order)
  int w;             // columns
  int r;             // rows of the front
  int parent;        // -1 for a root
  int child_begin;   // children in child_idx[child_begin, child_end)
  int child_end;
  int pad0;  // elimination-tree level of the front
  int pad1;
};

// info words written by the factorisation kernels
enum { INFO_ZERO_PIVOT = 0, INFO_NEG_PIVOT = 1, INFO_TIMEOUT = 2, INFO_WORDS = 4 };

}  // namespace hipfact
