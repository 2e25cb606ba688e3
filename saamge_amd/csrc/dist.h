// Row-partitioned solve phase: ownership ranges, halo exchange, reductions (see dist.hip).
#pragma once
#include "hierarchy.h"

namespace saamge_amd {

// Decide whether level `lev` is applied by row blocks and, if so, build its ownership ranges and
// halo-exchange lists (collective: every rank must call it for the same levels).
bool dist_setup_level(Hierarchy &H, int lev);
// refresh the halo entries of the global-length vector x from their owners
void halo_exchange(Hierarchy &H, Level::Dist &D, double *x);
// in-place sum over ranks of `count` doubles on the device
void dist_allreduce(Hierarchy &H, double *buf, long long count);
// in-place all-gather of the own row ranges of the global-length vector x
void dist_allgather_rows(Hierarchy &H, Level::Dist &D, double *x);

}  // namespace saamge_amd
