// Row-partitioned solve phase: ownership ranges, halo exchange, reductions (see dist.hip).
#pragma once
#include <functional>

#include "hierarchy.h"

namespace saamge_amd {

// Decide whether level `lev` is applied by row blocks and, if so, build its ownership ranges and
// halo-exchange lists (collective: every rank must call it for the same levels).
bool dist_setup_level(Hierarchy &H, int lev);
// refresh the halo entries of the global-length vector x from their owners
void halo_exchange(Hierarchy &H, Level::Dist &D, double *x);
// The pattern of every row-partitioned SpMV: refresh the halo of x, then apply `op(stream, rows)` (an SpMV-family
// launch that reads x and writes the given rows) to the own rows.  The rows that read no halo entry are applied on
// a side stream while the exchange is in flight, the others after it (hypre overlaps its ParCSR communication
// with the `diag` product the same way).  D == nullptr: one launch over all rows.
void halo_then(Hierarchy &H, Level::Dist *D, double *x, const std::function<void(hipStream_t, RowRange)> &op);
// in-place sum over ranks of `count` doubles on the device
void dist_allreduce(Hierarchy &H, double *buf, long long count);
// in-place all-gather of the own row ranges of the global-length vector x
void dist_allgather_rows(Hierarchy &H, Level::Dist &D, double *x);
void dist_reduce_scatter_rows(Hierarchy &H, Level::Dist &D, double *buf);

}  // namespace saamge_amd
