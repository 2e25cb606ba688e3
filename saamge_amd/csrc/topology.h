// Integer topology of one level: the agg_partitioning_relations_t of the reference
// (amg/inc/aggregates.hpp:120-179), single rank, built on the host and mirrored to
// the device for the assembly / MIS kernels.
#pragma once
#include "common.h"

namespace saamge_amd {

struct Table {  // mfem::Table: CSR of ints
    hvec<int> I, J;
    int ncols = 0;
    int nrows() const { return (int)I.size() - 1; }
    int row_size(int i) const { return I[i + 1] - I[i]; }
    const int *row(int i) const { return J.data() + I[i]; }
};

Table table_transpose(const Table &T);                   // rows ascending (mfem::Transpose)
Table table_mult(const Table &A, const Table &B);        // first-encounter order (mfem::Mult)

constexpr signed char FLAG_BETWEEN_AES = 0x01;           // amg/inc/aggregates.hpp:102
constexpr signed char FLAG_ON_ESS_BORDER = 0x02;         // amg/inc/aggregates.hpp:103

struct Relations {
    int ND = 0, NE = 0, nparts = 0, num_mises = 0;
    Table elem_to_dof, dof_to_elem, AE_to_elem, AE_to_dof, dof_to_AE;
    Table mis_to_dof, mis_to_AE, AE_to_mis;
    hvec<int> partitioning;   // elem -> AE
    hvec<int> dof_id_inAE;    // aligned with dof_to_AE.J
    hvec<int> elem_ldof;      // aligned with elem_to_dof.J: index of that dof in the element's AE
    hvec<int> mises;          // dof -> MIS
    hvec<int> dof_row_in_mis; // dof -> position inside its MIS
    hvec<signed char> agg_flags;
    // (MIS, AE) incidence pairs, MIS-major (== mis_to_AE entries): local AE indices of the
    // MIS's dofs, used to restrict AE eigenvectors to the MIS and to build P_loc.
    hvec<int64_t> pair_loc_off;  // [npairs+1] offsets into pair_loc
    hvec<int> pair_loc;          // AE-local index of each MIS dof
    hvec<int> ae_pair;           // aligned with AE_to_mis.J: pair id of (AE, mis)
    // device build (build_relations_ae_device): AE_to_dof.J, dof_to_AE, dof_id_inAE and agg_flags (0.4 GB at
    // 256^3) stay on the device until a host phase asks for them (fetch_relations_ae_host): the host MIS
    // build, the host build of the next level's elements, the inspection getters
    bool ae_host_pending = false;
};

// agg_create_partitioning_tables + agg_produce_mises + agg_construct_agg_flags
// (amg/src/aggregates.cpp:1357-1443, :501-653, :198-216).  bdr may be null (coarse levels).
// Split in two so that the MIS half can run on a host thread while the GPU already works on the
// AE matrices (which only need the first half).
void build_relations_ae(Relations &r, Table &&elem_to_dof, const hvec<int> &partitioning,
                        int nparts, int ND, const signed char *bdr);
// `aggregates_A` (host copy of the level matrix): do_aggregates on the last coarsening --
// aggregates with arbitration instead of MISes (agg_construct_aggregate_mises,
// amg/src/aggregates.cpp:324-487; Arbitrator::suggest, amg/src/arbitrator.cpp:93-204).
struct HostCsr {
    int nrows = 0;
    std::vector<roff_t> rowptr;
    std::vector<int> col;
    std::vector<double> val;
};
void build_relations_mis(Relations &r, const HostCsr *aggregates_A = nullptr);

struct DevRelations {
    DBuf<int> e2d_I, e2d_J, elem_ldof, part;
    DBuf<int> d2e_I, d2e_J;
    DBuf<int> ae2d_I, ae2d_J;
    DBuf<int> d2ae_I, d2ae_J, dof_id_inAE;
    DBuf<int> mis2d_I, mis2d_J, mis2ae_I, mis2ae_J, ae2mis_I, ae2mis_J, ae_pair;
    DBuf<int> mises, dof_row_in_mis;
    DBuf<int64_t> pair_loc_off;
    DBuf<int> pair_loc;
    DBuf<signed char> flags;
};
void upload_relations_ae(DevRelations &d, const Relations &r, hipStream_t s);
// build_relations_ae + upload_relations_ae entirely on the device for device-resident inputs with
// a fixed number of dofs per element (level 0); the host receives the tables the MIS stage and
// the next level need, never elem_to_dof.  Returns false (nothing built) when an agglomerate is
// too large for the LDS kernels: the caller then takes the host path.
bool build_relations_ae_device(Relations &r, DevRelations &d, const int *e2d_dev, int NE, int nde,
                               const int *part_dev, int nparts, int ND, const signed char *bdr_dev,
                               hipStream_t s);
void fetch_relations_ae_host(Relations &r, const DevRelations &d, hipStream_t s);
void upload_relations_mis(DevRelations &d, const Relations &r, hipStream_t s);
// build_relations_mis + upload_relations_mis on the device (topology_mis.hip): same tables bit for bit; needs the AE
// half of `d`.  Returns false when the level has to take the host path (hash collision, invalid partition).
bool build_relations_mis_device(Relations &r, DevRelations &d, hipStream_t s);

}  // namespace saamge_amd
