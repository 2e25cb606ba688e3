// Host-side integer topology (single rank).  Output is bit-identical to the reference's
// tables; the reference's O(#MIS x ND) MIS loop (amg/src/aggregates.cpp:541-607) is replaced
// by a signature hash with the same numbering (first appearance scanning dofs upward,
// dofs inside a MIS ascending).
#include "topology.h"

#include <algorithm>
#include <unordered_map>

namespace saamge_amd {

Table table_transpose(const Table &T) {
    Table R;
    const int nr = T.nrows();
    R.ncols = nr;
    R.I.assign((size_t)T.ncols + 1, 0);
    for (int v : T.J) R.I[(size_t)v + 1]++;
    for (int i = 0; i < T.ncols; ++i) R.I[i + 1] += R.I[i];
    R.J.resize(T.J.size());
    std::vector<int> pos(R.I.begin(), R.I.end() - 1);
    for (int i = 0; i < nr; ++i)
        for (int k = T.I[i]; k < T.I[i + 1]; ++k) R.J[pos[T.J[k]]++] = i;
    return R;
}

Table table_mult(const Table &A, const Table &B) {
    Table C;
    const int nr = A.nrows();
    C.ncols = B.ncols;
    C.I.assign((size_t)nr + 1, 0);
    std::vector<int> stamp((size_t)B.ncols, -1);
    // count
    for (int i = 0; i < nr; ++i) {
        int cnt = 0;
        for (int k = A.I[i]; k < A.I[i + 1]; ++k) {
            const int j = A.J[k];
            for (int q = B.I[j]; q < B.I[j + 1]; ++q) {
                const int c = B.J[q];
                if (stamp[c] != i) { stamp[c] = i; ++cnt; }
            }
        }
        C.I[i + 1] = C.I[i] + cnt;
    }
    C.J.resize((size_t)C.I[nr]);
    std::fill(stamp.begin(), stamp.end(), -1);
    for (int i = 0; i < nr; ++i) {
        int p = C.I[i];
        for (int k = A.I[i]; k < A.I[i + 1]; ++k) {
            const int j = A.J[k];
            for (int q = B.I[j]; q < B.I[j + 1]; ++q) {
                const int c = B.J[q];
                if (stamp[c] != i) { stamp[c] = i; C.J[p++] = c; }
            }
        }
    }
    return C;
}

static inline uint64_t hash_row(const int *r, int n) {
    uint64_t h = 1469598103934665603ull ^ (uint64_t)n;
    for (int i = 0; i < n; ++i) {
        h ^= (uint64_t)(uint32_t)r[i] + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
        h *= 1099511628211ull;
    }
    return h;
}

void build_relations(Relations &r, Table &&elem_to_dof, const std::vector<int> &partitioning,
                     int nparts, int ND, const signed char *bdr) {
    r.ND = ND;
    r.nparts = nparts;
    r.elem_to_dof = std::move(elem_to_dof);
    r.elem_to_dof.ncols = ND;
    r.NE = r.elem_to_dof.nrows();
    SA_REQUIRE((int)partitioning.size() == r.NE, "partitioning size != number of elements");
    r.partitioning = partitioning;
    for (int e = 0; e < r.NE; ++e)
        SA_REQUIRE(partitioning[e] >= 0 && partitioning[e] < nparts, "partition id out of range");
    for (int v : r.elem_to_dof.J) SA_REQUIRE(v >= 0 && v < ND, "elem_to_dof entry out of range");
    r.dof_to_elem = table_transpose(r.elem_to_dof);
    // elem_to_AE / AE_to_elem (agg_construct_tables_from_arr)
    Table elem_to_AE;
    elem_to_AE.ncols = nparts;
    elem_to_AE.I.resize((size_t)r.NE + 1);
    for (int e = 0; e <= r.NE; ++e) elem_to_AE.I[e] = e;
    elem_to_AE.J = partitioning;
    r.AE_to_elem = table_transpose(elem_to_AE);
    for (int p = 0; p < nparts; ++p) SA_REQUIRE(r.AE_to_elem.row_size(p) > 0, "empty agglomerate");
    r.AE_to_dof = table_mult(r.AE_to_elem, r.elem_to_dof);   // first-encounter order
    r.dof_to_AE = table_transpose(r.AE_to_dof);               // ascending AE ids
    // dof_id_inAE (agg_build_glob_to_AE_id_map, :1202-1244)
    r.dof_id_inAE.assign(r.dof_to_AE.J.size(), -1);
    {
        std::vector<int> pos(r.dof_to_AE.I.begin(), r.dof_to_AE.I.end() - 1);
        // dof_to_AE rows are ascending in AE and AEs are visited ascending -> pos walks each row
        for (int p = 0; p < nparts; ++p)
            for (int k = r.AE_to_dof.I[p]; k < r.AE_to_dof.I[p + 1]; ++k) {
                const int dof = r.AE_to_dof.J[k];
                r.dof_id_inAE[pos[dof]++] = k - r.AE_to_dof.I[p];
            }
    }
    // elem_ldof: AE-local index of every element dof (each element lies in exactly one AE)
    r.elem_ldof.resize(r.elem_to_dof.J.size());
    for (int e = 0; e < r.NE; ++e) {
        const int p = partitioning[e];
        for (int k = r.elem_to_dof.I[e]; k < r.elem_to_dof.I[e + 1]; ++k) {
            const int dof = r.elem_to_dof.J[k];
            const int *row = r.dof_to_AE.row(dof);
            const int rs = r.dof_to_AE.row_size(dof);
            const int *it = std::lower_bound(row, row + rs, p);
            r.elem_ldof[k] = r.dof_id_inAE[r.dof_to_AE.I[dof] + (int)(it - row)];
        }
    }
    // MISes
    r.mises.assign((size_t)ND, -1);
    r.dof_row_in_mis.assign((size_t)ND, 0);
    std::vector<int> single((size_t)nparts, -1);          // MIS of dofs living in exactly one AE
    std::unordered_map<uint64_t, std::vector<int>> multi;  // hash -> candidate MIS ids
    std::vector<int> mis_rep;                              // representative dof of each MIS
    std::vector<int> mis_size;
    for (int i = 0; i < ND; ++i) {
        const int rs = r.dof_to_AE.row_size(i);
        const int *row = r.dof_to_AE.row(i);
        SA_REQUIRE(rs > 0, "dof without any element");
        int mid = -1;
        if (rs == 1) {
            mid = single[row[0]];
            if (mid < 0) {
                mid = (int)mis_rep.size();
                single[row[0]] = mid;
                mis_rep.push_back(i);
                mis_size.push_back(0);
            }
        } else {
            std::vector<int> &cands = multi[hash_row(row, rs)];
            for (int c : cands) {
                const int rep = mis_rep[c];
                if (r.dof_to_AE.row_size(rep) == rs &&
                    std::equal(row, row + rs, r.dof_to_AE.row(rep))) { mid = c; break; }
            }
            if (mid < 0) {
                mid = (int)mis_rep.size();
                cands.push_back(mid);
                mis_rep.push_back(i);
                mis_size.push_back(0);
            }
        }
        r.mises[i] = mid;
        r.dof_row_in_mis[i] = mis_size[mid]++;
    }
    r.num_mises = (int)mis_rep.size();
    r.mis_to_dof.ncols = ND;
    r.mis_to_dof.I.assign((size_t)r.num_mises + 1, 0);
    for (int m = 0; m < r.num_mises; ++m) r.mis_to_dof.I[m + 1] = r.mis_to_dof.I[m] + mis_size[m];
    r.mis_to_dof.J.resize((size_t)ND);
    for (int i = 0; i < ND; ++i) r.mis_to_dof.J[r.mis_to_dof.I[r.mises[i]] + r.dof_row_in_mis[i]] = i;
    // mis_to_AE = mis_to_dof x dof_to_AE == the (ascending) AE list of any member dof
    r.mis_to_AE.ncols = nparts;
    r.mis_to_AE.I.assign((size_t)r.num_mises + 1, 0);
    for (int m = 0; m < r.num_mises; ++m)
        r.mis_to_AE.I[m + 1] = r.mis_to_AE.I[m] + r.dof_to_AE.row_size(mis_rep[m]);
    r.mis_to_AE.J.resize((size_t)r.mis_to_AE.I[r.num_mises]);
    for (int m = 0; m < r.num_mises; ++m)
        std::copy(r.dof_to_AE.row(mis_rep[m]), r.dof_to_AE.row(mis_rep[m]) + r.dof_to_AE.row_size(mis_rep[m]),
                  r.mis_to_AE.J.begin() + r.mis_to_AE.I[m]);
    r.AE_to_mis = table_transpose(r.mis_to_AE);  // ascending MIS ids == the "sorted" order of elmat.cpp:121-123
    // flags
    r.agg_flags.assign((size_t)ND, 0);
    for (int i = 0; i < ND; ++i) {
        signed char f = bdr ? bdr[i] : 0;
        if (r.dof_to_AE.row_size(i) > 1) f |= FLAG_BETWEEN_AES;
        r.agg_flags[i] = f;
    }
    // (MIS, AE) pairs and AE-local indices
    const size_t npairs = r.mis_to_AE.J.size();
    r.pair_loc_off.assign(npairs + 1, 0);
    for (int m = 0; m < r.num_mises; ++m)
        for (int q = r.mis_to_AE.I[m]; q < r.mis_to_AE.I[m + 1]; ++q)
            r.pair_loc_off[(size_t)q + 1] = r.pair_loc_off[q] + mis_size[m];
    r.pair_loc.resize((size_t)r.pair_loc_off[npairs]);
    for (int m = 0; m < r.num_mises; ++m) {
        const int na = r.mis_to_AE.row_size(m);
        for (int t = 0; t < na; ++t) {
            const int q = r.mis_to_AE.I[m] + t;
            int *dst = r.pair_loc.data() + r.pair_loc_off[q];
            for (int k = 0; k < mis_size[m]; ++k) {
                const int dof = r.mis_to_dof.J[r.mis_to_dof.I[m] + k];
                // every dof of the MIS has the same AE list: the t-th AE of the dof's row
                dst[k] = r.dof_id_inAE[r.dof_to_AE.I[dof] + t];
            }
        }
    }
    r.ae_pair.resize(r.AE_to_mis.J.size());
    {
        std::vector<int> pos(r.AE_to_mis.I.begin(), r.AE_to_mis.I.end() - 1);
        for (int m = 0; m < r.num_mises; ++m)
            for (int q = r.mis_to_AE.I[m]; q < r.mis_to_AE.I[m + 1]; ++q) r.ae_pair[pos[r.mis_to_AE.J[q]]++] = q;
    }
}

void upload_relations(DevRelations &d, const Relations &r, hipStream_t s) {
    d.e2d_I.from_host(r.elem_to_dof.I, s);
    d.e2d_J.from_host(r.elem_to_dof.J, s);
    d.elem_ldof.from_host(r.elem_ldof, s);
    d.part.from_host(r.partitioning, s);
    d.d2e_I.from_host(r.dof_to_elem.I, s);
    d.d2e_J.from_host(r.dof_to_elem.J, s);
    d.ae2d_I.from_host(r.AE_to_dof.I, s);
    d.ae2d_J.from_host(r.AE_to_dof.J, s);
    d.d2ae_I.from_host(r.dof_to_AE.I, s);
    d.d2ae_J.from_host(r.dof_to_AE.J, s);
    d.dof_id_inAE.from_host(r.dof_id_inAE, s);
    d.mis2d_I.from_host(r.mis_to_dof.I, s);
    d.mis2d_J.from_host(r.mis_to_dof.J, s);
    d.mis2ae_I.from_host(r.mis_to_AE.I, s);
    d.mis2ae_J.from_host(r.mis_to_AE.J, s);
    d.ae2mis_I.from_host(r.AE_to_mis.I, s);
    d.ae2mis_J.from_host(r.AE_to_mis.J, s);
    d.ae_pair.from_host(r.ae_pair, s);
    d.mises.from_host(r.mises, s);
    d.dof_row_in_mis.from_host(r.dof_row_in_mis, s);
    d.pair_loc_off.from_host(r.pair_loc_off, s);
    d.pair_loc.from_host(r.pair_loc, s);
    d.flags.from_host(r.agg_flags, s);
}

}  // namespace saamge_amd
