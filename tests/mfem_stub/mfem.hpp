// TEST INFRASTRUCTURE -- NOT MFEM.  A declaration-only stand-in for the handful of MFEM / hypre types
// that include/saamge_amd_mfem.hpp touches, so that the adaptor layer can be COMPILED (syntax, overloads,
// const-correctness) in the build container, where MFEM and hypre are not installed.  Signatures follow
// MFEM's public headers (linalg/vector.hpp, linalg/operator.hpp, linalg/sparsemat.hpp, linalg/densemat.hpp,
// linalg/hypre.hpp, general/table.hpp, general/array.hpp, fem/pbilinearform.hpp); nothing here has a body
// beyond trivial inline accessors, nothing can be linked or run, and nothing under saamge_amd/ or oracle/
// includes it.  It is not used to build the reference.
#ifndef SAAMGE_AMD_TEST_MFEM_STUB
#define SAAMGE_AMD_TEST_MFEM_STUB
#include <algorithm>

typedef int HYPRE_Int;
typedef int MPI_Comm;
typedef int MPI_Datatype;
typedef int MPI_Op;
extern const MPI_Datatype MPI_BYTE, MPI_DOUBLE, MPI_INT;
extern const MPI_Op MPI_SUM;
int MPI_Comm_rank(MPI_Comm comm, int *rank);
int MPI_Comm_size(MPI_Comm comm, int *size);
int MPI_Allgather(const void *sendbuf, int sendcount, MPI_Datatype sendtype, void *recvbuf, int recvcount, MPI_Datatype recvtype, MPI_Comm comm);
int MPI_Allgatherv(const void *sendbuf, int sendcount, MPI_Datatype sendtype, void *recvbuf, const int *recvcounts, const int *displs,
                   MPI_Datatype recvtype, MPI_Comm comm);
int MPI_Allreduce(const void *sendbuf, void *recvbuf, int count, MPI_Datatype datatype, MPI_Op op, MPI_Comm comm);
int MPI_Alltoallv(const void *sendbuf, const int *sendcounts, const int *sdispls, MPI_Datatype sendtype, void *recvbuf,
                  const int *recvcounts, const int *rdispls, MPI_Datatype recvtype, MPI_Comm comm);

namespace mfem {

void mfem_error(const char *msg = 0);

template <class T>
class Array {
public:
    Array();
    void SetSize(int n);
    int Size() const;
    T &operator[](int i);
    const T &operator[](int i) const;
};

class Vector {
public:
    Vector();
    Vector(double *data, int size);
    double *GetData() const;
    int Size() const;
    Vector &operator=(double value);
};

class Operator {
public:
    explicit Operator(int s = 0);
    virtual ~Operator();
    int Height() const;
    int Width() const;
    virtual void Mult(const Vector &x, Vector &y) const = 0;
protected:
    int height, width;
};

class Solver : public Operator {
public:
    bool iterative_mode;
    explicit Solver(int s = 0, bool iter_mode = false);
    virtual void SetOperator(const Operator &op) = 0;
};

class Matrix : public Operator {
public:
    virtual double &Elem(int i, int j) = 0;
    virtual const double &Elem(int i, int j) const = 0;
};

class DenseMatrix : public Matrix {
public:
    DenseMatrix();
    virtual double &Elem(int i, int j);
    virtual const double &Elem(int i, int j) const;
    virtual void Mult(const Vector &x, Vector &y) const;
};

class SparseMatrix : public Matrix {
public:
    SparseMatrix();
    SparseMatrix(int *i, int *j, double *data, int m, int n);
    int *GetI() const;
    int *GetJ() const;
    double *GetData() const;
    virtual double &Elem(int i, int j);
    virtual const double &Elem(int i, int j) const;
    virtual void Mult(const Vector &x, Vector &y) const;
};

class Table {
public:
    Table();
    Table(const Table &);
    ~Table();
    void SetDims(int rows, int nnz);
    int Size() const;
    int RowSize(int i) const;
    int *GetI();
    int *GetJ();
    const int *GetI() const;
    const int *GetJ() const;
    void MakeI(int nrows);
    void AddAColumnInRow(int r);
    void MakeJ();
    void AddConnection(int r, int c);
    void ShiftUpI();
};
Table *Transpose(const Table &A);
Table *Mult(const Table &A, const Table &B);

class HypreParMatrix : public Operator {
public:
    HypreParMatrix(MPI_Comm comm, HYPRE_Int global_num_rows, HYPRE_Int global_num_cols, HYPRE_Int *row_starts,
                   HYPRE_Int *col_starts, SparseMatrix *diag);
    ~HypreParMatrix();
    MPI_Comm GetComm() const;
    HYPRE_Int GetGlobalNumRows() const;
    HYPRE_Int GetGlobalNumCols() const;
    void GetDiag(SparseMatrix &diag) const;
    void GetOffd(SparseMatrix &offd, HYPRE_Int *&cmap) const;
    HYPRE_Int *RowPart();
    HYPRE_Int *ColPart();
    const HYPRE_Int *RowPart() const;
    const HYPRE_Int *ColPart() const;
    virtual void Mult(const Vector &x, Vector &y) const;
};

class HypreParVector : public Vector {
public:
    HypreParVector();
};

class ParMesh {
public:
    int GetNE() const;
    const Table &ElementToElementTable();
};

class ParFiniteElementSpace {
public:
    ParMesh *GetParMesh() const;
    const Table &GetElementToDofTable() const;
    void GetEssentialVDofs(const Array<int> &bdr_attr_is_ess, Array<int> &ess_dofs) const;
};

class ParBilinearForm {
public:
    void ComputeElementMatrix(int i, DenseMatrix &elmat);
    ParFiniteElementSpace *ParFESpace() const;
    SparseMatrix &SpMat();
};

}  // namespace mfem
#endif
