/*
 * dft_solver.h -- C-ABI of the MI355X-native XC/Fock engine (libdft.so).
 *
 * Drop-in boundary: the first four entry points are exactly the four
 * `extern "C"` symbols of the reference (src/dft_solver.h:66-88, defined at
 * src/dft_solver.cu:675-719) that its ctypes wrapper binds (dft.py:24-50):
 * same names, argument order, types and error behaviour.  Everything after
 * them is a non-breaking extension (new symbols only).
 *
 * All pointers are raw *device* addresses passed as 64-bit integers, exactly
 * as the reference's caller does (`cupy_array.data.ptr`, dft.py:69-95).  All
 * arrays are fp64, C-contiguous:
 *   dm (nao,nao) | ao (ngrid,nao) | ao_grad (3,ngrid,nao) planar | weights (ngrid)
 *   vxc (nao,nao) overwritten | eri (nao^2,nao^2) | J, K (nao,nao) overwritten.
 * Work is enqueued on the solver's stream (default: the null stream, like the
 * reference); DFT_ComputeXC returns Exc once the call's last kernel -- a one-block
 * finishing kernel that stream order starts after every other kernel of the call
 * has completed -- has published it: on return Vxc is complete for consumers on
 * any stream, as after the reference's blocking copy (dft_solver.cu:575-582).
 * (Option "fuse_finish" = 1 trades that for one launch less: see DFT_SetOption.)
 * Every entry point runs on the device that was current at DFT_CreateSolver.
 * No function throws or aborts; failures print one line to stderr, are
 * retrievable with DFT_GetLastError(), and make DFT_ComputeXC return NaN.
 */
#ifndef QCDFT_AMD_DFT_SOLVER_H
#define QCDFT_AMD_DFT_SOLVER_H

#ifdef __cplusplus
extern "C" {
#endif

/* Opaque handle; replaces the reference's `class XCSolver` hierarchy
 * (src/dft_solver.h:7-63: XCSolver / LDASolver / GGASolver / B3LYPSolver). */
typedef struct XCSolver XCSolver;

/* src/dft_solver.h:67-71 */
enum SolverType { SOLVER_LDA = 0, SOLVER_GGA = 1, SOLVER_B3LYP = 2 };

/* ---- reference ABI ------------------------------------------------------ */

/* src/dft_solver.h:73, src/dft_solver.cu:677-682.  Unknown type -> NULL. */
XCSolver *DFT_CreateSolver(int type);

/* src/dft_solver.h:75, src/dft_solver.cu:684-686.  NULL is a no-op. */
void DFT_DestroySolver(XCSolver *solver);

/* src/dft_solver.h:77-82, src/dft_solver.cu:688-704 -> {LDA,GGA,B3LYP}Solver::
 * compute_xc (:559-672).  Returns Exc = sum_g w_g rho_g eps_xc(g) and
 * overwrites vxc:  LDA/B3LYP symmetric; GGA the reference's one-sided matrix
 * (the caller averages it, dft.py:212).  d_ao_grad_ptr may be 0 for LDA.
 * NULL solver -> 0.0 (src/dft_solver.cu:695). */
double DFT_ComputeXC(XCSolver *solver, int ngrid, int nao,
                     unsigned long long d_dm_ptr,
                     unsigned long long d_ao_ptr,
                     unsigned long long d_ao_grad_ptr,
                     unsigned long long d_weights_ptr,
                     unsigned long long d_vxc_ptr);

/* src/dft_solver.h:84-87, src/dft_solver.cu:706-718 -> XCSolver::
 * compute_coulomb (:550-555).  J.ravel() = ERI^T . D.ravel() (the cublasDgemv
 * OP_N call on the row-major buffer).  Asynchronous on the solver's stream.
 * NULL solver -> no-op (src/dft_solver.cu:711). */
void DFT_ComputeCoulomb(XCSolver *solver, int nao,
                        unsigned long long d_eri_ptr,
                        unsigned long long d_dm_ptr,
                        unsigned long long d_J_ptr);

/* ---- extensions (not in the reference) ---------------------------------- */

/* ABI version of this library (bumped when an extension changes). */
int DFT_GetVersion(void);

/* Same as DFT_ComputeXC with a 64-bit grid count (the reference's `int`
 * products overflow once ngrid*nao >= 2^30, src/dft_solver.cu:597,634). */
double DFT_ComputeXC64(XCSolver *solver, long long ngrid, int nao,
                       unsigned long long d_dm_ptr,
                       unsigned long long d_ao_ptr,
                       unsigned long long d_ao_grad_ptr,
                       unsigned long long d_weights_ptr,
                       unsigned long long d_vxc_ptr);

/* DFT_ComputeXC with the OCCUPIED ORBITALS of the density: d_cocc_ptr (nao, nocc) f64 C-order with
 * dm = cocc . cocc^T (occupation folded in: sqrt(2) C_occ for the closed-shell dm of dft.py:181-182; the
 * same contract as DFT_ComputeJKFactorized).  Same results as DFT_ComputeXC(dm) -- same Exc, same Vxc
 * conventions per solver type -- but the density step runs as Y = AO . cocc, rho = rowsum(Y^2),
 * X = Y . cocc^T, grad rho = 2 rowsum(X * dAO): 4 nao nocc flops per grid point on the matrix cores instead
 * of the 2 nao^2 of the reference's contraction with the full matrix (dft_solver.cu:294-307, 346-380).
 * d_dm_ptr may be 0; where the occupied form does not do fewer matrix instructions (nocc close to nao/2:
 * minimal basis sets) the library takes the dm path, with the caller's dm if given, else with cocc . cocc^T
 * formed on the device.  A dm that is NOT cocc . cocc^T is the caller's error (the result then follows cocc
 * or dm depending on the path taken).  Returns Exc like DFT_ComputeXC. */
double DFT_ComputeXCOcc(XCSolver *solver, long long ngrid, int nao, int nocc,
                        unsigned long long d_cocc_ptr,
                        unsigned long long d_dm_ptr,
                        unsigned long long d_ao_ptr,
                        unsigned long long d_ao_grad_ptr,
                        unsigned long long d_weights_ptr,
                        unsigned long long d_vxc_ptr);

/* Asynchronous form of DFT_ComputeXCOcc (see DFT_ComputeXCAsync). */
int DFT_ComputeXCOccAsync(XCSolver *solver, long long ngrid, int nao, int nocc,
                          unsigned long long d_cocc_ptr,
                          unsigned long long d_dm_ptr,
                          unsigned long long d_ao_ptr,
                          unsigned long long d_ao_grad_ptr,
                          unsigned long long d_weights_ptr,
                          unsigned long long d_vxc_ptr,
                          unsigned long long d_exc_ptr);

/* Asynchronous form: Exc is written to the device double at d_exc_ptr; no
 * host synchronisation.  Returns 0 on success. */
int DFT_ComputeXCAsync(XCSolver *solver, long long ngrid, int nao,
                       unsigned long long d_dm_ptr,
                       unsigned long long d_ao_ptr,
                       unsigned long long d_ao_grad_ptr,
                       unsigned long long d_weights_ptr,
                       unsigned long long d_vxc_ptr,
                       unsigned long long d_exc_ptr);

/* Exact exchange on the dense ERI, replaces the driver's
 * cp.einsum('ijkl,jl->ik', eri4d, dm) (dft.py:218).  Asynchronous. */
void DFT_ComputeExchange(XCSolver *solver, int nao,
                         unsigned long long d_eri_ptr,
                         unsigned long long d_dm_ptr,
                         unsigned long long d_K_ptr);

/* J and K from ONE pass over the dense ERI (dft.py:203 + dft.py:218).
 * Either output pointer may be 0. */
void DFT_ComputeJK(XCSolver *solver, int nao,
                   unsigned long long d_eri_ptr,
                   unsigned long long d_dm_ptr,
                   unsigned long long d_J_ptr,
                   unsigned long long d_K_ptr);

/* The same contractions restricted to ERI rows (i, j) with i_lo <= i < i_hi: d_eri_rows points at row
 * (i_lo, 0) of the (nao^2, nao^2) matrix, i.e. at a rank's resident ROW BLOCK when the dense ERI is
 * sharded over GPUs (the reference is single-GPU, dft_solver.cu:550-555 / dft.py:218 are the
 * whole-matrix forms).  J receives this block's partial sum over rows for EVERY column (the reference's
 * OP_N dgemv is a sum over rows), K receives rows [i_lo, i_hi) and zeros elsewhere: summing the outputs of
 * all blocks (one all-reduce) gives the whole-matrix J and K.  Either output may be 0.  Asynchronous;
 * returns 0 or -1 (DFT_GetLastError). */
int DFT_ComputeJKRows(XCSolver *solver, int nao, int i_lo, int i_hi,
                      unsigned long long d_eri_rows,
                      unsigned long long d_dm,
                      unsigned long long d_J,
                      unsigned long long d_K);

/* J and K from a factorised ERI, (ij|kl) ~= sum_P L[P][i][j] L[P][k][l] (pivoted
 * Cholesky vectors, d_chol_ptr: (naux, nao, nao) f64 C-order, every L[P] symmetric).
 * Same matrices as DFT_ComputeCoulomb (dft_solver.cu:550-555, dft.py:203) and the
 * exchange einsum (dft.py:218) to within the factorisation threshold, for basis sizes
 * whose dense ERI (8 nao^4 bytes) does not fit in HBM; K runs as fp64 MFMA GEMMs.
 * d_cocc_ptr: (nao, nocc) f64 C-order with dm = cocc . cocc^T (occupation folded in,
 * i.e. sqrt(2) C_occ for the closed-shell dm of dft.py:181-182).  d_J_ptr or d_K_ptr
 * may be 0; J needs d_dm_ptr, K needs d_cocc_ptr.  When BOTH are requested the contraction
 * L_P : dm that J needs rides in the K build (taken from the half-transformed vectors and
 * cocc) as long as dm IS cocc . cocc^T -- checked on the device in every call; for any other
 * dm (damped, mixed, fractional occupations) J contracts dm itself in a pass of its own.
 * Asynchronous; returns 0 or -1 (DFT_GetLastError). */
int DFT_ComputeJKFactorized(XCSolver *solver, int nao, int naux, int nocc,
                            unsigned long long d_chol_ptr,
                            unsigned long long d_dm_ptr,
                            unsigned long long d_cocc_ptr,
                            unsigned long long d_J_ptr,
                            unsigned long long d_K_ptr);

/* AO values (and Cartesian gradients) on the grid: replaces PySCF's
 * dft.numint.eval_ao(mol, coords, deriv=0/1) at grid.py:30,38.
 * Shell table (host pointers, copied to the device on first use / change):
 *   nshell shells; shell s: centre (x,y,z) bohr in shl_xyz[3s..], angular
 *   momentum shl_l[s] (0..3), shl_nprim[s] primitives starting at
 *   shl_off[s] in prim_exp / prim_coef (coefficients already include the
 *   primitive and contraction normalisation), first AO column shl_ao[s].
 * coords (ngrid,3) device; ao (ngrid,nao) device out; ao_grad (3,ngrid,nao)
 * device out or 0.  Returns 0 on success. */
int DFT_EvalAO(XCSolver *solver, long long ngrid, int nao, int nshell,
               const double *shl_xyz, const int *shl_l, const int *shl_nprim,
               const int *shl_off, const int *shl_ao,
               const double *prim_exp, const double *prim_coef, int nprim_total,
               unsigned long long d_coords_ptr,
               unsigned long long d_ao_ptr,
               unsigned long long d_ao_grad_ptr);

/* The whole AO -> rho -> XC -> Vxc sweep without resident AO planes ("fused mode"): what
 * grid.py:30,38 + dft.py:155,172 + DFT_ComputeXC do together, chunk by chunk -- the AO values and gradients of
 * `chunk_points` grid points (0 = automatic: ~96 MB of planes, an Infinity-Cache-sized working set) are evaluated into
 * a workspace of the solver, swept, and overwritten by the next chunk; Vxc and Exc add up over the chunks (the
 * sweep is linear in the grid points).  Memory: chunk_points*nao*(1 or 4) doubles instead of ngrid*nao*(1 or 4)
 * (BASELINE config 5: 53 GB resident); cost: the AO evaluation is repeated in every call.  Shell table as for
 * DFT_EvalAO; coords (ngrid,3), weights (ngrid), dm (nao,nao) device in; vxc (nao,nao) and exc (1 double, may be 0)
 * device out, valid after the work queued on the solver's stream completes (as DFT_ComputeXCAsync).  0 on success. */
int DFT_ComputeXCDirect(XCSolver *solver, long long ngrid, int nao, int nshell,
                        const double *shl_xyz, const int *shl_l, const int *shl_nprim,
                        const int *shl_off, const int *shl_ao,
                        const double *prim_exp, const double *prim_coef, int nprim_total,
                        unsigned long long d_coords_ptr,
                        unsigned long long d_weights_ptr,
                        unsigned long long d_dm_ptr,
                        unsigned long long d_vxc_ptr,
                        unsigned long long d_exc_ptr,
                        long long chunk_points);

/* Columns of the electron-repulsion matrix on the device: all (ij|kl) with k in shell C and l in shell D, for every
 * i >= j -- what the integral-direct pivoted Cholesky factorisation of the ERI asks for per pivot (cholesky.py; the
 * reference builds the whole tensor on the host with PySCF, `mol.intor('int2e')` at grid.py:65).  Device counterpart
 * of the host engine's column routine (csrc/integrals.c::qc_eri_cols2: same McMurchie-Davidson formulation, s-f
 * shells, same Schwarz screening).  Shell table as for DFT_EvalAO (host arrays, copied once); qmax_pairs: the Schwarz
 * bounds sqrt(max (ab|ab)) of the nshell (nshell + 1) / 2 shell pairs a >= b in the order a (a + 1) / 2 + b.
 * DFT_EriColumns clears and fills d_out: ((2 l_C + 1)(2 l_D + 1), nao, nao) f64, matrix (k, l) at index
 * k (2 l_D + 1) + l, elements i >= j only (each matrix is symmetric; the other triangle stays zero).  Asynchronous on
 * the handle's stream (default: the null stream).  Returns 0 or -1 (DFT_EriColumnsLastError). */
void *DFT_EriColumnsOpen(int nshell, const double *shl_xyz, const int *shl_l, const int *shl_nprim,
                         const int *shl_off, const int *shl_ao, const double *prim_exp, const double *prim_coef,
                         int nao, int nprim_total, const double *qmax_pairs);
int DFT_EriColumns(void *handle, int shell_C, int shell_D, double screen, unsigned long long d_out_ptr);
/* Several ket shell pairs in one call (their kernels run side by side): block k, laid out as DFT_EriColumns does, starts
 * offsets[k] doubles into d_out; blocks must not overlap, [0, largest end) is cleared as a whole. */
int DFT_EriColumnsMany(void *handle, int npairs, const int *shell_C, const int *shell_D, double screen,
                       unsigned long long d_out_ptr, const long long *offsets);
int DFT_EriColumnsSetStream(void *handle, unsigned long long hip_stream);
const char *DFT_EriColumnsLastError(void *handle);
void DFT_EriColumnsClose(void *handle);

/* The rest of an SCF cycle on the device: between the cycle's J / K / Vxc and the next density the
 * reference's loop runs on the host -- Fock assembly (dft.py:212-223), Pulay DIIS (dft.py:225), eigh(F, S) (dft.py:227),
 * dm = 2 C_occ C_occ^T (dft.py:228) and the energy traces (dft.py:231-234).  DFT_ScfTailStep queues all of it behind
 * the kernels that produced J, K and Vxc (csrc/scf_tail.hip): the eigenproblem as the occupied-subspace rotation of
 * scf.OccupiedRotation from the basis in d_basis.  For 2 <= nao <= 512, 1 <= nocc <= 64 (NULL from Open otherwise; up to
 * 128 x 32 the rotation's matrices live in LDS, above in memory).
 *   Open    hcore, overlap: (nao, nao) device, read in every step; d_basis: (nao, nao) device, columns = an
 *           S-orthonormal basis whose first nocc columns span the occupied space (the eigenvectors of the last full
 *           diagonalisation, uploaded by the caller; replaced by the rotated basis after every successful step);
 *           d_fock_out: (nao, nao) device, receives the DIIS-extrapolated Fock matrix of every step; d_mo_energy:
 *           nao doubles or 0 (occupied: exact levels of the rotated block, virtual: diagonal estimates).
 *   Step    rotate = 0: DIIS only (status 1).  c_hf: exact-exchange fraction (K may be 0).  tol: residual at which the
 *           rotation's fixed point stops; canon_tol: the Jacobi sweeps that make the occupied block diagonal end with
 *           the first sweep that met no column pair with |cos| above it (the pairs are then below ~canon_tol^2;
 *           <= 0: 1e-3); max_inner: fixed-point steps allowed (<= 0: 60).  slot: ring slot (0..7) that
 *           receives this cycle's (F, e); hist: the nhist <= 8 live slots, `slot` among them; coef: NULL, or the nhist
 *           Pulay coefficients to use instead of solving for them.  d_J, d_K, d_vraw: this cycle's matrices (vraw as
 *           DFT_ComputeXC leaves it: symmetrised here, dft.py:212); d_dm (nao, nao), d_cocc (nao, nocc) with
 *           dm = cocc cocc^T: the CURRENT density in, the NEXT one out (status 0 only).  d_exc: 0, or the device scalar
 *           a DFT_ComputeXC*Async call queued before this step writes: Wait then returns it as out[7], and the whole
 *           cycle needs one host wait.
 *   Finish  after status 1: the caller has diagonalised d_fock_out and put the eigenvectors into d_basis; writes
 *           cocc = sqrt(2) basis[:, :nocc], dm and the traces.
 *   Wait    blocks (polling host-mapped memory) until the last Step / Finish has completed; out[0..6] = tr(dm' Hcore),
 *           tr(dm' J)/2, -c_hf tr(dm' K)/4, |dm' - dm|_F, status, fixed-point steps, Jacobi sweeps, Exc (see d_exc).  status 0: done;
 *           1: diagonalise d_fock_out yourself, then Finish; 2: the DIIS system was singular (DFT_ScfTailGram copies the
 *           8 x 8 Gram matrix of the ring to the host: solve it there and repeat the Step with `coef`); 3: see DFT_ScfTailMore.
 * Everything is asynchronous on the handle's stream except Wait and Gram.  0 on success, -1 on error. */
void *DFT_ScfTailOpen(int nao, int nocc, unsigned long long d_hcore, unsigned long long d_overlap, unsigned long long d_basis,
                      unsigned long long d_fock_out, unsigned long long d_mo_energy);
int DFT_ScfTailSetStream(void *handle, unsigned long long hip_stream);
int DFT_ScfTailStep(void *handle, int rotate, double c_hf, double tol, double canon_tol, int max_inner, int slot, int nhist,
                    const int *hist, const double *coef, unsigned long long d_J, unsigned long long d_K,
                    unsigned long long d_vraw, unsigned long long d_dm, unsigned long long d_cocc, unsigned long long d_exc);
int DFT_ScfTailFinish(void *handle, double c_hf, unsigned long long d_J, unsigned long long d_K, unsigned long long d_dm,
                      unsigned long long d_cocc);
/* Above 128 functions / 32 occupied orbitals the rotation's fixed-point steps are launches of their own, queued in advance
 * (SetStepsHint: how many per Step, 1..60, default 6); status 3 from Wait = they were not enough: More queues `nsteps` more and
 * the rest of the step again (same matrices as the Step it continues), then Wait again. */
int DFT_ScfTailMore(void *handle, int nsteps, unsigned long long d_J, unsigned long long d_K, unsigned long long d_dm,
                    unsigned long long d_cocc, unsigned long long d_exc);
int DFT_ScfTailSetStepsHint(void *handle, int nsteps);
int DFT_ScfTailWait(void *handle, double *out8);
int DFT_ScfTailGram(void *handle, double *host_out64);
int DFT_ScfTailStamps(void *handle, long long *host_out16);   /* diagnostics: 100 MHz stamps of the rotation kernel's phases */
const char *DFT_ScfTailLastError(void *handle);
void DFT_ScfTailClose(void *handle);

/* Options: "quirks" (1 = reference formulas as shipped, default; 0 = corrected
 * VWN5 / PBE-c derivatives), "path" (0 = auto: wave-specialised
 * persistent MFMA kernels for nao <= 128, generic MFMA kernels above; 1 =
 * plain-VALU validation kernels; 2 = generic MFMA kernels always), "profile"
 * (1 = record per-kernel HIP events for DFT_GetTimings), "ksplit" (grid chunks
 * of the generic Vxc contraction; 0 = auto), "spin_wait" (1, default: the host
 * polls the host-mapped Exc word written by the last kernel instead of sleeping
 * in hipStreamSynchronize), "fuse_finish" (0, default; 1 = the Vxc reduce kernel publishes Exc itself, one launch
 * fewer: DFT_ComputeXC may then return while that kernel's last blocks still store Vxc, which only
 * consumers on the solver's own stream are ordered behind), "strict_sync" (with fuse_finish = 1: 1 =
 * DFT_ComputeXC also waits for the stream to report complete before returning), "occ" (DFT_ComputeXCOcc:
 * 0 = auto, 1 = always the occupied-orbital density step, 2 = never), "eri_symmetric" (0, default: DFT_ComputeCoulomb is
 * eri^T . vec(dm) for any matrix, the reference's GEMV; 1 = the caller vouches that eri is symmetric as an (nao^2, nao^2)
 * matrix, (ij|kl) = (kl|ij), as every real ERI is: only its upper triangle is read, half the bytes and half the time; 2 = ... and
 * in each index pair, (ij|kl) = (ji|kl) = (ij|lk), and dm = dm^T: only the unique eighth is read, 51 against 240 us at Benzene/def2-SVP; below 48 functions both values keep the full pass), "graph" (the synchronous calls: -1 = auto, default:
 * a call repeated with the same pointers and sizes is replayed as one recorded HIP graph where it is launch-bound,
 * planes of at most 2e6 doubles; 1 = always; 0 = never.  Same kernels, same results bit for bit), "tiny" (bases of at most 32 functions: -1 = auto, default: the
 * whole sweep -- density, functional, Vxc contraction of a 16-point sub-tile -- in ONE kernel plus the slab sum where that is
 * the faster call, which it is at every size measured (0.71-1.00 of the four launches, 20 k-300 k points: profiles/r03_tiny_scan_final.txt);
 * 1 = whenever nao <= 32; 0 = never.  Results agree with the four-launch path to the
 * rounding of the sums, not bit for bit), "ao_pt" (grid points per workgroup of DFT_EvalAO:
 * 8, 16, or 0 = auto), "rho_rows" (grid rows per workgroup of the large-basis
 * density kernel: 64, default, or 128).  Returns 0 if the key is known. */
int DFT_SetOption(XCSolver *solver, const char *key, double value);

/* Run subsequent work on `hip_stream` (a hipStream_t cast to an integer);
 * 0 restores the null stream. */
int DFT_SetStream(XCSolver *solver, unsigned long long hip_stream);

/* Last error text ("" if none).  The pointer stays valid until the next call
 * on the same solver. */
const char *DFT_GetLastError(XCSolver *solver);

/* With option "profile"=1: durations (ms) of the kernels of the last
 * DFT_ComputeXC* call, in launch order; names[i] (if non-NULL) receives a
 * static string.  Returns the number of entries written (<= max_entries). */
int DFT_GetTimings(XCSolver *solver, double *ms, const char **names, int max_entries);

#ifdef __cplusplus
}
#endif
#endif /* QCDFT_AMD_DFT_SOLVER_H */
