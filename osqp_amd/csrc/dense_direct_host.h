// dense_direct_host.h -- host side of the dense-direct solve (kernels: dense_direct.h); included by engine.hip after build_elim.
// small_only: the call ahead of the resident PCG -- reduced systems of at most OSQP_AMD_DENSE_SMALL (1024) dense unknowns, whatever
// their sparsity: there a solve is three launches and 15-20 us, which neither a resident launch (one exchange per PCG iteration,
// 90 us) nor launch-per-step PCG iterations come near (tools/small_paths.py: 125-iteration solves in 3.6 / 4.1 / 4.9 / 6.1 ms at
// n = 150 / 300 / 600 / 1000 against 18 / 11.3 / 11.1 / 11.2 ms; at n = 2000 the 5.6 ms refresh of every rho update eats the gain).
static int build_dense_direct(hipeng *e, const csc *P, const csc *A, bool small_only) {
  int want = 1;
  if (const char *x = getenv("OSQP_AMD_DENSE_DIRECT")) want = atoi(x);
  int small_max = 1024;
  if (const char *x = getenv("OSQP_AMD_DENSE_SMALL")) small_max = std::max(0, atoi(x));
  if (small_only && small_max == 0) return 0;
  const int n = e->n, m = e->m;
  if (!want || e->res_on || e->res_kind != 0 || n == 0 || m == 0) return 0;
  auto gone = [&](int j) { return !e->erow.empty() && e->erow[j] >= 0; };
  // rows of A over the unknowns of the reduced system: dense (>= DD_DENSE_ROW entries: a row of R) or short (scattered)
  std::vector<char> isdense(m, 0), indense(n, 0), coupledP(n, 0);
  std::vector<int> drow;
  for (int i = 0; i < m; i++) {
    int len = 0;
    for (int k = e->A.rowptr[i]; k < e->A.rowptr[i + 1]; k++) len += !gone(e->A.col[k]);
    if (len >= DD_DENSE_ROW) { isdense[i] = 1; drow.push_back(i); for (int k = e->A.rowptr[i]; k < e->A.rowptr[i + 1]; k++) indense[e->A.col[k]] = 1; }
  }
  for (int j = 0; j < n; j++)
    for (long long k = P->p[j]; k < P->p[j + 1]; k++) if (P->i[k] != j) { coupledP[j] = 1; coupledP[(int)P->i[k]] = 1; }
  // B2: a greedy independent set of sparse variables (ascending index; a chosen variable blocks its neighbours)
  std::vector<int> vidx(n, -1), alist, blist, bnbr_var;
  std::vector<char> blocked(n, 0), chosen(n, 0);
  std::vector<int> nb;
  for (int j = 0; j < n; j++) {
    if (gone(j) || indense[j] || coupledP[j] || blocked[j]) continue;
    nb.clear();
    bool ok = true;
    for (long long k = A->p[j]; k < A->p[j + 1] && ok; k++) {
      const int i = (int)A->i[k];
      for (int q = e->A.rowptr[i]; q < e->A.rowptr[i + 1]; q++) {
        const int v = e->A.col[q];
        if (v == j || gone(v)) continue;
        if (chosen[v]) { ok = false; break; }
        if (std::find(nb.begin(), nb.end(), v) == nb.end()) { nb.push_back(v); if ((int)nb.size() > DD_NBR) { ok = false; break; } }
      }
    }
    if (!ok) continue;
    chosen[j] = 1;
    for (int v : nb) blocked[v] = 1;
    blist.push_back(j);
    for (int s = 0; s < DD_NBR; s++) bnbr_var.push_back(s < (int)nb.size() ? nb[s] : -1);
  }
  for (int j = 0; j < n; j++) if (!gone(j) && !chosen[j]) { vidx[j] = (int)alist.size(); alist.push_back(j); }
  for (size_t b = 0; b < blist.size(); b++) vidx[blist[b]] = -((int)b + 2);
  const int na = (int)alist.size(), nb2 = (int)blist.size(), nd = (int)drow.size();
  if (na == 0 || na > DD_MAX) return 0;
  const int nap = (na + DD_NB - 1) / DD_NB * DD_NB;
  if (small_only && na > small_max) return 0;
  // does it pay?  a PCG solve streams A twice and P once per iteration, rarely fewer than eight of them; this one streams the
  // inverse once (and the refresh costs a few such solves)
  const double nnzA = (double)e->A.val.size(), nnzP = 2.0 * (double)e->P_toM_up.size();
  const double bytes_pcg = 8.0 * (2.0 * nnzA + nnzP) * 12.0, bytes_dd = 3.0 * 8.0 * (double)nap * nap;
  bool pays = (small_only || bytes_pcg > bytes_dd) && (double)nd * nap * 8.0 < 6e9;
  if (const char *x = getenv("OSQP_AMD_DENSE_DIRECT")) if (atoi(x) == 2) pays = (double)nd * nap * 8.0 < 6e9;      // 2: whenever it fits (tests)
  if (!pays) return 0;
  std::vector<int> bnbr(bnbr_var.size());
  for (size_t q = 0; q < bnbr_var.size(); q++) bnbr[q] = bnbr_var[q] < 0 ? -1 : vidx[bnbr_var[q]];
  // the adjacency seen from the dense side: for dense index a the {b, slot} pairs with bnbr[b][slot] == a, b ascending
  std::vector<int> aptr((size_t)nap + 1, 0);
  std::vector<int2> aadj;
  {
    std::vector<int> cnt((size_t)nap, 0);
    for (size_t q = 0; q < bnbr.size(); q++) if (bnbr[q] >= 0) cnt[bnbr[q]]++;
    for (int a = 0; a < nap; a++) aptr[a + 1] = aptr[a] + cnt[a];
    aadj.resize((size_t)aptr[nap]);
    std::vector<int> pos(aptr.begin(), aptr.end() - 1);
    for (size_t q = 0; q < bnbr.size(); q++) if (bnbr[q] >= 0) aadj[pos[bnbr[q]]++] = int2{(int)(q / DD_NBR), (int)(q % DD_NBR)};
  }
  // per variable: the slots in M of its entries in short rows of A (what k_dd_scatter walks)
  std::vector<int> sptr((size_t)n + 1, 0), spos;
  for (int j = 0; j < n; j++) {
    for (int k = e->M.split[j]; k < e->M.rowptr[j + 1]; k++) if (!isdense[e->M.col[k] - n]) spos.push_back(k);
    sptr[j + 1] = (int)spos.size();
  }
  DdCtx dd{};
  dd.na = na; dd.nap = nap; dd.nb2 = nb2; dd.nd = nd;
  int *d_vidx = nullptr, *d_alist = nullptr, *d_blist = nullptr, *d_bnbr = nullptr, *d_drow = nullptr, *d_aptr = nullptr, *d_sptr = nullptr, *d_spos = nullptr; int2 *d_aadj = nullptr; char *d_isdense = nullptr;
  if (dev_alloc(e, &d_vidx, (size_t)n) || dev_alloc(e, &d_alist, (size_t)na) || dev_alloc(e, &d_blist, (size_t)nb2) || dev_alloc(e, &d_bnbr, bnbr.size()) ||
      dev_alloc(e, &d_drow, (size_t)nd) || dev_alloc(e, &d_aptr, aptr.size()) || dev_alloc(e, &d_aadj, aadj.size()) || dev_alloc(e, &d_sptr, sptr.size()) || dev_alloc(e, &d_spos, spos.size()) || dev_alloc(e, &d_isdense, (size_t)m) || dev_alloc(e, &dd.bval, bnbr.size()) || dev_alloc(e, &dd.bdiag, (size_t)nb2) ||
      dev_alloc(e, &dd.S, (size_t)nap * nap) || dev_alloc(e, &dd.S0, (size_t)nap * nap) || dev_alloc(e, &dd.R, (size_t)nd * nap) || dev_alloc(e, &dd.dw, (size_t)nd) || dev_alloc(e, &dd.rr, (size_t)nap) ||
      dev_alloc(e, &dd.vv, (size_t)nap) || dev_alloc(e, &dd.D, (size_t)DD_NB * DD_NB) || dev_alloc(e, &dd.Bp, (size_t)DD_NB * nap) || dev_alloc(e, &dd.T, (size_t)DD_NB * nap) || dev_alloc(e, &dd.Tp, (size_t)DD_NB * nap) || dev_alloc(e, &dd.X2, (size_t)nap * nap) ||
      dev_alloc(e, &dd.flag, (size_t)4)) return HIPENG_ERR_HIP;
#define DDUP(dst, src) if (!(src).empty()) HIPCHK(hipMemcpyAsync(dst, (src).data(), (src).size() * sizeof((src)[0]), hipMemcpyHostToDevice, e->stream))
  {
    int symv = nap >= 2048;
    if (const char *x = getenv("OSQP_AMD_DENSE_SYMV")) symv = atoi(x) != 0;
    const size_t tiles = (size_t)(nap / DD_NB) * (nap / DD_NB + 1) / 2;
    if (symv && (dev_alloc(e, &dd.rowpart, tiles * DD_NB) || dev_alloc(e, &dd.colpart, tiles * DD_NB))) return HIPENG_ERR_HIP;
  }
  DDUP(d_vidx, vidx); DDUP(d_alist, alist); DDUP(d_blist, blist); DDUP(d_bnbr, bnbr); DDUP(d_drow, drow); DDUP(d_isdense, isdense); DDUP(d_aptr, aptr); DDUP(d_aadj, aadj); DDUP(d_sptr, sptr); DDUP(d_spos, spos);
#undef DDUP
  HIPCHK(hipStreamSynchronize(e->stream));        // (the sources are locals)
  dd.vidx = d_vidx; dd.alist = d_alist; dd.blist = d_blist; dd.bnbr = d_bnbr; dd.drow = d_drow; dd.isdense = d_isdense; dd.aptr = d_aptr; dd.aadj = d_aadj; dd.sptr = d_sptr; dd.spos = d_spos;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_dd_gemv), hipFuncAttributeMaxDynamicSharedMemorySize, nap * (int)sizeof(double)) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  e->dd = dd;
  e->dd_init_r = e->c.init_r; e->dd_init_stride = e->c.init_stride;       // (what the launch-per-step kernels use: direct_disable puts them back)
  e->c.init_r = e->c.r; e->c.init_stride = 1;
  e->c.fin_wave_rows = 1;
  {
    int one_gather = 1;
    if (const char *x = getenv("OSQP_AMD_DENSE_ONE_GATHER")) one_gather = atoi(x) != 0;
    // (only where k_pcg_init's pass over [P | A'] is more than a launch: on small problems k_vd's own launch would cost more than it saves)
    if (one_gather && nnzA >= 2e5 && dev_alloc(e, &e->c.vd, (size_t)(n + m))) return HIPENG_ERR_HIP;
  }
  e->res_kind = 4; e->res_on = e->res_use = true;
  if (e->trace) fprintf(stderr, "[osqp_amd] dense-direct solve: %d dense unknowns (%d x %d inverse, %.0f MB), %d sparse unknowns by Schur complement, %d dense rows of A\n",
                        na, nap, nap, 8e-6 * nap * nap, nb2, nd);
  return 0;
}

// New rho, sigma or matrix values: form the Schur complement again, invert it, and check the inverse against the matrix as formed.
// The block sweeps are Gauss-Jordan in blocks: on a positive definite matrix they cannot break down, but their error grows like
// cond(S)^2 eps (measured: |S^-1 S - I| = 5e-15 on the Lasso's S, 0.7 at cond 1e8) where a Cholesky factorisation has cond eps.  So every
// fresh inverse multiplies a fixed probe vector and the formed matrix multiplies the result; more than DD_CHECK off (or a pivot that
// was not positive) and the inverse is computed again by the Cholesky route; if that fails too the engine drops the dense solve.
#define DD_CHECK 1e-6
static int dd_refresh(hipeng *e) {
  const DdCtx &dd = e->dd;
  const int nap = dd.nap;
  HIPCHK(hipMemsetAsync(dd.flag, 0, 4 * sizeof(int), e->stream));
  if (dd.nd) HIPCHK(hipMemsetAsync(dd.R, 0, (size_t)dd.nd * nap * sizeof(double), e->stream));
  else HIPCHK(hipMemsetAsync(dd.S, 0, (size_t)nap * nap * sizeof(double), e->stream));
  if (dd.nb2) {
    HIPCHK(hipMemsetAsync(dd.bval, 0, (size_t)dd.nb2 * DD_NBR * sizeof(double), e->stream));
    HIPCHK(hipMemsetAsync(dd.bdiag, 0, (size_t)dd.nb2 * sizeof(double), e->stream));
  }
  if (dd.nd) {
    hipLaunchKernelGGL(k_dd_fill_R, dim3(dd.nd), dim3(TB), 0, e->stream, e->c, dd);
    dd_gemm(e->stream, dim3(nap / 128, nap / 128), dd.S, nap, (const double *)dd.R, nap, (const double *)dd.R, nap, (const double *)dd.dw,
                       nap, nap, dd.nd, 1.0, 0.0, -1, -1, -1, -1, 1, 0);
  }
  hipLaunchKernelGGL(k_dd_scatter, dim3(elem_grid(e->n)), dim3(TB), 0, e->stream, e->c, dd);
  hipLaunchKernelGGL(k_dd_schur, dim3(elem_grid(std::max(1, std::max(dd.nb2, nap)))), dim3(TB), 0, e->stream, dd);
  HIPCHK(hipMemcpyAsync(dd.S0, dd.S, (size_t)nap * nap * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
  hipLaunchKernelGGL(k_dd_mirror, dim3(nap / 64, nap / 64), dim3(TB), 0, e->stream, dd.S0, nap, 0);
  // The inverse: block sweeps first (21 ms at n_a = 5 000, error ~ cond^2 eps), checked against the matrix as formed:
  //   y = S^-1 u (into vv), z = S0 y (into Bp), err = max |z - u|.
  // Failed (or a pivot was not positive): the matrix as formed goes through the blocked Cholesky route (cond eps; five times slower
  // until its triangular inversion is parallelised better) and is checked again.  OSQP_AMD_DENSE_CHOL=1 takes that route at once.
  static int chol_first = -1;
  if (chol_first < 0) { chol_first = 0; if (const char *x = getenv("OSQP_AMD_DENSE_CHOL")) chol_first = atoi(x) != 0; }
  const dim3 gg(std::min(1024, nap / 4));
  const size_t lds = (size_t)nap * sizeof(double);
  double err = 0.0; int flag = 0;
  int attempt = chol_first ? 1 : 0;
  for (; attempt < 2; attempt++) {
    if (attempt == 1) {
      HIPCHK(hipMemcpyAsync(dd.S, dd.S0, (size_t)nap * nap * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
      HIPCHK(hipMemsetAsync(dd.flag, 0, 4 * sizeof(int), e->stream));
    }
    if (int rc = attempt ? dd_invert_chol(e->stream, dd.S, nap, dd.Bp, dd.T, dd.Tp, dd.X2, dd.flag) : dd_invert_sweep(e->stream, dd.S, nap, dd.D, dd.Bp, dd.T, dd.flag)) return rc;
    hipLaunchKernelGGL(k_dd_probe_fill, dim3(elem_grid(nap)), dim3(TB), 0, e->stream, dd);
    hipLaunchKernelGGL(k_dd_gemv, gg, dim3(TB), lds, e->stream, e->c, nap, (const double *)dd.S, (const double *)dd.rr, dd.vv, 0);
    hipLaunchKernelGGL(k_dd_gemv, gg, dim3(TB), lds, e->stream, e->c, nap, (const double *)dd.S0, (const double *)dd.vv, dd.Bp, 0);
    hipLaunchKernelGGL(k_dd_probe_err, dim3(1), dim3(TB), 0, e->stream, dd, (const double *)dd.Bp, dd.D);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&err, dd.D, sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(&flag, dd.flag, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (!flag && err <= DD_CHECK) break;
    if (e->trace) fprintf(stderr, "[osqp_amd] dense-direct: the %s inverse failed its check (%.2e%s)\n", attempt ? "Cholesky" : "block-sweep", err, flag ? ", a pivot was not positive" : "");
  }
  e->dd_chol = attempt >= 1;                // (the inverse in use came from the Cholesky route)
  e->dd_check = err;
  if (flag || !(err <= DD_CHECK)) direct_disable(e, "dense-direct", flag ? "a pivot was not positive" : "the inverse failed its check", err);
  return 0;
}

// (Tried for systems of at most 1024 unknowns, where a solve is launch-bound: gather, product and update in ONE launch, every workgroup
// forming the whole reduced residual itself and multiplying 32 rows of the inverse -- 125-iteration solves went from 3.4 / 4.1 / 4.7 /
// 6.1 ms to 3.6 / 4.5 / 5.6 / 7.5 ms at n = 150 / 300 / 600 / 1000: with nap / 32 workgroups the product is no longer spread over the machine.)
static void launch_dense_direct(hipeng *e) {
  const DdCtx &dd = e->dd;
  hipLaunchKernelGGL(k_dd_gather, dim3(elem_grid(dd.nap)), dim3(TB), 0, e->stream, e->c, dd);
  if (dd.rowpart) {
    hipLaunchKernelGGL(k_dd_symv_tiles, dim3(dd.nap / DD_NB, dd.nap / DD_NB), dim3(TB), 0, e->stream, e->c, dd.nap, (const double *)dd.S, (const double *)dd.rr, dd.rowpart, dd.colpart, 1);
  } else
  hipLaunchKernelGGL(k_dd_gemv, dim3(std::min(1024, dd.nap / 4)), dim3(TB), (size_t)dd.nap * sizeof(double), e->stream, e->c, dd.nap, (const double *)dd.S, (const double *)dd.rr, dd.vv, 1);
  hipLaunchKernelGGL(k_dd_finish, dim3(elem_grid(e->n)), dim3(TB), 0, e->stream, e->c, dd);
}

// Test / measurement hook: the blocked inversion alone.  A: n x n row-major symmetric positive definite on the host (n a multiple
// of 128); Ainv receives the inverse, ms[0] the time of the inversion on the device, ms[1] that of one n x n x n TN GEMM
// (C = A' A on the matrix cores: the kernel's rate).  Returns 0, or 1 when a pivot was not positive.
extern "C" int hipeng_dense_invert_selftest(int n, const double *A, double *Ainv, double *ms) {
  if (n <= 0 || n % DD_NB || !A || !Ainv) return HIPENG_ERR_ARG;
  double *dA = nullptr, *dC = nullptr, *D = nullptr, *Bp = nullptr, *T = nullptr, *Tp = nullptr; int *flag = nullptr;
  const size_t nn = (size_t)n * n;
  HIPCHK(hipMalloc(&dA, nn * 8)); HIPCHK(hipMalloc(&dC, nn * 8)); HIPCHK(hipMalloc(&D, DD_NB * DD_NB * 8)); HIPCHK(hipMalloc(&Bp, (size_t)DD_NB * n * 8));
  HIPCHK(hipMalloc(&T, (size_t)DD_NB * n * 8)); HIPCHK(hipMalloc(&Tp, (size_t)DD_NB * n * 8)); HIPCHK(hipMalloc(&flag, 16));
  HIPCHK(hipMemcpy(dA, A, nn * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemset(flag, 0, 16));
  hipEvent_t e0, e1, e2; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventCreate(&e2));
  HIPCHK(hipEventRecord(e0, 0));
  dd_gemm(0, dim3(n / 128, n / 128), dC, n, (const double *)dA, n, (const double *)dA, n, (const double *)nullptr, n, n, n, 1.0, 0.0, -1, -1, -1, -1, 0, 0);
  HIPCHK(hipEventRecord(e1, 0));
  int chol = 1;
  if (const char *x = getenv("OSQP_AMD_DENSE_CHOL")) chol = atoi(x) != 0;
  if (int rc = chol ? dd_invert_chol(0, dA, n, Bp, T, Tp, dC, flag) : dd_invert_sweep(0, dA, n, D, Bp, T, flag)) return rc;
  HIPCHK(hipEventRecord(e2, 0)); HIPCHK(hipEventSynchronize(e2));
  float t0 = 0, t1 = 0; HIPCHK(hipEventElapsedTime(&t0, e0, e1)); HIPCHK(hipEventElapsedTime(&t1, e1, e2));
  if (ms) { ms[0] = t1; ms[1] = t0; }
  int hf = 0;
  HIPCHK(hipMemcpy(Ainv, dA, nn * 8, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(&hf, flag, 4, hipMemcpyDeviceToHost));
  (void)hipFree(dA); (void)hipFree(dC); (void)hipFree(D); (void)hipFree(Bp); (void)hipFree(T); (void)hipFree(Tp); (void)hipFree(flag);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
  return hf ? 1 : 0;
}
