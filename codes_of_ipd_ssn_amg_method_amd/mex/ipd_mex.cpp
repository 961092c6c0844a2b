// ipd_mex.cpp -- MEX gateway: MATLAB <-> the C ABI of libipdamg.so (include/ipd_amg.h).
//
// Build (on a machine with MATLAB; not compilable in this pipeline, which has no mex.h):
//   mex -R2018a -I../../include ipd_mex.cpp -L.. -lipdamg -output ipd_mex
// Call convention: out = ipd_mex('<function>', args...), used by the same-named .m shims
// in this directory, which keep the reference's signatures (ASAt.m:2, Ax.m:2, ...).
//
// Conventions honoured here: sparse mxArrays are CSC with mwIndex
// (uint64) Jc/Ir -- binary compatible with the int64 arrays of ipd_csc for sizes < 2^63;
// logical arrays are mxLogical bytes; MATLAB owns the inputs; errors are raised through
// mexErrMsgIdAndTxt with ipd_last_error(); one process-wide context and one process-wide
// "current hierarchy" emulate the reference's `global Ack Prok J smoth_it Rk`.
#include <cstring>
#include <string>
#include <vector>

#include "ipd_amg.h"
#include "mex.h"

static ipd_ctx* g_ctx = nullptr;
static ipd_rng* g_rng = nullptr;   // MATLAB-compatible stream; reseed with ipd_mex('rng', seed)
static ipd_amg* g_h = nullptr;     // the "global" hierarchy of Class_AMG / MG_Vcycle / MG_Wcycle
static ipd_apd* g_apd = nullptr;   // the driver workspace of apd_create / apd_warmup / apd_run

static void at_exit() {
    if (g_apd) ipd_apd_destroy(g_apd);
    g_apd = nullptr;
    if (g_h) ipd_amg_destroy(g_h);
    if (g_rng) ipd_rng_destroy(g_rng);
    if (g_ctx) ipd_ctx_destroy(g_ctx);
    g_h = nullptr; g_rng = nullptr; g_ctx = nullptr;
}
static void chk(int rc) {
    if (rc != IPD_OK) mexErrMsgIdAndTxt("ipdamg:error", "%s", ipd_last_error());
}
static void ensure() {
    if (!g_ctx) { chk(ipd_ctx_create(0, &g_ctx)); mexAtExit(at_exit); }
    if (!g_rng) chk(ipd_rng_create(5489u, &g_rng));
}
static ipd_csc csc_of(const mxArray* a) {
    if (!mxIsSparse(a) || !mxIsDouble(a)) mexErrMsgIdAndTxt("ipdamg:arg", "sparse double expected");
    ipd_csc c;
    c.nrows = (int64_t)mxGetM(a); c.ncols = (int64_t)mxGetN(a);
    c.jc = reinterpret_cast<const int64_t*>(mxGetJc(a));
    c.ir = reinterpret_cast<const int64_t*>(mxGetIr(a));
    c.pr = mxGetDoubles(a);
    c.nnz = c.jc[c.ncols];
    return c;
}
static mxArray* to_mx(ipd_csc_out* o) {
    mxArray* a = mxCreateSparse((mwSize)o->nrows, (mwSize)o->ncols, (mwSize)(o->nnz ? o->nnz : 1), mxREAL);
    std::memcpy(mxGetJc(a), o->jc, sizeof(int64_t) * (size_t)(o->ncols + 1));
    if (o->nnz) {
        std::memcpy(mxGetIr(a), o->ir, sizeof(int64_t) * (size_t)o->nnz);
        std::memcpy(mxGetDoubles(a), o->pr, sizeof(double) * (size_t)o->nnz);
    }
    ipd_csc_free(o);
    return a;
}
static mxArray* col(size_t n) { return mxCreateDoubleMatrix((mwSize)n, 1, mxREAL); }
static mxArray* logical_col(const std::vector<uint8_t>& v) {
    mxArray* a = mxCreateLogicalMatrix((mwSize)v.size(), 1);
    std::memcpy(mxGetLogicals(a), v.data(), v.size());
    return a;
}
static double field(const mxArray* s, const char* f, double dflt) {
    const mxArray* v = mxIsStruct(s) ? mxGetField(s, 0, f) : nullptr;
    if (!v || mxIsEmpty(v)) return dflt;                    // isempty -> default (Class_AMG.m:26-34)
    return mxIsChar(v) ? (double)(unsigned char)mxArrayToString(v)[0] : mxGetScalar(v);
}
static ipd_amg_opts opts_of(const mxArray* s) {
    ipd_amg_opts o; ipd_amg_opts_init(&o);
    if (!s || !mxIsStruct(s)) {   // Class_AMG.m:22-23, nargin == 2
        o.retol = 1e-12; o.bigph = 0; o.maxit = 20; o.theta = 0.25; o.smoth = 10; o.cycle = 1;
        o.isnsp = 1; o.inter = 1; return o;
    }
    o.retol = field(s, "retol", -1); o.theta = field(s, "theta", -1);
    o.bigph = (int32_t)field(s, "bigph", -1); o.maxit = (int32_t)field(s, "maxit", -1);
    o.smoth = (int32_t)field(s, "smoth", -1); o.cycle = (int32_t)field(s, "cycle", -1);
    o.isnsp = (int32_t)field(s, "isnsp", -1); o.inter = (int32_t)field(s, "inter", -1);
    o.fnode = (int64_t)field(s, "fnode", -1);
    return o;
}
static const double* opt_vec(const mxArray* s, const char* f) {
    const mxArray* v = (s && mxIsStruct(s)) ? mxGetField(s, 0, f) : nullptr;
    return (v && !mxIsEmpty(v)) ? mxGetDoubles(v) : nullptr;
}
static std::vector<double> diag_of_sparse(const mxArray* T, size_t M, bool* any) {
    std::vector<double> t(M, 0.0); *any = false;
    if (!T || mxIsEmpty(T)) return t;
    ipd_csc c = csc_of(T);
    for (int64_t j = 0; j < c.ncols; ++j)
        for (int64_t k = c.jc[j]; k < c.jc[j + 1]; ++k)
            if (c.ir[k] == j) { t[(size_t)j] = c.pr[k]; *any = *any || c.pr[k] != 0.0; }
    return t;
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    if (nrhs < 1 || !mxIsChar(prhs[0])) mexErrMsgIdAndTxt("ipdamg:arg", "ipd_mex('<name>', ...)");
    const std::string fn = mxArrayToString(prhs[0]);
    const mxArray** a = prhs + 1;
    ensure();
    if (fn == "rng") {               // keep the library stream in step with MATLAB's rng(seed)
        ipd_rng_destroy(g_rng); g_rng = nullptr;
        chk(ipd_rng_create((uint32_t)mxGetScalar(a[0]), &g_rng));
    } else if (fn == "Ax" || fn == "Aty") {      // y = Ax(x,p,q) / z = Aty(y,p,q)
        const int64_t m = (int64_t)mxGetNumberOfElements(a[1]), n = (int64_t)mxGetNumberOfElements(a[2]);
        if (fn == "Ax") { plhs[0] = col((size_t)(m + n));
            chk(ipd_ax(g_ctx, mxGetDoubles(a[0]), mxGetDoubles(a[1]), mxGetDoubles(a[2]), m, n, mxGetDoubles(plhs[0])));
        } else { plhs[0] = col((size_t)(m * n));
            chk(ipd_aty(g_ctx, mxGetDoubles(a[0]), mxGetDoubles(a[1]), mxGetDoubles(a[2]), m, n, mxGetDoubles(plhs[0]))); }
    } else if (fn == "ASAt") {                   // H = ASAt(s,p,q), s logical
        const int64_t m = (int64_t)mxGetNumberOfElements(a[1]), n = (int64_t)mxGetNumberOfElements(a[2]);
        if (!mxIsLogical(a[0])) mexErrMsgIdAndTxt("ipdamg:arg", "ASAt: s must be logical");
        ipd_csc_out H; chk(ipd_asat(g_ctx, (const uint8_t*)mxGetLogicals(a[0]), mxGetDoubles(a[1]), mxGetDoubles(a[2]), m, n, &H));
        plhs[0] = to_mx(&H);
    } else if (fn == "invAAt") {                 // shim resolves nargin (invAAt.m:7-12)
        const int64_t m = (int64_t)mxGetNumberOfElements(a[1]), n = (int64_t)mxGetNumberOfElements(a[2]);
        plhs[0] = col((size_t)(m + n));
        chk(ipd_inv_aat(g_ctx, mxGetDoubles(a[0]), mxGetDoubles(a[1]), mxGetDoubles(a[2]), m, n, mxGetScalar(a[3]), mxGetScalar(a[4]), mxGetDoubles(plhs[0])));
    } else if (fn == "invHHt") {
        const int64_t m = (int64_t)mxGetNumberOfElements(a[1]), n = (int64_t)mxGetNumberOfElements(a[2]);
        plhs[0] = col((size_t)(m + n + 1));
        chk(ipd_inv_hht(g_ctx, mxGetDoubles(a[0]), mxGetDoubles(a[1]), mxGetDoubles(a[2]), m, n, mxGetScalar(a[3]), mxGetDoubles(a[4]), mxGetDoubles(plhs[0])));
    } else if (fn == "strength") {
        ipd_csc A = csc_of(a[0]); ipd_csc_out S;
        chk(ipd_strength(g_ctx, &A, nrhs > 2 ? (int)mxGetScalar(a[1]) : 2, &S)); plhs[0] = to_mx(&S);
    } else if (fn == "cf_split" || fn == "mis_set") {
        ipd_csc A = csc_of(a[0]); std::vector<uint8_t> c((size_t)A.nrows), f((size_t)A.nrows);
        if (fn == "cf_split") chk(ipd_cf_split(g_ctx, &A, c.data(), f.data()));
        else { ipd_csc_out As; chk(ipd_mis_set(g_ctx, &A, nrhs > 2 ? mxGetScalar(a[1]) : 0.025, g_rng, c.data(), f.data(), &As));
               if (nlhs > 2) plhs[2] = to_mx(&As); else ipd_csc_free(&As); }
        plhs[0] = logical_col(c); if (nlhs > 1) plhs[1] = logical_col(f);
    } else if (fn == "transfer") {               // [Ac,Pro,~,indC] = transfer(A,opts,J)
        ipd_csc A = csc_of(a[0]); ipd_amg_opts o = opts_of(nrhs > 2 ? a[1] : nullptr);
        ipd_csc_out Ac, Pro; std::vector<uint8_t> c((size_t)A.nrows);
        chk(ipd_transfer(g_ctx, &A, &o, nrhs > 3 ? (int)mxGetScalar(a[2]) : 2, g_rng, &Ac, &Pro, c.data()));
        plhs[0] = to_mx(&Ac); if (nlhs > 1) plhs[1] = to_mx(&Pro); else ipd_csc_free(&Pro);
        if (nlhs > 3) { plhs[2] = mxCreateDoubleMatrix(0, 0, mxREAL); plhs[3] = logical_col(c); }
    } else if (fn == "Class_AMG") {              // [x,it,rel_res,rel_resk,rhok] = Class_AMG(A,b,opts)
        ipd_csc A = csc_of(a[0]); const mxArray* so = nrhs > 3 ? a[2] : nullptr;
        ipd_amg_opts o = opts_of(so);
        if (g_h) { ipd_amg_destroy(g_h); g_h = nullptr; }
        chk(ipd_amg_setup(g_ctx, &A, &o, g_rng, &g_h));
        const int maxit = o.maxit >= 0 ? o.maxit : 50;
        plhs[0] = col((size_t)A.nrows); int32_t it = 0; double rel = 0;
        std::vector<double> rk((size_t)maxit + 2), rho((size_t)maxit + 2);
        chk(ipd_amg_solve(g_h, mxGetDoubles(a[1]), opt_vec(so, "guess"), mxGetDoubles(plhs[0]), &it, &rel, rk.data(), rho.data()));
        if (nlhs > 1) plhs[1] = mxCreateDoubleScalar(it);
        if (nlhs > 2) plhs[2] = mxCreateDoubleScalar(rel);
        if (nlhs > 3) { plhs[3] = col((size_t)it + 1); std::memcpy(mxGetDoubles(plhs[3]), rk.data(), sizeof(double) * ((size_t)it + 1)); }
        if (nlhs > 4) { plhs[4] = col((size_t)it + 1); std::memcpy(mxGetDoubles(plhs[4]), rho.data(), sizeof(double) * ((size_t)it + 1)); }
    } else if (fn == "MG_Vcycle" || fn == "MG_Wcycle") {   // e = MG_?cycle(r,isnsp,k[,e]) on the current hierarchy
        if (!g_h) mexErrMsgIdAndTxt("ipdamg:state", "no hierarchy: call Class_AMG first (global Ack Prok J Rk)");
        const int isnsp = nrhs > 2 ? (int)mxGetScalar(a[1]) : 0, k = nrhs > 3 ? (int)mxGetScalar(a[2]) : 1;
        plhs[0] = col(mxGetNumberOfElements(a[0]));
        if (fn == "MG_Vcycle") chk(ipd_amg_vcycle(g_h, mxGetDoubles(a[0]), isnsp, k, mxGetDoubles(plhs[0])));
        else chk(ipd_amg_wcycle(g_h, mxGetDoubles(a[0]), isnsp, k, nrhs > 4 ? mxGetDoubles(a[3]) : nullptr, mxGetDoubles(plhs[0])));
    } else if (fn == "PCG") {                    // [d,it,res,resk] = PCG(H,e,pcg_options)
        ipd_csc H = csc_of(a[0]); ipd_pcg_opts o; ipd_pcg_opts_init(&o); const mxArray* so = nrhs > 3 ? a[2] : nullptr;
        if (so) { o.retol = field(so, "retol", -1); o.maxit = (int64_t)field(so, "maxit", -1); o.precd = (int32_t)field(so, "precd", -1); o.nf = (int64_t)field(so, "nf", 0); }
        const int64_t maxit = o.maxit >= 0 ? o.maxit : 10000;
        plhs[0] = col((size_t)H.nrows); int64_t it = 0; double res = 0; std::vector<double> rk((size_t)maxit + 1);
        chk(ipd_pcg(g_ctx, &H, mxGetDoubles(a[1]), opt_vec(so, "guess"), &o, mxGetDoubles(plhs[0]), &it, &res, rk.data()));
        if (nlhs > 1) plhs[1] = mxCreateDoubleScalar((double)it);
        if (nlhs > 2) plhs[2] = mxCreateDoubleScalar(res);
        if (nlhs > 3) { plhs[3] = col((size_t)maxit); std::memcpy(mxGetDoubles(plhs[3]), rk.data(), sizeof(double) * (size_t)maxit); }
    } else if (fn == "components") {             // 1-based outputs for MATLAB
        ipd_csc A = csc_of(a[0]); const size_t N = (size_t)A.nrows; int64_t nc = 0;
        std::vector<int64_t> b(N), sz(N), p(N), r(N + 1);
        chk(ipd_components(g_ctx, &A, b.data(), sz.data(), p.data(), r.data(), &nc));
        plhs[0] = mxCreateDoubleMatrix(1, (mwSize)N, mxREAL);
        for (size_t i = 0; i < N; ++i) mxGetDoubles(plhs[0])[i] = (double)(b[i] + 1);
        if (nlhs > 1) { plhs[1] = mxCreateDoubleMatrix(1, (mwSize)nc, mxREAL); for (int64_t c = 0; c < nc; ++c) mxGetDoubles(plhs[1])[c] = (double)sz[(size_t)c]; }
        if (nlhs > 2) { plhs[2] = mxCreateDoubleMatrix(1, (mwSize)N, mxREAL); for (size_t i = 0; i < N; ++i) mxGetDoubles(plhs[2])[i] = (double)(p[i] + 1); }
        if (nlhs > 3) { plhs[3] = mxCreateDoubleMatrix(1, (mwSize)nc + 1, mxREAL); for (int64_t c = 0; c <= nc; ++c) mxGetDoubles(plhs[3])[c] = (double)(r[(size_t)c] + 1); }
    } else if (fn == "Hybrid_AMG" || fn == "AMG4POT" || fn == "Hybrid_twogrid" || fn == "AMG4POT_twogrid" ||
               fn == "aug_PCG" || fn == "PCG4POT") {   // [zeta,it,res,info] = f(prob_data,options)
        const mxArray* pd = a[0]; const bool pot = fn == "AMG4POT" || fn == "AMG4POT_twogrid" || fn == "PCG4POT";
        const mxArray *p = mxGetField(pd, 0, "p"), *q = mxGetField(pd, 0, "q");
        ipd_prob P; std::memset(&P, 0, sizeof(P));
        P.m = (int64_t)mxGetNumberOfElements(p); P.n = (int64_t)mxGetNumberOfElements(q);
        P.bk1 = mxGetScalar(mxGetField(pd, 0, "bk1")); P.tk = mxGetScalar(mxGetField(pd, 0, "tk"));
        P.p = mxGetDoubles(p); P.q = mxGetDoubles(q);
        ipd_csc H0 = csc_of(mxGetField(pd, 0, "H0")); P.H0 = &H0; P.z = mxGetDoubles(mxGetField(pd, 0, "z"));
        bool anyT = false; std::vector<double> t = diag_of_sparse(mxGetField(pd, 0, "T"), (size_t)(P.m + P.n), &anyT);
        P.t = anyT ? t.data() : nullptr;
        if (pot) { P.s = (const uint8_t*)mxGetLogicals(mxGetField(pd, 0, "s")); P.phi = mxGetDoubles(mxGetField(pd, 0, "phi")); }
        int32_t it = 0; double res = 0; int64_t info[2] = {0, 0};
        plhs[0] = col((size_t)(P.m + P.n + (pot ? 1 : 0)));
        if (fn == "aug_PCG" || fn == "PCG4POT") {          // pcg_options: retol, maxit (precd forced to 2)
            ipd_pcg_opts po; ipd_pcg_opts_init(&po); int64_t it64 = 0;
            po.retol = field(a[1], "retol", -1); po.maxit = (int64_t)field(a[1], "maxit", -1);
            chk(pot ? ipd_pcg4pot(g_ctx, &P, &po, mxGetDoubles(plhs[0]), &it64, &res, info)
                    : ipd_aug_pcg(g_ctx, &P, &po, mxGetDoubles(plhs[0]), &it64, &res, info));
            it = (int32_t)it64;
        } else {
            ipd_amg_opts o = opts_of(a[1]);
            if (fn == "Hybrid_AMG") chk(ipd_hybrid_amg(g_ctx, &P, &o, g_rng, mxGetDoubles(plhs[0]), &it, &res, info));
            else if (fn == "AMG4POT") chk(ipd_amg4pot(g_ctx, &P, &o, g_rng, mxGetDoubles(plhs[0]), &it, &res, info));
            else if (fn == "Hybrid_twogrid") chk(ipd_hybrid_twogrid(g_ctx, &P, &o, g_rng, mxGetDoubles(plhs[0]), &it, &res, info));
            else chk(ipd_amg4pot_twogrid(g_ctx, &P, &o, g_rng, mxGetDoubles(plhs[0]), &it, &res, info));
        }
        if (nlhs > 1) plhs[1] = mxCreateDoubleScalar(it);
        if (nlhs > 2) plhs[2] = mxCreateDoubleScalar(res);
        if (nlhs > 3) { plhs[3] = mxCreateDoubleMatrix(1, 2, mxREAL); mxGetDoubles(plhs[3])[0] = (double)info[0]; mxGetDoubles(plhs[3])[1] = (double)info[1]; }
    } else if (fn == "apd_create") {   // ipd_mex('apd_create', cls, c,r,l,p,q, gama | mu,phi): the loaded workspace
        ipd_apd_data d; std::memset(&d, 0, sizeof(d));
        d.cls = (int32_t)mxGetScalar(a[0]);
        d.c = mxGetDoubles(a[1]); d.r = mxGetDoubles(a[2]); d.l = mxGetDoubles(a[3]);
        d.p = mxGetDoubles(a[4]); d.q = mxGetDoubles(a[5]);
        d.m = (int64_t)mxGetNumberOfElements(a[4]); d.n = (int64_t)mxGetNumberOfElements(a[5]);
        if (d.cls == 1) {
            if (mxGetNumberOfElements(a[6]) == 1) d.gama_scalar = mxGetScalar(a[6]);
            else d.gama = mxGetDoubles(a[6]);
        } else {
            d.mu = mxGetScalar(a[6]); d.phi = mxGetDoubles(a[7]);
        }
        if (g_apd) { ipd_apd_destroy(g_apd); g_apd = nullptr; }
        chk(ipd_apd_create(g_ctx, &d, &g_apd));
    } else if (fn == "apd_warmup" || fn == "apd_state") {   // [uk,lk] = ... (warmup_class1.m:2 / warmup_class2.m:1)
        if (!g_apd) mexErrMsgIdAndTxt("ipdamg:state", "no workspace: call apd_create first");
        if (fn == "apd_warmup") {
            const double res = mxGetScalar(a[0]), mi = mxGetScalar(a[1]);
            chk(ipd_apd_warmup(g_apd, res, mxIsInf(mi) ? -1 : (int64_t)mi));
        }
        int64_t U = 0, L = 0; chk(ipd_apd_dims(g_apd, &U, &L));
        plhs[0] = col((size_t)U); mxArray* lk = col((size_t)L); double bk = 0;
        chk(ipd_apd_get_state(g_apd, mxGetDoubles(plhs[0]), nullptr, mxGetDoubles(lk), &bk));
        if (nlhs > 1) plhs[1] = lk; else mxDestroyArray(lk);
        if (nlhs > 2) plhs[2] = mxCreateDoubleScalar(bk);
    } else if (fn == "apd_run") {      // out = ipd_mex('apd_run', amg_options[, iters]): the main loop, inner_solver = 4
        if (!g_apd) mexErrMsgIdAndTxt("ipdamg:state", "no workspace: call apd_create first");
        ipd_amg_opts o = opts_of(a[0]); ipd_apd_result r;
        chk(ipd_apd_run(g_apd, nullptr, &o, g_rng, nrhs > 2 ? (int32_t)mxGetScalar(a[1]) : 1000000, &r));
        const char* names[] = {"converged", "k", "fval", "rr", "SumAMG", "TotalAMG", "FailAMG", "MaxAMG",
                               "fxk", "KKT_xk", "KKT_lk", "KKT_yk", "KKT_zk", "SsN_itnum"};
        plhs[0] = mxCreateStructMatrix(1, 1, 14, names);
        const double sc[8] = {(double)r.converged, (double)r.k, r.fval, r.rr, (double)r.sum_amg,
                              (double)r.total_amg, (double)r.fail_amg, (double)r.max_amg};
        for (int i = 0; i < 8; ++i) mxSetFieldByNumber(plhs[0], 0, i, mxCreateDoubleScalar(sc[i]));
        for (int w = 0; w < 6; ++w) {   // histories as column vectors, like the script's variables
            int64_t cnt = 0; chk(ipd_apd_history(g_apd, w, nullptr, 0, &cnt));
            mxArray* v = col((size_t)cnt);
            chk(ipd_apd_history(g_apd, w, mxGetDoubles(v), cnt, &cnt));
            mxSetFieldByNumber(plhs[0], 0, 8 + w, v);
        }
    } else {
        mexErrMsgIdAndTxt("ipdamg:arg", "unknown function '%s'", fn.c_str());
    }
}
