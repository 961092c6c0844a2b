/* TEST STUB -- NOT MATLAB's mex.h.
 *
 * Declarations (no definitions) of the handful of MATLAB C Matrix / MEX API entry points that
 * codes_of_ipd_ssn_amg_method_amd/mex/ipd_mex.cpp uses, with the signatures of MATLAB's
 * documented interleaved-complex API (-R2018a), so that the gateway can be SYNTAX- and
 * TYPE-checked in a pipeline that has no MATLAB (tests/test_mex_gateway_compiles.py:
 * `g++ -fsyntax-only`).  Nothing links against this header and nothing is executed.        */
#ifndef IPD_TEST_MEX_STUB_H
#define IPD_TEST_MEX_STUB_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef bool mxLogical;
typedef double mxDouble;
typedef enum { mxREAL, mxCOMPLEX } mxComplexity;

mxDouble* mxGetDoubles(const mxArray*);
double mxGetScalar(const mxArray*);
size_t mxGetNumberOfElements(const mxArray*);
size_t mxGetM(const mxArray*);
size_t mxGetN(const mxArray*);
mxArray* mxGetField(const mxArray*, mwIndex, const char*);
void mxSetFieldByNumber(mxArray*, mwIndex, int, mxArray*);
mxArray* mxCreateDoubleScalar(double);
mxArray* mxCreateDoubleMatrix(mwSize, mwSize, mxComplexity);
mxArray* mxCreateSparse(mwSize, mwSize, mwSize, mxComplexity);
mxArray* mxCreateLogicalMatrix(mwSize, mwSize);
mxArray* mxCreateStructMatrix(mwSize, mwSize, int, const char**);
void mxDestroyArray(mxArray*);
bool mxIsStruct(const mxArray*);
bool mxIsEmpty(const mxArray*);
bool mxIsChar(const mxArray*);
bool mxIsSparse(const mxArray*);
bool mxIsLogical(const mxArray*);
bool mxIsDouble(const mxArray*);
bool mxIsInf(double);
mxLogical* mxGetLogicals(const mxArray*);
mwIndex* mxGetJc(const mxArray*);
mwIndex* mxGetIr(const mxArray*);
char* mxArrayToString(const mxArray*);
void mxFree(void*);
void mexErrMsgIdAndTxt(const char*, const char*, ...);
int mexAtExit(void (*)(void));
void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);
#ifdef __cplusplus
}
#endif
#endif
