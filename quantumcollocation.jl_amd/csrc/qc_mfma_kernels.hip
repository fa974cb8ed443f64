// MFMA (v_mfma_f64_16x16x4_f64) kernels for 2N in {16, 32}. Placeholder until the register-resident
// path lands; qc_mfma_supported() == false routes every descriptor to the LDS kernels.
#include "qc_internal.h"

bool qc_mfma_supported(const QcParams&) { return false; }
size_t qc_mfma_gx_doubles(const QcParams&) { return 0; }
void qc_mfma_pack_G(const QcParams&, const double*, double*) {}
hipError_t qc_launch_mfma_F_jac(const QcParams&, const double*, double*, double*, hipStream_t) { return hipErrorNotSupported; }
hipError_t qc_launch_mfma_hess(const QcParams&, const double*, const double*, double*, hipStream_t) { return hipErrorNotSupported; }
