// G1 instantiation of the MSM pipeline (/root/reference/src/g1.rs:602-632) and of normalize_batch (src/g1.rs:537-543).
#include "msm_curve.hpp"

namespace mi {

int g1_set_bases(mi_ctx* ctx, const void* bases, size_t n, unsigned precompute_c) { return set_bases_impl<msmk::G1C>(ctx, bases, n, precompute_c); }
int g1_set_bases_device(mi_ctx* ctx, const void* d_bases, size_t n) { return set_bases_impl<msmk::G1C>(ctx, d_bases, n, 0, BaseSrc::DeviceAffine); }
int g1_set_bases_from_jacobian(mi_ctx* ctx, const void* jac, size_t n) { return set_bases_impl<msmk::G1C>(ctx, jac, n, 0, BaseSrc::HostJacobian); }
void g1_install_resident(mi_ctx* ctx, size_t k, const void* d_affine, size_t lo, size_t n, bool validated) { install_resident<msmk::G1C>(ctx, k, d_affine, lo, n, validated); }
int g1_msm(mi_ctx* ctx, const void* bases, const uint8_t* scalars, bool scalars_on_device, size_t n, unsigned fmt, void* out) {
    return msm_impl<msmk::G1C>(ctx, bases, scalars, scalars_on_device, n, fmt, out);
}
int g1_msm_windows(mi_ctx* ctx, const uint8_t* d_scalars, size_t n, unsigned fmt, void* d_out, mi_window_info* info) {
    return msm_windows_impl<msmk::G1C>(ctx, d_scalars, n, fmt, d_out, info);
}
int g1_msm_batch(mi_ctx* ctx, const uint8_t* const* scalars, bool scalars_on_device, size_t k, size_t n, unsigned fmt, mi_g1* out) {
    return msm_batch_impl<msmk::G1C>(ctx, scalars, scalars_on_device, k, n, fmt, out);
}
int g1_normalize(mi_ctx* ctx, const mi_g1* in, bool on_device, size_t n, mi_g1_affine* out) { return normalize_impl<msmk::G1C>(ctx, in, on_device, n, out); }

CurveCost g1_cost() { return HostCurve<msmk::G1C>::cost(); }

}  // namespace mi
