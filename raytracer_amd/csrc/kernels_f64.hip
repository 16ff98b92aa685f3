// kernels_f64.hip -- FP64 closest-hit check kernels for gfx950 (product code).
// Built with -ffp-contract=off: every * and + rounds separately, / and sqrt are IEEE, so prim ids and t values are
// bit-identical to the reference's hittable_list_hit (hittable_list.c:125-143) run on an x86-64 host.
// One lane per ray; the primitive loop index is wave-uniform, so the scene tables arrive through scalar loads.
#include <hip/hip_runtime.h>

#include "device/rt_core.hpp"
#include "kernels.h"

namespace {

struct NoMedium { __device__ bool operator()(int, double, double &) const { return false; } };

__device__ void first_hit_one(const SceneT<double> &sc, const RayT<double> &r, double tmin, double tmax,
                              int32_t *prim_id, double *t_out, double *rec12) {
    NoMedium nm;
    const HitT<double> h = closest_hit_ref<double>(sc, r, tmin, tmax, nm);   // the reference's tree and culling, not the production BVH
    *prim_id = h.prim;
    *t_out = h.prim >= 0 ? h.t : 0.0;
    if (rec12) {
        for (int k = 0; k < 12; ++k) rec12[k] = 0.0;
        rec12[11] = -1.0;
        if (h.prim >= 0) {
            const PrimT<double> p = sc.prims[h.prim];
            const RecT<double> rec = resolve_hit<double>(sc, r, p, h.t, true);
            rec12[0] = 1.0; rec12[1] = h.t; rec12[2] = rec.p.x; rec12[3] = rec.p.y; rec12[4] = rec.p.z;
            rec12[5] = rec.n.x; rec12[6] = rec.n.y; rec12[7] = rec.n.z; rec12[8] = rec.u; rec12[9] = rec.v;
            rec12[10] = rec.front; rec12[11] = sc.materials[rec.mat].type;
        }
    }
}

__global__ void __launch_bounds__(256) k_first_hit_f64(SceneT<double> sc, const double *__restrict__ rays7, size_t n, double tmin, double tmax,
                                                       int32_t *__restrict__ prim_id, double *__restrict__ t_out, double *__restrict__ rec12) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *a = rays7 + 7 * i;
    RayT<double> r; r.o = mk3<double>(a[0], a[1], a[2]); r.d = mk3<double>(a[3], a[4], a[5]); r.time = a[6];
    first_hit_one(sc, r, tmin, tmax, prim_id + i, t_out + i, rec12 ? rec12 + 12 * i : nullptr);
}

__global__ void __launch_bounds__(256) k_primary_first_hit_f64(SceneT<double> sc, CameraT<double> cam, const double *__restrict__ st2,
                                                               const double *__restrict__ lens3, size_t n, double tmin, double tmax,
                                                               int32_t *__restrict__ prim_id, double *__restrict__ t_out, double *__restrict__ rays7_out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double dx = lens3 ? lens3[3 * i] : 0.0, dy = lens3 ? lens3[3 * i + 1] : 0.0, tm = lens3 ? lens3[3 * i + 2] : 0.0;
    const RayT<double> r = camera_ray<double>(cam, st2[2 * i], st2[2 * i + 1], dx, dy, tm);
    if (rays7_out) { double *o = rays7_out + 7 * i; o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z; o[6] = r.time; }
    first_hit_one(sc, r, tmin, tmax, prim_id + i, t_out + i, nullptr);
}

}  // namespace

hipError_t launch_first_hit_f64(const SceneT<double> &sc, const double *rays7, size_t n, double tmin, double tmax,
                                int32_t *prim_id, double *t, double *rec12, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_first_hit_f64, dim3(blocks), dim3(256), 0, stream, sc, rays7, n, tmin, tmax, prim_id, t, rec12);
    return hipGetLastError();
}
hipError_t launch_primary_first_hit_f64(const SceneT<double> &sc, const CameraT<double> &cam, const double *st2, const double *lens3,
                                        size_t n, double tmin, double tmax, int32_t *prim_id, double *t, double *rays7_out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_primary_first_hit_f64, dim3(blocks), dim3(256), 0, stream, sc, cam, st2, lens3, n, tmin, tmax, prim_id, t, rays7_out);
    return hipGetLastError();
}
