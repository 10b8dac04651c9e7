// Train-time augmentation on the device, following the reference's
// segdino3d/datasets/transform/point_cloud_transforms.py: CustomRandomFlip3D :36-157, CustomGlobalRotScaleTrans
// :167-354 (rotation about z, then scale, then translation), NormalizePointsColor :357-389 and ElasticTransfrom
// :392-473 (two granularities of blurred Gaussian noise, trilinear interpolation, applied to the voxel-unit coordinates
// of the points and of the 2D-query centres).  The random draws stay on the host (numpy.random in the reference's order,
// segdino3d_amd/augment.py); everything that touches N points or the noise volumes runs here.  All of it is streaming
// work: one read + one write per point, the noise volumes (a few hundred KB) live in L2.
#include "common.h"
#include "../../include/segdino3d_hip.h"

struct AffineParams { int flip_x, flip_y; float c, s, scale, t[3]; int color; float mean[3], stdv[3]; };

// in place on points[:, 0:3] (and [:, 3:6] when color): the reference's operation order, one fp32 rounding per step
__global__ __launch_bounds__(256) void augment_points_kernel(float* __restrict__ pts, int ld, int64_t n, const AffineParams p) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float* r = pts + i * ld;
    float x = r[0], y = r[1], z = r[2];
    if (p.flip_x) x = -x;
    if (p.flip_y) y = -y;
    // points @ [[c, s, 0], [-s, c, 0], [0, 0, 1]]
    const float xr = __fadd_rn(__fmul_rn(x, p.c), __fmul_rn(y, -p.s));
    const float yr = __fadd_rn(__fmul_rn(x, p.s), __fmul_rn(y, p.c));
    r[0] = __fadd_rn(__fmul_rn(xr, p.scale), p.t[0]);
    r[1] = __fadd_rn(__fmul_rn(yr, p.scale), p.t[1]);
    r[2] = __fadd_rn(__fmul_rn(z, p.scale), p.t[2]);
    if (p.color) {
#pragma unroll
        for (int c = 0; c < 3; ++c) r[3 + c] = __fdiv_rn(__fsub_rn(r[3 + c], p.mean[c]), p.stdv[c]);
    }
}

// out[i, 0:3] = xyz[i] / voxel_size (fp32 division, as numpy does for a float32 array and a Python float)
__global__ __launch_bounds__(256) void voxel_units_kernel(const float* __restrict__ pts, int ld, int64_t n, float voxel_size, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) out[i * 3 + c] = __fdiv_rn(pts[i * ld + c], voxel_size);
}

// scipy.ndimage.convolve(n, ones(3) / 3 along `axis`, mode='constant', cval=0) on `grids` volumes [D0, D1, D2]:
// float32 weights, double accumulation, float32 result
__global__ __launch_bounds__(256) void box_blur3_kernel(const float* __restrict__ in, float* __restrict__ out, int D0, int D1, int D2, int axis,
                                                        int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int64_t vol = (int64_t)D0 * D1 * D2;
    const int64_t e = i % vol;
    const int i2 = (int)(e % D2), i1 = (int)((e / D2) % D1), i0 = (int)(e / ((int64_t)D1 * D2));
    const int idx = axis == 0 ? i0 : (axis == 1 ? i1 : i2);
    const int dim = axis == 0 ? D0 : (axis == 1 ? D1 : D2);
    const int64_t stride = axis == 0 ? (int64_t)D1 * D2 : (axis == 1 ? D2 : 1);
    const double w = (double)(1.0f / 3.0f);
    double a = w * (double)in[i];
    if (idx + 1 < dim) a += w * (double)in[i + stride];
    if (idx > 0) a += w * (double)in[i - stride];
    out[i] = (float)a;
}

// coords[i] += mag * (trilinear interpolation of the three noise volumes at coords[i]); grid axis d = linspace(-(b_d - 1) gran,
// (b_d - 1) gran, b_d); points outside the grid are not moved (RegularGridInterpolator(bounds_error=0, fill_value=0), :463-468)
__global__ __launch_bounds__(256) void elastic_kernel(float* __restrict__ coords, int64_t n, const float* __restrict__ noise, int D0, int D1, int D2,
                                                      float gran, float mag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int dim[3] = {D0, D1, D2};
    double x[3], f[3];
    int lo[3];
    bool inside = true;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        x[d] = (double)coords[i * 3 + d];
        const double half = (double)(dim[d] - 1) * (double)gran;
        inside = inside && x[d] >= -half && x[d] <= half;
        // node spacing of linspace(-half, half, b) = 2 gran
        const double u = (x[d] + half) / (2.0 * (double)gran);
        int l = (int)floor(u);
        l = l < 0 ? 0 : (l > dim[d] - 2 ? dim[d] - 2 : l);
        lo[d] = l; f[d] = u - (double)l;
    }
    if (!inside) return;
    const int64_t vol = (int64_t)D0 * D1 * D2;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* g = noise + c * vol;
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int a = k >> 2, b = (k >> 1) & 1, e = k & 1;
            const double w = (a ? f[0] : 1.0 - f[0]) * (b ? f[1] : 1.0 - f[1]) * (e ? f[2] : 1.0 - f[2]);
            v += w * (double)g[((int64_t)(lo[0] + a) * D1 + (lo[1] + b)) * D2 + (lo[2] + e)];
        }
        coords[i * 3 + c] = (float)(x[c] + v * (double)mag);
    }
}

#define ST ((hipStream_t)stream)
extern "C" {

int sd3d_augment_points(float* points, int ld, int64_t n, int flip_x, int flip_y, float angle, float scale, const float* trans3,
                        const float* color_mean3, const float* color_std3, void* stream) {
    if (n <= 0) return SD3D_OK;
    if (ld < 3 || (color_mean3 && ld < 6)) return sd3d_set_error(SD3D_ERR_ARG, "augment_points: row stride too small");
    AffineParams p;
    p.flip_x = flip_x; p.flip_y = flip_y; p.c = cosf(angle); p.s = sinf(angle); p.scale = scale;
    for (int c = 0; c < 3; ++c) p.t[c] = trans3 ? trans3[c] : 0.f;
    p.color = color_mean3 != nullptr;
    for (int c = 0; c < 3; ++c) { p.mean[c] = color_mean3 ? color_mean3[c] : 0.f; p.stdv[c] = color_std3 ? color_std3[c] : 1.f; }
    augment_points_kernel<<<(unsigned)cdiv(n, 256), 256, 0, ST>>>(points, ld, n, p);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

int sd3d_voxel_units(const float* points, int ld, int64_t n, float voxel_size, float* coords, void* stream) {
    if (n <= 0) return SD3D_OK;
    if (voxel_size <= 0.f) return sd3d_set_error(SD3D_ERR_ARG, "voxel_units: voxel size must be positive");
    voxel_units_kernel<<<(unsigned)cdiv(n, 256), 256, 0, ST>>>(points, ld, n, voxel_size, coords);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

int sd3d_box_blur3(const float* in, float* out, int grids, int D0, int D1, int D2, int axis, void* stream) {
    if (grids <= 0 || D0 <= 0 || D1 <= 0 || D2 <= 0 || axis < 0 || axis > 2 || in == out) return sd3d_set_error(SD3D_ERR_ARG, "box_blur3: bad arguments");
    const int64_t total = (int64_t)grids * D0 * D1 * D2;
    box_blur3_kernel<<<(unsigned)cdiv(total, 256), 256, 0, ST>>>(in, out, D0, D1, D2, axis, total);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

int sd3d_elastic_displace(float* coords, int64_t n, const float* noise, int D0, int D1, int D2, float gran, float mag, void* stream) {
    if (n <= 0) return SD3D_OK;
    if (D0 < 2 || D1 < 2 || D2 < 2 || gran <= 0.f) return sd3d_set_error(SD3D_ERR_ARG, "elastic_displace: noise volume must be at least 2 nodes per axis");
    elastic_kernel<<<(unsigned)cdiv(n, 256), 256, 0, ST>>>(coords, n, noise, D0, D1, D2, gran, mag);
    SD3D_CHECK_LAUNCH();
    return SD3D_OK;
}

}  // extern "C"
