// frames.hip -- the step right after the solver in the reference's render loop: simulated vertices
// -> per-face frames (MeshGaussianModel.set_mesh_by_verts, /root/reference/scene/mesh_gaussian_model.py:137-146 with
// compute_face_orientation, utils/graphics_utils.py:88-106) -> position / rotation / scale of the Gaussians bound to the
// faces (GaussianModel.get_xyz / get_rotation / get_scaling, scene/gaussian_model.py:112-151).  Pure per-face and
// per-Gaussian maps, HBM-bound: 24 B of indices + gathered vertices in, 68 B out per face; 36 B in + 32 B of gathered
// frame, 44 B out per Gaussian.  Keeps the simulated vertices on the device instead of the reference's
// OBJ -> disk -> Blender -> reload detour for everything but the AO map.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/mpmhip.h"

namespace {

constexpr int TPB = 256;

struct F3 { float x, y, z; };
__device__ __forceinline__ F3 ld3(const float *p, int i) { return F3{p[3 * (size_t)i], p[3 * (size_t)i + 1], p[3 * (size_t)i + 2]}; }
__device__ __forceinline__ F3 sub(F3 a, F3 b) { return F3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ F3 cross(F3 a, F3 b) { return F3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// graphics_utils.py:82-86: x / sqrt(clamp(dot(x, x), min = 1e-20))
__device__ __forceinline__ float length(F3 a) { return sqrtf(fmaxf(dot(a, a), 1e-20f)); }
__device__ __forceinline__ F3 safe_normalize(F3 a) { float l = length(a); return F3{a.x / l, a.y / l, a.z / l}; }

// roma.rotmat_to_unitquat (= SciPy's Rotation.from_matrix): XYZW, the largest of (R00, R11, R22, trace) picks the
// branch (first maximum wins, like argmax), no sign canonicalisation
__device__ __forceinline__ void rotmat_to_quat_xyzw(const float R[3][3], float q[4]) {
  float d[4] = {R[0][0], R[1][1], R[2][2], 0.0f};
  d[3] = d[0] + d[1] + d[2];
  int c = 0;
  for (int k = 1; k < 4; ++k)
    if (d[k] > d[c]) c = k;
  if (c != 3) {
    int i = c, j = (i + 1) % 3, k = (j + 1) % 3;
    q[i] = 1.0f - d[3] + 2.0f * R[i][i];
    q[j] = R[j][i] + R[i][j];
    q[k] = R[k][i] + R[i][k];
    q[3] = R[k][j] - R[j][k];
  } else {
    q[0] = R[2][1] - R[1][2];
    q[1] = R[0][2] - R[2][0];
    q[2] = R[1][0] - R[0][1];
    q[3] = 1.0f + d[3];
  }
  float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int k = 0; k < 4; ++k) q[k] = q[k] / n;
}

__global__ void k_face_frames(const float *verts, const int32_t *faces, int n_f, float *center, float *mat, float *quat,
                              float *scale) {
  int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n_f) return;
  int i0 = faces[3 * (size_t)f], i1 = faces[3 * (size_t)f + 1], i2 = faces[3 * (size_t)f + 2];
  F3 v0 = ld3(verts, i0), v1 = ld3(verts, i1), v2 = ld3(verts, i2);
  // triangles.mean(dim=-2): ((v0 + v1) + v2) / 3
  center[3 * (size_t)f] = ((v0.x + v1.x) + v2.x) / 3.0f;
  center[3 * (size_t)f + 1] = ((v0.y + v1.y) + v2.y) / 3.0f;
  center[3 * (size_t)f + 2] = ((v0.z + v1.z) + v2.z) / 3.0f;
  F3 e1 = sub(v1, v0), e2 = sub(v2, v0);
  F3 a0 = safe_normalize(e1);
  F3 a1 = safe_normalize(cross(a0, e2));
  F3 a2 = safe_normalize(cross(a1, a0));
  a2 = F3{-a2.x, -a2.y, -a2.z};  // "will have artifacts without negation", graphics_utils.py:99
  float R[3][3] = {{a0.x, a1.x, a2.x}, {a0.y, a1.y, a2.y}, {a0.z, a1.z, a2.z}};  // columns a0 a1 a2
  float *m = mat + 9 * (size_t)f;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) m[3 * r + c] = R[r][c];
  scale[f] = (length(e1) + fabsf(dot(a2, e2))) / 2.0f;
  float q[4];
  rotmat_to_quat_xyzw(R, q);
  float *o = quat + 4 * (size_t)f;  // stored WXYZ (quat_xyzw_to_wxyz)
  o[0] = q[3]; o[1] = q[0]; o[2] = q[1]; o[3] = q[2];
}

// torch.nn.functional.normalize(q, dim=-1): q / max(|q|, 1e-12)
__device__ __forceinline__ void normalize4(float q[4]) {
  float n = fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
  for (int k = 0; k < 4; ++k) q[k] = q[k] / n;
}

__global__ void k_bind_gaussians(int n_g, const int32_t *binding, const float *xyz_local, const float *rot_raw,
                                 const float *scaling_raw, const float *center, const float *mat, const float *quat,
                                 const float *fscale, float *xyz, float *rot, float *scaling) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_g) return;
  int f = binding[g];
  float s = fscale[f];
  if (xyz) {  // bmm(face_orien_mat[binding], _xyz) * face_scaling + face_center, gaussian_model.py:149-150
    const float *m = mat + 9 * (size_t)f;
    F3 p = ld3(xyz_local, g), c = ld3(center, f);
    xyz[3 * (size_t)g] = (m[0] * p.x + m[1] * p.y + m[2] * p.z) * s + c.x;
    xyz[3 * (size_t)g + 1] = (m[3] * p.x + m[4] * p.y + m[5] * p.z) * s + c.y;
    xyz[3 * (size_t)g + 2] = (m[6] * p.x + m[7] * p.y + m[8] * p.z) * s + c.z;
  }
  if (rot) {  // quat_product(normalize(face quat), normalize(_rotation)), WXYZ in and out, :133-136
    float a[4], b[4];
    for (int k = 0; k < 4; ++k) { a[k] = quat[4 * (size_t)f + k]; b[k] = rot_raw[4 * (size_t)g + k]; }
    normalize4(a);
    normalize4(b);
    float pw = a[0], px = a[1], py = a[2], pz = a[3], qw = b[0], qx = b[1], qy = b[2], qz = b[3];
    float *o = rot + 4 * (size_t)g;
    o[0] = pw * qw - px * qx - py * qy - pz * qz;
    o[1] = pw * qx + px * qw + py * qz - pz * qy;
    o[2] = pw * qy - px * qz + py * qw + pz * qx;
    o[3] = pw * qz + px * qy - py * qx + pz * qw;
  }
  if (scaling)  // exp(_scaling) * face_scaling[binding], :121-122
    for (int k = 0; k < 3; ++k) scaling[3 * (size_t)g + k] = expf(scaling_raw[3 * (size_t)g + k]) * s;
}

// The argument lists of the rasteriser call (gaussian_renderer/__init__.py:52-103), assembled in ONE launch: rows [0, n_g) are the
// mesh-bound Gaussians -- means3D = get_xyz, rotations = get_rotation, scales = get_scaling (as k_bind_gaussians), opacities =
// sigmoid(_opacity) (gaussian_model.py:39,158-160) -- rows [n_g, n_g + n_x) the caller's `extra` primitives copied behind them
// (:84-91: the five torch.cat of the render call); means2D (zeros, :27) is cleared here too.
__global__ void k_render_inputs(int n_g, int n_x, const int32_t *binding, const float *xyz_local, const float *rot_raw,
                                const float *scaling_raw, const float *opacity_raw, const float *center, const float *mat,
                                const float *quat, const float *fscale, const float *x_xyz, const float *x_opacity,
                                const float *x_scales, const float *x_rot, float *means3D, float *means2D, float *opacities,
                                float *scales, float *rotations) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_g + n_x) return;
  for (int k = 0; k < 3; ++k) means2D[3 * (size_t)g + k] = 0.0f;
  if (g >= n_g) {
    int e = g - n_g;
    for (int k = 0; k < 3; ++k) { means3D[3 * (size_t)g + k] = x_xyz[3 * (size_t)e + k]; scales[3 * (size_t)g + k] = x_scales[3 * (size_t)e + k]; }
    for (int k = 0; k < 4; ++k) rotations[4 * (size_t)g + k] = x_rot[4 * (size_t)e + k];
    opacities[g] = x_opacity[e];
    return;
  }
  int f = binding[g];
  float s = fscale[f];
  const float *m = mat + 9 * (size_t)f;
  F3 p = ld3(xyz_local, g), c = ld3(center, f);
  means3D[3 * (size_t)g] = (m[0] * p.x + m[1] * p.y + m[2] * p.z) * s + c.x;
  means3D[3 * (size_t)g + 1] = (m[3] * p.x + m[4] * p.y + m[5] * p.z) * s + c.y;
  means3D[3 * (size_t)g + 2] = (m[6] * p.x + m[7] * p.y + m[8] * p.z) * s + c.z;
  float a[4], b[4];
  for (int k = 0; k < 4; ++k) { a[k] = quat[4 * (size_t)f + k]; b[k] = rot_raw[4 * (size_t)g + k]; }
  normalize4(a);
  normalize4(b);
  float pw = a[0], px = a[1], py = a[2], pz = a[3], qw = b[0], qx = b[1], qy = b[2], qz = b[3];
  float *o = rotations + 4 * (size_t)g;
  o[0] = pw * qw - px * qx - py * qy - pz * qz;
  o[1] = pw * qx + px * qw + py * qz - pz * qy;
  o[2] = pw * qy - px * qz + py * qw + pz * qx;
  o[3] = pw * qz + px * qy - py * qx + pz * qw;
  for (int k = 0; k < 3; ++k) scales[3 * (size_t)g + k] = expf(scaling_raw[3 * (size_t)g + k]) * s;
  opacities[g] = 1.0f / (1.0f + expf(-opacity_raw[g]));  // torch.sigmoid
}

// compute_cov_from_F (/root/reference/warp_mpm/mpm_utils.py:1108-1132): cov = F_trial * sym(cov0) * F_trial^T per
// particle, upper triangle out (xx, xy, xz, yy, yz, zz).  Products summed left to right as Warp's mat33 product does.
__global__ void k_cov_from_F(const float *F_trial, const float *cov0, int n, float *out) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  float F[3][3], S[3][3], T[3][3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) F[r][c] = F_trial[9 * (size_t)p + 3 * r + c];
  const float *q = cov0 + 6 * (size_t)p;
  S[0][0] = q[0]; S[0][1] = q[1]; S[0][2] = q[2];
  S[1][0] = q[1]; S[1][1] = q[3]; S[1][2] = q[4];
  S[2][0] = q[2]; S[2][1] = q[4]; S[2][2] = q[5];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) T[r][c] = F[r][0] * S[0][c] + F[r][1] * S[1][c] + F[r][2] * S[2][c];
  float *o = out + 6 * (size_t)p;
  int k = 0;
  for (int r = 0; r < 3; ++r)
    for (int c = r; c < 3; ++c) o[k++] = T[r][0] * F[c][0] + T[r][1] * F[c][1] + T[r][2] * F[c][2];
}

int check(hipError_t e) { return e == hipSuccess ? MPMHIP_OK : MPMHIP_ERR_HIP; }

}  // namespace

extern "C" {

int mpmhip_cov_from_F(int32_t device, void *stream, const float *particle_F_trial, const float *particle_cov, int32_t n,
                      float *new_cov) {
  if (n < 0 || (n > 0 && (!particle_F_trial || !particle_cov || !new_cov))) return MPMHIP_ERR_INVALID;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0 || device < 0 || device >= n_dev) return MPMHIP_ERR_NO_DEVICE;
  if (n == 0) return MPMHIP_OK;
  if (int rc = check(hipSetDevice(device))) return rc;
  hipLaunchKernelGGL(k_cov_from_F, (unsigned)((n + TPB - 1) / TPB), TPB, 0, (hipStream_t)stream, particle_F_trial, particle_cov, n,
                     new_cov);
  return check(hipGetLastError());
}


int mpmhip_face_frames(int32_t device, void *stream, const float *verts, const int32_t *faces, int32_t n_faces,
                       float *face_center, float *face_orien_mat, float *face_orien_quat, float *face_scaling) {
  if (n_faces < 0 || (n_faces > 0 && (!verts || !faces || !face_center || !face_orien_mat || !face_orien_quat || !face_scaling)))
    return MPMHIP_ERR_INVALID;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0 || device < 0 || device >= n_dev) return MPMHIP_ERR_NO_DEVICE;
  if (n_faces == 0) return MPMHIP_OK;
  if (int rc = check(hipSetDevice(device))) return rc;
  hipLaunchKernelGGL(k_face_frames, (unsigned)((n_faces + TPB - 1) / TPB), TPB, 0, (hipStream_t)stream, verts, faces, n_faces,
                     face_center, face_orien_mat, face_orien_quat, face_scaling);
  return check(hipGetLastError());
}

int mpmhip_bind_gaussians(int32_t device, void *stream, int32_t n_gaussians, const int32_t *binding, const float *xyz_local,
                          const float *rotation_raw, const float *scaling_raw, const float *face_center,
                          const float *face_orien_mat, const float *face_orien_quat, const float *face_scaling, float *xyz,
                          float *rotation, float *scaling) {
  if (n_gaussians < 0) return MPMHIP_ERR_INVALID;
  if (n_gaussians > 0 && (!binding || !face_scaling || (xyz && (!xyz_local || !face_center || !face_orien_mat)) ||
                          (rotation && (!rotation_raw || !face_orien_quat)) || (scaling && !scaling_raw)))
    return MPMHIP_ERR_INVALID;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0 || device < 0 || device >= n_dev) return MPMHIP_ERR_NO_DEVICE;
  if (n_gaussians == 0) return MPMHIP_OK;
  if (int rc = check(hipSetDevice(device))) return rc;
  hipLaunchKernelGGL(k_bind_gaussians, (unsigned)((n_gaussians + TPB - 1) / TPB), TPB, 0, (hipStream_t)stream, n_gaussians,
                     binding, xyz_local, rotation_raw, scaling_raw, face_center, face_orien_mat, face_orien_quat,
                     face_scaling, xyz, rotation, scaling);
  return check(hipGetLastError());
}

int mpmhip_render_inputs(int32_t device, void *stream, int32_t n_gaussians, int32_t n_extra, const int32_t *binding,
                         const float *xyz_local, const float *rotation_raw, const float *scaling_raw, const float *opacity_raw,
                         const float *face_center, const float *face_orien_mat, const float *face_orien_quat,
                         const float *face_scaling, const float *extra_xyz, const float *extra_opacity, const float *extra_scales,
                         const float *extra_rotations, float *means3D, float *means2D, float *opacities, float *scales,
                         float *rotations) {
  if (n_gaussians < 0 || n_extra < 0) return MPMHIP_ERR_INVALID;
  if (n_gaussians + n_extra > 0 && (!means3D || !means2D || !opacities || !scales || !rotations)) return MPMHIP_ERR_INVALID;
  if (n_gaussians > 0 && (!binding || !xyz_local || !rotation_raw || !scaling_raw || !opacity_raw || !face_center || !face_orien_mat ||
                          !face_orien_quat || !face_scaling))
    return MPMHIP_ERR_INVALID;
  if (n_extra > 0 && (!extra_xyz || !extra_opacity || !extra_scales || !extra_rotations)) return MPMHIP_ERR_INVALID;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0 || device < 0 || device >= n_dev) return MPMHIP_ERR_NO_DEVICE;
  if (n_gaussians + n_extra == 0) return MPMHIP_OK;
  if (int rc = check(hipSetDevice(device))) return rc;
  hipLaunchKernelGGL(k_render_inputs, (unsigned)((n_gaussians + n_extra + TPB - 1) / TPB), TPB, 0, (hipStream_t)stream, n_gaussians,
                     n_extra, binding, xyz_local, rotation_raw, scaling_raw, opacity_raw, face_center, face_orien_mat, face_orien_quat,
                     face_scaling, extra_xyz, extra_opacity, extra_scales, extra_rotations, means3D, means2D, opacities, scales,
                     rotations);
  return check(hipGetLastError());
}

}  // extern "C"
