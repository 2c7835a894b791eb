// bc.hpp -- grid boundary conditions evaluated per node (shared by both back ends).
// Reference: the `collide` closures registered by add_surface_collider (mpm_solver.py:600-655),
// set_velocity_on_cuboid (:950-973), add_bounding_box (:993-1050), enforce_grid_velocity_by_mask
// (:1341-1352).  Node position = index * dx.
#pragma once
#include "ctx.hpp"
#include "mpm_math.hpp"

namespace mpm {

// node position minus a BC's reference point, with the product and the difference rounded separately: a node exactly on
// the face of a cuboid / on a plane must be classified the same way by both kernel back ends and by the CPU oracle
// (built with -ffp-contract=off), whatever the surrounding code lets the compiler fuse.  HIP's __fmul_rn / __fsub_rn
// are plain operators and do get contracted, hence the pragma (the `contract` flag is per instruction and survives
// inlining).
__device__ __forceinline__ float node_off(int g, float dx, float p) {
#pragma clang fp contract(off)
  float pos = (float)g * dx;
  return pos - p;
}
__device__ __forceinline__ float plane_side(V3 off, const float *n) {
#pragma clang fp contract(off)
  float a = off.x * n[0], b = off.y * n[1], c = off.z * n[2];
  return (a + b) + c;
}

// returns true when v was (re)written
__device__ __forceinline__ bool apply_bc(const BC &bc, V3 &v, int gx, int gy, int gz, int G, float dx, float time,
                                         float dt, size_t dense_index) {
  if (bc.type == BC_GRIDMASK) {
    if (bc.mask[dense_index] >= 1) { v = v3(0, 0, 0); return true; }
    return false;
  }
  bool in_window = time >= bc.start_time && time < bc.end_time;
  if (bc.type == BC_SURFACE) {
    if (!in_window) return false;
    V3 off = v3(node_off(gx, dx, bc.point[0]), node_off(gy, dx, bc.point[1]), node_off(gz, dx, bc.point[2]));
    float dp = plane_side(off, bc.normal);
    if (!(dp < 0.0f)) return false;
    if (bc.surface_type == 11) {
      float z = node_off(gz, dx, 0.0f);
      if (z < 0.4f || z > 0.53f) v = v3(0, 0, 0);
      else v = v3(v.x * 0.3f, 0.0f, v.z * 0.3f);
    } else {
      // sticky; slip / frictional surfaces also end in a zero write (quirk Q1, mpm_solver.py:653-655)
      v = v3(0, 0, 0);
    }
    return true;
  }
  if (bc.type == BC_CUBOID) {
    if (in_window) {
      V3 off = v3(node_off(gx, dx, bc.point[0]), node_off(gy, dx, bc.point[1]), node_off(gz, dx, bc.point[2]));
      if (fabsf(off.x) < bc.size[0] && fabsf(off.y) < bc.size[1] && fabsf(off.z) < bc.size[2]) {
        v = v3(bc.velocity[0], bc.velocity[1], bc.velocity[2]);
        return true;
      }
      return false;
    }
    if (bc.reset == 1 && time < bc.end_time + 15.0f * dt) { v = v3(0, 0, 0); return true; }
    return false;
  }
  if (bc.type == BC_BBOX) {
    if (!in_window) return false;
    const int padding = 3;
    bool ch = false;
    if (gx < padding && v.x < 0.0f) { v.x = 0.0f; ch = true; }
    if (gx >= G - padding && v.x > 0.0f) { v.x = 0.0f; ch = true; }
    if (gy < padding && v.y < 0.0f) { v.y = 0.0f; ch = true; }
    if (gy >= G - padding && v.y > 0.0f) { v.y = 0.0f; ch = true; }
    if (gz < padding && v.z < 0.0f) { v.z = 0.0f; ch = true; }
    if (gz >= G - padding && v.z > 0.0f) { v.z = 0.0f; ch = true; }
    return ch;
  }
  return false;
}

// Conservative box test for apply_bc: false only if bc cannot change any node with indices in [lo, hi] (per axis) at this
// time.  g2p evaluates nodes tile by tile and skips the BC loop for tiles no BC can reach.
__device__ __forceinline__ bool bc_may_touch(const BC &bc, int lox, int loy, int loz, int hix, int hiy, int hiz, int G,
                                             float dx, float time, float dt) {
  if (bc.type == BC_GRIDMASK) return true;
  bool in_window = time >= bc.start_time && time < bc.end_time;
  if (bc.type == BC_SURFACE) {
    if (!in_window) return false;
    // dp is affine in the node index: its minimum over the box is taken at a corner (same float ops as apply_bc)
    float lo[3] = {node_off(lox, dx, bc.point[0]), node_off(loy, dx, bc.point[1]), node_off(loz, dx, bc.point[2])};
    float hi[3] = {node_off(hix, dx, bc.point[0]), node_off(hiy, dx, bc.point[1]), node_off(hiz, dx, bc.point[2])};
    float dpmin = 0.0f, mag = 0.0f;
    for (int a = 0; a < 3; ++a) {
      dpmin += fminf(lo[a] * bc.normal[a], hi[a] * bc.normal[a]);
      mag += fmaxf(fabsf(lo[a] * bc.normal[a]), fabsf(hi[a] * bc.normal[a]));
    }
    return dpmin < 1e-5f * mag + 1e-30f;  // margin for the different summation order
  }
  if (bc.type == BC_CUBOID) {
    if (!in_window) return bc.reset == 1 && time < bc.end_time + 15.0f * dt;
    // |g dx - p| < size somewhere in [lo, hi]: g dx - p is monotone in g
    return (node_off(lox, dx, bc.point[0]) < bc.size[0]) && (node_off(hix, dx, bc.point[0]) > -bc.size[0]) &&
           (node_off(loy, dx, bc.point[1]) < bc.size[1]) && (node_off(hiy, dx, bc.point[1]) > -bc.size[1]) &&
           (node_off(loz, dx, bc.point[2]) < bc.size[2]) && (node_off(hiz, dx, bc.point[2]) > -bc.size[2]);
  }
  if (bc.type == BC_BBOX) {
    const int padding = 3;
    return in_window && (lox < padding || loy < padding || loz < padding || hix >= G - padding || hiy >= G - padding ||
                         hiz >= G - padding);
  }
  return false;
}

// host-side `modify` of set_velocity_on_cuboid, mpm_solver.py:975-981
inline void bc_host_modify(BC &bc, float time, float dt) {
  if (bc.type == BC_CUBOID && time >= bc.start_time && time < bc.end_time)
    for (int a = 0; a < 3; ++a) bc.point[a] = bc.point[a] + dt * bc.velocity[a];
}

}  // namespace mpm
